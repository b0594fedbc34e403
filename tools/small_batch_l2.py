"""The reference's 256-query batch under the L2 metric (FAISS's default `metric_type`) over 1.5M x 768 rows: ms per search through the
streaming kernel -- which serves d = 768 since round 6 (the row term enters the MFMA chain as fp32, csrc/knn_small8.inc) -- and through
the 256 x 256 tile kernel, same process; bench.py reports it as `secondary.small_batch_l2`.  usage: python tools/small_batch_l2.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main(rows=1_500_000, d=768, nq=256, k=100, reps=50):
    from viquae_amd import _lib
    from viquae_amd.index import MI355XFlatIndex
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(0)
    idx = MI355XFlatIndex(device=dev.index, string_factory="Flat", metric_type=1, screen=True)
    for s in range(0, rows, 1 << 16):
        idx.add(torch.randn((min(1 << 16, rows - s), d), generator=g, device=dev), total_hint=rows)
    Q = torch.randn((nq, d), generator=g, device=dev)
    out = {"workload": f"{nq} queries x {rows}x{d} KB, exact L2 top-{k} (FAISS's BLAS form), MI355XFlatIndex.search_device"}
    ref = None
    for small, key in ((1, "streaming"), (0, "tile")):
        with _lib.knn_option(_lib.KNN_OPT_SMALL_SCAN, small):
            kind = idx.scan_kind(nq, k)
            for _ in range(10):
                D, I = idx.search_device(Q, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                idx.search_device(Q, k)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            st = idx.screen_stats(nq, k)
            out[key] = {"scan_kind": kind, "ms": round(ms, 4), "queries_per_s": round(nq / ms * 1e3, 1),
                        "query_tiles_recomputed_exactly": int(st[0]), "candidates_rescored_per_query": round(st[1] / nq, 1)}
            if ref is None:
                ref = (D.clone(), I.clone())
            else:
                out["results_identical"] = bool(torch.equal(I, ref[1]) and torch.equal(D.view(torch.int32), ref[0].view(torch.int32)))
    del idx
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    print(json.dumps(main()))
