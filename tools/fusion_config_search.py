#!/usr/bin/env python3
"""bench.py's `secondary.fusion_config_search`: the four dense indexes of the paper's headline fusion run,
/root/reference/experiments/ir/viquae/dpr+arcface+clip+imagenet/config_test.json:5-56 --

    DPR_few_shot_dp   768-d   "Flat"          metric_type 0      (data/viquae_passages)
    resnet           2048-d   "L2norm,Flat"   metric_type 0      (imagenet-RN50, non_humans)
    clip-RN50        1024-d   "L2norm,Flat"   metric_type 0      (non_humans)
    arcface           512-d   "L2norm,Flat"   metric_type 0      (first_face_embedding, humans_with_faces)

all with `device: null` (-> FAISS's own NormalizationTransform arithmetic, l2norm_form "faiss") and searched in 256-question
batches (map_kwargs.batch_size, :59-62), each over a synthetic 1.5M-row KB (BASELINE's KB size; the real tables hold 11.9M
passages / ~1M articles / ~0.5M articles) through the same C-ABI call the index makes (mq_knn_search_screened_f32), HIP events
around the scan kernel.  Per index: which scan serves it (`scan_kind`), the scan's time, the fraction of the 8 TB/s HBM
roofline that reading the bf16 screening copy ONCE amounts to and the fraction of the 2.5 PFLOP/s bf16 peak its 2 nq N dp
flops amount to (256 questions x 2048 columns: the larger of the two), the whole call, and the screen's own statistics.

Data: every vector = a shared component + isotropic noise (``shared`` : 1 in norm ratio; image / DPR embeddings are not centred),
the same generator for KB rows and questions."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

PEAK_HBM_GBPS = 8000.0
PEAK_BF16_TFLOPS = 2500.0
INDEXES = [  # (index name, column, d, string_factory) in config order
    ("DPR_few_shot_dp", "DPR_few_shot", 768, "Flat"),
    ("resnet", "imagenet-RN50", 2048, "L2norm,Flat"),
    ("clip-RN50", "clip-RN50", 1024, "L2norm,Flat"),
    ("arcface", "first_face_embedding", 512, "L2norm,Flat"),
]


def _fill(idx, rows, d, seed, shared, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    mu = torch.randn((1, d), generator=g, device=device)
    mu = shared * mu / mu.norm() * d ** 0.5        # `shared` x the noise's expected norm
    step = max(1 << 12, (1 << 26) // d // 64 * 64)
    for s in range(0, rows, step):
        idx.add(mu + torch.randn((min(step, rows - s), d), generator=g, device=device), total_hint=rows)
    return lambda n: mu + torch.randn((n, d), generator=g, device=device)


def one_index(name, column, d, factory, rows, nq, k, reps, shared, device):
    from viquae_amd import _lib
    from viquae_amd.index import MI355XFlatIndex
    lib = _lib.load()
    idx = MI355XFlatIndex(string_factory=factory, metric_type=0, screen=True, l2norm_form="faiss")
    make_queries = _fill(idx, rows, d, seed=d, shared=shared, device=device)
    Q = make_queries(nq).contiguous()
    stream = torch.cuda.current_stream(device)
    ws_bytes = int(lib.mq_knn_workspace_bytes_metric(rows, d, nq, k, idx._screen_metric))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    D = torch.empty((nq, k), dtype=torch.float32, device=device)
    I = torch.empty((nq, k), dtype=torch.int64, device=device)
    flags = idx._search_flags()

    def call(e0=None, e1=None):
        _lib.check(lib.mq_knn_search_screened_f32(
            idx._packed.data_ptr() if idx._packed is not None else None, idx._sqnorm.data_ptr(), idx._rowmajor.data_ptr(),
            idx._bf16.data_ptr(), idx._xmax2.data_ptr(), rows, d, Q.data_ptr(), nq, k, idx._screen_metric, flags, 0, D.data_ptr(), I.data_ptr(),
            ws.data_ptr(), ws_bytes, stream.cuda_stream, e0.cuda_event if e0 else None, e1.cuda_event if e1 else None),
            "mq_knn_search_screened_f32")

    for _ in range(25):  # (the first calls after the index build run 10 - 30 % slow -- clocks settling; 3 warm-up calls left that in the mean)
        call()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(stream)
        b.record(stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        call(a, b)
    torch.cuda.synchronize()
    call_ms = (time.perf_counter() - t0) / reps * 1e3
    scan_ms = sum(a.elapsed_time(b) for a, b in evs) / reps
    idx._ws = idx._last_ws = ws
    idx._last_call_nq = nq
    st = idx.screen_stats(nq, k)
    # the answer is the exact one: every returned score is the k-ordered fp32 product of its row, best first, ids unique
    Xn = idx._rowmajor[:rows]
    rows_of = Xn[I[:8].reshape(-1)].reshape(8, k, d)
    qn = Q[:8]
    if "L2norm" in factory:
        qn = qn * (1.0 / qn.double().pow(2).sum(1, keepdim=True).sqrt()).float()
    ok = bool(((rows_of.double() * qn.double()[:, None, :]).sum(-1) - D[:8].double()).abs().max() < 1e-3 * max(1.0, float(D[:8].abs().max()))
              and (D[:, 1:] <= D[:, :-1]).all() and (I >= 0).all())
    dp = (d + (2 if idx._screen_metric else 0) + 63) // 64 * 64
    kb_bytes = rows * dp * 2
    out = {"index": name, "column": column, "d": d, "string_factory": factory, "metric_type": 0, "l2norm_form": "faiss" if "L2norm" in factory else None,
           "queries_centred": idx._screen_metric == 2, "bf16_columns": dp,
           "scan_kind": idx.scan_kind(nq, k), "scan_ms": round(scan_ms, 4), "call_ms": round(call_ms, 4),
           "queries_per_s": round(nq / call_ms * 1e3, 1),
           "algorithmic_hbm_bytes": kb_bytes, "hbm_frac": round(kb_bytes / (scan_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
           # 256 questions against every row: at 1024 / 2048 columns the scan is nearer the matrix pipe's roofline than HBM's
           "bf16_tflops": round(2.0 * nq * rows * dp / (scan_ms * 1e-3) / 1e12, 1),
           "mfma_frac": round(2.0 * nq * rows * dp / (scan_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
           "screen": {"query_tiles_recomputed_exactly": st[0], "candidates_rescored_per_query": round(st[1] / nq, 1),
                      "max_candidates_of_a_query": st[2], "pool_keys_per_query": round(st[4] / nq, 1)},
           "sanity_scores_sorted_and_rescored": ok}
    del idx, ws, D, I, Q
    torch.cuda.empty_cache()
    return out


def main(rows=1_500_000, nq=256, k=100, reps=20, shared=1.0, only=None):
    device = torch.device("cuda", torch.cuda.current_device())
    out = {"workload": f"the four dense indexes of experiments/ir/viquae/dpr+arcface+clip+imagenet/config_test.json, {rows} synthetic rows each "
                       f"(shared component : noise = {shared} : 1), {nq}-question batches (map_kwargs.batch_size), exact IP top-{k}, one C-ABI call per batch",
           "indexes": []}
    for name, column, d, factory in INDEXES:
        if only and name != only:
            continue
        try:
            out["indexes"].append(one_index(name, column, d, factory, rows, nq, k, reps, shared, device))
        except Exception as e:  # one index failing must not lose the others
            out["indexes"].append({"index": name, "d": d, "error": repr(e)})
    ms = [r["call_ms"] for r in out["indexes"] if "call_ms" in r]
    if len(ms) == len(INDEXES) and not only:
        out["all_four_ms_per_batch"] = round(sum(ms), 4)
        out["questions_per_s_all_four"] = round(nq / sum(ms) * 1e3, 1)
    return out


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_500_000)
    ap.add_argument("--shared", type=float, default=1.0)
    ap.add_argument("--nq", type=int, default=256)
    ap.add_argument("--only", default=None, help="one index name (profiling)")
    a = ap.parse_args()
    print(json.dumps(main(rows=a.rows, nq=a.nq, shared=a.shared, only=a.only)))
