"""256-query searches over 1.5M x 768 rows: inner product against the L2 metric, streaming kernel against the tile kernel -- ms per
search and per scan (HIP events around the scan through the C ABI).  Round 6: the L2 metric at d = 768 takes the streaming kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from viquae_amd import _lib
from viquae_amd.index import MI355XFlatIndex

rows, d, nq, k = 1_500_000, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 100
dev = torch.device("cuda")
for metric, name in ((0, "inner product"), (1, "L2")):
    g = torch.Generator(device=dev).manual_seed(0)
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=metric)
    for s in range(0, rows, 1 << 16):
        idx.add(torch.randn((min(1 << 16, rows - s), d), generator=g, device=dev), total_hint=rows)
    Q = torch.randn((nq, d), generator=g, device=dev)
    ref = None
    for small in (1, 0):
        with _lib.knn_option(_lib.KNN_OPT_SMALL_SCAN, small):
            kind = idx.scan_kind(nq, k)
            for _ in range(5):
                D, I = idx.search_device(Q, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                idx.search_device(Q, k)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 50 * 1e3
            scan = idx.last_scan_ms() if hasattr(idx, "last_scan_ms") else float("nan")
            st = idx.screen_stats(nq, k)
            same = "" if ref is None else f"  same results as the streaming kernel: {bool(torch.equal(I, ref[1]) and torch.equal(D.view(torch.int32), ref[0].view(torch.int32)))}"
            ref = ref or (D.clone(), I.clone())
            print(f"{name:14s} {kind:6s}: {ms:.3f} ms per search ({nq / ms * 1e3:.0f} queries/s), tiles recomputed {st[0]}, candidates per query {st[1] / nq:.0f}{same}")
    del idx
