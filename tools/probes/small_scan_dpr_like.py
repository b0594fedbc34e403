"""The 256-query streaming scan on DPR-like data (a shared direction of norm 9 + N(0, 0.25^2) noise, bench.py's `dpr_like_data`) and on
clustered data: query tiles recomputed exactly, candidates per query, ms per search -- against the tile kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from viquae_amd import _lib
from viquae_amd.index import MI355XFlatIndex
dev = torch.device("cuda"); rows, d, nq, k = 1_500_000, 768, 256, 100
g = torch.Generator(device=dev).manual_seed(7)
mu = torch.randn((1, d), generator=g, device=dev); mu = 9.0 * mu / mu.norm()
for name in ("dpr_like", "clusters_1000"):
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0)
    if name == "dpr_like":
        for s0 in range(0, rows, 1 << 16):
            idx.add(mu + 0.25 * torch.randn((min(1 << 16, rows - s0), d), generator=g, device=dev), total_hint=rows)
        Q = mu + 0.25 * torch.randn((nq, d), generator=g, device=dev)
    else:
        c = torch.randn((1000, d), generator=g, device=dev)
        for s0 in range(0, rows, 1 << 16):
            n = min(1 << 16, rows - s0)
            idx.add(c[torch.randint(0, 1000, (n,), generator=g, device=dev)] + 0.3 * torch.randn((n, d), generator=g, device=dev), total_hint=rows)
        Q = c[torch.randint(0, 1000, (nq,), generator=g, device=dev)] + 0.3 * torch.randn((nq, d), generator=g, device=dev)
    for small in ("1", "0"):
      with _lib.knn_option(_lib.KNN_OPT_SMALL_SCAN, int(small)):   # through the C ABI: the environment is read once at load
        D, I = idx.search_device(Q, k); torch.cuda.synchronize()
        st = idx.screen_stats(nq, k)
        t0 = time.perf_counter()
        for _ in range(20): idx.search_device(Q, k)
        torch.cuda.synchronize()
        print(name, "streaming" if small == "1" else "tile", f"{(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", "flagged", st[0], "cand/query", st[1] / nq, "max pool", st[3], "keys", st[4])
    del idx
