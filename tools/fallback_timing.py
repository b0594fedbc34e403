"""Cost of the exact-scan fallback of a screened index, with and without the panel copy: 1.5M x 768 integer KB whose 3-hot
queries make every query tile overflow (all 16 tiles recomputed by knn_scan_kernel).  usage: python tools/fallback_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viquae_amd.index import MI355XFlatIndex

N, D, NQ, K = 1_500_000, 768, 4096, 100
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(9)
idx = {kp: MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True, keep_panel=kp) for kp in (False, True)}
for s in range(0, N, 1 << 16):
    x = torch.randint(-2, 3, (min(1 << 16, N - s), D), generator=g, device=dev).float()
    for i in idx.values():
        i.add(x, total_hint=N)
Q = torch.randint(-2, 3, (NQ, D), generator=g, device=dev).float()
keep = torch.zeros((NQ, D), device=dev)
keep.scatter_(1, torch.rand((NQ, D), generator=g, device=dev).topk(3, dim=1).indices, 1.0)
Q = torch.where(Q == 0, torch.ones_like(Q), Q) * keep
res = {}
for kp, i in idx.items():
    i.search_device(Q, K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = i.search_device(Q, K)
    torch.cuda.synchronize()
    res[kp] = out
    print(f"keep_panel={kp}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per 4096-query search, tiles recomputed {i.screen_stats(NQ, K)[0]}")
assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
print("identical results")
