import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viquae_amd.index import MI355XFlatIndex
N, d, nq, k = 1_500_000, 768, 4096, 100
dev = torch.device("cuda")
res = {}
for screen in (True, False):
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=screen)
    g = torch.Generator(device=dev); g.manual_seed(0)
    for s in range(0, N, 1 << 16):
        idx.add(torch.randn((min(1 << 16, N - s), d), generator=g, device=dev), total_hint=N)
    Q = torch.randn((nq, d), generator=g, device=dev)
    for _ in range(2): D, I = idx.search_device(Q, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): D, I = idx.search_device(Q, k)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    res[screen] = (D.clone(), I.clone())
    extra = idx.screen_stats(nq, k) if screen else ()
    print(f"screen={screen}: {ms:.2f} ms -> {nq / ms * 1e3:.0f} q/s", extra, flush=True)
    del idx; torch.cuda.empty_cache()
print("identical:", torch.equal(res[True][0], res[False][0]), torch.equal(res[True][1], res[False][1]))
