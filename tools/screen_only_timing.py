import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viquae_amd.index import MI355XFlatIndex
N, d, nq, k = int(os.environ.get('N', 1500000)), int(os.environ.get("D", 768)), int(os.environ.get("NQ", 4096)), 100
dev = torch.device("cuda")
METRIC = int(os.environ.get("METRIC", 0))
idx = MI355XFlatIndex(string_factory="Flat", metric_type=METRIC, screen=os.environ.get("SCREEN", "1") == "1")
g = torch.Generator(device=dev); g.manual_seed(0)
for s in range(0, N, 1 << 16):
    idx.add(torch.randn((min(1 << 16, N - s), d), generator=g, device=dev), total_hint=N)
Q = torch.randn((nq, d), generator=g, device=dev)
for _ in range(2): D, I = idx.search_device(Q, k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): D, I = idx.search_device(Q, k)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"metric={METRIC} N={N} QPX={os.environ.get('MQ_KNN_QPX')} nq={nq} d={d}: {ms:.2f} ms -> {nq / ms * 1e3:.0f} q/s", idx.screen_stats(nq, k) if idx.screen else "", flush=True)
