"""search_device at the reference's map batch size (256 queries, k = 100) over the 1.5M x 768 KB: ms per batch, for a
rocprofv3 --kernel-trace --stats run (python3 tools/small_batch_search.py [nq] [steps] [k])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viquae_amd.index import MI355XFlatIndex

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows, d, k = 1_500_000, 768, (int(sys.argv[3]) if len(sys.argv) > 3 else 100)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
idx = MI355XFlatIndex(string_factory="Flat", metric_type=0)
for s in range(0, rows, 1 << 16):
    idx.add(torch.randn((min(1 << 16, rows - s), d), generator=g, device=dev), total_hint=rows)
Q = torch.randn((nq, d), generator=g, device=dev)
for _ in range(3):
    idx.search_device(Q, k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    idx.search_device(Q, k)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / steps
print(f"nq={nq} k={k}: {t * 1e3:.3f} ms per batch = {nq / t:.0f} queries/s")
