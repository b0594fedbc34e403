"""FAISS's small-batch L2 path (csrc/knn_direct.inc) over a 1.5M x 768 shard: ms per search for 1 / 8 / 19 queries, and the
20-query BLAS-form search (screened path) next to it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viquae_amd.index import MI355XFlatIndex
N, d = 1_500_000, 768
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(0)
idx = MI355XFlatIndex(string_factory="Flat", metric_type=1)
for s in range(0, N, 1 << 16):
    idx.add(torch.randn((min(1 << 16, N - s), d), generator=g, device=dev), total_hint=N)
for nq in (1, 8, 19, 20, 256):
    Q = torch.randn((nq, d), generator=g, device=dev)
    idx.search_device(Q, 100); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        idx.search_device(Q, 100)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 20
    print(f"L2 nq={nq:3d}: {t * 1e3:.3f} ms per search ({'direct (q-x)^2 form' if nq < 20 else 'BLAS form, screened'}); "
          f"KB bytes / t = {N * d * 4 / t / 1e12:.2f} TB/s" if nq < 20 else f"L2 nq={nq:3d}: {t * 1e3:.3f} ms per search (BLAS form, screened)")
