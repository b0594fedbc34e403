"""Do the tails of one screened search (candidate selection, re-scoring: memory / latency-bound) overlap the scan of the next
one (matrix-bound) when consecutive searches alternate between two streams with their own workspaces?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from viquae_amd import _lib
from viquae_amd.index import MI355XFlatIndex
lib = _lib.load()
dev = torch.device("cuda")
rows, d, nq, k = 1_500_000, 768, 4096, 100
g = torch.Generator(device=dev); g.manual_seed(0)
idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
for s in range(0, rows, 1 << 16):
    idx.add(torch.randn((min(1 << 16, rows - s), d), generator=g, device=dev), total_hint=rows)
Q = torch.randn((nq, d), generator=g, device=dev)
nb = int(lib.mq_knn_workspace_bytes_metric(rows, d, nq, k, 0))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ws = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(2)]
D = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(2)]
I = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(2)]
def step(j, st):
    _lib.check(lib.mq_knn_search_screened_f32(None, idx._sqnorm.data_ptr(), idx._rowmajor.data_ptr(), idx._bf16.data_ptr(), idx._xmax2.data_ptr(),
                                              rows, d, Q.data_ptr(), nq, k, 0, 0, 0, D[j].data_ptr(), I[j].data_ptr(), ws[j].data_ptr(), nb,
                                              st.cuda_stream, None, None))
for mode in ("one stream", "two streams", "one stream", "two streams"):
    for _ in range(3): step(0, streams[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 40
    for i in range(n):
        j = i & 1 if mode == "two streams" else 0
        step(j, streams[j])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{mode}: {dt * 1e3:.3f} ms per 4096-query search -> {nq / dt:.0f} queries/s")
print("results equal:", bool(torch.equal(D[0], D[1]) and torch.equal(I[0], I[1])))
