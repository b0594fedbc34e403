"""First GPU contact: a few parity cases + a timing of the scan at full size (scratch tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from viquae_amd.index import MI355XFlatIndex
from oracle import knn as ok

def case(n, d, nq, k, metric):
    rng = np.random.default_rng(n + d)
    X = rng.standard_normal((n, d), dtype=np.float32); Q = rng.standard_normal((nq, d), dtype=np.float32)
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=metric); idx.add_vectors(X)
    D, I = idx.search_batch(Q, k)
    Do, Io = ok.knn(X, Q, k, metric=metric)
    print(f"n={n} d={d} nq={nq} k={k} metric={metric}: idx_equal={np.array_equal(I, Io)} score_equal={np.array_equal(D, Do)}", flush=True)
    if not np.array_equal(I, Io):
        bad = np.nonzero((I != Io).any(1))[0]
        print(" bad queries", bad[:10], "\n", I[bad[0]][:12], "\n", Io[bad[0]][:12], "\n", D[bad[0]][:6], Do[bad[0]][:6])

case(2048, 64, 37, 10, 0)
case(10000, 768, 256, 100, 0)
case(10000, 768, 256, 100, 1)
case(777, 100, 3, 1, 0)

# full-size timing
N, d, nq, k = int(os.environ.get("N", 1500000)), 768, int(os.environ.get("NQ", 4096)), 100
dev = torch.device("cuda")
idx = MI355XFlatIndex(string_factory="Flat", metric_type=0)
g = torch.Generator(device=dev); g.manual_seed(0)
t0 = time.time()
for s in range(0, N, 1 << 16):
    n = min(1 << 16, N - s)
    idx.add(torch.randn((n, d), generator=g, device=dev), total_hint=N)
torch.cuda.synchronize(); print("index build s", time.time() - t0, flush=True)
Q = torch.randn((nq, d), generator=g, device=dev)
for it in range(2):
    D, I = idx.search_device(Q, k)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
iters = 5
for it in range(iters):
    D, I = idx.search_device(Q, k)
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / iters
print(f"search {nq} q over {N}x{d}: {ms:.2f} ms -> {nq / ms * 1e3:.0f} q/s, {2 * nq * N * d / ms / 1e9:.1f} TFLOP/s", flush=True)
# sanity on the big run: sample queries against torch
S = (Q[:8] @ torch.cat([idx.reconstruct_rows_torch(i, 1 << 16) for i in range(0, N, 1 << 16)]).T) if hasattr(idx, "reconstruct_rows_torch") else None
