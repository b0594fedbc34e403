import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import knn as ok
rng = np.random.default_rng(0)
N = 1 << 18
X = rng.standard_normal((N, 768), dtype=np.float32); Q = rng.standard_normal((4096, 768), dtype=np.float32)
print("omp threads", ok.num_threads(), os.environ.get("OMP_NUM_THREADS"), os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES"))
for blk in (4096, 16384, 65536):
    ok.knn_blas(X[:8192], Q, 100, backend="c", block=blk)
    t = time.perf_counter(); ok.knn_blas(X, Q, 100, backend="c", block=blk); dt = time.perf_counter() - t
    print("c block", blk, round(2 * 4096 * N * 768 / dt / 1e9), "GFLOP/s", flush=True)
