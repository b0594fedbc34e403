"""Index build time: 1.5M x 768 fp32 rows from host memory (numpy) into the screened index."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from viquae_amd.index import MI355XFlatIndex
N, d = int(os.environ.get("N", 1500000)), 768
rng = np.random.default_rng(0)
X = rng.standard_normal((N, d), dtype=np.float32)
for screen in (True, False):
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=screen)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx.add_vectors(X)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"screen={screen}: add_vectors {N}x{d} from numpy: {t:.2f} s ({N * d * 4 / t / 1e9:.1f} GB/s of fp32 rows)")
    del idx; torch.cuda.empty_cache()
