import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from viquae_amd.image.decode_pool import DecodePool
from viquae_amd.image.preprocess import CLIPImageProcessorHIP
rng = np.random.default_rng(1)
work = tempfile.mkdtemp()
for i in range(64):
    Image.fromarray(rng.integers(0, 256, (375, 500, 3), dtype=np.uint8)).save(os.path.join(work, f"{i}.bmp"))
paths = [os.path.join(work, f"{int(i)}.bmp") for i in rng.integers(0, 64, 3072)]
torch.zeros(1, device="cuda")
tr = CLIPImageProcessorHIP()
t0 = time.perf_counter(); pool = DecodePool(32, 3072 * 768 * 1024, 2); print("pool create", time.perf_counter() - t0)
side = torch.cuda.Stream()
for rep in range(3):
    t0 = time.perf_counter(); sizes = pool.sizes(paths); t1 = time.perf_counter()
    geom, totals = tr.plan(np.array(sizes, dtype=np.int64)); t2 = time.perf_counter()
    slot = pool.take_slot(); failed = pool.decode(slot, {k: int(g[0]) for k, g in enumerate(geom)}); t3 = time.perf_counter()
    with torch.cuda.stream(side):
        packed = pool.tensors[slot]
        ta = time.perf_counter(); src = packed[:int(totals[0])].to("cuda", non_blocking=True); side.synchronize(); tb = time.perf_counter()
        out = tr.run_packed(packed, geom, totals, len(paths)); t4 = time.perf_counter()
    print(f"rep {rep}: sizes {t1-t0:.3f} plan {t2-t1:.3f} decode {t3-t2:.3f} h2d-only {tb-ta:.3f} ({totals[0]/1e9:.2f} GB) run_packed {t4-tb:.3f}")
pool.close()
