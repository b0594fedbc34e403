"""Decode phase of viquae_amd.image.decode_pool alone: ms per 3072-image batch against the number of worker processes,
six consecutive batches each (first-touch effects show in the first two = one per slot).  No GPU work besides registration."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from PIL import Image
from viquae_amd.image.decode_pool import DecodePool
from viquae_amd.image.preprocess import CLIPImageProcessorHIP
rng = np.random.default_rng(1)
work = tempfile.mkdtemp()
for i in range(64):
    Image.fromarray(rng.integers(0, 256, (375, 500, 3), dtype=np.uint8)).save(os.path.join(work, f"{i}.bmp"))
paths = [os.path.join(work, f"{int(i)}.bmp") for i in rng.integers(0, 64, 3072)]
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
torch.zeros(1, device="cuda")
tr = CLIPImageProcessorHIP()
for pin in ("1", "0"):
    os.environ["MQ_IMAGE_PIN"] = pin
    for procs in (16, 32, 64, 128):
        pool = DecodePool(procs, 3072 * 768 * 1024, 2)
        line = []
        for rep in range(6):
            t0 = time.perf_counter(); sizes = pool.sizes(paths); t1 = time.perf_counter()
            geom, totals = tr.plan(np.array(sizes, dtype=np.int64))
            slot = pool.take_slot(); t2 = time.perf_counter()
            pool.decode(slot, {k: int(g[0]) for k, g in enumerate(geom)}); t3 = time.perf_counter()
            line.append(f"{(t1 - t0) * 1e3:.0f}+{(t3 - t2) * 1e3:.0f}")
        print(f"pin={pin} procs={procs}: sizes+decode ms per batch: " + " ".join(line), flush=True)
        pool.close()
