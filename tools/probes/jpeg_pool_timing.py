"""The decode workers on JPEG files alone (no tower): ms per 3072-image batch for the two phases, with the split decoder
(workers Huffman-decode into staging areas, viquae_amd/image/jpeg.py) and without (Pillow decodes to RGB), 16 / 32 workers,
six consecutive batches each.  Files as in tools/bench_encode_surface.py (500 x 375, quality 90)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from PIL import Image
from viquae_amd.image import jpeg as dj
from viquae_amd.image.decode_pool import DecodePool
from viquae_amd.image.preprocess import CLIPImageProcessorHIP
rng = np.random.default_rng(1)
work = tempfile.mkdtemp()
yy, xx = np.mgrid[0:375, 0:500]
for i in range(64):
    f = rng.uniform(0.005, 0.05, (3, 2))
    a = np.stack([127 + 100 * np.sin(f[c, 0] * xx + i) * np.cos(f[c, 1] * yy) for c in range(3)], axis=2)
    Image.fromarray(np.clip(a + rng.normal(0, 6, a.shape), 0, 255).astype(np.uint8)).save(os.path.join(work, f"{i}.jpg"), quality=90)
paths = [os.path.join(work, f"{int(i)}.jpg") for i in rng.integers(0, 64, 3072)]
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip(), "cpus", os.cpu_count(), flush=True)
except Exception as e:
    print("no cgroup cpu.max", e)
data = open(paths[0], "rb").read()
p = dj.probe(data)
st = np.zeros(p[4], dtype=np.uint8)
t0 = time.perf_counter()
for _ in range(200):
    dj.stage(data, st.ctypes.data, st.size)
t1 = time.perf_counter()
for _ in range(200):
    np.asarray(Image.open(paths[0]).convert("RGB"))
t2 = time.perf_counter()
print(f"one thread, one file of {len(data)} bytes: scan into staging {(t1 - t0) / 200 * 1e3:.3f} ms, Pillow {(t2 - t1) / 200 * 1e3:.3f} ms", flush=True)
if "--no-gpu" not in sys.argv:
    torch.zeros(1, device="cuda")
tr = CLIPImageProcessorHIP()
for mode in ("1", "0"):
    os.environ["MQ_IMAGE_DEVICE_JPEG"] = mode
    for procs in (16, 32):
        pool = DecodePool(procs, 3072 * 768 * 1024, 2)
        line = []
        for rep in range(6):
            t0 = time.perf_counter(); sizes = pool.sizes(paths); t1 = time.perf_counter()
            geom, totals = tr.plan(np.array(sizes, dtype=np.int64))
            layout = dj.plan_layout(geom, totals, dict(pool.last_jpeg)) if pool.last_jpeg else None
            slot = pool.take_slot(); t2 = time.perf_counter()
            pool.decode(slot, {k: int(layout["staging"][k] if layout else g[0]) for k, g in enumerate(geom)}); t3 = time.perf_counter()
            line.append(f"{(t1 - t0) * 1e3:.0f}+{(t2 - t1) * 1e3:.0f}+{(t3 - t2) * 1e3:.0f}")
        print(f"split decoder={mode} procs={procs}: sizes + plan + decode ms per batch: " + " ".join(line), flush=True)
        pool.close()
