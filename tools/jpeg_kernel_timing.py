"""The device half of the split JPEG decoder alone (csrc/jpeg.hip: jpeg_idct_kernel + jpeg_rgb_kernel) on one batch of the image
job: 3072 references to 64 synthetic 500 x 375 4:2:0 files (tools/bench_encode_surface.py's content, quality 90), staged by this
process.  Prints ms per batch and the achieved rate against the algorithmic bytes (per image: coefficient blocks read + sample
bytes written by the inverse DCT, sample bytes read + RGB bytes written by the colour kernel).  Run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split (profiles/r06_jpeg_kernel_stats.csv)."""
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from viquae_amd.image import jpeg as dj

rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:375, 0:500]
files = []
for i in range(64):
    f = rng.uniform(0.005, 0.05, (3, 2))
    a = np.stack([127 + 100 * np.sin(f[c, 0] * xx + i) * np.cos(f[c, 1] * yy) for c in range(3)], axis=2)
    buf = io.BytesIO()
    Image.fromarray(np.clip(a + rng.normal(0, 6, a.shape), 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=90)
    files.append(buf.getvalue())
B = 3072
refs = rng.integers(0, 64, B)
infos = [dj.probe(d) for d in files]
st_off, off = [], 0
for r in refs:
    st_off.append(off)
    off += infos[r][4]
h2d = off
rgb_off = []
for r in refs:
    rgb_off.append(off)
    off += (infos[r][0] * infos[r][1] * 3 + 15) & ~15
host = torch.empty(h2d, dtype=torch.uint8, pin_memory=True)
hnp = host.numpy()
staged = {}
for n, r in enumerate(refs):   # one Huffman decode per distinct file, then copies
    if r not in staged:
        assert dj.stage(files[r], hnp.ctypes.data + st_off[n], infos[r][4])
        staged[r] = st_off[n]
    else:
        hnp[st_off[n]:st_off[n] + infos[r][4]] = hnp[staged[r]:staged[r] + infos[r][4]]
pristine = host.cuda()
buf = torch.empty(off, dtype=torch.uint8, device="cuda")
items = np.array(list(zip(st_off, rgb_off)), dtype=np.int64)
mb, mp = max(i[3] for i in infos), max(dj.strips(i[0], i[1]) for i in infos)
times = []
for it in range(12):
    buf[:h2d].copy_(pristine)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dj.decode_staged(buf, items, mb, mp)
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1))
ms = float(np.median(times[2:]))
blocks = sum(infos[r][3] for r in refs)
pixels = sum(infos[r][0] * infos[r][1] for r in refs)
alg = blocks * 128 + blocks * 64 + (pixels * 3 // 2) + pixels * 3   # 4:2:0: 1.5 samples per pixel feed one RGB pixel
ref = np.asarray(Image.open(io.BytesIO(files[refs[5]])).convert("RGB"))
got = buf[rgb_off[5]:rgb_off[5] + ref.size].view(ref.shape).cpu().numpy()
print(json.dumps({"images": B, "ms_per_batch": round(ms, 3), "algorithmic_bytes": int(alg), "achieved_gbps": round(alg / ms / 1e6, 1),
                  "hbm_frac": round(alg / ms / 1e6 / 8000, 3), "identical_to_pillow": bool(np.array_equal(ref, got)),
                  "images_per_s_device_half": round(B / ms * 1e3)}))
