import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import image as oi
from viquae_amd.image.preprocess import CLIPImageProcessorHIP
rng = np.random.default_rng(0)
p = CLIPImageProcessorHIP(do_rescale=False, do_normalize=False)
for (h,w) in [(300,200),(224,224),(375,500),(500,375),(640,480),(225,1000),(100,80),(2000,1500),(60,9000)]:
    im = rng.integers(0,256,(h,w,3),dtype=np.uint8)
    got = p([im])["pixel_values"][0].cpu().numpy()
    want = oi.clip_preprocess([im], do_rescale=False, do_normalize=False)[0]
    d = got != want
    cols = np.where(d.any(axis=(0,1)))[0]; rows = np.where(d.any(axis=(0,2)))[0]
    print((h,w), "mismatch", d.mean(), "cols", cols[:6], "..", cols[-3:], "rows", rows[:4], "maxdiff", np.abs(got-want).max())
