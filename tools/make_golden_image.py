"""Golden vectors for the image preprocessing (oracle/image.py, csrc/image.hip), minted in THIS container from the
un-vendored dependencies that hold the arithmetic: Pillow (``Image.resize``) and transformers' PIL image processor
(``CLIPImageProcessor`` -> ``CLIPImageProcessorPil``), which is what the reference's ``transform(images,
return_tensors="pt")`` (meerqat/image/embedding.py:141-152) runs.  Writes tests/golden/image_resize.npz and
tests/golden/image_clip.npz.  Run: ``python tools/make_golden_image.py``."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

RESIZE_CASES = [(37, 53), (53, 37), (32, 32), (64, 48), (33, 150), (150, 33), (200, 31), (1, 1), (2, 90), (97, 131), (160, 120)]
CLIP_CONFIGS = {
    "default64": dict(size={"shortest_edge": 64}, crop_size={"height": 64, "width": 64}),
    "bilinear48": dict(size={"shortest_edge": 48}, crop_size={"height": 40, "width": 40}, resample=2),
    "exact": dict(size={"height": 50, "width": 70}, crop_size={"height": 44, "width": 60}),
    "raw": dict(size={"shortest_edge": 32}, crop_size={"height": 32, "width": 32}, do_normalize=False),
    "noscale": dict(size={"shortest_edge": 32}, crop_size={"height": 32, "width": 32}, do_rescale=False),
    "noresize": dict(do_resize=False, crop_size={"height": 30, "width": 30}),
}
CLIP_SIZES = [(150, 100), (64, 64), (65, 200), (37, 53), (128, 96)]


def images(rng, sizes):
    out = []
    for t, (h, w) in enumerate(sizes):
        if t % 2 == 0:
            out.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        else:  # smooth gradients with a hard edge: overshoot of the cubic filter -> the clip to [0, 255] matters
            yy, xx = np.mgrid[0:h, 0:w]
            im = np.stack([yy * 255 // max(h - 1, 1), xx * 255 // max(w - 1, 1), (yy + xx) % 256], -1).astype(np.uint8)
            im[h // 3: h // 2, w // 4: w // 2] = (255, 0, 255)
            out.append(im)
    return out


def main():
    from PIL import Image
    from transformers import CLIPImageProcessor
    import PIL
    import transformers
    rng = np.random.default_rng(2024)
    rec = {"pillow_version": np.array(PIL.__version__), "shortest_edge": np.array(32)}
    for c, im in enumerate(images(rng, RESIZE_CASES)):
        h, w = im.shape[:2]
        short, long = min(h, w), max(h, w)
        nl = int(32 * long / short)
        oh, ow = (nl, 32) if w <= h else (32, nl)
        rec[f"in_{c}"] = im
        for kind in (2, 3):
            rec[f"out_{c}_k{kind}"] = np.array(Image.fromarray(im).resize((ow, oh), resample=kind))
    np.savez_compressed(os.path.join(GOLD, "image_resize.npz"), **rec)

    rec = {"transformers_version": np.array(transformers.__version__), "pillow_version": np.array(PIL.__version__)}
    ims = images(rng, CLIP_SIZES)
    for c, im in enumerate(ims):
        rec[f"in_{c}"] = im
    for name, cfg in CLIP_CONFIGS.items():
        proc = CLIPImageProcessor(**cfg)
        use = ims if name != "noresize" else [im for im in ims if min(im.shape[:2]) >= 30]
        rec[f"pixel_values_{name}"] = proc([Image.fromarray(im) for im in use], return_tensors="np")["pixel_values"]
    np.savez_compressed(os.path.join(GOLD, "image_clip.npz"), **rec)
    for f in ("image_resize.npz", "image_clip.npz"):
        print(f, os.path.getsize(os.path.join(GOLD, f)), "bytes")


if __name__ == "__main__":
    main()
