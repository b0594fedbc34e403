"""is Pillow's JPEG decoding (libjpeg-turbo: ISLOW IDCT, fancy upsampling, integer YCbCr tables) reproducible
bit for bit outside the library?  Decodes N files with Pillow and with the numpy restatement oracle/jpeg.py and counts the
files (and pixels) that differ.

    python tools/jpeg_pillow_parity.py [N=1000] [--seed S] [--dir D]   # --dir: also every *.jpg / *.jpeg under D

The files are written by Pillow's own encoder from synthetic pictures (smooth gradients + texture + hard edges + noise, so that
every frequency band and the clamps are exercised): sizes 1 .. 640 per side (most not multiples of the MCU), qualities 5 .. 100,
sampling 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0 (Pillow's `subsampling` 0 / 1 / 2 + one hand-set case), grey, optimised Huffman tables
or the standard ones, restart intervals.  Files this restatement declines (progressive, CMYK, ...) are counted separately."""
import io
import os
import sys
import time

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import jpeg as oj  # noqa: E402


def picture(rng, h, w, grey=False):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    ch = []
    for _ in range(1 if grey else 3):
        a = 128 + 100 * np.sin(xx / rng.uniform(3, 60) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(3, 60) + rng.uniform(0, 6))
        a += rng.uniform(-60, 60) * ((xx // rng.integers(2, 40) + yy // rng.integers(2, 40)) % 2)       # hard edges
        a += rng.normal(0, rng.choice([0, 2, 10, 40]), (h, w))                                            # noise
        if rng.random() < 0.3:                                                                             # saturated patches: the clamps
            y0, x0 = rng.integers(0, h), rng.integers(0, w)
            a[y0:y0 + h // 3 + 1, x0:x0 + w // 3 + 1] = rng.choice([0, 255])
        ch.append(np.clip(a, 0, 255))
    a = np.stack(ch, -1).astype(np.uint8)
    return Image.fromarray(a[:, :, 0], "L") if grey else Image.fromarray(a, "RGB")


def encode(rng, im):
    kw = dict(quality=int(rng.choice([5, 20, 35, 50, 65, 75, 85, 90, 95, 100])), optimize=bool(rng.random() < 0.3))
    if im.mode == "RGB":
        kw["subsampling"] = int(rng.choice([0, 1, 2, 2]))
    if rng.random() < 0.2:
        kw["restart_marker_blocks"] = int(rng.integers(1, 9))
    buf = io.BytesIO()
    try:
        im.save(buf, "JPEG", **kw)
    except OSError:   # libjpeg's "Suspension not allowed here": Pillow's output buffer is too small for this option set
        kw.pop("restart_marker_blocks", None)
        kw["optimize"] = False
        buf = io.BytesIO()
        im.save(buf, "JPEG", **kw)
    return buf.getvalue(), kw


def sizes(rng):
    r = rng.random()
    if r < 0.1:
        return int(rng.integers(1, 20)), int(rng.integers(1, 20))
    if r < 0.8:
        return int(rng.integers(16, 260)), int(rng.integers(16, 260))
    return int(rng.integers(200, 641)), int(rng.integers(200, 641))


def compare(data):
    ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    got = oj.decode(data)
    return ref, got


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if args else 1000
    seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 0
    rng = np.random.default_rng(seed)
    t0 = time.time()
    files = []
    for i in range(n):
        h, w = sizes(rng)
        data, kw = encode(rng, picture(rng, h, w, grey=rng.random() < 0.1))
        files.append((f"synthetic[{i}] {h}x{w} {kw}", data))
    if "--dir" in sys.argv:
        root = sys.argv[sys.argv.index("--dir") + 1]
        for d, _, names in os.walk(root):
            for nm in sorted(names):
                if nm.lower().endswith((".jpg", ".jpeg")):
                    files.append((os.path.join(d, nm), open(os.path.join(d, nm), "rb").read()))
    same = differ = declined = 0
    pixels = bad_pixels = 0
    worst = 0
    kinds = {}
    for name, data in files:
        try:
            ref, got = compare(data)
        except oj.Unsupported as e:
            declined += 1
            print("declined:", name, e)
            continue
        f = oj.read_coefficients(data)
        key = "grey" if len(f["components"]) == 1 else "x".join(f"{c['h']}{c['v']}" for c in f["components"])
        kinds[key] = kinds.get(key, 0) + 1
        pixels += ref.size
        if ref.shape == got.shape and np.array_equal(ref, got):
            same += 1
        else:
            differ += 1
            d = np.abs(ref.astype(int) - got.astype(int)) if ref.shape == got.shape else None
            nb = int((d != 0).sum()) if d is not None else ref.size
            bad_pixels += nb
            worst = max(worst, int(d.max()) if d is not None else 255)
            print("DIFFERS:", name, "shape", ref.shape, got.shape, "values off:", nb, "max", None if d is None else int(d.max()))
    print(f"{len(files)} files in {time.time() - t0:.1f} s: identical {same}, different {differ}, declined {declined}; "
          f"sampling kinds {kinds}; {pixels} sample values compared, {bad_pixels} off (worst by {worst})")
    return 1 if differ else 0


if __name__ == "__main__":
    sys.exit(main())
