"""Damaged JPEG files: which of them the split decoder (csrc/jpeg.hip) accepts, and whether the accepted ones still decode like
Pillow.  One bit of the entropy-coded data of a good file is flipped per trial; `stage` (mq_jpeg_read_coefficients) either
declines (the product then uses Pillow) or accepts, and the accepted files' pixels -- oracle/jpeg.py's arithmetic on the
library's coefficients, which the device kernels reproduce (tests/test_jpeg_gpu.py) -- are compared with Pillow's.  Baseline and
progressive files.

    python tools/jpeg_damage_parity.py [files=12] [flips per file=300]"""
import io
import os
import sys
import warnings

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import jpeg as oj  # noqa: E402
import jpeg_pillow_parity as jp  # noqa: E402
from viquae_amd.image import jpeg as dj  # noqa: E402


def main():
    nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    flips = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    acc = rej = same = diff = perr = 0
    q = [85, 50, 95, 30, 75, 90, 10, 60, 100, 5, 70, 80]
    for seed in range(nfiles):
        rng = np.random.default_rng(seed)
        kw = dict(quality=q[seed % 12])
        grey = seed % 12 == 7
        if not grey:
            kw["subsampling"] = seed % 3
        if seed % 12 in (4, 9):
            kw["restart_marker_blocks"] = 2
        if seed % 12 in (2, 5, 7, 9, 10):
            kw["progressive"] = True
        buf = io.BytesIO()
        jp.picture(rng, 64 + (seed % 12) * 13, 80 + (seed % 12) * 7, grey=grey).save(buf, "JPEG", **kw)
        good = buf.getvalue()
        p = dj.probe(good)
        st = np.zeros(p[4], dtype=np.uint8)
        assert dj.stage(good, st.ctypes.data, st.size)
        sos = good.find(b"\xff\xda")
        for _ in range(flips):
            b = bytearray(good)
            pos = int(rng.integers(sos + 14, len(good) - 2))
            b[pos] ^= 1 << int(rng.integers(0, 8))
            b = bytes(b)
            if not dj.stage(b, st.ctypes.data, st.size):
                rej += 1
                continue
            acc += 1
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    ref = np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))
            except Exception as e:  # noqa: BLE001
                perr += 1
                print("Pillow raises on an accepted file:", e, "seed", seed, "pos", pos)
                continue
            if np.array_equal(ref, oj.decode_staging(st)):   # the oracle's arithmetic on the library's coefficients
                same += 1
            else:
                diff += 1
                print("DIFFERENT: seed", seed, "pos", pos)
    print(f"accepted {acc}, declined {rej}; of the accepted: identical to Pillow {same}, different {diff}, Pillow raised {perr}")
    return 1 if diff or perr else 0


if __name__ == "__main__":
    sys.exit(main())
