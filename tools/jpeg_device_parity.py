"""The split JPEG decoder END TO END (host Huffman scan + the GPU kernels of csrc/jpeg.hip) against Pillow over N files written by
Pillow's encoder: tools/jpeg_pillow_parity.py's pictures and option mix, a third of them progressive.  Needs an MI355X.

    python tools/jpeg_device_parity.py [N=1000] [--seed S]"""
import io
import os
import sys
import time

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import jpeg_pillow_parity as jp  # noqa: E402
from viquae_amd.image import jpeg as dj  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if args else 1000
    seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 0
    rng = np.random.default_rng(seed)
    same = differ = values = 0
    kinds = {}
    t0 = time.time()
    for lo in range(0, n, 100):
        files = []
        for i in range(lo, min(n, lo + 100)):
            h, w = jp.sizes(rng)
            im = jp.picture(rng, h, w, grey=rng.random() < 0.1)
            kw = dict(quality=int(rng.choice([5, 20, 35, 50, 65, 75, 85, 90, 95, 100])))
            if im.mode == "RGB":
                kw["subsampling"] = int(rng.choice([0, 1, 2, 2]))
            if i % 3 == 0:
                kw["progressive"] = True
            elif rng.random() < 0.3:
                kw["optimize"] = True
            buf = io.BytesIO()
            try:
                im.save(buf, "JPEG", **kw)
            except OSError:   # libjpeg's "Suspension not allowed here": Pillow's output buffer is too small for this picture / option set
                kw.pop("optimize", None)
                kw["quality"] = min(kw["quality"], 75)
                try:
                    buf = io.BytesIO()
                    im.save(buf, "JPEG", **kw)
                except OSError:
                    kw.pop("progressive", None)
                    buf = io.BytesIO()
                    im.save(buf, "JPEG", **kw)
            files.append(buf.getvalue())
            key = ("progressive " if kw.get("progressive") else "") + ("grey" if im.mode == "L" else {0: "4:4:4", 1: "4:2:2", 2: "4:2:0"}[kw["subsampling"]])
            kinds[key] = kinds.get(key, 0) + 1
        for d, g in zip(files, dj.decode_files(files)):
            ref = np.asarray(Image.open(io.BytesIO(d)).convert("RGB"))
            values += ref.size
            if np.array_equal(ref, g.cpu().numpy()):
                same += 1
            else:
                differ += 1
                print("DIFFERS:", ref.shape)
    print(f"{n} files in {time.time() - t0:.1f} s: identical {same}, different {differ}; {values} sample values compared; kinds {kinds}")
    return 1 if differ else 0


if __name__ == "__main__":
    sys.exit(main())
