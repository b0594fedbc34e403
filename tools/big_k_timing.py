"""Cost of a search by k at BASELINE configs[1]'s size (1.5M x 768, 4096 queries, IP): the screen's own range (k <= 224), the
row-range path beyond it (225 ... 1792: P = ceil(k / 112) ranges, merged and proved), the exact rounds (MQ_KNN_PARTITIONS=0, and
k > 1792).  usage: python tools/big_k_timing.py [k ...]   (MQ_KNN_PARTITIONS=0 python tools/big_k_timing.py for the old path)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viquae_amd.index import MI355XFlatIndex


def main():
    ks = [int(a) for a in sys.argv[1:]] or [100, 224, 256, 384, 512, 1024, 1792]
    g = torch.Generator(device="cuda").manual_seed(0)
    X = torch.randn((1_500_000, 768), generator=g, device="cuda")
    Q = torch.randn((4096, 768), generator=g, device="cuda")
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
    idx.add(X)
    del X
    out = {"partitions": os.environ.get("MQ_KNN_PARTITIONS", "1") != "0", "rows": 1_500_000, "d": 768, "nq": 4096, "ms": {}}
    for k in ks:
        idx.search_device(Q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3 if k > 224 and not out["partitions"] else 5
        for _ in range(n):
            idx.search_device(Q, k)
        torch.cuda.synchronize()
        out["ms"][k] = round((time.perf_counter() - t0) / n * 1e3, 2)
        print(k, out["ms"][k], "ms", flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
