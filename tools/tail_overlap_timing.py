"""A search of several 4096-query chunks over 1.5M x 768 (BASELINE configs[4]'s per-GPU work: 16,384 queries): the serial chunk
loop against the pipelined one (second halves on a second stream under the next scan).  usage: python tools/tail_overlap_timing.py [nq]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viquae_amd.index import MI355XFlatIndex


def timed(fn, steps=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    g = torch.Generator(device="cuda").manual_seed(0)
    X = torch.randn((1_500_000, 768), generator=g, device="cuda")
    Q = torch.randn((nq, 768), generator=g, device="cuda")
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
    idx.add(X)
    del X
    rec = {"nq": nq, "rows": 1_500_000, "d": 768, "k": 100}
    os.environ["MQ_KNN_TAIL_OVERLAP"] = "0"
    D0, I0 = idx.search_device(Q, 100)
    rec["one_chunk_ms"] = timed(lambda: idx.search_device(Q[:4096], 100))
    rec["serial_ms"] = timed(lambda: idx.search_device(Q, 100))
    os.environ["MQ_KNN_TAIL_OVERLAP"] = "1"
    D1, I1 = idx.search_device(Q, 100)
    rec["pipelined_ms"] = timed(lambda: idx.search_device(Q, 100))
    rec["bit_identical"] = bool(torch.equal(D0, D1) and torch.equal(I0, I1))
    rec["queries_per_s_serial"] = nq / rec["serial_ms"] * 1e3
    rec["queries_per_s_pipelined"] = nq / rec["pipelined_ms"] * 1e3
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
