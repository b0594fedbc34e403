"""Probe (round 5): one DPR forward of 2048 x 100 tokens against the same batch cut into parts whose forwards are enqueued on
separate streams -- do the parts fill each other's gaps (the partial last round of 256 x 256 GEMM tiles, the memory-bound
LayerNorm / attention kernels beside MFMA-bound GEMMs)?   usage: python tools/probes/two_stream_forward.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from tools.bench_encoders import BERT_BASE, random_bert_state
from viquae_amd.encoders import DPRContextEncoder

dev = "cuda"
B, L = 2048, 100
model = DPRContextEncoder.from_state_dict(dict(BERT_BASE), random_bert_state(BERT_BASE, 0)).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
ids = torch.randint(1000, 30000, (B, L), generator=g, device=dev)
mask = torch.ones((B, L), dtype=torch.int64, device=dev)
tt = torch.zeros((B, L), dtype=torch.int64, device=dev)


def fwd(sl):
    return model(input_ids=ids[sl], token_type_ids=tt[sl], attention_mask=mask[sl])["pooler_output"]


def whole():
    return fwd(slice(0, B))


def parts(n_parts, streams):
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    outs = []
    step = B // n_parts
    for i in range(n_parts):
        with torch.cuda.stream(streams[i % len(streams)]):
            outs.append(fwd(slice(i * step, (i + 1) * step)))
    for s in streams:
        cur.wait_stream(s)
    return torch.cat(outs)


def time_it(fn, steps=5, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


ref = whole()
print("whole batch: %.2f ms" % time_it(whole))
for n_parts, n_streams in ((2, 2), (4, 2), (4, 4), (2, 1), (8, 2)):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    out = parts(n_parts, streams)
    torch.cuda.synchronize()
    print("%d parts on %d streams: %.2f ms, identical %s" % (n_parts, n_streams, time_it(lambda: parts(n_parts, streams)), bool(torch.equal(out, ref))))
print("whole batch again: %.2f ms" % time_it(whole))
