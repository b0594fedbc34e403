"""Probe: is forking worker processes (that never touch the GPU) from a process that HAS initialised HIP safe on this pool?
Bounded by the caller's `timeout`."""
import multiprocessing as mp
import mmap
import os
import time

import numpy as np
import torch


def work(args):
    i, n = args
    a = np.frombuffer(BUF, dtype=np.uint8)
    a[i * n:(i + 1) * n] = i + 1
    return os.getpid()


if __name__ == "__main__":
    x = torch.randn(1 << 20, device="cuda")
    print("gpu ok", float(x.sum()))
    BUF = mmap.mmap(-1, 1 << 24)
    t = torch.frombuffer(BUF, dtype=torch.uint8)
    rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel(), 0)
    print("cudaHostRegister rc", rc, "is_pinned", t.is_pinned())
    ctx = mp.get_context("fork")
    t0 = time.time()
    with ctx.Pool(8) as pool:
        pids = pool.map(work, [(i, 1 << 20) for i in range(16)])
    print("pool done", len(set(pids)), "workers", round(time.time() - t0, 2), "s")
    d = t.cuda(non_blocking=True)
    torch.cuda.synchronize()
    print("h2d of the shared buffer ok", int(d[:1 << 20].max()), int(d[15 << 20:].min()))
    y = torch.randn(1 << 20, device="cuda")
    print("gpu still ok", float((x + y).abs().mean()))
