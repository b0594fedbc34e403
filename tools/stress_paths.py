"""Randomised checks of the round-2 paths on the GPU:
  * FAISS's small-batch L2 form (csrc/knn_direct.inc) against the CPU oracle's direct form, bit for bit;
  * row-sharded search (LocalShardsFlatIndex: shard records + merge) against one unsharded index, bit for bit, with random
    shard counts, both metrics, "L2norm,Flat", tiny KBs with empty shards, ragged adds.
usage: python tools/stress_paths.py [n_cases] [first_seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import knn as ok
from viquae_amd.index import MI355XFlatIndex
from viquae_amd.sharded import LocalShardsFlatIndex


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    t0 = time.time()
    for seed in range(first, first + n_cases):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([37, 64, 1000, 5000, 20000, 60000]))
        d = int(rng.choice([8, 30, 64, 100, 256, 768]))
        k = int(rng.choice([1, 10, 100, 128]))
        kind = rng.choice(["normal", "ties", "dups", "scaled"])
        X = rng.standard_normal((n, d), dtype=np.float32)
        if kind == "ties":
            X = rng.integers(-2, 3, (n, d)).astype(np.float32)
        elif kind == "dups":
            X[rng.integers(0, n, n // 2)] = X[rng.integers(0, n, n // 2)]
        elif kind == "scaled":
            X *= np.exp(rng.standard_normal((n, 1))).astype(np.float32)
        # 1. small-batch L2 against the oracle
        nq = int(rng.integers(1, 20))
        Q = rng.standard_normal((nq, d), dtype=np.float32) if kind != "ties" else rng.integers(-2, 3, (nq, d)).astype(np.float32)
        Q[0] = X[n // 2]
        factory = str(rng.choice(["Flat", "L2norm,Flat"])) if kind != "ties" else "Flat"
        idx = MI355XFlatIndex(string_factory=factory, metric_type=1, screen=bool(rng.integers(0, 2)))
        at = 0
        while at < n:                                   # ragged adds
            step = int(rng.integers(1, max(2, n // 3)))
            idx.add(X[at:at + step])
            at += step
        D, I = idx.search_batch(Q, k)
        Do, Io = ok.knn(X, Q, k, metric=1, l2norm=factory != "Flat", l2_form="direct")
        ok1 = np.array_equal(I, Io) and np.array_equal(D, Do)
        # 2. sharded against unsharded (20..300 queries, both metrics)
        metric = int(rng.integers(0, 2))
        nq2 = int(rng.choice([20, 100, 300]))
        Q2 = rng.standard_normal((nq2, d), dtype=np.float32) if kind != "ties" else rng.integers(-2, 3, (nq2, d)).astype(np.float32)
        shards = int(rng.integers(2, 9))
        sh = LocalShardsFlatIndex([0] * shards, string_factory=factory, metric_type=metric, allow_repeated_devices=True)
        sh.add_vectors(X)
        one = MI355XFlatIndex(string_factory=factory, metric_type=metric)
        one.add(X)
        a, b = sh.search_batch(Q2, k), one.search_batch(Q2, k)
        ok2 = np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
        print(f"seed {seed:4d} {kind:7s} N={n:6d} d={d:3d} k={k:3d} {factory:12s} direct-L2 nq={nq:2d} {'ok ' if ok1 else 'MISMATCH'} | "
              f"{shards} shards metric {metric} nq={nq2:3d} {'ok ' if ok2 else 'MISMATCH'}", flush=True)
        bad += (not ok1) + (not ok2)
        del idx, sh, one
    print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
