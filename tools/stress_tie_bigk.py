"""Randomised checks of the round-3 boundary parameters on the GPU, bit for bit against oracle/knn_oracle.c: tie order
(id_asc / id_desc) and k anywhere in 1 .. 2048 on every search path -- exact fp32 scan, screened search (with its fallback and
FAISS's small-batch L2 form), row shards + record merge -- over random shapes (d not a multiple of 16, ragged N and nq, k near
and above N), free-form / tie-heavy / duplicated / mixed-scale data, both metrics, "L2norm,Flat".
usage: python tools/stress_tie_bigk.py [n_cases] [first_seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import knn as ok
from viquae_amd.index import MI355XFlatIndex
from viquae_amd.sharded import LocalShardsFlatIndex


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad, t0 = 0, time.time()
    for seed in range(first, first + n_cases):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([5, 64, 300, 1000, 5000, 20000]))
        d = int(rng.choice([3, 16, 30, 64, 100, 257]))
        nq = int(rng.choice([1, 7, 19, 20, 33, 257, 600]))
        k = int(rng.choice([1, 100, 128, 129, 200, 256, 300, 700, 1024, 2048]))
        metric = int(rng.integers(0, 2))
        tie = str(rng.choice(["id_asc", "id_desc"]))
        factory = "L2norm,Flat" if rng.random() < 0.2 else "Flat"
        kind = str(rng.choice(["normal", "ties", "dups", "scaled"]))
        if kind == "ties" and factory == "Flat":
            X = rng.integers(-2, 3, (n, d)).astype(np.float32)
            Q = rng.integers(-2, 3, (nq, d)).astype(np.float32)
        else:
            X = rng.standard_normal((n, d), dtype=np.float32)
            Q = rng.standard_normal((nq, d), dtype=np.float32)
            if kind == "dups":
                X[rng.integers(0, n, n // 2)] = X[rng.integers(0, n, n // 2)]
            elif kind == "scaled":
                X *= np.exp(rng.standard_normal((n, 1))).astype(np.float32)
        want = ok.knn(X, Q, k, metric=metric, l2norm=factory != "Flat", tie_order=tie)
        paths = {}
        for screen in (False, True):
            idx = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=screen, tie_order=tie)
            idx.add_vectors(X)
            paths["screened" if screen else "exact"] = idx.search_batch(Q, k)
        ns = int(rng.integers(2, 6))
        sh = LocalShardsFlatIndex([0] * ns, string_factory=factory, metric_type=metric, allow_repeated_devices=True, tie_order=tie)
        sh.add_vectors(X)
        paths[f"{ns} shards"] = sh.search_batch(Q, k)
        for name, (D, I) in paths.items():
            if not (np.array_equal(I, want[1]) and np.array_equal(D, want[0], equal_nan=True)):
                bad += 1
                print(f"MISMATCH seed {seed} {name}: n={n} d={d} nq={nq} k={k} metric={metric} tie={tie} {factory} {kind}", flush=True)
    print(f"{n_cases} cases x 3 paths, {bad} mismatches, {time.time() - t0:.0f} s")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
