"""Mints tests/golden/fuse.json by running the REFERENCE's own ``default_minimum`` and ``gzmuv_norm``
(meerqat/ir/fuse.py:86-146) in the build container (numba / ranx replaced by inert stand-ins, see
tools/ref_import.py) on small seeded runs.  The expected fused run (``wsum``) is produced by the oracle's
restatement of ranx's published algorithm on top of the reference-normalised runs: ranx is not installed
here, so that last step is "parity unpinned" and the file says so.

    python tools/make_golden_fuse.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import ref_import  # noqa: E402
from oracle import fuse as ofuse  # noqa: E402


def seeded_runs(seed, nq, n_runs, k, n_docs, empty_every=0, f32=True):
    rng = np.random.default_rng(seed)
    runs = []
    for r in range(n_runs):
        run = {}
        for q in range(nq):
            if empty_every and (q + r) % empty_every == 0:
                run[f"q{q}"] = {}
                continue
            kk = int(rng.integers(1, k + 1))
            docs = rng.choice(n_docs, size=kk, replace=False)
            scores = np.sort(rng.standard_normal(kk) * (1 + 3 * r) + 10 * r)[::-1]
            if f32:
                scores = scores.astype(np.float32)
            run[f"q{q}"] = {str(int(d)): float(s) for d, s in zip(docs, scores)}
        runs.append(run)
    return runs


def main():
    ref = ref_import.import_reference_fuse()
    Run = sys.modules["ranx"].Run
    cases = []
    for name, kw, weights in [
        ("two_runs", dict(seed=0, nq=6, n_runs=2, k=12, n_docs=40), [0.5, 0.5]),
        ("four_runs_some_empty", dict(seed=1, nq=9, n_runs=4, k=20, n_docs=50, empty_every=4), [0.3, 0.2, 0.2, 0.2]),
        ("three_runs_f64", dict(seed=2, nq=5, n_runs=3, k=30, n_docs=35, f32=False), [0.6, 0.2, 0.2]),
    ]:
        runs = seeded_runs(**kw)
        # reference: Fusion.__init__ applies default_minimum in place on Run objects (fuse.py:176-177)
        ref_runs = [Run(json.loads(json.dumps(run)), name=f"r{i}") for i, run in enumerate(runs)]
        ref_defmin = ref.default_minimum(ref_runs)
        defmin = [json.loads(json.dumps(r.run)) for r in ref_defmin]
        gz = [dict(ref.gzmuv_norm(r).run) for r in ref_defmin]
        gz = [{q: dict(res) for q, res in run.items()} for run in gz]
        gz_nodefmin = [{q: dict(res) for q, res in ref.gzmuv_norm(Run(json.loads(json.dumps(run)))).run.items()}
                       for run in runs]
        # the oracle restatement must agree with the reference on both steps, exactly
        o_defmin = ofuse.default_minimum(runs)
        assert o_defmin == defmin, name
        o_gz = [ofuse.gzmuv_norm(r) for r in o_defmin]
        assert o_gz == gz, name
        assert [ofuse.gzmuv_norm(r) for r in runs] == gz_nodefmin, name
        cases.append({
            "name": name, "weights": weights, "runs": runs,
            "reference_default_minimum": defmin,
            "reference_gzmuv_after_defmin": gz,
            "reference_gzmuv_no_defmin": gz_nodefmin,
            "unpinned_wsum_gzmuv_defmin": ofuse.wsum(gz, weights),
            "unpinned_wsum_gzmuv_nodefmin": ofuse.wsum(gz_nodefmin, weights),
        })
    out = os.path.join(ROOT, "tests", "golden", "fuse.json")
    with open(out, "wt") as file:
        json.dump({"note": "default_minimum / gzmuv: outputs of the reference's own functions; wsum: oracle "
                           "restatement of ranx (absent) on those -- parity unpinned", "cases": cases}, file)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
