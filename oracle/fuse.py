"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's late fusion.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module;
the product (``viquae_amd.ir.fuse``) runs the HIP kernels in ``viquae_amd/csrc/fuse.hip``.

What is restated, with the lines it follows:

* ``default_minimum``  -- meerqat/ir/fuse.py:129-146.  Union of the retrieved documents per query over all
  runs; every run that has at least one result for that query gets the missing documents at its own minimum
  score for that query.
* ``gzmuv_norm``       -- meerqat/ir/fuse.py:86-126.  ONE mean / standard deviation per run, over all the
  scores of all its queries (``np.mean`` / ``np.std`` of a float64 array, population std), then
  ``(s - mean) / max(std, 1e-9)``.
* ``zmuv_norm``        -- ranx (requirements.txt:15, ``ranx>=0.3.2``; third-party, NOT vendored and NOT
  installed here): the same formula with the moments taken per query.  Restated from ranx's published
  ``normalization/zmuv_norm.py``; parity unpinned.
* ``wsum``             -- ranx ``fusion/wsum.py`` + ``fusion/comb_sum.py``: each run's scores times its weight,
  then per query the sum over the runs that hold the document, accumulated in run order starting from 0.0;
  the fused run is sorted by score, best first.  Restated from the published algorithm; parity unpinned
  (ranx absent).  The order of documents with EQUAL fused scores is not defined by the reference's call
  site; this restatement (and the HIP kernel) break ties by ascending integer document id.
* ``fusion_test``      -- ``Fusion.__init__`` + ``Fusion.test`` (meerqat/ir/fuse.py:158-186,215-236): defmin
  on the raw runs first, then the norm, then the weighted sum -- the only configuration the shipped
  experiments use is ``norm="gzmuv", defmin=true, method="wsum"`` (experiments/ir/viquae/dpr+clip/config.json:38-45).

Pinned: ``default_minimum`` and ``gzmuv_norm`` against the reference's own functions run in the build
container (tools/make_golden_fuse.py -> tests/golden/fuse.json).  ``wsum`` / ``zmuv``: parity unpinned.

Runs are plain ``{q_id: {doc_id: score}}`` dicts, like ``Searcher.runs[index_name]``.
"""
import numpy as np


def default_minimum(runs):
    union = {}
    for run in runs:
        for q_id, results in run.items():
            union.setdefault(q_id, set())
            union[q_id] |= results.keys()
    out = []
    for run in runs:
        new_run = {}
        for q_id, results in run.items():
            results = dict(results)
            if results:
                minimum = min(results.values())
                for d_id in union[q_id]:
                    results.setdefault(d_id, minimum)
            new_run[q_id] = results
        out.append(new_run)
    return out


def gzmuv_norm(run):
    scores = np.array([v for results in run.values() for v in results.values()], dtype=np.float64)
    mean, std = np.mean(scores), np.std(scores)
    den = max(std, 1e-9)
    return {q_id: {d: (s - mean) / den for d, s in results.items()} for q_id, results in run.items()}


def zmuv_norm(run):
    out = {}
    for q_id, results in run.items():
        if not results:
            out[q_id] = {}
            continue
        scores = np.array(list(results.values()), dtype=np.float64)
        mean, std = np.mean(scores), np.std(scores)
        den = max(std, 1e-9)
        out[q_id] = {d: (s - mean) / den for d, s in results.items()}
    return out


NORMS = {None: lambda run: run, "gzmuv": gzmuv_norm, "zmuv": zmuv_norm}


def wsum(runs, weights):
    fused = {}
    for q_id in runs[0]:
        docs = set()
        for run in runs:
            docs |= run.get(q_id, {}).keys()
        acc = {}
        for d in docs:
            s = 0.0
            for run, w in zip(runs, weights):
                results = run.get(q_id, {})
                if d in results:
                    s = s + float(w) * float(results[d])
            acc[d] = s
        fused[q_id] = dict(sorted(acc.items(), key=lambda kv: (-kv[1], int(kv[0]))))
    return fused


def fusion_test(runs, weights, norm="gzmuv", defmin=False):
    if defmin:
        runs = default_minimum(runs)
    runs = [NORMS[norm](run) for run in runs]
    return wsum(runs, weights)
