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


# ---------------------------------------------------------------------------------------------------------------
# Rank metrics and the weight search (round 6).  ranx (requirements.txt:15) is NOT installed here and not vendored:
# what follows restates its PUBLISHED definitions -- parity unpinned vs ranx.
#
# * ``rank_metric``   -- ranx ``metrics/{reciprocal_rank,precision,hit_rate,recall}.py``: the run is taken best first
#   (here: in the order given, which is how every run of this build is stored), the cut is ``k`` (``k = 0``: the
#   length of the query's run), a document is relevant when its judgement is >= 1; ``precision`` divides by the
#   cut itself, ``recall`` by the number of relevant documents of the query (0 when it has none); the job-level
#   figure is ``np.mean`` over the queries of the qrels (``ranx.evaluate``; called at meerqat/ir/search.py:500-504
#   and meerqat/ir/fuse.py:233-234 with ["mrr@100", "precision@1", "precision@20", "hit_rate@20"]).
# * ``wsum_trials``   -- ranx ``fusion/wsum.py`` + ``fusion/common.py``: the candidate weights are
#   ``[round(x, 2) for x in np.arange(0, 1 + step, step)]`` with step 0.1 and a trial is every tuple of
#   ``itertools.product`` whose Python ``sum`` equals 1.0 EXACTLY -- so the tuples whose floating-point sum misses
#   1.0 (4 of the 66 for three runs, 30 of 286 for four) are not tried; restated as published, quirk included.
# * ``fusion_fit``    -- ``Fusion.fit`` (meerqat/ir/fuse.py:193-217): default-minimum on the raw runs (done by
#   ``Fusion.__init__``, :176-177), the custom norm as a preprocessing (:198-201), then ``optimize_fusion``: every
#   trial is fused with ``wsum`` and scored with ``metric``; the FIRST trial that reaches the best score wins.
# ---------------------------------------------------------------------------------------------------------------
import itertools


def parse_metric(name):
    base, _, k = name.partition("@")
    return base, int(k) if k else 0


def rank_metric(run, qrels, name):
    """np.mean over the queries of ``run`` of one metric; returns (mean, per-query list)."""
    base, k = parse_metric(name)
    values = []
    for q_id, results in run.items():
        docs = list(results)
        relevant = {d for d, r in qrels.get(q_id, {}).items() if r >= 1}
        cut = len(docs) if k == 0 else k
        top = docs[:cut]
        flags = [d in relevant for d in top]
        if cut == 0:
            v = 0.0
        elif base == "mrr":
            v = 1.0 / (flags.index(True) + 1) if any(flags) else 0.0
        elif base == "precision":
            v = sum(flags) / cut
        elif base == "hit_rate":
            v = 1.0 if any(flags) else 0.0
        elif base == "recall":
            v = sum(flags) / len(relevant) if relevant else 0.0
        else:
            raise ValueError(f"metric '{name}' is not restated")
        values.append(v)
    return (float(np.mean(np.array(values, dtype=np.float64))) if values else 0.0), values


def wsum_trials(n_runs, step=0.1):
    weights = [round(float(x), 2) for x in np.arange(0, 1 + step, step)]
    return [seq for seq in itertools.product(*[weights] * n_runs) if sum(seq) == 1.0]


def fusion_fit(runs, qrels, norm="gzmuv", defmin=False, metric="mrr@100", step=0.1):
    """-> (best_params, [(weights, score)] in trial order)."""
    if defmin:
        runs = default_minimum(runs)
    runs = [NORMS[norm](run) for run in runs]
    report = []
    best, best_score = None, None
    for weights in wsum_trials(len(runs), step):
        score, _ = rank_metric(wsum(runs, weights), qrels, metric)
        report.append((weights, score))
        if best_score is None or score > best_score:
            best, best_score = weights, score
    return {"weights": list(best)}, report
