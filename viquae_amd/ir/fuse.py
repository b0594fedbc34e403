"""Mirror of ``meerqat.ir.fuse`` for the configuration the shipped experiments use:
``Fusion(qrels, runs, norm="gzmuv", method="wsum", output, defmin=True).test(best_params={"weights": [...]})``
(meerqat/ir/fuse.py:158-186,215-236; experiments/ir/viquae/dpr+clip/config.json:38-45), called by
``dataset_search`` (meerqat/ir/search.py:514-524).

The arithmetic -- ``default_minimum`` (:129-146), ``gzmuv_norm`` (:86-126), ranx's ``zmuv`` norm and ``wsum``
fusion -- runs on the MI355X (``viquae_amd/csrc/fuse.hip`` through ``mq_fuse_wsum_f64``): runs are laid out as
padded ``[n_runs, nq, K]`` tables of (integer document id, f64 score), one workgroup fuses one query.  There is
no CPU fallback: without the HIP library and a GPU the call raises.

``Fusion.fit`` (:193-217, ranx's ``optimize_fusion`` for ``wsum``: a grid of weights in steps of 0.1, each trial
fused and scored by a rank metric) runs on the device too, all trials in one launch (``mq_fuse_fit_wsum_f64``); its trial
set and tie rule restate ranx's published code -- parity unpinned vs ranx, which is not installed.

What differs from the reference: ``method`` other than ``"wsum"`` and ranx norms other
than ``None`` / ``"zmuv"`` are not implemented; documents with EQUAL fused scores are ordered by ascending
document id (the reference leaves that order to ranx's sort).  Results are plain ``{q_id: {doc_id: score}}``
dicts, best first (wrapped in ``ranx.Run`` when ranx is importable).
"""
import ctypes
import json
from pathlib import Path

import numpy as np
import torch

from .. import _lib
from .runs import ArrayRun, dump_run

NORM_CODES = {None: 0, "gzmuv": 1, "zmuv": 2}
MAX_ENTRIES = 4096  # n_runs * K limit of the kernel (include/meerqat_hip.h)


def fuse_tables(ids, scores, weights, norm="gzmuv", defmin=False):
    """Device-resident fusion.  ``ids`` int64 / ``scores`` f64 ``[n_runs, nq, K]`` on a GPU (-1 = empty slot) ->
    ``(fused_ids [nq, n_runs*K], fused_scores, counts [nq] int32)``, best first."""
    if norm not in NORM_CODES:
        raise NotImplementedError(f"norm '{norm}': only None, 'gzmuv' and 'zmuv' run on the device")
    lib = _lib.load()
    _lib.require_gpu()
    if ids.dim() != 3 or ids.shape != scores.shape:
        raise ValueError("ids and scores must both be [n_runs, nq, K]")
    if not ids.is_cuda or not scores.is_cuda:
        raise ValueError("fuse_tables works on device tensors")
    n_runs, nq, K = ids.shape
    if len(weights) != n_runs:
        raise ValueError(f"{len(weights)} weights for {n_runs} runs")
    ids = ids.to(torch.int64).contiguous()
    scores = scores.to(torch.float64).contiguous()
    dev = ids.device
    out_ids = torch.empty((nq, n_runs * K), dtype=torch.int64, device=dev)
    out_scores = torch.empty((nq, n_runs * K), dtype=torch.float64, device=dev)
    counts = torch.empty((nq,), dtype=torch.int32, device=dev)
    if nq == 0:
        return out_ids, out_scores, counts
    ws_bytes = lib.mq_fuse_workspace_bytes(n_runs, nq, K)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    w = (ctypes.c_double * n_runs)(*[float(x) for x in weights])
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.mq_fuse_wsum_f64(ids.data_ptr(), scores.data_ptr(), n_runs, nq, K, w, NORM_CODES[norm],
                                        int(bool(defmin)), out_ids.data_ptr(), out_scores.data_ptr(),
                                        counts.data_ptr(), ws.data_ptr(), ws_bytes, stream), "mq_fuse_wsum_f64")
    return out_ids, out_scores, counts


def _as_dict(run):
    if isinstance(run, ArrayRun):
        return run   # a mapping already; its rows go to the device as arrays (runs_to_tables)
    if isinstance(run, (str, Path)):
        with open(run, "rt") as file:
            return json.load(file)
    if hasattr(run, "to_dict"):
        return run.to_dict()
    if hasattr(run, "run"):
        return {q: dict(results) for q, results in run.run.items()}
    return run


def runs_to_tables(runs, device="cuda:0"):
    """``[{q_id: {doc_id: score}}]`` -> ``(q_ids, doc_names, ids, scores)`` with ids / scores ``[n_runs, nq, K]`` on
    ``device``.  Document ids that are all decimal strings keep their integer value; otherwise they are numbered
    in sorted order and ``doc_names`` holds the inverse."""
    q_ids = list(runs[0].keys())
    for run in runs[1:]:
        if run.keys() != runs[0].keys():
            raise ValueError("all runs must hold the same queries (Searcher fills every index for every question)")
    if all(isinstance(run, ArrayRun) for run in runs):
        # runs kept as arrays since the search (viquae_amd/ir/runs.py): no dict is built on the way to the device
        tabs = [run.tables(q_ids) for run in runs]
        if all(t is not None for t in tabs):
            K = max([1] + [t[1].shape[1] for t in tabs])
            if len(runs) * K > MAX_ENTRIES:
                raise ValueError(f"{len(runs)} runs x {K} results per query exceed the kernel's {MAX_ENTRIES} entries")
            ids = np.full((len(runs), len(q_ids), K), -1, dtype=np.int64)
            scores = np.zeros((len(runs), len(q_ids), K), dtype=np.float64)
            for r, (_, ti, ts) in enumerate(tabs):
                ids[r, :, :ti.shape[1]] = ti
                scores[r, :, :ti.shape[1]] = ts
            return q_ids, None, torch.from_numpy(ids).to(device), torch.from_numpy(scores).to(device)
    docs = set()
    for run in runs:
        for results in run.values():
            docs |= results.keys()
    numeric = all(isinstance(d, str) and d.isdecimal() for d in docs)
    doc_names = None
    if numeric:
        to_int = int
    else:
        doc_names = sorted(docs, key=str)
        lookup = {d: i for i, d in enumerate(doc_names)}
        to_int = lookup.__getitem__
    K = max([1] + [len(results) for run in runs for results in run.values()])
    if len(runs) * K > MAX_ENTRIES:
        raise ValueError(f"{len(runs)} runs x {K} results per query exceed the kernel's {MAX_ENTRIES} entries")
    ids = np.full((len(runs), len(q_ids), K), -1, dtype=np.int64)
    scores = np.zeros((len(runs), len(q_ids), K), dtype=np.float64)
    for r, run in enumerate(runs):
        for q, q_id in enumerate(q_ids):
            results = run[q_id]
            n = len(results)
            if n:
                ids[r, q, :n] = [to_int(d) for d in results.keys()]
                scores[r, q, :n] = list(results.values())
    return q_ids, doc_names, torch.from_numpy(ids).to(device), torch.from_numpy(scores).to(device)


def tables_to_run(q_ids, doc_names, fused_ids, fused_scores, counts, as_arrays=False):
    fused_ids, fused_scores, counts = fused_ids.cpu().numpy(), fused_scores.cpu().numpy(), counts.cpu().numpy()
    if as_arrays and len(set(q_ids)) == len(q_ids):
        # the fused run stays rows of the fused tables (an ArrayRun: dicts on demand, run file straight from the arrays)
        ids = np.where(np.arange(fused_ids.shape[1])[None, :] < counts[:, None], fused_ids, -1)
        run = ArrayRun()
        run.add_block(list(q_ids), ids, fused_scores, doc_names)
        return run
    run = {}
    for q, q_id in enumerate(q_ids):
        n = int(counts[q])
        names = fused_ids[q, :n].tolist()
        names = [str(i) for i in names] if doc_names is None else [doc_names[i] for i in names]
        run[q_id] = dict(zip(names, fused_scores[q, :n].tolist()))
    return run


def fuse_runs(runs, weights, norm="gzmuv", defmin=False, device="cuda:0"):
    """dict runs in, fused dict run out; the arithmetic happens in ``fuse_tables``.  Runs that are still arrays
    (``ArrayRun``) go in as arrays and the fused run comes back as one."""
    runs = [_as_dict(run) for run in runs]
    q_ids, doc_names, ids, scores = runs_to_tables(runs, device)
    as_arrays = all(isinstance(run, ArrayRun) for run in runs)
    return tables_to_run(q_ids, doc_names, *fuse_tables(ids, scores, weights, norm=norm, defmin=defmin), as_arrays=as_arrays)


def wsum_trials(n_runs, step=0.1):
    """ranx's trial set for ``wsum`` (``fusion/wsum.py`` + ``fusion/common.py``, restated as published -- parity unpinned vs
    ranx, which is not installed): candidate weights ``round(x, 2)`` for x in ``np.arange(0, 1 + step, step)``, a trial is every
    tuple of ``itertools.product`` whose Python ``sum`` is EXACTLY 1.0 (so tuples whose floating-point sum misses 1.0 -- 4 of 66
    for three runs, 30 of 286 for four -- are not tried, as there)."""
    import itertools
    weights = [round(float(x), 2) for x in np.arange(0, 1 + step, step)]
    return [seq for seq in itertools.product(*[weights] * n_runs) if sum(seq) == 1.0]


def fit_tables(ids, scores, trials, rel_ptr, rel_ids, metric_code, metric_k, norm="gzmuv", defmin=False):
    """Every trial fused and scored on the device.  ``ids`` / ``scores`` [n_runs, nq, K] as for ``fuse_tables``, ``trials`` f64
    [T, n_runs], qrels as CSR (``viquae_amd.ir.metrics.qrels_to_csr``) -> (mean f64 [T], per-query f64 [T, nq]), on the device."""
    if norm not in NORM_CODES:
        raise NotImplementedError(f"norm '{norm}': only None, 'gzmuv' and 'zmuv' run on the device")
    lib = _lib.load()
    _lib.require_gpu()
    if ids.dim() != 3 or ids.shape != scores.shape:
        raise ValueError("ids and scores must both be [n_runs, nq, K]")
    n_runs, nq, K = ids.shape
    if trials.dim() != 2 or trials.shape[1] != n_runs:
        raise ValueError(f"trials must be [T, {n_runs}]")
    dev = ids.device
    ids = ids.to(torch.int64).contiguous()
    scores = scores.to(torch.float64).contiguous()
    trials = trials.to(device=dev, dtype=torch.float64).contiguous()
    T = trials.shape[0]
    per_q = torch.empty((T, nq), dtype=torch.float64, device=dev)
    mean = torch.empty((T,), dtype=torch.float64, device=dev)
    if nq == 0 or T == 0:
        return mean.zero_(), per_q
    ws_bytes = lib.mq_fuse_workspace_bytes(n_runs, nq, K)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.mq_fuse_fit_wsum_f64(ids.data_ptr(), scores.data_ptr(), n_runs, nq, K, trials.data_ptr(), T,
                                            NORM_CODES[norm], int(bool(defmin)), rel_ptr.data_ptr(), rel_ids.data_ptr(),
                                            int(metric_code), int(metric_k), per_q.data_ptr(), mean.data_ptr(), ws.data_ptr(),
                                            ws_bytes, stream), "mq_fuse_fit_wsum_f64")
    return mean, per_q


def fit_wsum(runs, qrels, norm="gzmuv", defmin=False, metric="mrr@100", step=0.1, device="cuda:0"):
    """-> (``{"weights": [...]}``, [(weights, score)] in trial order): the first trial with the best score wins."""
    from .metrics import parse_metric, qrels_to_csr
    code, k = parse_metric(metric)
    runs = [_as_dict(run) for run in runs]
    q_ids, doc_names, ids, scores = runs_to_tables(runs, device)
    lookup = None if doc_names is None else {d: i for i, d in enumerate(doc_names)}
    ptr, rel = qrels_to_csr(qrels, q_ids, lookup)
    trials = wsum_trials(len(runs), step)
    if not trials:
        raise ValueError(f"no trial for {len(runs)} runs at step {step}")
    dev = torch.device(device)
    mean, _ = fit_tables(ids, scores, torch.tensor(trials, dtype=torch.float64), torch.from_numpy(ptr).to(dev),
                         torch.from_numpy(rel if rel.size else np.zeros(1, np.int64)).to(dev), code, k, norm=norm, defmin=defmin)
    mean = mean.cpu().numpy()
    best = int(np.argmax(mean))   # np.argmax: the first of equal maxima
    return {"weights": list(trials[best])}, [(list(w), float(s)) for w, s in zip(trials, mean)]


class Fusion:
    """Same constructor as the reference's (meerqat/ir/fuse.py:158-186).  ``runs``: dicts, ranx ``Run`` objects or
    paths to JSON runs; ``qrels`` (a dict, a ranx ``Qrels`` or a path) is used by ``fit`` and by the metric line of ``test``."""

    def __init__(self, qrels=None, runs=None, norm="zmuv", method="wsum", output=None, defmin=False, device="cuda:0"):
        self.qrels = qrels
        self.runs = [_as_dict(run) for run in runs]
        self.norm = norm
        self.method = method
        self.defmin = defmin
        self.device = device
        if output is not None:
            output = Path(output)
            output.mkdir(exist_ok=True)
        self.output = output

    def fit(self, metric="mrr@100", step=0.1):
        """Finds the best fusion weights (meerqat/ir/fuse.py:193-217 -> ``ranx.optimize_fusion(method="wsum")``): every trial
        of ranx's grid is fused and scored on the device in one launch (``mq_fuse_fit_wsum_f64``); the first trial that reaches
        the best ``metric`` wins and is written to ``{norm}_{method}_best_params.yaml`` like the reference does.  Returns
        ``{(norm, method): (best_params, report)}`` (the reference returns None and prints)."""
        import yaml
        from .metrics import parse_metric
        norms = [self.norm] if self.norm is None or isinstance(self.norm, str) else self.norm
        methods = [self.method] if self.method is None or isinstance(self.method, str) else self.method
        if self.qrels is None:
            raise ValueError("Fusion.fit needs qrels")
        out = {}
        for norm in norms:
            for method in methods:
                if method != "wsum":
                    raise NotImplementedError(f"method '{method}': only 'wsum' (the shipped configs' method) is implemented")
                best_params, report = fit_wsum(self.runs, _as_dict(self.qrels), norm=norm, defmin=self.defmin,
                                               metric=metric, step=step, device=self.device)
                lines = "\n".join(f"  {tuple(w)}: {s:.6f}" for w, s in report)
                print(f"Norm: {norm}, Method: {method}. Best parameters: {best_params}.\n{metric} of every trial:\n{lines}")
                if self.output is not None:
                    with open(self.output / f"{norm}_{method}_best_params.yaml", "wt") as file:
                        yaml.dump(json.loads(json.dumps(best_params)), file)
                out[(norm, method)] = (best_params, report)
        self.best = out
        return out

    def test(self, best_params, metrics=None):
        if self.method != "wsum":
            raise NotImplementedError(f"method '{self.method}': only 'wsum' (the shipped configs' method) is implemented")
        weights = best_params["weights"]
        fused = fuse_runs(self.runs, weights, norm=self.norm, defmin=self.defmin, device=self.device)
        if self.output is not None:
            dump_run(fused, self.output / "test_run.json")
        if self.qrels is not None:
            from .metrics import evaluate
            if metrics is None:
                metrics = ["mrr@100", "precision@1", "precision@20", "hit_rate@20"]
            self.test_scores = evaluate(_as_dict(self.qrels), fused, metrics, device=self.device)
            print(" & ".join(f"{m}: {v:.3f}" for m, v in self.test_scores.items()))
        try:
            import ranx
        except ImportError:
            return fused
        return ranx.Run(fused.to_dict() if isinstance(fused, ArrayRun) else fused, name="fusion")
