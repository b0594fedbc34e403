"""Rank metrics of search runs on the MI355X: the report ``dataset_search`` ends with (meerqat/ir/search.py:397,500-512,
``ranx.compare(qrels, runs, metrics=["mrr@100", "precision@1", "precision@20", "hit_rate@20"])`` -> ``metrics.json`` /
``metrics.tex``) and the objective of ``Fusion.fit`` (meerqat/ir/fuse.py:193-217), without ranx.

``ranx`` (requirements.txt:15) is third-party and not vendored; ``mrr`` / ``precision`` / ``hit_rate`` / ``recall`` with a
cut ``@k`` are restated from its published definitions in ``csrc/fuse.hip`` (``mq_run_metrics_f64``) -- **parity unpinned
vs ranx**.  What this report does NOT hold: ranx's pairwise statistical tests (``comparisons`` / ``win_tie_loss``; Fisher's
randomisation test draws random permutations) -- ``metrics.json`` carries the scores only and says so.

Runs go to the device as ``[nq, K]`` id tables (``ArrayRun.tables`` when the run is still arrays), qrels as a CSR table
of the relevant ids of every query.  No CPU fallback: without the library and a GPU the call raises."""
import ctypes
import json

import numpy as np
import torch

from .. import _lib
from .runs import ArrayRun

_EMPTY = np.zeros(0, dtype=np.int64)
METRIC_CODES = {"mrr": 0, "precision": 1, "hit_rate": 2, "recall": 3}   # MQ_RANK_METRIC_*
MAX_METRICS = 16
DEFAULT_METRICS = ["mrr@100", "precision@1", "precision@20", "hit_rate@20"]


def parse_metric(name):
    """'mrr@100' -> (code, 100); 'precision' -> (code, 0): the whole run."""
    base, _, k = str(name).partition("@")
    if base not in METRIC_CODES:
        raise NotImplementedError(f"metric '{name}': only {sorted(METRIC_CODES)} (with an optional @k) run on the device")
    k = int(k) if k else 0
    if k < 0:
        raise ValueError(f"metric '{name}': negative cut")
    return METRIC_CODES[base], k


def qrels_to_csr(qrels, q_ids, doc_lookup=None):
    """``{q_id: {doc: judgement}}`` -> (rel_ptr int64 [nq + 1], rel_ids int64) of the documents judged >= 1, ascending inside a
    query.  Documents are decimal strings (the reference's ``str(i)``) or go through ``doc_lookup`` (name -> integer); a
    relevant document that no run can hold (no integer id) still counts in ``recall``'s denominator: it gets an id no run uses."""
    ptr = np.zeros(len(q_ids) + 1, dtype=np.int64)
    rows = []
    for n, q in enumerate(q_ids):
        rel = []
        unknown = 0
        judged = qrels.get(q)
        if not judged:
            rows.append(_EMPTY)
            ptr[n + 1] = ptr[n]
            continue
        for d, r in judged.items():
            if r < 1:
                continue
            if doc_lookup is not None:
                i = doc_lookup.get(d)
            else:
                i = int(d) if isinstance(d, (int, np.integer)) or (isinstance(d, str) and d.isdecimal()) else None
            if i is None:
                unknown += 1
                i = (1 << 62) + unknown
            rel.append(i)
        rel.sort()
        rows.append(np.asarray(rel, dtype=np.int64))
        ptr[n + 1] = ptr[n] + len(rel)
    ids = np.concatenate(rows) if rows else np.zeros(0, np.int64)
    return ptr, np.ascontiguousarray(ids, dtype=np.int64)


def run_tables(run, q_ids=None):
    """(q_ids, ids int64 [nq, K], doc_lookup | None) of a run given as ``ArrayRun`` or ``{q: {doc: score}}`` (best first)."""
    if isinstance(run, ArrayRun):
        tabs = run.tables(q_ids, ids_only=True)
        if tabs is not None:
            return tabs[0], tabs[1], None
    if q_ids is None:
        q_ids = list(run.keys())
    rows = [list(run[q].keys()) for q in q_ids]
    numeric = all(isinstance(d, str) and d.isdecimal() for row in rows for d in row)
    lookup = None
    if not numeric:
        names = sorted({d for row in rows for d in row}, key=str)
        lookup = {d: i for i, d in enumerate(names)}
    K = max([1] + [len(row) for row in rows])
    ids = np.full((len(rows), K), -1, dtype=np.int64)
    for n, row in enumerate(rows):
        if row:
            ids[n, :len(row)] = [int(d) for d in row] if lookup is None else [lookup[d] for d in row]
    return q_ids, ids, lookup


def evaluate(qrels, run, metrics=None, device="cuda:0", return_per_query=False):
    """``ranx.evaluate(qrels, run, metrics)``: ``{metric: mean over the run's queries}`` (a float for a single metric name)."""
    single = isinstance(metrics, str)
    names = [metrics] if single else list(DEFAULT_METRICS if metrics is None else metrics)
    if len(names) > MAX_METRICS:
        raise ValueError(f"at most {MAX_METRICS} metrics per call")
    parsed = [parse_metric(m) for m in names]
    lib = _lib.load()
    _lib.require_gpu()
    q_ids, ids, lookup = run_tables(run)
    nq = len(q_ids)
    if nq == 0:
        means, per_query = np.zeros(len(names)), np.zeros((len(names), 0))
    else:
        ptr, rel = qrels_to_csr(qrels, q_ids, lookup)
        dev = torch.device(device)
        ids_d = torch.from_numpy(ids).to(dev)
        ptr_d = torch.from_numpy(ptr).to(dev)
        rel_d = torch.from_numpy(rel if rel.size else np.zeros(1, np.int64)).to(dev)
        per_q = torch.empty((len(names), nq), dtype=torch.float64, device=dev)
        mean = torch.empty((len(names),), dtype=torch.float64, device=dev)
        codes = (ctypes.c_int * len(names))(*[c for c, _ in parsed])
        ks = (ctypes.c_int * len(names))(*[k for _, k in parsed])
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            _lib.check(lib.mq_run_metrics_f64(ids_d.data_ptr(), nq, ids.shape[1], ptr_d.data_ptr(), rel_d.data_ptr(), len(names),
                                              codes, ks, per_q.data_ptr(), mean.data_ptr(), stream), "mq_run_metrics_f64")
        means, per_query = mean.cpu().numpy(), per_q.cpu().numpy()
    out = {m: float(v) for m, v in zip(names, means)}
    if single:
        out = out[names[0]]
    if return_per_query:
        return out, {m: per_query[i] for i, m in enumerate(names)}
    return out


class Report:
    """What ``ranx.compare`` hands ``dataset_search``: printable, ``save(path)`` -> metrics.json, ``to_latex()``.  Scores only:
    no statistical tests (module docstring)."""

    def __init__(self, model_names, metrics, results):
        self.model_names, self.metrics, self.results = list(model_names), list(metrics), results

    def to_dict(self):
        d = {"stat_test": None, "metrics": self.metrics, "model_names": self.model_names}
        for m in self.model_names:
            d[m] = {"scores": self.results[m], "comparisons": {}, "win_tie_loss": {}}
        return d

    def save(self, path):
        with open(path, "wt") as file:
            json.dump(self.to_dict(), file, indent=4)

    def to_table(self):
        head = ["#", "Model"] + self.metrics
        rows = [[chr(ord("a") + i) if i < 26 else str(i), m] + [f"{self.results[m][x]:.3f}" for x in self.metrics]
                for i, m in enumerate(self.model_names)]
        widths = [max(len(str(r[c])) for r in [head] + rows) for c in range(len(head))]
        line = lambda r: "  ".join(str(v).ljust(w) for v, w in zip(r, widths))
        return "\n".join([line(head), line(["-" * w for w in widths])] + [line(r) for r in rows])

    __str__ = to_table

    def to_latex(self):
        cols = "c|l" + "|c" * len(self.metrics)
        esc = lambda s: str(s).replace("_", r"\_")
        lines = [r"\begin{table*}[ht]", r"\centering", r"\caption{Overall effectiveness of the models.}",
                 r"\begin{tabular}{" + cols + "}", r"\toprule",
                 " & ".join([r"\textbf{\#}", r"\textbf{Model}"] + [r"\textbf{" + esc(m) + "}" for m in self.metrics]) + r" \\",
                 r"\midrule"]
        for i, m in enumerate(self.model_names):
            lines.append(" & ".join([chr(ord("a") + i) if i < 26 else str(i), esc(m)]
                                    + [f"{self.results[m][x]:.3f}" for x in self.metrics]) + r" \\")
        lines += [r"\bottomrule", r"\end{tabular}", r"\label{tab:results}", r"\end{table*}"]
        return "\n".join(lines)


def compare(qrels, runs, metrics=None, device="cuda:0", **ignored):
    """``ranx.compare(qrels, runs=..., metrics=...)`` for the scores.  ``runs``: ``{name: run}`` or a list of (name, run);
    extra ranx arguments (``stat_test``, ``max_p``, ...) are accepted and ignored: no tests are run."""
    metrics = list(DEFAULT_METRICS if metrics is None else ([metrics] if isinstance(metrics, str) else metrics))
    items = list(runs.items()) if hasattr(runs, "items") else list(runs)
    results = {name: evaluate(qrels, run, metrics, device=device) for name, run in items}
    return Report([name for name, _ in items], metrics, results)
