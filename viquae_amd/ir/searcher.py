"""Mirror of ``meerqat.ir.search.Searcher`` / ``dataset_search`` for dense (FAISS-kind) indexes
(SURVEY.md section 8 f.1): ``Searcher.__init__`` (meerqat/ir/search.py:332-399), ``Searcher.__call__``
(:401-459), ``dataset_search`` (:462-524).

What is the same: constructor arguments, the ``runs[index_name][q_id][doc_id] = score`` layout (doc ids are
strings), the article->passage fan-out through ``index_mapping`` with the 1e-8 penalty per passage
(:421-427) or ``many2one: "max"`` (:429-431), the cut at ``k`` entries per query (:439-440), the on-the-fly
relevance judgement of retrieved passages against the reference KB (:442-457, ``find_relevant`` =
meerqat/ir/metrics.py:79-124 for string answers).
What differs: no Elasticsearch client is constructed (sparse kinds are outside this build); ``ranx`` is
not needed -- the metric report (``metrics.json`` / ``metrics.tex``, scores without ranx's statistical tests) is computed from
the result arrays on the device (``viquae_amd/ir/metrics.py``), and so is ``Fusion.fit``; numerical
(InfoSeek) question types are not judged here.  With several indexes the runs are fused like the reference
does (:514-524) through ``viquae_amd.ir.fuse.Fusion`` (HIP kernels); the fused run is ``searcher.fusion``.

The reference builds the run with a Python triple loop per batch (256 x 100 dict inserts).  Here a batch's [nq, k]
result arrays are KEPT AS ARRAYS to the end of the job: ``searcher.runs[index_name]`` is an ``ArrayRun``
(viquae_amd/ir/runs.py), the same ``{q_id: {str(doc): score}}`` mapping whose entries stay rows of the result arrays until
somebody indexes them; the run files are written from the arrays (``mq_format_run_json``, byte for byte what ``json.dump``
of the dicts writes) and the late fusion takes its tables from them.  Only the on-the-fly relevance judgement (a reference
KB is given: it needs each query's run at once) fills the dicts batch by batch as before.  With an ``index_mapping`` the hits are
expanded with numpy (CSR gather) before the pass that applies the reference's insertion rules.
"""
import json
import os
import re
import string
import warnings
from pathlib import Path

import numpy as np

from ..index import L2_DIRECT_BELOW
from .runs import ArrayRun, dump_run
from .search import KnowledgeBase


_PUNCTUATION = str.maketrans("", "", string.punctuation)
_ARTICLES = re.compile(r"\b(a|an|the)\b")


def answer_preprocess(answer):
    """Lower-case, drop punctuation, articles and extra whitespace (meerqat/data/loading.py:152-164, in its order: lower ->
    punctuation -> articles -> whitespace).  Same string as the reference's character loop, through ``str.translate`` and a
    compiled pattern: ~15 x faster on a 100-word passage, and the relevance judgement calls it once per retrieved passage."""
    return " ".join(_ARTICLES.sub(" ", answer.lower().translate(_PUNCTUATION)).split())


class PassageTexts:
    """The reference KB's passages, preprocessed once each.  ``find_relevant`` (meerqat/ir/metrics.py:79-124) reads
    ``kb[i][reference_key]`` -- one Arrow row materialised as a Python dict per retrieved passage -- and preprocesses it, 100
    times per question and index; here the misses of a whole request are fetched with ONE ``take`` on the Arrow column and the
    preprocessed strings are kept (up to ``capacity`` of them: popular passages come back for many questions)."""

    def __init__(self, kb, reference_key="passage", capacity=1 << 20):
        self.kb, self.key, self.capacity = kb, reference_key, int(capacity)
        self.texts = {}
        self._column = None
        try:
            if getattr(kb, "_indices", None) is None and reference_key in kb.column_names:
                self._column = kb.data.column(reference_key)
        except Exception:  # not an Arrow-backed datasets.Dataset (tests pass plain lists of dicts)
            self._column = None

    def get_many(self, ids):
        texts = self.texts
        missing = [i for i in ids if i not in texts]
        if missing:
            if len(texts) + len(missing) > self.capacity:
                # evicting drops ids of THIS request that were cached too: fetch the whole request again
                texts.clear()
                missing = list(dict.fromkeys(ids))
            if self._column is not None:
                raw = self._column.take(missing).to_pylist()
            else:
                raw = [self.kb[i][self.key] for i in missing]
            for i, t in zip(missing, raw):
                texts[i] = answer_preprocess(t)
        return [texts[i] for i in ids]


def find_relevant(retrieved, original_answer, alternative_answers, kb, reference_key="passage", question_type="String", passages=None):
    """Which retrieved passages contain the answer (or one of its aliases) as whole words (meerqat/ir/metrics.py:79-124).
    ``passages``: a :class:`PassageTexts` over ``kb`` (the searcher keeps one); without it every passage is read and preprocessed
    here.  A passage is relevant when ANY alias matches, which is all the reference's first-match loop decides."""
    if question_type not in ("String", None):
        raise NotImplementedError("numerical / time question types (InfoSeek) are not judged by this mirror")
    original_relevant, relevant = [], []
    ids = [int(i) for i in retrieved]
    if not ids:
        return original_relevant, relevant
    # After answer_preprocess an answer holds no ASCII punctuation, hence no regex metacharacter: `\b{answer}\b` is the literal
    # answer between word boundaries, and `answer in passage` (a C substring search, ~100 x cheaper than the regex on a 100-word
    # passage) is a necessary condition -- the pattern only runs on the passages that contain the answer at all.
    answer0 = answer_preprocess(original_answer)
    first = re.compile(rf"\b{answer0}\b")
    aliases = [(a, re.compile(rf"\b{a}\b")) for a in dict.fromkeys(answer_preprocess(a) for a in alternative_answers)]
    texts = passages.get_many(ids) if passages is not None else [answer_preprocess(kb[i][reference_key]) for i in ids]
    for i, passage in zip(ids, texts):
        if answer0 in passage and first.search(passage) is not None:
            original_relevant.append(i)
            relevant.append(i)
        elif any(a in passage and pat.search(passage) is not None for a, pat in aliases):
            relevant.append(i)
    return original_relevant, relevant


class _Mapping:
    """index_mapping (dict article -> list of passage ids) as CSR arrays for vectorised fan-out."""

    def __init__(self, mapping):
        keys = sorted(mapping)
        self.lookup = {k: n for n, k in enumerate(keys)}
        lens = np.array([len(mapping[k]) for k in keys], dtype=np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(lens)])
        self.values = np.concatenate([np.asarray(mapping[k], dtype=np.int64) for k in keys]) if keys else np.zeros(0, np.int64)
        # for whole blocks of hits: the keys as a sorted array (when they are integers) and whether no passage belongs to two
        # articles -- then distinct hits can never write one run entry twice, and a block needs no per-query dict semantics
        try:
            self.keys = np.asarray(keys, dtype=np.int64) if all(isinstance(k, (int, np.integer)) for k in keys) else None
        except (TypeError, ValueError, OverflowError):
            self.keys = None
        self.disjoint = bool(self.values.size == np.unique(self.values).size) and bool((self.values >= 0).all())

    def expand(self, indices, scores, many2one):
        """(passage ids, scores) of one query's hits, in hit order."""
        rows = np.array([self.lookup[int(i)] for i in indices], dtype=np.int64)
        starts, ends = self.offsets[rows], self.offsets[rows + 1]
        counts = ends - starts
        pos = np.arange(counts.sum()) - np.repeat(np.cumsum(counts) - counts, counts)
        ids = self.values[np.repeat(starts, counts) + pos]
        sc = np.repeat(np.asarray(scores, dtype=np.float32), counts)
        if many2one is None:
            # penalty grows with the passage's rank inside its article.  `np.float32 score - python float penalty`
            # is a float32 subtraction under NumPy >= 2 (what this container's reference run does; it was float64
            # under NumPy 1.x): the penalty is cast to float32 first
            sc = sc - (1e-8 * pos).astype(np.float32)
        return ids, sc, np.cumsum(counts)

    def expand_block(self, indices, scores, k, many2one):
        """A whole block of hits at once: ``indices`` / ``scores`` [nq, kk] -> (passage ids [nq, W] padded with -1, scores
        [nq, W] float32) after the reference's loop (meerqat/ir/search.py:419-440): every hit's passages in order, the 1e-8
        penalty per rank inside the article (``many2one`` None) or the hit's score (``"max"``), cut behind the first HIT at
        which the run holds k entries.  Only for blocks where no run entry can be written twice -- a disjoint mapping and
        distinct hits per query, which the caller checks -- so that the dict semantics reduce to a running count.  Returns None
        when a hit is not a key of the mapping (the caller's per-query path raises the reference's KeyError)."""
        nq, kk = indices.shape
        flat = indices.ravel()
        rows = np.searchsorted(self.keys, flat)
        rows[rows >= self.keys.size] = 0
        if self.keys.size == 0 or not bool((self.keys[rows] == flat).all()):
            return None
        counts = (self.offsets[rows + 1] - self.offsets[rows]).reshape(nq, kk)
        cum = np.cumsum(counts, axis=1)
        # the loop stops behind the first hit with cum >= k (checked after EVERY hit, also one that maps to no passage)
        reached = cum >= k
        last = np.where(reached.any(axis=1), reached.argmax(axis=1), kk - 1)
        keep_hit = np.arange(kk)[None, :] <= last[:, None]
        counts = np.where(keep_hit, counts, 0)
        per_q = counts.sum(axis=1)
        W = int(per_q.max()) if nq else 0
        ids = np.full((nq, max(W, 1)), -1, dtype=np.int64)
        sc = np.zeros((nq, max(W, 1)), dtype=np.float32)
        c = counts.ravel()
        total = int(c.sum())
        if total:
            first = np.cumsum(c) - c                                  # position of each hit's first passage in the flat expansion
            pos = np.arange(total) - np.repeat(first, c)              # rank of a passage inside its article
            src = np.repeat(self.offsets[rows], c) + pos
            qrow = np.repeat(np.repeat(np.arange(nq), kk), c)
            col = np.arange(total) - np.repeat(np.cumsum(per_q) - per_q, per_q)
            ids[qrow, col] = self.values[src]
            val = np.repeat(scores.ravel().astype(np.float32), c)
            if many2one is None:
                val = val - (1e-8 * pos).astype(np.float32)
            sc[qrow, col] = val
        return ids, sc


class Searcher:
    def __init__(self, kb_kwargs, k=100, reference_kb_path=None, reference_key="passage", qrels=None, request_timeout=1000,
                 es_client_kwargs={}, fusion_kwargs={}, metrics_kwargs={}, do_fusion=None, qnonrels=None, kbs=None,
                 reference_kb=None):
        self.k = k
        self.kbs = {}
        self.qrels = {}
        self.qnonrels = {}
        if qrels is not None:
            with open(qrels, "rt") as file:
                self.qrels = json.load(file)
        if qnonrels is not None:
            with open(qnonrels, "rt") as file:
                self.qnonrels = json.load(file)
        self._runs = {}
        self._pending = []  # (kb, index_name, q_ids, scores [nq, k], indices [nq, k]) blocks not yet turned into dicts
        resolved = {}
        for kb_path, kb_kwarg in kb_kwargs.items():
            real = Path(kb_path).expanduser().resolve()
            if real in resolved:
                raise ValueError(f"'{kb_path}' and '{resolved[real]}' resolve to the same path")
            resolved[real] = kb_path
            kb = kbs[kb_path] if kbs and kb_path in kbs else KnowledgeBase(kb_path, **kb_kwarg)
            self.kbs[kb_path] = kb
            assert not (kb.indexes.keys() & self._runs.keys()), "All KBs should have unique index names"
            for index_name in kb.indexes:
                self._runs[index_name] = ArrayRun()
        assert not ({"search", "fusion"} & self._runs.keys()), "'search', 'fusion' are reserved names"
        self.do_fusion = True if (do_fusion is None and len(self._runs) > 1) else do_fusion
        if self.do_fusion:
            assert len(self._runs) > 1
        if reference_kb is not None:
            self.reference_kb = reference_kb
        elif reference_kb_path is None:
            assert qrels is not None
            warnings.warn("Didn't get a reference KB -> will not be able to extend the annotation coverage "
                          "so results should be interpreted carefully.\n")
            self.reference_kb = None
        else:
            from datasets import load_from_disk
            ref = load_from_disk(reference_kb_path)
            self.reference_kb = ref.remove_columns([c for c in ref.column_names if c != reference_key])
        self.reference_key = reference_key
        self.fusion_kwargs = fusion_kwargs
        self.metrics_kwargs = dict(metrics=["mrr@100", "precision@1", "precision@20", "hit_rate@20"])
        self.metrics_kwargs.update(metrics_kwargs)
        # where the metric report and the late fusion run: this process's GPU (one process per GPU under torch.distributed.run)
        self.metric_device = f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}"
        self._csr = {}

    @property
    def runs(self):
        """``runs[index_name][q_id][doc_id] = score`` (meerqat/ir/search.py:386,413-440).  Each run is an ``ArrayRun``
        (viquae_amd/ir/runs.py): a mapping whose entries stay rows of the [nq, k] result arrays until they are indexed; reading
        ``runs`` files the result blocks kept since the last read into it, in arrival order."""
        self._flush()
        return self._runs

    @runs.setter
    def runs(self, value):
        self._pending = []
        self._runs = value

    def _as_block(self, kb, run, q_ids, scores, indices):
        """(ids [nq, W] padded with -1, scores [nq, W]) of a result block that can enter ``run`` as rows of arrays -- new, distinct
        question ids whose hits cannot write one run entry twice -- else None (the per-query loop with the reference's dict
        semantics then fills the run).  Without an ``index_mapping``: distinct non-negative hits, cut at k; with one: a disjoint
        mapping and distinct hits (``_Mapping.expand_block``)."""
        nq, kk = indices.shape
        if not isinstance(run, ArrayRun) or not nq or not kk or len(set(q_ids)) != nq or any(q in run for q in q_ids):
            return None
        if kb.index_mapping is None:
            cut = min(self.k, kk)
            if cut <= 0:
                return None
            srt = np.sort(indices[:, :cut], axis=1)
            if not bool((srt[:, 0] >= 0).all() and (srt[:, 1:] != srt[:, :-1]).all()):
                return None
            # distinct hits of new questions: the reference's loop keeps exactly the first k of them, in order
            return indices[:, :cut], scores[:, :cut]
        if kb.many2one not in (None, "max"):
            return None
        # article -> passage fan-out of a whole block (the image indexes of the fusion configs): when the mapping is disjoint and a
        # query's hits are distinct, no run entry is written twice and the reference's loop is a running count
        csr = self._csr_of(kb)
        srt = np.sort(indices, axis=1)
        if csr.keys is None or not csr.disjoint or not bool((srt[:, 1:] != srt[:, :-1]).all()):
            return None
        return csr.expand_block(indices, scores, self.k, kb.many2one)

    def _flush(self):
        pending, self._pending = self._pending, []
        waiting = {}   # index_name -> (q_ids, set(q_ids), [ids blocks], [score blocks]): consecutive plain blocks filed as ONE

        def file_waiting(index_name):
            w = waiting.pop(index_name, None)
            if w is not None:
                self._runs[index_name].add_block(w[0], np.concatenate(w[2]) if len(w[2]) > 1 else w[2][0],
                                                 np.concatenate(w[3]) if len(w[3]) > 1 else w[3][0])

        for kb, index_name, q_ids, scores, indices in pending:
            run = self._runs[index_name]
            indices = np.asarray(indices)
            scores = np.asarray(scores, dtype=np.float32)
            block = self._as_block(kb, run, q_ids, scores, indices)
            w = waiting.get(index_name)
            if block is not None and kb.index_mapping is None and (w is None or (w[1].isdisjoint(q_ids) and w[2][0].shape[1] == block[0].shape[1])):
                # the block stays a block of rows (viquae_amd/ir/runs.py): dicts are built when somebody indexes the run
                if w is None:
                    w = waiting[index_name] = ([], set(), [], [])
                w[0].extend(q_ids)
                w[1].update(q_ids)
                w[2].append(block[0])
                w[3].append(block[1])
                continue
            file_waiting(index_name)
            if block is not None and not any(q in run for q in q_ids):
                run.add_block(q_ids, block[0], block[1])
                continue
            for q_id, sc, idx in zip(q_ids, scores.tolist(), indices.tolist()):
                self._fill_run(kb, run.setdefault(q_id, {}), sc, idx)
        for index_name in list(waiting):
            file_waiting(index_name)

    def _passage_texts(self):
        texts = getattr(self, "_texts", None)
        if texts is None or texts.kb is not self.reference_kb:
            texts = self._texts = PassageTexts(self.reference_kb, self.reference_key)
        return texts

    def _csr_of(self, kb):
        csr = self._csr.get(id(kb))
        if csr is None:  # (dict.setdefault would build the CSR arrays again for EVERY query: its argument is evaluated first)
            csr = self._csr[id(kb)] = _Mapping(kb.index_mapping)
        return csr

    def _fill_run(self, kb, run_q, scores, indices):
        """One query's hits -> run dict, cut at k entries (reference: search.py:413-440)."""
        if kb.index_mapping is None:
            if not run_q and len(set(indices[:self.k])) == min(self.k, len(indices)):
                # distinct hits into an empty run: the loop below keeps exactly the first k of them
                run_q.update(zip(map(str, indices[:self.k]), scores))
                return
            for score, i in zip(scores, indices):
                run_q[str(i)] = score
                if len(run_q) >= self.k:
                    break
            return
        csr = self._csr_of(kb)
        ids, sc, ends = csr.expand(indices, scores, kb.many2one)
        if kb.many2one not in (None, "max"):
            raise ValueError(f"Invalid value for many2one: '{kb.many2one}'. Choose from {{None, 'max'}}")
        keys, vals = list(map(str, ids.tolist())), sc.tolist()
        start = 0
        for end in ends.tolist():  # one hit = its passages [start, end), possibly none (an empty index_mapping entry)
            if kb.many2one is None:
                for n in range(start, end):
                    run_q[keys[n]] = vals[n]
            else:
                for n in range(start, end):
                    j = keys[n]
                    if j not in run_q or run_q[j] < vals[n]:
                        run_q[j] = vals[n]
            start = end
            # the reference tests the cut after each HIT, all passages of the article inserted first -- also after a hit
            # that maps to no passage at all, and also when the run was already full on entry (a repeated question id)
            if len(run_q) >= self.k:
                return

    arrow_queries = None  # set by dataset_search: query vectors come from the Arrow table, not from `batch`

    def __call__(self, batch, row_indices=None):
        question_types = batch.get("question_type", ["String"] * len(batch["id"]))
        # the answers are only read when relevance is judged on the fly (dataset_search does not even decode them otherwise)
        outputs = batch["output"] if self.reference_kb is not None else [None] * len(batch["id"])
        for kb in self.kbs.values():
            for index_name, index in kb.indexes.items():
                if self.arrow_queries is not None and row_indices is not None and index.key in self.arrow_queries.columns:
                    # vectors straight from the Arrow table (no None among them), searched a window of rows at a time
                    scores_batch, indices_batch = self.arrow_queries.search(kb, index_name, index.key, row_indices, self.k)
                else:
                    queries = batch[index.key]
                    if any(query is None for query in queries):
                        scores_batch, indices_batch = kb.search_batch_if_not_None(index_name, queries, k=self.k)
                    else:
                        scores_batch, indices_batch = kb.search_batch(index_name, queries, k=self.k)
                if self.reference_kb is None and isinstance(scores_batch, np.ndarray) and isinstance(indices_batch, np.ndarray):
                    # nothing reads this batch's run before the end of the job: keep the arrays (see `runs`)
                    self._pending.append((kb, index_name, list(batch["id"]), scores_batch, indices_batch))
                    continue
                self._flush()
                run = self._runs[index_name]
                block = None
                if isinstance(scores_batch, np.ndarray) and isinstance(indices_batch, np.ndarray) and indices_batch.ndim == 2:
                    # the judged job keeps its runs as arrays too (round 5): what the judgement needs of a question's run is the SET
                    # of retrieved documents, which is the row of the block
                    block = self._as_block(kb, run, list(batch["id"]), np.asarray(scores_batch, dtype=np.float32), indices_batch)
                    if block is not None:
                        run.add_block(list(batch["id"]), block[0], block[1])
                for row, (q_id, scores, indices, gt, question_type) in enumerate(zip(batch["id"], scores_batch, indices_batch, outputs,
                                                                                     question_types)):
                    if block is None:
                        run_q = run.setdefault(q_id, {})
                        scores = np.asarray(scores).tolist()
                        indices = np.asarray(indices).tolist()
                        self._fill_run(kb, run_q, scores, indices)
                        run_keys = run_q.keys()
                    else:
                        ids_row = block[0][row]
                        run_keys = {str(i) for i in ids_row[ids_row >= 0].tolist()}
                    if self.reference_kb is not None:
                        self.qrels.setdefault(q_id, {})
                        self.qnonrels.setdefault(q_id, {})
                        retrieved = run_keys - (self.qrels[q_id].keys() | self.qnonrels[q_id].keys())
                        _, relevant = find_relevant(retrieved, gt["original_answer"], gt["answer"], self.reference_kb,
                                                    reference_key=self.reference_key, question_type=question_type,
                                                    passages=self._passage_texts())
                        self.qrels[q_id].update({str(i): 1 for i in relevant})
                        self.qnonrels[q_id].update({i: 0 for i in retrieved - self.qrels[q_id].keys()})
        return batch


class ArrowQueryColumns:
    """The query columns of the searcher's dense indexes, read straight from the dataset's Arrow table.

    Under the shipped configs' ``"format": {}`` (meerqat/ir/search.py:535-536) ``Dataset.map`` decodes every embedding of a
    batch into Python floats -- 256 x 768 objects, ~50 ms, then ~5 ms more for ``np.array`` in ``search_batch`` (:143) --
    against ~1 ms for the search itself.  ``dataset_search`` therefore drops these columns from what ``map`` decodes and
    lets the searcher fetch a batch's vectors by row index as one float32 [batch, d] array (a zero-copy view of the Arrow
    buffer, ~0.05 ms).  Left to the ordinary path: columns holding None (missing faces, :148-171), datasets with an
    indices mapping (select / shuffle) and datasets whose format the user chose."""

    def __init__(self, dataset, searcher):
        import pyarrow as pa
        self.columns = {}
        self._ahead, self._width = {}, {}
        self.window = int(os.environ.get("MQ_SEARCH_WINDOW", "4096"))
        fmt = getattr(dataset, "format", None) or {}
        if fmt.get("type") is not None or getattr(dataset, "_indices", None) is not None:
            return
        for kb in searcher.kbs.values():
            for index in kb.indexes.values():
                key = index.key
                if key in self.columns or key not in dataset.column_names:
                    continue
                col = dataset.data.column(key)
                t = col.type
                if not (pa.types.is_list(t) or pa.types.is_large_list(t) or pa.types.is_fixed_size_list(t)):
                    continue
                if not pa.types.is_floating(t.value_type) or col.null_count:
                    continue
                self.columns[key] = col

    def __bool__(self):
        return bool(self.columns)

    # Search-ahead.  `Dataset.map` hands the searcher `map_kwargs.batch_size` = 256 rows at a time (the shipped configs), but
    # one search of 4096 queries costs 2.05 us per query on the device against 3.4 us in batches of 256 (the KB is streamed once
    # per search either way) and one trip through the index wrappers instead of sixteen.  When the query vectors come from the
    # Arrow table anyway, the searcher therefore searches a WINDOW of consecutive rows at the first batch that needs them and
    # serves the following batches from the result arrays.  With the inner product a query's exact top-k does not depend on
    # what else is in its batch, so every batch gets exactly the arrays it would have got.  With the L2 metric it does depend
    # on the SIZE of the call: FAISS (and this library: MQ_KNN_L2_DIRECT_BELOW) computes a batch of fewer than 20 queries
    # with the direct sums of (q - x)^2 and a larger one with the BLAS form ||q||^2 + ||x||^2 - 2 q.x -- other score bits,
    # possibly another order of near-ties.  An L2 index is therefore only searched ahead when the batch AND the window call
    # both have at least 20 rows (both then use the BLAS form, as the per-batch calls would); a short batch -- a small
    # batch_size, or the last few rows of the dataset -- is searched by its own call.  MQ_SEARCH_WINDOW=<rows> (0 = off).  (A helper thread
    # searching window j + 1 while `map` hands out window j measured nothing -- 307 k against 312 k queries/s: the map loop holds
    # the GIL, the helper waits a switch interval for each of its Python steps -- and was dropped.)

    def search(self, kb, index_name, key, indices, k):
        """(scores, ids) of the rows `indices` of query column `key`, as ``kb.search_batch(index_name, vectors, k)`` returns."""
        n = len(indices)
        first = int(indices[0]) if n else 0
        slot = (id(kb), index_name, key, k)
        width = self._width.setdefault(slot, self.window // n * n if n else 0)  # whole batches: windows start where batches do
        if not n or int(indices[-1]) - first + 1 != n or width <= n:
            return kb.search_batch(index_name, self.batch(key, indices), k=k)
        size_matters = self._call_size_matters(kb, index_name)
        if size_matters and n < L2_DIRECT_BELOW:
            return kb.search_batch(index_name, self.batch(key, indices), k=k)
        hit = self._ahead.get(slot)
        if hit is None or not (hit[0] <= first and first + n <= hit[1]):
            stop = min(first + width, len(self.columns[key]))
            if size_matters and stop - first < L2_DIRECT_BELOW:  # the tail of the column: a call of its own form
                return kb.search_batch(index_name, self.batch(key, indices), k=k)
            scores, ids = kb.search_batch(index_name, self.batch(key, range(first, stop)), k=k)
            if not (isinstance(scores, np.ndarray) and isinstance(ids, np.ndarray)):
                return scores[:n], ids[:n]  # an index that answers with lists: no slicing guarantees, no cache
            hit = self._ahead[slot] = (first, stop, scores, ids)
        lo = first - hit[0]
        return hit[2][lo:lo + n], hit[3][lo:lo + n]

    @staticmethod
    def _call_size_matters(kb, index_name):
        """True unless the index is known to score by inner product (whose arithmetic is the same for any batch size)."""
        try:
            index = kb.dataset._indexes[index_name]
        except (AttributeError, KeyError, TypeError):
            return True
        return getattr(index, "metric_type", 1) != 0

    def close(self):
        self._ahead.clear()

    def batch(self, key, indices):
        """float32 [len(indices), d] of column ``key`` (indices: what ``map(with_indices=True)`` hands over)."""
        col = self.columns[key]
        n = len(indices)
        first = int(indices[0])
        if n and int(indices[-1]) - first + 1 == n:
            part = col.slice(first, n)
        else:
            part = col.take(list(map(int, indices)))
        part = part.combine_chunks()
        flat = part.flatten().to_numpy(zero_copy_only=False)
        if flat.size % max(n, 1):
            raise ValueError(f"column '{key}' holds vectors of different lengths")
        return np.ascontiguousarray(flat.reshape(n, -1), dtype=np.float32)


def dataset_search(dataset, k=100, metric_save_path=None, map_kwargs={}, report=True, **kwargs):
    """Searcher over ``dataset.map``; saves qrels / runs (JSON) under ``metric_save_path``, computes and saves the
    metric report like the reference (meerqat/ir/search.py:500-512), then runs the fusion subcommand (``fit`` or ``test``)."""
    searcher = Searcher(k=k, **kwargs)
    queries = ArrowQueryColumns(dataset, searcher)
    # the mapped dataset is not kept (neither does the reference keep it): no point in fingerprinting the searcher
    map_kwargs = dict(map_kwargs)
    if "new_fingerprint" not in map_kwargs:
        from datasets.fingerprint import generate_random_fingerprint
        map_kwargs["new_fingerprint"] = generate_random_fingerprint()
    # Only what Searcher.__call__ reads is decoded by `map` (ids; answers and question types when relevance is judged on the
    # fly; query columns the Arrow transport does not serve), and nothing is written back: the reference discards the mapped
    # dataset too (meerqat/ir/search.py:482), so the searcher is wrapped to return None ("no update" for Dataset.map).
    needed = {"id"} | {index.key for kb in searcher.kbs.values() for index in kb.indexes.values()} - set(queries.columns)
    if searcher.reference_kb is not None:
        needed |= {"output", "question_type"}
    dataset = dataset.remove_columns([c for c in dataset.column_names if c not in needed])
    if queries:
        searcher.arrow_queries = queries

        def run_batch(batch, row_indices):
            searcher(batch, row_indices)

        try:
            dataset.map(run_batch, batched=True, with_indices=True, **map_kwargs)
        finally:
            queries.close()
            searcher.arrow_queries = None
    else:
        def run_batch(batch):
            searcher(batch)

        dataset.map(run_batch, batched=True, **map_kwargs)
    from .embedding import process_rank_and_world
    if process_rank_and_world()[0] != 0:
        metric_save_path = None  # sharded search: every rank holds the same runs, rank 0 writes them
    if metric_save_path is not None:
        metric_save_path = Path(metric_save_path)
        metric_save_path.mkdir(exist_ok=True)
        with open(metric_save_path / "qrels.json", "wt") as file:
            json.dump(searcher.qrels, file)
        with open(metric_save_path / "qnonrels.json", "wt") as file:
            json.dump(searcher.qnonrels, file)
        for index_name, run in searcher.runs.items():
            dump_run(run, metric_save_path / f"{index_name}.json")  # straight from the result arrays (mq_format_run_json)
    # the metric report (reference: search.py:500-512, ranx.compare) from the result arrays, on the device; scores only, no
    # statistical tests (viquae_amd/ir/metrics.py)
    # `report=False` (extra key) skips it: the search and the run files are then all the job does
    searcher.report = None
    if report:
        from .metrics import compare
        searcher.report = compare(searcher.qrels, searcher.runs, device=searcher.metric_device, **searcher.metrics_kwargs)
        print(searcher.report)
        if metric_save_path is not None:
            searcher.report.save(metric_save_path / "metrics.json")
            with open(metric_save_path / "metrics.tex", "wt") as file:
                file.write(searcher.report.to_latex())
    # late fusion of the searches (reference: search.py:514-524), on the device
    if searcher.do_fusion:
        from .fuse import Fusion
        fusion_kwargs = dict(searcher.fusion_kwargs)
        subcommand = fusion_kwargs.pop("subcommand")
        subcommand_kwargs = fusion_kwargs.pop("subcommand_kwargs", {})
        fuser = Fusion(qrels=searcher.qrels, runs=list(searcher.runs.values()), output=metric_save_path,
                       **{"device": searcher.metric_device, **fusion_kwargs})
        searcher.fusion = getattr(fuser, subcommand)(**subcommand_kwargs)
    return searcher


def main(dataset_path, config_path, k=100, metrics=None, disable_caching=False):
    """The reference's ``python -m meerqat.ir.search <dataset> <config> [--k=<k> --metrics=<path> --disable_caching]``
    (meerqat/ir/search.py:527-543): load the dataset and the JSON config, apply its ``format`` entry, search, save."""
    import datasets
    from datasets import load_from_disk
    from .embedding import init_process_group_from_env
    if disable_caching:
        datasets.disable_caching()
    # under `python -m torch.distributed.run --nproc-per-node N -m viquae_amd.ir.searcher ...`: one process per GPU, every
    # KB index row-sharded over the ranks (viquae_amd.sharded.make_flat_index); every rank runs the same questions and ends
    # with the same runs, rank 0 writes them
    init_process_group_from_env()
    dataset = load_from_disk(dataset_path)
    with open(config_path, "rt") as file:
        config = json.load(file)
    dataset.set_format(**config.pop("format", {}))
    return dataset_search(dataset, int(k), metric_save_path=Path(metrics) if metrics is not None else None, **config)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="search a dataset's embedded questions in HIP-backed KBs (meerqat.ir.search)")
    ap.add_argument("dataset")
    ap.add_argument("config")
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--metrics")
    ap.add_argument("--disable_caching", action="store_true")
    a = ap.parse_args()
    main(a.dataset, a.config, a.k, a.metrics, a.disable_caching)
