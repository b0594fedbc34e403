"""Runs kept as arrays to the end of a search job.

The reference fills ``runs[index_name][q_id][str(doc)] = score`` with a Python triple loop per batch
(meerqat/ir/search.py:413-440) and ends the job with ``run.save(...)`` / ``Fusion(runs=...)`` over those dicts (:485-524).
``ArrayRun`` is the same mapping -- ``run[q_id]`` is a ``{str(doc): float(score)}`` dict, iteration is in arrival order,
``run == {...}`` compares like a dict -- whose entries stay rows of the ``[nq, k]`` result arrays until somebody indexes
them:

* ``run[q_id]`` / ``run.items()`` build the dicts asked for (and keep them: a caller may mutate what it got);
* ``dump_json(file)`` writes the run file straight from the arrays -- the text ``json.dump(run.to_dict(), file)`` would
  write, byte for byte -- through the library's host formatter (``mq_format_run_json``: all host cores, no per-hit Python
  object);
* ``tables()`` hands the late fusion (``viquae_amd.ir.fuse``) the id / score tables without the dict round trip.

Entries that needed the reference's insertion rules (an ``index_mapping`` fan-out, a question id seen twice, duplicate hits)
are ordinary dicts inside the same mapping."""
import json
from collections.abc import MutableMapping

import numpy as np

MAX_PARTS = 32  # MQ_RUN_JSON_MAX_PARTS


def _json_key(q):
    """A question id as the JSON object key ``json.dump`` writes for it: str as is; int / float / bool / None coerced the way
    ``json.encoder`` coerces dict keys (``{5: ...}`` -> ``"5"``, ``True`` -> ``"true"``, ``None`` -> ``"null"``)."""
    if isinstance(q, str):
        return json.dumps(q)
    if q is True:
        return '"true"'
    if q is False:
        return '"false"'
    if q is None:
        return '"null"'
    if isinstance(q, int):
        return json.dumps(int.__repr__(q))
    if isinstance(q, float):
        return json.dumps(json.dumps(q))      # float.__repr__, or NaN / Infinity as the encoder spells them
    raise TypeError(f"keys must be str, int, float, bool or None, not {type(q).__name__}")


class ArrayRun(MutableMapping):
    def __init__(self, mapping=None):
        self._entries = {}   # q_id -> dict | (block number, row)
        self._blocks = []    # (ids int64 [n, K], scores [n, K] f32 | f64, doc_names | None)
        if mapping:
            self.update(mapping)

    # ---------------------------------------------------------------- filling
    def add_block(self, q_ids, ids, scores, doc_names=None):
        """Rows of a result block whose q_ids are new to the run: ``ids`` int64 [n, K] (a row ends at its first negative id),
        ``scores`` [n, K] float32 or float64.  ``doc_names``: names of non-numeric document ids (ids index into it)."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        scores = np.ascontiguousarray(scores)
        if scores.dtype not in (np.float32, np.float64):
            scores = scores.astype(np.float64)
        if ids.ndim != 2 or ids.shape != scores.shape or len(q_ids) != ids.shape[0]:
            raise ValueError("add_block: q_ids [n], ids [n, K] and scores [n, K] must agree")
        b = len(self._blocks)
        self._blocks.append((ids, scores, doc_names))
        entries = self._entries
        for row, q in enumerate(q_ids):
            if q in entries:
                raise ValueError(f"add_block: question id {q!r} is already in the run")
            entries[q] = (b, row)

    def _row_dict(self, ref):
        ids, scores, names = self._blocks[ref[0]]
        i, s = ids[ref[1]], scores[ref[1]]
        neg = np.flatnonzero(i < 0)
        n = int(neg[0]) if neg.size else len(i)
        keys = map(str, i[:n].tolist()) if names is None else (names[j] for j in i[:n].tolist())
        return dict(zip(keys, s[:n].tolist()))

    # ---------------------------------------------------------------- the mapping
    def __getitem__(self, q):
        e = self._entries[q]
        if type(e) is tuple:
            e = self._entries[q] = self._row_dict(e)
        return e

    def __setitem__(self, q, results):
        self._entries[q] = results

    def __delitem__(self, q):
        del self._entries[q]

    def __iter__(self):
        return iter(self._entries)

    def __len__(self):
        return len(self._entries)

    def __contains__(self, q):
        return q in self._entries

    def __eq__(self, other):
        if isinstance(other, ArrayRun):
            other = other.to_dict()
        if not isinstance(other, dict):
            return NotImplemented
        return self.to_dict() == other

    __hash__ = None

    def __repr__(self):
        lazy = sum(type(e) is tuple for e in self._entries.values())
        return f"ArrayRun({len(self._entries)} questions, {lazy} still rows of {len(self._blocks)} result blocks)"

    def is_filled(self, q):
        """True when the run holds at least one hit for ``q`` (without building its dict)."""
        e = self._entries.get(q)
        if e is None:
            return False
        if type(e) is tuple:
            ids = self._blocks[e[0]][0]
            return bool(ids.shape[1]) and bool(ids[e[1], 0] >= 0)
        return bool(e)

    def lazy_questions(self):
        return sum(type(e) is tuple for e in self._entries.values())

    def to_dict(self):
        """The plain ``{q_id: {doc: score}}`` dict (every entry built, block by block: ONE str() and ONE float pass per block)."""
        cache = {}
        out = {}
        for q, e in self._entries.items():
            if type(e) is not tuple:
                out[q] = e
                continue
            b = e[0]
            if b not in cache:
                ids, scores, names = self._blocks[b]
                full = bool(ids.size) and bool((ids >= 0).all())
                if full and names is None:
                    cache[b] = (list(map(str, ids.ravel().tolist())), scores.ravel().tolist(), ids.shape[1])
                else:
                    cache[b] = None
            c = cache[b]
            if c is None:
                out[q] = self._row_dict(e)
            else:
                keys, vals, K = c
                out[q] = dict(zip(keys[e[1] * K:(e[1] + 1) * K], vals[e[1] * K:(e[1] + 1) * K]))
        return out

    # ---------------------------------------------------------------- run file
    def _spans(self):
        """The entries in order, cut into maximal spans: ('rows', block, first row, q_ids) of consecutive rows of one block with
        numeric document ids, or ('dicts', [(q_id, dict)])."""
        spans = []
        for q, e in self._entries.items():
            if type(e) is tuple and self._blocks[e[0]][2] is None:
                last = spans[-1] if spans else None
                if last is not None and last[0] == "rows" and last[1] == e[0] and last[2] + len(last[3]) == e[1]:
                    last[3].append(q)
                else:
                    spans.append(("rows", e[0], e[1], [q]))
            else:
                d = e if type(e) is not tuple else self._row_dict(e)
                if spans and spans[-1][0] == "dicts":
                    spans[-1][1].append((q, d))
                else:
                    spans.append(("dicts", [(q, d)]))
        return spans

    def _json_chunks(self, n_threads=0):
        """The run text as a list of bytes-like chunks (to be written one after the other): no copy of the formatted text."""
        from .. import _lib
        lib = _lib.load()
        chunks = [b"{"]
        sep = False
        for span in self._spans():
            if sep:
                chunks.append(b", ")
            sep = True
            if span[0] == "dicts":
                chunks.append(", ".join(f"{_json_key(q)}: {json.dumps(d)}" for q, d in span[1]).encode())
                continue
            _, b, r0, q_ids = span
            ids, scores, _ = self._blocks[b]
            n = len(q_ids)
            ids, scores = ids[r0:r0 + n], scores[r0:r0 + n]
            enc = [_json_key(q).encode() for q in q_ids]
            off = np.zeros(n + 1, dtype=np.int64)
            np.cumsum([len(x) for x in enc], out=off[1:])
            blob = b"".join(enc)
            K = ids.shape[1]
            cap = int(off[-1]) + n * (8 + 52 * K)           # the worst case (include/meerqat_hip.h)
            out = np.empty(cap, dtype=np.uint8)
            parts = np.zeros(2 * MAX_PARTS, dtype=np.int64)
            rc = int(lib.mq_format_run_json(blob, off.ctypes.data, n, ids.ctypes.data, scores.ctypes.data,
                                            int(scores.dtype == np.float64), K, None, out.ctypes.data, cap, int(n_threads),
                                            parts.ctypes.data))
            if rc < 0:
                raise ValueError(f"mq_format_run_json failed: {rc}")
            view = memoryview(out)
            for p in range(rc):
                if p:
                    chunks.append(b", ")
                chunks.append(view[int(parts[2 * p]):int(parts[2 * p] + parts[2 * p + 1])])
        chunks.append(b"}")
        return chunks

    def json_bytes(self, n_threads=0):
        """The bytes ``json.dumps(self.to_dict())`` encodes to, without building the dicts of entries that are still rows."""
        return b"".join(self._json_chunks(n_threads))

    def dump_json(self, file, n_threads=0):
        """Write the run file: a path, or a file object opened for bytes or text."""
        chunks = self._json_chunks(n_threads)
        if hasattr(file, "write"):
            try:
                for c in chunks:
                    file.write(c)
            except TypeError:
                file.write(b"".join(chunks).decode("ascii"))
            return
        with open(file, "wb", buffering=0) as f:
            for c in chunks:
                f.write(c)

    # ---------------------------------------------------------------- fusion tables
    def tables(self, q_ids=None, ids_only=False):
        """(q_ids, ids int64 [nq, K], scores float64 [nq, K]) with -1 / 0 in unused slots, when every entry is still a row of a
        result block with numeric document ids; None otherwise (the caller then goes through the dicts).  ``ids_only``: the
        scores come back as None (the rank metrics only need the order)."""
        if q_ids is None:
            q_ids = list(self._entries)
        refs = [self._entries.get(q) for q in q_ids]
        if not refs or any(type(e) is not tuple or self._blocks[e[0]][2] is not None for e in refs):
            return None
        K = max(self._blocks[b][0].shape[1] for b in {e[0] for e in refs})
        ids = np.full((len(refs), K), -1, dtype=np.int64)
        scores = None if ids_only else np.zeros((len(refs), K), dtype=np.float64)
        blk = np.fromiter((e[0] for e in refs), dtype=np.int64, count=len(refs))
        row = np.fromiter((e[1] for e in refs), dtype=np.int64, count=len(refs))
        for b in np.unique(blk).tolist():
            sel = np.flatnonzero(blk == b)
            bi, bs, _ = self._blocks[b]
            ids[sel, :bi.shape[1]] = bi[row[sel]]
            if scores is not None:
                scores[sel, :bi.shape[1]] = bs[row[sel]]
        # a row ends at its first negative id: clear what follows it (nothing to do for full rows -- the usual case)
        neg = ids < 0
        if neg.any():
            first = np.where(neg.any(axis=1), neg.argmax(axis=1), K)
            dead = np.arange(K)[None, :] >= first[:, None]
            ids[dead] = -1
            if scores is not None:
                scores[dead] = 0.0
        return q_ids, ids, scores


def dump_run(run, path):
    """Write ``run`` (an ``ArrayRun`` or a plain dict) as the reference's run file."""
    if isinstance(run, ArrayRun):
        run.dump_json(path)
    else:
        with open(path, "wt") as file:
            json.dump(run, file)
