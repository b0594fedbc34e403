"""Row-sharded exact indexes: the KB rows split contiguously over several MI355X.

New functionality with no reference counterpart (the reference's search path has no collective,
SURVEY.md section 2 "Collective call sites"; FAISS's own multi-GPU mode is only reached through
``device=-1`` / a device list, datasets/search.py:315-347); both classes must return exactly what
one MI355XFlatIndex over the whole matrix returns.

``ShardedFlatIndex``  -- one PROCESS per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).
  Per search, for every chunk of <= 4096 queries:
  1. the rank scans its own shard for the chunk's queries with global ids (``id_offset`` = first
     global row of the shard), writing scores and ids straight into its *shard record*
     ``{fp32 scores [nq,k] | int64 ids [nq,k]}`` (include/meerqat_hip.h, mq_shard_record_bytes);
  2. ONE all-gather of the records (12 B per entry, 4.9 MB per rank at nq=4096, k=100: latency-bound
     on xGMI, never link-bound), issued asynchronously so that it overlaps the scan of the NEXT chunk;
  3. every rank merges the ``world`` sorted lists per query (mq_topk_merge_records_f32) with the same
     (score, id) order as the single-GPU path, so "lower id wins ties" holds across shards.

``LocalShardsFlatIndex`` -- ONE process driving several GPUs (what ``device=-1`` / ``device=[0,1,..]``
  mean in the reference's configs): one MI355XFlatIndex per device, the records are copied
  peer-to-peer to the first device and merged there.  No collective library involved.

``local_index`` / ``merge_fn`` are injection points used by the CPU (gloo) tests, which cannot run
HIP kernels; the defaults are the HIP implementations and there is no CPU fallback.
"""
import os
import struct
from typing import Optional

import numpy as np

from .index import (BatchedSearchResults, BaseIndex, SearchResults, MI355XFlatIndex, METRIC_L2, MAX_K, FLAG_PHASE_FRONT,
                    FLAG_PHASE_TAIL, _SCREEN_QUERY_CHUNK, index_file_format, index_file_header, parse_string_factory, query_chunks)


_FLT_MAX = float(np.finfo(np.float32).max)


def shard_bounds(n_total: int, world: int, rank: int):
    """Contiguous row range [lo, hi) of ``rank``: ceil(N/world) rows per shard, rounded up to a
    multiple of 64 (the panel height) so every shard but the last holds whole panels.  Trailing
    ranks get an EMPTY range when N is small (N=520, world=8: ranks 5-7) -- an empty shard takes
    part in every collective and contributes nothing."""
    per = -(-n_total // world)
    per = -(-per // 64) * 64
    lo = min(rank * per, n_total)
    hi = min(lo + per, n_total)
    return lo, hi


def record_layout(nq: int, k: int):
    """(record bytes, byte offset of the ids) -- mirrors mq_shard_record_bytes / _ids_offset."""
    ids = -(-(nq * k * 4) // 8) * 8
    rec = -(-(nq * k * 12) // 16) * 16
    return rec, ids


def _record_views(buf, nq, k):
    """(D [nq,k] f32, I [nq,k] i64) views into one record (a uint8 tensor of record_layout()[0] bytes)."""
    import torch
    _, ids = record_layout(nq, k)
    D = buf[: nq * k * 4].view(torch.float32).view(nq, k)
    I = buf[ids: ids + nq * k * 8].view(torch.int64).view(nq, k)
    return D, I


def _gathered_views(buf, world, nq, k):
    """([world,nq,k] f32, [world,nq,k] i64) strided views into the all-gathered records."""
    import torch
    rec, ids = record_layout(nq, k)
    m = buf.view(world, rec)
    Ds = m[:, : nq * k * 4].view(torch.float32).view(world, nq, k)
    Is = m[:, ids: ids + nq * k * 8].view(torch.int64).view(world, nq, k)
    return Ds, Is


def _merge_metric(metric, tie_order):
    """`metric` argument of the mq_topk_merge_* entries: the metric, OR-ed with MQ_MERGE_TIE_ID_DESC."""
    from .index import MERGE_TIE_ID_DESC
    return int(metric) | (MERGE_TIE_ID_DESC if tie_order == "id_desc" else 0)


def _hip_merge_records(records, world, nq, k, metric, tie_order="id_asc"):
    """All-gathered records (CUDA uint8 [world * record_bytes]) -> merged (D, I) through the C ABI."""
    import torch
    from . import _lib
    lib = _lib.load()
    D = torch.empty((nq, k), dtype=torch.float32, device=records.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=records.device)
    with torch.cuda.device(records.device):
        _lib.check(lib.mq_topk_merge_records_f32(records.data_ptr(), world, nq, k, _merge_metric(metric, tie_order), D.data_ptr(), I.data_ptr(),
                                                 torch.cuda.current_stream(records.device).cuda_stream),
                   "mq_topk_merge_records_f32")
    return D, I


def _hip_merge(Ds, Is, metric, tie_order="id_asc"):
    """[W,nq,k] contiguous CUDA tensors -> merged (D, I) through mq_topk_merge_f32."""
    import torch
    from . import _lib
    lib = _lib.load()
    W, nq, k = Ds.shape
    Ds, Is = Ds.contiguous(), Is.contiguous()
    D = torch.empty((nq, k), dtype=torch.float32, device=Ds.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=Ds.device)
    with torch.cuda.device(Ds.device):
        _lib.check(lib.mq_topk_merge_f32(Ds.data_ptr(), Is.data_ptr(), W, nq, k, _merge_metric(metric, tie_order), D.data_ptr(),
                                         I.data_ptr(), torch.cuda.current_stream(Ds.device).cuda_stream),
                   "mq_topk_merge_f32")
    return D, I


def _fill_empty(D, I, metric):
    """What a shard without rows reports: no neighbour in any slot (FAISS's heap neutral values)."""
    D.fill_(_FLT_MAX if metric == METRIC_L2 else -_FLT_MAX)
    I.fill_(-1)


class _ShardedBase(BaseIndex):
    """search / search_batch in terms of search_device (shared by both sharded classes)."""

    def _query_device(self):
        raise NotImplementedError

    def __reduce__(self):  # see MI355XFlatIndex.__reduce__
        raise TypeError(f"{type(self).__name__} lives in HBM and cannot be pickled: use save()")

    def search_batch(self, queries, k: int = 10, **kwargs) -> BatchedSearchResults:
        """The host boundary of MI355XFlatIndex.search_batch: page-locked staging buffers kept by the index, asynchronous copies
        on the search stream, ONE synchronisation (not a pageable upload and two blocking downloads per batch)."""
        import torch
        from .index import _PINNED_IO_MAX_QUERIES
        queries = np.asarray(queries)
        if len(queries.shape) != 2:
            raise ValueError("Shape of query must be 2D")
        dev = self._query_device()
        nq, d = queries.shape
        on_gpu = torch.device(dev).type == "cuda"  # (the gloo CPU tests serve the local index from the oracle: plain copies there)
        if not on_gpu or nq == 0 or nq > _PINNED_IO_MAX_QUERIES:
            q = torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32)).to(dev)
            D, I = self.search_device(q, k)
            if on_gpu:
                torch.cuda.current_stream(dev).synchronize()
            return BatchedSearchResults(D.cpu().numpy(), I.cpu().numpy().astype(int))
        io = getattr(self, "_io", None)
        if io is None or io[0].shape[0] < nq or io[0].shape[1] != d or io[2].shape[1] != k or io[1].device != dev:
            cap = max(256, 1 << (int(nq) - 1).bit_length())
            io = self._io = (torch.empty((cap, d), dtype=torch.float32).pin_memory(), torch.empty((cap, d), dtype=torch.float32, device=dev),
                             torch.empty((cap, k), dtype=torch.float32).pin_memory(), torch.empty((cap, k), dtype=torch.int64).pin_memory())
        q_pin, q_dev, D_pin, I_pin = io
        np.copyto(q_pin[:nq].numpy(), queries, casting="same_kind" if queries.dtype.kind == "f" else "unsafe")
        with torch.cuda.device(dev):
            q_dev[:nq].copy_(q_pin[:nq], non_blocking=True)
            D, I = self.search_device(q_dev[:nq], k)
            D_pin[:nq].copy_(D, non_blocking=True)
            I_pin[:nq].copy_(I, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        return BatchedSearchResults(D_pin[:nq].numpy().copy(), I_pin[:nq].numpy().astype(int))

    def _chunk_buffers(self, n, k, dev, slot):
        """(record, gathered) of one <= 4096-query chunk, kept per (n, k) in two alternating slots: chunk i's buffers are still
        being gathered / merged while chunk i + 1 scans into the other pair."""
        import torch
        cache = self.__dict__.setdefault("_chunk_cache", {})
        key = (n, k, str(dev), slot)
        buf = cache.get(key)
        if buf is None:
            if len(cache) > 16:
                cache.clear()
            rec_bytes, _ = record_layout(n, k)
            record = torch.empty(rec_bytes, dtype=torch.uint8, device=dev)
            gathered = torch.empty(self.world * rec_bytes, dtype=torch.uint8, device=dev) if (self.world > 1 or getattr(self, "always_gather", False)) else record
            buf = cache[key] = (record, gathered)
        return buf

    def search(self, query, k: int = 10, **kwargs) -> SearchResults:
        query = np.asarray(query)
        if len(query.shape) != 1 and (len(query.shape) != 2 or query.shape[0] != 1):
            raise ValueError("Shape of query is incorrect, it has to be either a 1D array or 2D (1, N)")
        scores, indices = self.search_batch(query.reshape(1, -1), k)
        return SearchResults(scores[0], indices[0].astype(int))


class ShardedFlatIndex(_ShardedBase):
    def __init__(self, string_factory: Optional[str] = None, metric_type: Optional[int] = None, group=None,
                 local_index=None, merge_fn=None, device=None, screen=None, always_gather=False, tie_order=None, l2norm_form=None):
        import torch.distributed as dist
        self.group = group
        # always_gather: run the collective and the record merge even with ONE rank (tests / 1-GPU profiling of the N>1 path)
        self.always_gather = bool(always_gather) and dist.is_initialized()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.metric_type = METRIC_L2 if metric_type is None else int(metric_type)
        self.string_factory = string_factory
        self.do_l2norm = parse_string_factory(string_factory)
        self._l2norm_form = l2norm_form  # None: the index class's default on add, FAISS's form for a loaded "L2norm,Flat" file
        if local_index is None:
            from .index import _resolve_device
            local_index = MI355XFlatIndex(device=device, string_factory=string_factory, metric_type=metric_type, screen=screen,
                                          tie_order=tie_order, l2norm_form=l2norm_form)
            # resolved NOW: a rank whose shard turns out empty still needs a device for its records
            local_index._torch_device = _resolve_device(device)
        self.local = local_index
        self.tie_order = getattr(local_index, "tie_order", None) or tie_order or "id_asc"
        self.merge_fn = merge_fn  # None -> mq_topk_merge_records_f32 on the gathered buffer
        self.ntotal = 0
        self.d = getattr(local_index, "d", None)

    # ---------------------------------------------------------------- construction
    def _set_total(self, n, d=None):
        self.ntotal = int(n)
        if d is not None:
            self.d = int(d)

    def add_global(self, vectors):
        """Every rank is handed the same [N,d] matrix (e.g. a memory-mapped Arrow column) and keeps
        rows shard_bounds(N, world, rank)."""
        n = len(vectors)
        lo, hi = shard_bounds(n, self.world, self.rank)
        self.local.id_offset = lo
        if hi > lo:
            self.local.add(np.asarray(vectors[lo:hi], dtype=np.float32), total_hint=hi - lo)
        self._set_total(n, np.shape(vectors)[1] if n else None)

    def add_local(self, rows, id_offset: int, n_total: int):
        """This rank's shard, already selected by the caller (rows numpy or device tensor)."""
        self.local.id_offset = int(id_offset)
        self.local.add(rows)
        self._set_total(n_total, rows.shape[1])

    def add_vectors(self, vectors, column: Optional[str] = None, **kwargs):
        if column is not None:
            from .index import iter_arrow_column
            n = len(vectors)
            lo, hi = shard_bounds(n, self.world, self.rank)
            self.local.id_offset = lo
            pend = None
            d = None
            # only THIS rank's rows are decoded (host time and IO scale with N, not world x N)
            for part in iter_arrow_column(vectors, column, lo, hi):
                d = part.shape[1]
                pend = part if pend is None else np.concatenate([pend, part])
                full = (pend.shape[0] // 64) * 64
                if full:
                    self.local.add(pend[:full], total_hint=hi - lo)
                    pend = pend[full:]
            if d is None and n:  # an empty shard still needs the dimension
                for first in iter_arrow_column(vectors, column, 0, 1):
                    d = first.shape[1]
            if pend is not None and pend.shape[0]:
                self.local.add(pend, total_hint=hi - lo)
            self._set_total(n, d)
        else:
            self.add_global(vectors)

    # ---------------------------------------------------------------- search
    def _query_device(self):
        return getattr(self.local, "_torch_device", None) or "cpu"

    def _scan(self, q, k, record):
        """Local shard scan of one chunk, results written into the record's D / I views."""
        D, I = _record_views(record, q.shape[0], k)
        if getattr(self.local, "ntotal", 1) == 0:
            _fill_empty(D, I, self.metric_type)
        elif isinstance(self.local, MI355XFlatIndex):
            self.local.search_device(q, k, out=(D, I))
        else:  # injected local index (CPU tests)
            Dl, Il = self.local.search_device(q, k)
            D.copy_(Dl)
            I.copy_(Il)

    def _merge(self, gathered, nq, k):
        if self.merge_fn is None:
            return _hip_merge_records(gathered, self.world, nq, k, self.metric_type, self.tie_order)
        Ds, Is = _gathered_views(gathered, self.world, nq, k)
        if self.tie_order != "id_asc":
            return self.merge_fn(Ds, Is, self.metric_type, tie_order=self.tie_order)
        return self.merge_fn(Ds, Is, self.metric_type)

    def search_device(self, queries, k, chunk=None):
        """queries replicated on every rank (same tensor) -> merged (D, I) on every rank.

        Chunks of <= 4096 queries are software-pipelined: the all-gather of chunk i (async, RCCL's own
        stream) runs while the shard scan of chunk i+1 occupies the compute stream; only the last
        chunk's collective and merge are exposed."""
        import torch
        import torch.distributed as dist
        if k < 1:
            raise ValueError("k must be >= 1")
        if k > MAX_K:
            raise NotImplementedError(f"k={k} > {MAX_K}")
        nq = queries.shape[0]
        dev = queries.device
        if (self.world == 1 and not self.always_gather and isinstance(self.local, MI355XFlatIndex) and self.local.ntotal > 0
                and self.merge_fn is None):
            return self.local.search_device(queries, k)
        chunk = int(chunk or _SCREEN_QUERY_CHUNK)
        D = torch.empty((nq, k), dtype=torch.float32, device=dev)
        I = torch.empty((nq, k), dtype=torch.int64, device=dev)
        pending = None  # (work, gathered, s, n)

        def finish(p):
            work, gathered, s, n = p
            if work is not None:
                work.wait()  # the compute stream waits for the collective; the host does not block
            Dm, Im = self._merge(gathered, n, k)
            D[s:s + n].copy_(Dm)
            I[s:s + n].copy_(Im)

        pieces = query_chunks(nq, chunk)
        if (len(pieces) > 1 and chunk <= _SCREEN_QUERY_CHUNK and dev.type == "cuda" and isinstance(self.local, MI355XFlatIndex)
                and self.local.screen and self.local.ntotal > 0 and queries.dtype == torch.float32 and queries.is_contiguous()
                and queries.dim() == 2 and queries.shape[1] == self.local.d
                and os.environ.get("MQ_KNN_TAIL_OVERLAP", "0") == "1"):
            # Screened local shard: only the scans stay on the caller's stream.  What follows the scan of chunk i (candidates,
            # exact re-scoring, top-k), its all-gather and the merge of chunk i-1 run on the index's second stream under the
            # scan of chunk i+1 (MI355XFlatIndex.search_phase); same kernels per chunk, bit-identical results.
            main = torch.cuda.current_stream(dev)
            spaces, tail = self.local.pipeline_workspaces(max(e - s for s, e in pieces), k)
            freed = []  # freed[j]: recorded on `tail` once chunk j's all-gather has been waited for and its merge enqueued
            for ci, (s, e) in enumerate(pieces):
                q, n, ws = queries[s:e], e - s, spaces[ci & 1]
                record, gathered = self._chunk_buffers(n, k, dev, ci & 1)
                out = _record_views(record, n, k)
                if ci >= 2:
                    # the workspace AND the record / gather buffers of chunk i-2 are free again: its second half has run, its
                    # collective has finished reading the record (work.wait() in finish()) and its merge has read `gathered`.
                    # (Round 4 waited only for the second half: for k beyond the screen FRONT writes D / I straight into the
                    # record, which RCCL could still have been reading -- ADVICE r4.)
                    main.wait_event(freed[ci - 2])
                self.local.search_phase(q, k, out, ws, FLAG_PHASE_FRONT, main)
                scanned = torch.cuda.Event()
                scanned.record(main)
                tail.wait_event(scanned)
                with torch.cuda.stream(tail):
                    self.local.search_phase(q, k, out, ws, FLAG_PHASE_TAIL, tail)
                    if self.world > 1 or self.always_gather:
                        work = dist.all_gather_into_tensor(gathered, record, group=self.group, async_op=True)
                    else:
                        work = None
                    if pending is not None:
                        finish(pending)
                        ev = torch.cuda.Event()
                        ev.record(tail)
                        freed.append(ev)
                    pending = (work, gathered, s, n)
            with torch.cuda.stream(tail):
                finish(pending)
            main.wait_stream(tail)
            return D, I
        for ci, (s, e) in enumerate(pieces):
            q = queries[s:e]
            n = e - s
            # the chunk two back has been merged (finish() below runs before the next scan is enqueued on the same stream)
            record, gathered = self._chunk_buffers(n, k, dev, ci & 1)
            self._scan(q, k, record)
            if self.world > 1 or self.always_gather:
                work = dist.all_gather_into_tensor(gathered, record, group=self.group, async_op=True)
            else:
                work = None
            if pending is not None:
                finish(pending)
            pending = (work, gathered, s, n)
        if pending is not None:
            finish(pending)
        return D, I

    # ---------------------------------------------------------------- persistence
    def save(self, file, storage_options=None, format=None):
        """One file in one of MI355XFlatIndex.save's formats (FAISS's own for ``*.faiss`` / ``*.index``) holding the WHOLE
        matrix: rank 0 writes the header and sizes the file, every rank then writes its own rows at their offset (ranks of
        one node share the file system; the reference's save_path semantics, meerqat/ir/search.py:247-248)."""
        import torch.distributed as dist
        path = os.fspath(file)
        d = int(self.d or 0)
        head = index_file_header(self.ntotal, d, self.metric_type, self.do_l2norm, index_file_format(path, format))
        if self.rank == 0:
            with open(path, "wb") as f:
                f.write(head)
                f.truncate(len(head) + self.ntotal * d * 4)
        if self.world > 1:
            dist.barrier(group=self.group)
        if getattr(self.local, "ntotal", 0):
            rows = self.local.reconstruct_n()
            with open(path, "r+b") as f:
                f.seek(len(head) + int(self.local.id_offset) * d * 4)
                f.write(np.ascontiguousarray(rows, dtype=np.float32).tobytes())
        if self.world > 1:
            dist.barrier(group=self.group)

    def load_rows(self, file):
        """Fill this (empty) sharded index from a whole-matrix file: every rank reads only its row range."""
        from .index import read_index_file_header
        path = os.fspath(file)
        n, d, metric, l2norm, data_off = read_index_file_header(path)
        lo, hi = shard_bounds(n, self.world, self.rank)
        self.local.id_offset = lo
        # FaissIndex.load reads metric and transform from the FILE (datasets/search.py:399-416): a load config may omit
        # string_factory / metric_type, so the shard must scan with what the file says, like _merge and _fill_empty do
        if getattr(self.local, "ntotal", 0):
            raise ValueError("load_rows fills an EMPTY sharded index")
        self.metric_type, self.do_l2norm = int(metric), bool(l2norm)
        self.string_factory = "L2norm,Flat" if l2norm else "Flat"
        self.local.metric_type = int(metric)
        if hi > lo:
            rows = np.fromfile(path, dtype=np.float32, count=(hi - lo) * d, offset=data_off + lo * d * 4).reshape(hi - lo, d)
            self.local.do_l2norm = False  # stored rows are already normalised
            self.local.add(rows, total_hint=hi - lo)
        self.local.do_l2norm = bool(l2norm)
        if l2norm and hasattr(self.local, "l2norm_form"):  # like MI355XFlatIndex.load: a stored "L2norm,Flat" index is FAISS's object
            self.local.l2norm_form = self._l2norm_form or os.environ.get("MQ_KNN_L2NORM_FORM", "faiss")
        self._set_total(n, d)
        return self


class LocalShardsFlatIndex(_ShardedBase):
    """One process, several GPUs (``device=-1`` or ``device=[...]``): a MI355XFlatIndex per device, records
    copied peer-to-peer to the first device, merged there."""

    def __init__(self, devices, string_factory: Optional[str] = None, metric_type: Optional[int] = None, screen=None,
                 allow_repeated_devices=False, tie_order=None, l2norm_form=None):
        import torch
        from . import _lib
        _lib.require_gpu()
        devices = [int(x) for x in devices]
        # allow_repeated_devices: several shards on one GPU (exercises the N-shard path on a 1-GPU box)
        if not devices or min(devices) < 0 or (len(set(devices)) != len(devices) and not allow_repeated_devices):
            raise ValueError(f"expected a list of distinct non-negative GPU ids, got {devices}")
        if max(devices) >= torch.cuda.device_count():
            raise ValueError(f"GPU id {max(devices)} is not visible ({torch.cuda.device_count()} GPUs)")
        self.devices = devices
        self.world = len(devices)
        self.metric_type = METRIC_L2 if metric_type is None else int(metric_type)
        self.string_factory = string_factory
        self.do_l2norm = parse_string_factory(string_factory)
        self._l2norm_form = l2norm_form
        self.shards = [MI355XFlatIndex(device=g, string_factory=string_factory, metric_type=metric_type, screen=screen,
                                       tie_order=tie_order, l2norm_form=l2norm_form) for g in devices]
        self.tie_order = self.shards[0].tie_order
        for sh, g in zip(self.shards, devices):
            sh._torch_device = torch.device("cuda", g)
        self.ntotal = 0
        self.d = None

    def _query_device(self):
        return self.shards[0]._torch_device

    def add_vectors(self, vectors, column: Optional[str] = None, **kwargs):
        if column is not None:
            from .index import iter_arrow_column
            blocks = iter_arrow_column(vectors, column)
        else:
            mat = np.asarray(vectors, dtype=np.float32)
            if mat.ndim != 2:
                raise ValueError("expected a 2-D matrix of vectors")
            blocks = iter([mat])
        n = len(vectors)
        bounds = [shard_bounds(n, self.world, r) for r in range(self.world)]
        for sh, (lo, _) in zip(self.shards, bounds):
            sh.id_offset = lo
        seen = 0
        pend = [None] * self.world
        for block in blocks:
            self.d = block.shape[1]
            b_lo, b_hi = seen, seen + block.shape[0]
            seen = b_hi
            for r, (lo, hi) in enumerate(bounds):
                s, e = max(lo, b_lo), min(hi, b_hi)
                if s >= e:
                    continue
                part = block[s - b_lo:e - b_lo]
                pend[r] = part if pend[r] is None else np.concatenate([pend[r], part])
                full = (pend[r].shape[0] // 64) * 64
                if full:
                    self.shards[r].add(pend[r][:full], total_hint=hi - lo)
                    pend[r] = pend[r][full:]
        for r, (lo, hi) in enumerate(bounds):
            if pend[r] is not None and pend[r].shape[0]:
                self.shards[r].add(pend[r], total_hint=hi - lo)
        self.ntotal = n

    def load_rows(self, file):
        """Fill this (empty) index from a whole-matrix file (own format or a FAISS Flat file): metric and the "L2norm,"
        transform come from the file, like FaissIndex.load; the stored rows are already normalised, the QUERIES of every
        shard are normalised at search time."""
        from .index import read_index_file_header
        path = os.fspath(file)
        n, d, metric, l2norm, data_off = read_index_file_header(path)
        if self.ntotal:
            raise ValueError("load_rows fills an EMPTY index")
        self.metric_type, self.do_l2norm = int(metric), bool(l2norm)
        self.string_factory = "L2norm,Flat" if l2norm else "Flat"
        for sh in self.shards:
            sh.metric_type, sh.do_l2norm = int(metric), False
        if n:
            self.add_vectors(np.memmap(path, dtype=np.float32, mode="r", offset=data_off, shape=(n, d)))
        for sh in self.shards:
            sh.do_l2norm = bool(l2norm)
            if l2norm:  # like MI355XFlatIndex.load: a stored "L2norm,Flat" index is FAISS's object
                sh.l2norm_form = self._l2norm_form or os.environ.get("MQ_KNN_L2NORM_FORM", "faiss")
        self.ntotal, self.d = n, (d or None)
        return self

    def search_device(self, queries, k):
        import torch
        if k < 1:
            raise ValueError("k must be >= 1")
        if k > MAX_K:
            raise NotImplementedError(f"k={k} > {MAX_K}")
        nq = queries.shape[0]
        dev0 = self.shards[0]._torch_device
        rec_bytes, _ = record_layout(nq, k)
        gathered = torch.empty(self.world * rec_bytes, dtype=torch.uint8, device=dev0)
        records = []
        for r, sh in enumerate(self.shards):  # every device gets its scan enqueued before any result is collected
            dev = sh._torch_device
            with torch.cuda.device(dev):
                q = queries.to(dev, non_blocking=True)
                # a shard on the merging device writes its record in place; the others are copied in afterwards
                record = gathered[r * rec_bytes:(r + 1) * rec_bytes] if dev == dev0 else \
                    torch.empty(rec_bytes, dtype=torch.uint8, device=dev)
                D, I = _record_views(record, nq, k)
                if sh.ntotal == 0:
                    _fill_empty(D, I, self.metric_type)
                else:
                    sh.search_device(q, k, out=(D, I))
                records.append(record)
        for r, record in enumerate(records):
            if record.device != dev0:
                gathered[r * rec_bytes:(r + 1) * rec_bytes].copy_(record, non_blocking=True)  # peer-to-peer over xGMI
        return _hip_merge_records(gathered, self.world, nq, k, self.metric_type, self.tie_order)

    def save(self, file, storage_options=None, format=None):
        path = os.fspath(file)
        d = int(self.d or 0)
        with open(path, "wb") as f:
            f.write(index_file_header(self.ntotal, d, self.metric_type, self.do_l2norm, index_file_format(path, format)))
            for sh in self.shards:
                if sh.ntotal:
                    f.write(np.ascontiguousarray(sh.reconstruct_n(), dtype=np.float32).tobytes())


def visible_gpus():
    import torch
    return list(range(torch.cuda.device_count()))


def make_flat_index(device=None, string_factory=None, metric_type=None, screen=None, tie_order=None, l2norm_form=None):
    """The index ``KnowledgeBase.add_or_load_faiss_index`` builds, chosen as the reference's ``device``
    key is documented (datasets/search.py:315-347: int >= 0 -> that GPU, int < 0 -> all GPUs, list ->
    those GPUs; None = CPU FAISS there, the process's current GPU here):

    * a ``torch.distributed`` job with more than one rank (one process per GPU): ``ShardedFlatIndex`` over
      the default group, each rank's shard on its current device, whatever ``device`` says;
    * ``device=-1`` / a list of several GPUs in a single process: ``LocalShardsFlatIndex``;
    * otherwise one ``MI355XFlatIndex``."""
    import torch.distributed as dist
    kw = dict(string_factory=string_factory, metric_type=metric_type, screen=screen)
    if tie_order is not None:  # only forwarded when asked for: "id_asc" is every class's default
        kw["tie_order"] = tie_order
    if l2norm_form is not None:  # arithmetic of the "L2norm," prefix (viquae_amd.index.MI355XFlatIndex)
        kw["l2norm_form"] = l2norm_form
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return ShardedFlatIndex(device=None, **kw)
    if isinstance(device, int) and not isinstance(device, bool) and device < 0:
        gpus = visible_gpus()
        if len(gpus) > 1:
            return LocalShardsFlatIndex(gpus, **kw)
        device = 0
    elif isinstance(device, (list, tuple)) and len(device) > 1:
        return LocalShardsFlatIndex(list(device), **kw)
    return MI355XFlatIndex(device=device, **kw)
