"""MI355XFlatIndex -- exact (brute-force) dense index resident in HBM.

Drop-in for ``datasets.search.FaissIndex`` (site-packages/datasets/search.py:225-425) restricted to
what the reference uses: ``string_factory`` in {None, "Flat", "L2norm,Flat"}, ``metric_type`` in
{None, 0 (inner product), 1 (L2)} (meerqat/ir/search.py:207-249, experiments/ir/**/config.json).
It subclasses ``datasets.search.BaseIndex`` so that ``Dataset.search_batch`` /
``get_nearest_examples_batch`` (datasets/search.py:713-784) work once the object is registered in
``dataset._indexes``.

All arithmetic happens in libmeerqat_hip.so (csrc/knn.hip) through the C ABI; torch only owns the
device buffers and the stream.  No CPU fallback.
"""
import os
import struct
from pathlib import PurePath
from typing import Optional, Union

import numpy as np

try:  # datasets is the reference's host framework; BaseIndex is the seam
    from datasets.search import BaseIndex, BatchedSearchResults, SearchResults
except Exception:  # pragma: no cover - datasets is installed in the target image
    from typing import NamedTuple

    class BaseIndex:  # type: ignore
        pass

    class SearchResults(NamedTuple):  # type: ignore
        scores: list
        indices: list

    class BatchedSearchResults(NamedTuple):  # type: ignore
        total_scores: list
        total_indices: list

from . import _lib

METRIC_INNER_PRODUCT = 0  # faiss.METRIC_INNER_PRODUCT
METRIC_L2 = 1  # faiss.METRIC_L2
METRIC_IP_CENTRED = 2  # MQ_METRIC_IP_CENTRED: the inner product behind the centred-query screen (screen entry points only)
MAX_K = 2048  # MQ_KNN_MAX_K; the screened search serves k <= 224 itself, k <= 1792 over row ranges, beyond: ceil(k / 128) exact scans (include/meerqat_hip.h)
FLAG_L2NORM_QUERIES, FLAG_TIE_ID_DESC, MERGE_TIE_ID_DESC = 1, 2, 0x100  # MQ_KNN_FLAG_*, MQ_MERGE_TIE_ID_DESC
FLAG_L2NORM_FAISS = 4                            # MQ_KNN_FLAG_L2NORM_FAISS
FLAG_PHASE_FRONT, FLAG_PHASE_TAIL = 8, 16        # MQ_KNN_FLAG_PHASE_*: the two halves of one screened search
L2NORM_FORMS = {"numpy": 1, "faiss": 2}         # MQ_L2NORM_NUMPY, MQ_L2NORM_FAISS
TIE_ORDERS = ("id_asc", "id_desc")
_MAGIC = b"MQFLAT01"
_UPLOAD_ROWS = 1 << 16  # rows per host->device staging copy (multiple of 64)
_QUERY_CHUNK = 1 << 14
_SCREEN_QUERY_CHUNK = 1 << 12
_PINNED_IO_MAX_QUERIES = 1 << 15  # larger batches use one-off pageable copies (their transfer time is small beside the scan)


L2_DIRECT_BELOW = 20  # faiss::distance_compute_blas_threshold (include/meerqat_hip.h, MQ_KNN_L2_DIRECT_BELOW)


def query_chunks(nq, chunk):
    """[(start, stop)] pieces of at most ``chunk`` queries, one C-ABI call each.  FAISS picks the form of the L2
    distance from the size of the WHOLE batch (direct sum below 20 queries, BLAS form otherwise) and the C ABI
    from the size of the call, so a batch of 20 or more queries is never cut into a piece of fewer than 20."""
    bounds = [(s, min(s + chunk, nq)) for s in range(0, nq, chunk)]
    if len(bounds) > 1 and bounds[-1][1] - bounds[-1][0] < L2_DIRECT_BELOW and chunk >= 4 * L2_DIRECT_BELOW:
        cut = bounds[-1][0] - 2 * L2_DIRECT_BELOW
        bounds[-2] = (bounds[-2][0], cut)
        bounds[-1] = (cut, nq)
    return bounds


def parse_string_factory(string_factory):
    """Returns do_l2norm for the factories the reference ships; raises for anything else.

    "Flat" -> IndexFlat; "L2norm,Flat" -> IndexPreTransform(NormalizationTransform(d, 2), IndexFlat)
    (FAISS index_factory grammar).  Approximate indexes (IVF/PQ/HNSW ...) are not exact search and
    are not provided."""
    if string_factory is None:
        return False
    parts = [p.strip() for p in string_factory.split(",") if p.strip()]
    l2norm = False
    for p in parts[:-1]:
        if p == "L2norm":
            l2norm = True
        else:
            raise ValueError(f"Unsupported FAISS factory component '{p}' in '{string_factory}' "
                             "(MI355XFlatIndex provides exact 'Flat' and 'L2norm,Flat')")
    if not parts or parts[-1] != "Flat":
        raise ValueError(f"Unsupported FAISS factory '{string_factory}' "
                         "(MI355XFlatIndex provides exact 'Flat' and 'L2norm,Flat')")
    return l2norm


def _resolve_device(device):
    """HF convention (datasets/search.py:315-347): None -> default device, int >= 0 -> that GPU.
    The reference's shipped configs all say ``"device": null`` (CPU FAISS); here that means "the
    GPU of this process" (LOCAL_RANK-aware) because there is no CPU path."""
    import torch
    _lib.require_gpu()
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    if isinstance(device, int):
        if device < 0:
            raise TypeError("device=-1 (all GPUs) is served by viquae_amd.sharded.make_flat_index / "
                            "LocalShardsFlatIndex; MI355XFlatIndex is a single-GPU shard")
        return torch.device("cuda", device)
    if isinstance(device, (list, tuple)):
        if len(device) == 1:
            return torch.device("cuda", int(device[0]))
        raise TypeError("a device list is served by viquae_amd.sharded.make_flat_index / LocalShardsFlatIndex")
    if isinstance(device, torch.device):
        return device
    raise TypeError(f"The argument type: {type(device)} is not expected. "
                    "Please pass in either nothing, a positive int, a negative int, or a list of positive ints.")


class MI355XFlatIndex(BaseIndex):
    """Exact IP / L2 index over an fp32 matrix held in HBM in the kernel's panel layout."""

    def __init__(self, device: Optional[Union[int, list]] = None, string_factory: Optional[str] = None,
                 metric_type: Optional[int] = None, custom_index=None, id_offset: int = 0, screen: Optional[bool] = None,
                 keep_panel: Optional[bool] = None, tie_order: Optional[str] = None, l2norm_form: Optional[str] = None):
        if custom_index is not None:
            raise ValueError("custom_index is a FAISS object; MI355XFlatIndex builds its own index")
        self.device = device
        self.string_factory = string_factory
        # FAISS: IndexFlat(d) and index_factory(d, "Flat") default to METRIC_L2 when no metric is given
        self.metric_type = METRIC_L2 if metric_type is None else int(metric_type)
        if self.metric_type not in (METRIC_INNER_PRODUCT, METRIC_L2):
            raise ValueError(f"Unsupported metric_type {metric_type} (0 = inner product, 1 = L2)")
        self.do_l2norm = parse_string_factory(string_factory)
        # Arithmetic of the "L2norm," prefix (include/meerqat_hip.h): "numpy" = x / sqrt(sum x^2), the reference's L2norm()
        # (meerqat/ir/search.py:43-46) and what its GPU work-around applies to the KB column when `device` is given (:238-244);
        # "faiss" = FAISS's NormalizationTransform (x * float(1.0 / sqrt(sum x^2)), rows of norm 0 left untouched), what
        # "L2norm,Flat" computes with `device: null` -- KnowledgeBase.add_or_load_faiss_index asks for it then.
        # MQ_KNN_L2NORM_FORM sets the default of a directly constructed index ("numpy").
        l2norm_form = l2norm_form or os.environ.get("MQ_KNN_L2NORM_FORM", "numpy")
        if l2norm_form not in L2NORM_FORMS:
            raise ValueError(f"l2norm_form must be one of {sorted(L2NORM_FORMS)}, got {l2norm_form!r}")
        self.l2norm_form = l2norm_form
        # Which of several EXACTLY tied rows is the better one: "id_asc" (default, the lower id) or "id_desc" (the higher
        # id), for membership at the k-th boundary and output order alike.  Both are this library's documented policies;
        # FAISS's own behaviour on exact ties depends on its version and on k (oracle/knn_oracle.c).
        # MQ_KNN_TIE_ORDER sets the default.
        tie_order = tie_order or os.environ.get("MQ_KNN_TIE_ORDER", "id_asc")
        if tie_order not in TIE_ORDERS:
            raise ValueError(f"tie_order must be one of {TIE_ORDERS}, got {tie_order!r}")
        self.tie_order = tie_order
        self.id_offset = int(id_offset)
        self.ntotal = 0
        self.d = None
        self._packed = None  # torch.float32 [padded_rows * padded_dim]
        self._sqnorm = None  # torch.float32 [padded_rows]
        self._capacity = 0
        self._ws = None
        self._torch_device = None
        # screened search (bf16 screening + exact re-scoring, same results, csrc/knn_screen.inc): inner-product
        # indexes only; costs 1.5x the shard's HBM footprint.  MQ_KNN_SCREEN=0/1 overrides the default.
        if screen is None:
            screen = os.environ.get("MQ_KNN_SCREEN", "1") != "0"
        self.screen = bool(screen)  # both metrics: the L2 screen ranks by q.x - ||x||^2/2 (two extra bf16 columns)
        # A screened index holds the fp32 rows row-major (re-scoring) and a bf16 copy (screening): 1.5x the matrix.  The
        # panel-layout fp32 copy only feeds the exact scan -- the fallback of query tiles whose screening buffers overflow --
        # which can read the row-major copy instead (same MFMA sequence, slower operand path), so it is not kept unless
        # asked for (keep_panel=True / MQ_KNN_KEEP_PANEL=1: 2.5x the matrix, full-speed fallback).
        if keep_panel is None:
            keep_panel = (not self.screen) or os.environ.get("MQ_KNN_KEEP_PANEL", "0") == "1"
        self.keep_panel = bool(keep_panel) or not self.screen
        self._rowmajor = None  # torch.float32 [capacity, d] (screened path only)
        self._bf16 = None      # torch.uint8 bf16 copy
        self._xmax2 = None     # torch.float32 [4 + d]: max ||x||^2, max ||xc - bf16(xc)||^2, max ||xc||^2, max |c . xc| (kept by
        #                        mq_knn_screen_prepare), then the centre (read by the search under MQ_METRIC_IP_CENTRED)
        self._center = None    # torch.float32 [d]: the vector the bf16 screening copy is centred on (a view of _xmax2[4:])
        self._screen_metric = self.metric_type   # what the mq_knn_screen_* entries are told: METRIC_IP_CENTRED (2) when the
        #                        queries are centred as well (decided at the first add, _prepare_screen)

    def __reduce__(self):
        # Like a FAISS GPU index: device-resident, not picklable.  It also keeps `datasets` from hashing the whole shard
        # (a device -> host copy of every buffer, ~9 s at 1.5M x 768) when it fingerprints a Dataset.map call whose function
        # captures the index (Searcher -> KnowledgeBase -> index): on this error it falls back to a random fingerprint.
        raise TypeError(f"{type(self).__name__} lives in HBM and cannot be pickled: use save() / load()")

    # ------------------------------------------------------------------ construction
    def _ensure_capacity(self, n_total, d, exact=True):
        import torch
        lib = _lib.load()
        if self._torch_device is None:
            self._torch_device = _resolve_device(self.device)
        if self.d is None:
            self.d = int(d)
        elif self.d != int(d):
            raise ValueError(f"dimension mismatch: index has d={self.d}, got {d}")
        cap = int(lib.mq_padded_rows(n_total))
        if cap <= self._capacity:
            return
        if not exact and self._capacity:
            # appending without a known total: grow by 1.5x so that repeated add() calls copy O(N) rows overall
            cap = int(lib.mq_padded_rows(max(n_total, self._capacity + self._capacity // 2)))
        dpad = int(lib.mq_padded_dim(self.d))
        new_packed = torch.zeros(cap * dpad, dtype=torch.float32, device=self._torch_device) if self.keep_panel else None
        new_sqnorm = torch.zeros(cap, dtype=torch.float32, device=self._torch_device)
        if self._sqnorm is not None and self.ntotal > 0:
            # panels are contiguous: the old buffer is a prefix of the new one
            if self.keep_panel:
                new_packed[: self._packed.numel()].copy_(self._packed)
            new_sqnorm[: self._sqnorm.numel()].copy_(self._sqnorm)
        if self.screen:
            new_rm = torch.empty((cap, self.d), dtype=torch.float32, device=self._torch_device)
            new_bf = torch.zeros(int(lib.mq_knn_screen_bytes(cap, self.d, self._screen_metric)), dtype=torch.uint8, device=self._torch_device)
            if self._rowmajor is not None and self.ntotal > 0:
                new_rm[: self.ntotal].copy_(self._rowmajor[: self.ntotal])
                new_bf[: self._bf16.numel()].copy_(self._bf16)
            self._rowmajor, self._bf16 = new_rm, new_bf
        self._packed, self._sqnorm, self._capacity = new_packed, new_sqnorm, cap

    def add(self, vecs, total_hint: Optional[int] = None):
        """Append rows (numpy [n,d] or a CUDA torch tensor), any number at a time like ``faiss.Index.add``.  The C ABI
        packs whole 64-row panels: when the rows already stored end inside a panel, that panel's stored rows are read
        back and re-packed together with the first new rows (bit-identical: they are stored after the "L2norm,"
        transform and are not transformed twice)."""
        import torch
        lib = _lib.load()
        if isinstance(vecs, torch.Tensor):
            if vecs.dim() != 2:
                raise ValueError("expected a 2-D matrix of vectors")
            n, d = vecs.shape
        else:
            vecs = np.asarray(vecs, dtype=np.float32)
            if vecs.ndim != 2:
                raise ValueError("expected a 2-D matrix of vectors")
            n, d = vecs.shape
        if n == 0:
            return
        if self.screen and self._xmax2 is None:
            self._prepare_screen(vecs[:_UPLOAD_ROWS], d)  # centre + screen metric: BEFORE the bf16 copy is sized
        self._ensure_capacity(max(self.ntotal + n, total_hint or 0), d, exact=bool(total_hint))
        stream = torch.cuda.current_stream(self._torch_device).cuda_stream
        if not self.keep_panel:
            return self._add_rows_only(vecs, n, stream)
        first = 0
        if self.ntotal % 64 != 0:
            # finish the open panel: stored tail rows (already transformed) + the first new rows (transformed here)
            floor = self.ntotal // 64 * 64
            first = min(n, 64 - (self.ntotal - floor))
            with torch.cuda.device(self._torch_device):
                tail = torch.empty((self.ntotal - floor, self.d), dtype=torch.float32, device=self._torch_device)
                _lib.check(lib.mq_unpack_rows_f32(self._packed.data_ptr(), self._capacity, self.d, floor, tail.shape[0],
                                                  tail.data_ptr(), stream), "mq_unpack_rows_f32")
                head = vecs[:first]
                head = (head if isinstance(head, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(head)))
                head = head.to(device=self._torch_device, dtype=torch.float32).contiguous().clone()
                if self.do_l2norm:
                    _lib.check(lib.mq_l2norm_rows_form_f32(head.data_ptr(), head.shape[0], self.d, L2NORM_FORMS[self.l2norm_form], stream),
                               "mq_l2norm_rows_form_f32")
                blk = torch.cat([tail, head]).contiguous()
                _lib.check(lib.mq_pack_rows_f32(blk.data_ptr(), blk.shape[0], self.d, floor, 0, self._packed.data_ptr(),
                                                self._capacity, self._sqnorm.data_ptr(), stream), "mq_pack_rows_f32")
                if self.screen:
                    _lib.check(lib.mq_knn_screen_prepare(self._packed.data_ptr(), self._sqnorm.data_ptr(), self._capacity, self.d,
                                                         self._screen_metric, floor, blk.shape[0], self._rowmajor.data_ptr(),
                                                         self._bf16.data_ptr(), self._xmax2.data_ptr(),
                                                         self._center.data_ptr() if self._center is not None else None, stream),
                               "mq_knn_screen_prepare")
                self.ntotal = floor + blk.shape[0]
                torch.cuda.current_stream(self._torch_device).synchronize()
        with torch.cuda.device(self._torch_device):
            for i in range(first, n, _UPLOAD_ROWS):
                part = vecs[i:i + _UPLOAD_ROWS]
                if isinstance(part, torch.Tensor):
                    dev = part.to(device=self._torch_device, dtype=torch.float32).contiguous()
                else:
                    part = np.ascontiguousarray(part)
                    if not part.flags.writeable:
                        part = part.copy()
                    dev = torch.from_numpy(part).to(self._torch_device, non_blocking=False)
                _lib.check(lib.mq_pack_rows_f32(dev.data_ptr(), dev.shape[0], self.d, self.ntotal, self._l2norm_arg(),
                                                self._packed.data_ptr(), self._capacity, self._sqnorm.data_ptr(),
                                                stream), "mq_pack_rows_f32")
                if self.screen:
                    _lib.check(lib.mq_knn_screen_prepare(self._packed.data_ptr(), self._sqnorm.data_ptr(), self._capacity, self.d,
                                                         self._screen_metric, self.ntotal, dev.shape[0], self._rowmajor.data_ptr(),
                                                         self._bf16.data_ptr(), self._xmax2.data_ptr(),
                                                         self._center.data_ptr() if self._center is not None else None, stream),
                               "mq_knn_screen_prepare")
                self.ntotal += dev.shape[0]
                # `dev` goes back to torch's caching allocator while the kernels may still be queued: they run on torch's
                # current stream, and the allocator only hands the block to later work of that same stream
            torch.cuda.current_stream(self._torch_device).synchronize()

    def _upload(self, part):
        import torch
        if isinstance(part, torch.Tensor):
            return part.to(device=self._torch_device, dtype=torch.float32).contiguous()
        part = np.ascontiguousarray(part)
        if not part.flags.writeable:
            part = part.copy()
        return torch.from_numpy(part).to(self._torch_device, non_blocking=False)

    def _add_rows_only(self, vecs, n, stream):
        """add() of a screened index without a panel copy: the rows go straight to the row-major store (any row offset, no
        open-panel bookkeeping) with the "L2norm," transform and ||x||^2 of the packing kernel, then to the bf16 copy."""
        import torch
        lib = _lib.load()
        with torch.cuda.device(self._torch_device):
            for i in range(0, n, _UPLOAD_ROWS):
                dev = self._upload(vecs[i:i + _UPLOAD_ROWS])
                _lib.check(lib.mq_knn_screen_add_rows_f32(
                    dev.data_ptr(), dev.shape[0], self.d, self.ntotal, self._l2norm_arg(), self._screen_metric, self._capacity,
                    self._sqnorm.data_ptr(), self._rowmajor.data_ptr(), self._bf16.data_ptr(), self._xmax2.data_ptr(),
                    self._center.data_ptr() if self._center is not None else None, stream), "mq_knn_screen_add_rows_f32")
                self.ntotal += dev.shape[0]  # `dev` is released stream-ordered (see add())
            torch.cuda.current_stream(self._torch_device).synchronize()

    def _prepare_screen(self, first_rows, d):
        """First add of a screened index: the centre of the bf16 copy (``_choose_center``) and whether the QUERIES are centred as
        well (``MQ_METRIC_IP_CENTRED``, include/meerqat_hip.h): with the inner product, a centre that carries a quarter or more of
        the rows' squared norm (image features: non-negative, far from centred -- the reference's 2048-d ``imagenet-RN50`` and
        1024-d ``clip-RN50`` columns) makes the margin follow ||q|| while the scores spread like ||q - c|| ||x - c||; the
        centred-query screen removes that at the price of two more bf16 columns.  It is chosen when those columns are free
        (d % 64 in 1 ... 62) or the index is wider than the streaming kernel's 768 columns anyway -- and, where they cost a K block
        (d = 64 ... 768, multiples of 64), when the centre carries three quarters of the squared norm; never for d = 767, where a
        13th K block would cost the one-query-tile search its streaming kernel (at d = 768 the two columns sit alone in that block
        and the streaming kernel takes the row term as fp32, round 6).  MQ_KNN_CENTER_QUERIES=0 / 1 overrides."""
        import torch
        if self._torch_device is None:
            self._torch_device = _resolve_device(self.device)
        dev = self._upload(first_rows)
        center = self._choose_center(dev)
        self._screen_metric = self.metric_type
        # (a first add of a handful of rows says nothing about a common component: their mean IS most of them)
        if center is not None and self.metric_type == METRIC_INNER_PRODUCT and dev.shape[0] >= 256:
            x = dev.to(torch.float32)
            if self.do_l2norm:
                x = torch.nn.functional.normalize(x, dim=1)
            x2 = float(torch.nan_to_num(x, nan=0.0, posinf=0.0, neginf=0.0).pow(2).sum(1).mean())
            share = float(center.pow(2).sum()) / x2 if x2 > 0 else 0.0
            dp_plain, dp_cols = (int(d) + 63) // 64, (int(d) + 2 + 63) // 64
            free = dp_cols == dp_plain
            # free columns or a width the tile kernel serves anyway: a quarter of the squared norms; otherwise (a K block more:
            # +8 ... 25 % of the scan at d = 64 ... 704) only when the common component dominates (3/4: shared : noise >= 1.7)
            want = share >= (0.25 if (free or int(d) > 768) else 0.75)
            env = os.environ.get("MQ_KNN_CENTER_QUERIES")
            if env is not None:
                want = env != "0"
            # d = 767: the 13th K block would cost the one-query-tile search its streaming kernel; d = 768: the two columns sit ALONE
            # in the 13th block and that kernel takes the row term as fp32 beside twelve (round 6, csrc/knn_small8.inc)
            if want and not (dp_plain <= 12 < dp_cols and int(d) % 64 != 0 and env is None):
                self._screen_metric = METRIC_IP_CENTRED
        self._xmax2 = torch.zeros(4 + int(d), dtype=torch.float32, device=self._torch_device)
        if center is not None:
            self._xmax2[4:].copy_(center)
            self._center = self._xmax2[4:]
        else:
            self._center = None

    def _l2norm_arg(self):
        """The `l2norm` argument of the row-ingest entry points: 0 or the MQ_L2NORM_* code of this index's arithmetic."""
        return L2NORM_FORMS[self.l2norm_form] if self.do_l2norm else 0

    def _choose_center(self, first_rows):
        """Centre of the bf16 screening copy: the mean of the first rows added (as stored, i.e. after "L2norm,").  Any
        fixed vector keeps the screen lossless -- q.(x - c) ranks like q.x -- and dense-retrieval embeddings share a large
        common component, so the rounding error (hence the margin, hence the candidates per query) then follows ||x - c||
        instead of ||x||.  Both metrics (the L2 screen ranks by q.x - ||x||^2/2); MQ_KNN_CENTER=0 disables it."""
        import torch
        if os.environ.get("MQ_KNN_CENTER", "1") == "0":
            return None
        x = first_rows.to(torch.float32)
        if self.do_l2norm:
            nrm = x.norm(dim=1, keepdim=True)
            x = x / (torch.where(nrm > 0, nrm, torch.ones_like(nrm)) if self.l2norm_form == "faiss" else nrm)
        c = torch.nan_to_num(x, nan=0.0, posinf=0.0, neginf=0.0).mean(dim=0)
        return c.contiguous() if bool(torch.isfinite(c).all()) else None

    def add_vectors(self, vectors, column: Optional[str] = None, batch_size: int = 1000,
                    train_size: Optional[int] = None, faiss_verbose: Optional[bool] = None):
        """Same signature as FaissIndex.add_vectors (datasets/search.py:255-313).

        ``vectors`` is a numpy matrix or a ``datasets.Dataset`` (then ``column`` names a
        ``list<float>`` column).  The reference walks the dataset 1000 rows at a time through
        Python lists; here the Arrow column is viewed as one contiguous fp32 buffer per chunk.
        ``batch_size`` / ``train_size`` / ``faiss_verbose`` are accepted for compatibility (a Flat
        index needs no training)."""
        if column is None:
            mat = np.asarray(vectors, dtype=np.float32)
            if mat.ndim != 2:
                raise ValueError("expected a 2-D matrix of vectors")
            self.add(mat, total_hint=self.ntotal + mat.shape[0])
            return
        n_total = len(vectors)
        added = 0
        for block in iter_arrow_column(vectors, column):
            # keep every add but the last a multiple of 64 rows
            self._pending = block if getattr(self, "_pending", None) is None else np.concatenate([self._pending, block])
            full = (self._pending.shape[0] // 64) * 64
            if full:
                self.add(self._pending[:full], total_hint=self.ntotal - added + n_total)
                added += full
                self._pending = self._pending[full:]
        if getattr(self, "_pending", None) is not None and self._pending.shape[0]:
            self.add(self._pending, total_hint=self.ntotal - added + n_total)
        self._pending = None

    # ------------------------------------------------------------------ search
    def _workspace(self, nbytes):
        import torch
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self._torch_device)
        self._last_ws = self._ws
        return self._ws

    def _second_workspace(self, nbytes):
        """The workspace of the odd chunks of a pipelined search (chunk i's second half reads its workspace while chunk i+1's
        scan fills the other one) + the stream those second halves run on."""
        import torch
        ws = getattr(self, "_ws2", None)
        if ws is None or ws.numel() < nbytes:
            self._ws2 = ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self._torch_device)
        if getattr(self, "_tail_stream", None) is None:
            self._tail_stream = torch.cuda.Stream(device=self._torch_device)
        self._last_ws = ws
        return ws

    def search_device(self, queries, k, out=None):
        """queries: CUDA float32 [nq,d] tensor on this index's device -> (D [nq,k] f32, I [nq,k] i64)
        CUDA tensors.  This is the raw hot path: one C-ABI search call per <= 4096 (screened) / 16384 queries.
        ``out`` = (D, I) preallocated contiguous CUDA tensors to write into (e.g. views of a shard record)."""
        import torch
        lib = _lib.load()
        if self._sqnorm is None:
            raise ValueError("the index is empty: call add_vectors first")
        if queries.dim() != 2 or queries.shape[1] != self.d:
            raise ValueError(f"Shape of query must be 2D with {self.d} columns, got {tuple(queries.shape)}")
        if k < 1:
            raise ValueError("k must be >= 1")
        if k > MAX_K:
            raise NotImplementedError(f"k={k} > {MAX_K} (MQ_KNN_MAX_K, FAISS-GPU's own limit)")
        nq = queries.shape[0]
        if self.ntotal == 0:
            raise ValueError("the index is empty: call add_vectors first")
        queries = queries.to(dtype=torch.float32).contiguous()
        if out is not None:
            D, I = out
            if (tuple(D.shape) != (nq, k) or tuple(I.shape) != (nq, k) or D.dtype != torch.float32 or I.dtype != torch.int64
                    or not D.is_contiguous() or not I.is_contiguous() or D.device != self._torch_device):
                raise ValueError("out must be contiguous (float32 [nq,k], int64 [nq,k]) tensors on the index's device")
        else:
            D = torch.empty((nq, k), dtype=torch.float32, device=self._torch_device)
            I = torch.empty((nq, k), dtype=torch.int64, device=self._torch_device)
        stream = torch.cuda.current_stream(self._torch_device).cuda_stream
        with torch.cuda.device(self._torch_device):
            # screened path: 16 query tiles x 16 KB slabs per call keep 256 stripe maxima per query (tightest thresholds)
            chunk = _SCREEN_QUERY_CHUNK if self.screen else _QUERY_CHUNK
            pieces = query_chunks(nq, chunk)
            if self.screen and len(pieces) > 1 and os.environ.get("MQ_KNN_TAIL_OVERLAP", "0") == "1":
                self._search_chunks_pipelined(lib, queries, pieces, k, D, I)
                return D, I
            for s, e in pieces:
                q = queries[s:e]
                self._last_call_nq = e - s  # screen_stats reads the workspace geometry of the LAST C-ABI call
                nb = int(lib.mq_knn_workspace_bytes_metric(self.ntotal, self.d, q.shape[0], k, self._screen_metric if self.screen else self.metric_type))
                ws = self._workspace(nb)
                Dq, Iq = D[s:e], I[s:e]
                flags = self._search_flags()
                if self.screen:
                    _lib.check(lib.mq_knn_search_screened_f32(
                        self._packed.data_ptr() if self._packed is not None else None, self._sqnorm.data_ptr(), self._rowmajor.data_ptr(), self._bf16.data_ptr(),
                        self._xmax2.data_ptr(), self.ntotal, self.d, q.data_ptr(), q.shape[0], k, self._screen_metric,
                        flags,
                        self.id_offset, Dq.data_ptr(), Iq.data_ptr(), ws.data_ptr(), ws.numel(), stream, None, None),
                        "mq_knn_search_screened_f32")
                else:
                    _lib.check(lib.mq_knn_search_f32(self._packed.data_ptr(), self._sqnorm.data_ptr(), self.ntotal, self.d,
                                                     q.data_ptr(), q.shape[0], k, self.metric_type, flags,
                                                     self.id_offset, Dq.data_ptr(), Iq.data_ptr(), ws.data_ptr(), ws.numel(),
                                                     stream), "mq_knn_search_f32")
        return D, I

    def _search_flags(self):
        flags = (FLAG_L2NORM_QUERIES if self.do_l2norm else 0) | (FLAG_TIE_ID_DESC if self.tie_order == "id_desc" else 0)
        if self.do_l2norm and self.l2norm_form == "faiss":
            flags |= FLAG_L2NORM_FAISS
        return flags

    def _search_chunks_pipelined(self, lib, queries, pieces, k, D, I):
        """A screened search of several query chunks: the scans run back to back on the caller's stream, and what follows
        the scan of chunk i (candidate selection, exact re-scoring, exact top-k, recomputation of flagged tiles: about a
        tenth of a chunk's time, mostly latency-bound launches of one workgroup per query) runs on a second stream under the
        scan of chunk i+1 (MQ_KNN_FLAG_PHASE_*, include/meerqat_hip.h).  Two workspaces alternate; the scan of chunk i+2 waits
        for the second half of chunk i before it reuses that one.  Same kernels on the same data in the same order per
        chunk: results are those of the serial loop bit for bit.  Opt-in (MQ_KNN_TAIL_OVERLAP=1): measured 34.43 -> 34.30 ms
        on 16,384 queries x 1.5M x 768, i.e. nothing -- the scan is ONE workgroup per CU holding the whole register file (4
        waves x 128 VGPRs per SIMD) and 138 of 160 KB LDS for its 7.8 ms, so no other workgroup becomes resident before it
        retires (profiles/r04_notes.md section 9)."""
        import torch
        main = torch.cuda.current_stream(self._torch_device)
        spaces, tail = self.pipeline_workspaces(max(e - s for s, e in pieces), k)
        done = []
        for c, (s, e) in enumerate(pieces):
            q, out, ws = queries[s:e], (D[s:e], I[s:e]), spaces[c & 1]
            if c >= 2:
                main.wait_event(done[c - 2])
            self.search_phase(q, k, out, ws, FLAG_PHASE_FRONT, main)
            scanned = torch.cuda.Event()
            scanned.record(main)
            tail.wait_event(scanned)
            self.search_phase(q, k, out, ws, FLAG_PHASE_TAIL, tail)
            ev = torch.cuda.Event()
            ev.record(tail)
            done.append(ev)
        for ev in done[-2:]:
            main.wait_event(ev)

    def pipeline_workspaces(self, nq, k):
        """-> ((workspace of even chunks, workspace of odd chunks), the stream second halves run on) for chunks of <= nq queries."""
        nb = int(_lib.load().mq_knn_workspace_bytes_metric(self.ntotal, self.d, int(nq), k, self._screen_metric))
        return (self._workspace(nb), self._second_workspace(nb)), self._tail_stream

    def search_phase(self, q, k, out, ws, phase, stream):
        """One half (FLAG_PHASE_FRONT / FLAG_PHASE_TAIL) of the screened search of ONE chunk (<= 4096 queries) on ``stream``;
        both halves take the same q, out = (D, I) and workspace, and the caller orders TAIL after FRONT."""
        self._last_call_nq, self._last_ws = q.shape[0], ws
        _lib.check(_lib.load().mq_knn_search_screened_f32(
            self._packed.data_ptr() if self._packed is not None else None, self._sqnorm.data_ptr(), self._rowmajor.data_ptr(),
            self._bf16.data_ptr(), self._xmax2.data_ptr(), self.ntotal, self.d, q.data_ptr(), q.shape[0], k, self._screen_metric,
            self._search_flags() | phase, self.id_offset, out[0].data_ptr(), out[1].data_ptr(), ws.data_ptr(), ws.numel(),
            stream.cuda_stream, None, None), "mq_knn_search_screened_f32")

    def search_batch(self, queries, k: int = 10, **kwargs) -> BatchedSearchResults:
        """FaissIndex.search_batch (datasets/search.py:369-385): numpy [nq,d] -> (scores f32 [nq,k],
        indices int [nq,k]), best first, -1 for unfilled slots."""
        import torch
        queries = np.asarray(queries)
        if len(queries.shape) != 2:
            raise ValueError("Shape of query must be 2D")
        if self._torch_device is None:
            self._torch_device = _resolve_device(self.device)
        nq = queries.shape[0]
        if nq == 0 or nq > _PINNED_IO_MAX_QUERIES:
            q = torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32)).to(self._torch_device)
            D, I = self.search_device(q, k)
            torch.cuda.current_stream(self._torch_device).synchronize()
            return BatchedSearchResults(D.cpu().numpy(), I.cpu().numpy().astype(int))
        # Host boundary of one batch (the reference sends 256 queries at a time): page-locked staging buffers kept by the
        # index, asynchronous copies on the search stream, ONE synchronisation -- instead of three blocking pageable copies.
        q_pin, q_dev, D_pin, I_pin = self._host_io(nq, queries.shape[1], k)
        np.copyto(q_pin[:nq].numpy(), queries, casting="same_kind" if queries.dtype.kind == "f" else "unsafe")
        stream = torch.cuda.current_stream(self._torch_device)
        with torch.cuda.device(self._torch_device):
            q_dev[:nq].copy_(q_pin[:nq], non_blocking=True)
            D, I = self.search_device(q_dev[:nq], k)
            D_pin[:nq].copy_(D, non_blocking=True)
            I_pin[:nq].copy_(I, non_blocking=True)
        stream.synchronize()
        return BatchedSearchResults(D_pin[:nq].numpy().copy(), I_pin[:nq].numpy().astype(int))

    def _host_io(self, nq, d, k):
        """Page-locked (q, D, I) staging + the device query buffer, grown geometrically and reused across calls."""
        import torch
        io = getattr(self, "_io", None)
        if io is None or io[0].shape[0] < nq or io[0].shape[1] != d or io[2].shape[1] != k:
            cap = max(256, 1 << (int(nq) - 1).bit_length())
            io = (torch.empty((cap, d), dtype=torch.float32).pin_memory(),
                  torch.empty((cap, d), dtype=torch.float32, device=self._torch_device),
                  torch.empty((cap, k), dtype=torch.float32).pin_memory(),
                  torch.empty((cap, k), dtype=torch.int64).pin_memory())
            self._io = io
        return io

    def search(self, query, k: int = 10, **kwargs) -> SearchResults:
        """FaissIndex.search (datasets/search.py:349-367)."""
        query = np.asarray(query)
        if len(query.shape) != 1 and (len(query.shape) != 2 or query.shape[0] != 1):
            raise ValueError("Shape of query is incorrect, it has to be either a 1D array or 2D (1, N)")
        scores, indices = self.search_batch(query.reshape(1, -1), k)
        return SearchResults(scores[0], indices[0].astype(int))

    def _stats_workspace(self):
        ws = getattr(self, "_last_ws", None)
        return self._ws if ws is None else ws

    def scan_kind(self, nq, k):
        """Which screening scan serves a ``search_device`` call of ``nq`` queries: "tile" (256 x 256 tiles), "stream" (one query
        tile, queries in registers: csrc/knn_small.inc) or "none" (exact rounds / FAISS's small-batch L2 form / exact index)."""
        if not self.screen or not self.ntotal:
            return "none"
        kind = int(_lib.load().mq_knn_screen_scan_kind(self.ntotal, self.d, int(nq), int(k), self._screen_metric))
        if kind < 0:
            raise ValueError(f"mq_knn_screen_scan_kind: invalid arguments (status {kind})")
        return ("none", "tile", "stream")[kind]

    # ------------------------------------------------------------------ persistence
    def screen_stats(self, nq, k):
        """(query tiles recomputed exactly, candidates re-scored, max per query, ...) of the LAST C-ABI call of the last
        screened search (a search of more than 4096 queries is several calls)."""
        import ctypes
        lib = _lib.load()
        if nq > _SCREEN_QUERY_CHUNK:
            # the size of the last piece search_device cut (query_chunks moves 40 queries into a short tail)
            nq = getattr(self, "_last_call_nq", None) or query_chunks(nq, _SCREEN_QUERY_CHUNK)[-1][1] - query_chunks(nq, _SCREEN_QUERY_CHUNK)[-1][0]
        out = (ctypes.c_int64 * 8)()
        _lib.check(lib.mq_knn_screen_stats(self.ntotal, self.d, nq, k, self._stats_workspace().data_ptr(), out,
                                           __import__("torch").cuda.current_stream(self._torch_device).cuda_stream))
        return tuple(int(x) for x in out)

    def reconstruct_n(self, start=0, n=None):
        """Stored rows (after any L2norm transform) as numpy [n,d]."""
        import torch
        lib = _lib.load()
        n = self.ntotal - start if n is None else n
        if self._packed is None:
            return self._rowmajor[start:start + n].cpu().numpy()
        out = torch.empty((n, self.d), dtype=torch.float32, device=self._torch_device)
        stream = torch.cuda.current_stream(self._torch_device).cuda_stream
        with torch.cuda.device(self._torch_device):
            _lib.check(lib.mq_unpack_rows_f32(self._packed.data_ptr(), self._capacity, self.d, start, n, out.data_ptr(),
                                              stream), "mq_unpack_rows_f32")
        torch.cuda.current_stream(self._torch_device).synchronize()
        return out.cpu().numpy()

    def save(self, file: Union[str, PurePath], storage_options: Optional[dict] = None, format: Optional[str] = None):
        """FaissIndex.save (datasets/search.py:387-397; the reference's ``save_path``, meerqat/ir/search.py:247-248).
        Two formats, both read back by :meth:`load` (see :func:`index_file_header`): "faiss" -- the file
        ``faiss.write_index`` produces for this index (IndexFlat, or IndexPreTransform + NormalizationTransform + IndexFlat
        for "L2norm,Flat"), readable by FAISS itself -- when the path ends in ``.faiss`` / ``.index`` or on request
        (``format="faiss"`` / MQ_INDEX_FORMAT=faiss); otherwise this build's own "mqflat": 8-byte magic, int64 N, int32 d,
        int32 metric, int32 l2norm, int32 reserved, then N*d fp32 row-major (rows as stored)."""
        rows = self.reconstruct_n() if self.ntotal else np.zeros((0, self.d or 0), np.float32)
        with open(os.fspath(file), "wb") as f:
            f.write(index_file_header(self.ntotal, self.d or 0, self.metric_type, self.do_l2norm, index_file_format(file, format)))
            f.write(rows.tobytes())

    @classmethod
    def load(cls, file: Union[str, PurePath], device=None, storage_options: Optional[dict] = None, tie_order=None, l2norm_form=None):
        """FaissIndex.load (datasets/search.py:399-416).  Reads this class's own files and the FAISS files
        the reference's ``save_path`` / ``dataset.save_faiss_index`` wrote for the factories it ships
        (IndexFlat, and IndexPreTransform(NormalizationTransform, IndexFlat) for "L2norm,Flat")."""
        path = os.fspath(file)
        n, d, metric, l2norm, data_off = read_index_file_header(path)
        if l2norm and l2norm_form is None:
            # a stored "L2norm,Flat" index is the reference's CPU-FAISS object (meerqat/ir/search.py:247-248 saves, :235 loads it):
            # its queries go through FAISS's own NormalizationTransform
            l2norm_form = os.environ.get("MQ_KNN_L2NORM_FORM", "faiss")
        idx = cls(device=device, string_factory="L2norm,Flat" if l2norm else "Flat", metric_type=metric, tie_order=tie_order,
                  l2norm_form=l2norm_form)
        if n:
            rows = np.fromfile(path, dtype=np.float32, count=n * d, offset=data_off).reshape(n, d)
            # rows were stored after normalisation: do not normalise twice on load
            idx.do_l2norm = False
            idx.add(rows, total_hint=n)
            idx.do_l2norm = bool(l2norm)
        else:
            idx.d = d or None
        return idx


def index_file_format(path, format=None):
    """"faiss" or "mqflat" for a file about to be written: the explicit ``format``, else MQ_INDEX_FORMAT, else by extension
    (``.faiss`` / ``.index``: what the reference's users call their ``save_path`` files)."""
    fmt = format or os.environ.get("MQ_INDEX_FORMAT")
    if fmt is None:
        fmt = "faiss" if str(path).endswith((".faiss", ".index")) else "mqflat"
    if fmt not in ("faiss", "mqflat"):
        raise ValueError(f"unknown index file format {fmt!r} (faiss | mqflat)")
    return fmt


def index_file_header(n, d, metric, l2norm, fmt):
    """The bytes in front of the row-major fp32 matrix of an index file (its length = the matrix's offset).  "faiss": what
    faiss/impl/index_write.cpp writes for IndexFlat / IndexPreTransform(NormalizationTransform(d, 2.0), IndexFlat), as
    published for faiss >= 1.7.1 (see _read_faiss_flat; no FAISS binary in this image: the writer is checked against this
    reader and against the hand-assembled files of tests/test_host_logic_cpu.py, not against FAISS itself)."""
    n, d, metric = int(n), int(d), int(metric)
    if fmt == "mqflat":
        return _MAGIC + struct.pack("<qiiii", n, d, metric, int(bool(l2norm)), 0)

    def hdr():  # write_index_header: d, ntotal, two dummies (1 << 20), is_trained, metric_type
        return struct.pack("<iqqq?i", d, n, 1 << 20, 1 << 20, True, metric)

    flat = (b"IxFI" if metric == METRIC_INNER_PRODUCT else b"IxF2") + hdr() + struct.pack("<Q", n * d)
    if not l2norm:
        return flat
    # IndexPreTransform: header, chain length, NormalizationTransform {"VNrm", norm, d_in, d_out, is_trained}, sub-index
    return b"IxPT" + hdr() + struct.pack("<i", 1) + b"VNrm" + struct.pack("<f", 2.0) + struct.pack("<ii?", d, d, True) + flat


def _read_faiss_flat(f, path):
    """FAISS index file (faiss/impl/index_write.cpp, index_read.cpp as published for faiss >= 1.7.1, the
    reference's pin; no FAISS binary exists in this image, so this reader is checked against files
    assembled by hand from that published layout, not against FAISS itself):

      IndexFlat             "IxFI" (inner product) | "IxF2" (L2) | "IxFl" (legacy), index header, then the
                            vectors as {size_t count_of_floats, fp32 data}
      index header          int d, int64 ntotal, int64 dummy, int64 dummy, bool is_trained, int metric_type
                            (+ float metric_arg when metric_type > 1)
      IndexPreTransform     "IxPT", index header, int n_transforms, the transforms, then the sub-index
      NormalizationTransform "VNrm", float norm, then int d_in, int d_out, bool is_trained

    Returns (n, d, metric, l2norm, offset of the fp32 matrix)."""
    def rd(fmt):
        size = struct.calcsize(fmt)
        b = f.read(size)
        if len(b) != size:
            raise ValueError(f"{path}: truncated FAISS index file")
        return struct.unpack(fmt, b)

    def header():
        d, ntotal, _, _, _trained, metric = rd("<iqqq?i")
        if metric > 1:
            rd("<f")
        return d, ntotal, metric

    l2norm = False
    fourcc = f.read(4)
    if fourcc == b"IxPT":
        header()
        (nt,) = rd("<i")
        for _ in range(nt):
            vt = f.read(4)
            if vt != b"VNrm":
                raise ValueError(f"{path}: FAISS pre-transform {vt!r} is not supported (only the NormalizationTransform "
                                 "of 'L2norm,Flat')")
            (norm,) = rd("<f")
            rd("<ii?")
            if norm != 2.0:
                raise ValueError(f"{path}: NormalizationTransform with norm {norm} (only L2 is supported)")
            l2norm = True
        fourcc = f.read(4)
    if fourcc not in (b"IxFI", b"IxF2", b"IxFl"):
        raise ValueError(f"{path}: FAISS index type {fourcc!r} is not an exact Flat index "
                         "(MI355XFlatIndex provides 'Flat' and 'L2norm,Flat')")
    d, n, metric = header()
    if metric not in (METRIC_INNER_PRODUCT, METRIC_L2):
        raise ValueError(f"{path}: unsupported FAISS metric_type {metric}")
    (count,) = rd("<Q")
    if count != n * d:
        raise ValueError(f"{path}: FAISS IndexFlat holds {count} floats, expected {n} x {d}")
    return int(n), int(d), int(metric), l2norm, f.tell()


def read_index_file_header(path):
    """(n, d, metric, l2norm, byte offset of the row-major fp32 matrix) of an index file: this build's
    "MQFLAT01" format or a FAISS IndexFlat / IndexPreTransform file (see _read_faiss_flat)."""
    with open(path, "rb") as f:
        magic = f.read(8)
        if magic == _MAGIC:
            n, d, metric, l2norm, _ = struct.unpack("<qiiii", f.read(24))
            return int(n), int(d), int(metric), bool(l2norm), 32
        if magic[:4] in (b"IxFI", b"IxF2", b"IxFl", b"IxPT"):
            f.seek(0)
            return _read_faiss_flat(f, path)
    raise ValueError(f"{path} is neither an MI355XFlatIndex file nor a FAISS Flat index file")


def iter_arrow_column(dataset, column, start=0, stop=None):
    """Yields numpy [rows,d] fp32 blocks of a ``list<float>`` column without going through Python
    lists (the reference's add loop: datasets/search.py:311-313 -> list -> ndarray).  ``start`` / ``stop``
    restrict the walk to a row range (a rank's shard): only those rows are decoded or read."""
    import pyarrow as pa
    table = getattr(dataset, "data", None)
    indices = getattr(dataset, "_indices", None)
    n_rows = len(dataset)
    stop = n_rows if stop is None else min(int(stop), n_rows)
    start = max(0, int(start))
    if start >= stop:
        return
    if table is None or indices is not None:
        # selected/shuffled dataset: fall back to formatted access, still batched
        ds = dataset.with_format("numpy", columns=[column])
        for i in range(start, stop, 1 << 15):
            block = ds[i:min(i + (1 << 15), stop)][column]
            yield np.ascontiguousarray(np.stack(block) if isinstance(block, list) else block, dtype=np.float32)
        return
    col = table.column(column)
    if start or stop < n_rows:
        col = col.slice(start, stop - start)  # zero-copy: the chunks outside the range are never touched
    for chunk in col.chunks:
        if len(chunk) == 0:
            continue
        if pa.types.is_fixed_size_list(chunk.type):
            d = chunk.type.list_size
            flat = chunk.flatten()
        else:
            offsets = chunk.offsets.to_numpy()
            d = int(offsets[1] - offsets[0])
            if not np.all(np.diff(offsets) == d):
                raise ValueError(f"column '{column}' holds vectors of different lengths")
            flat = chunk.flatten()
        if chunk.null_count:
            raise ValueError(f"column '{column}' holds null vectors: cannot be indexed")
        arr = flat.to_numpy(zero_copy_only=False)
        yield np.ascontiguousarray(arr.reshape(len(chunk), d), dtype=np.float32)
