"""Row-sharded exact index: one process per GPU, KB rows split contiguously across ranks.

New functionality with no reference counterpart (the reference's search path has no collective,
SURVEY.md section 2 "Collective call sites"); it must return exactly what one MI355XFlatIndex over
the whole matrix returns.  Per search:

  1. every rank scans its own shard for ALL queries (csrc/knn.hip) with global ids
     (``id_offset`` = first global row of the shard);
  2. ONE all-gather (RCCL over xGMI when the backend is "nccl") of the per-shard ``[nq,k]`` scores
     (fp32) and ids (int64) -- 12 B per entry, 4.9 MB per rank at nq=4096, k=100: latency-bound,
     never link-bound;
  3. every rank merges the ``world`` sorted lists per query (shard_merge_kernel) with the same
     (score, id) order as the single-GPU path, so "lower id wins ties" holds across shards.

``local_index`` / ``merge_fn`` are injection points used by the CPU (gloo) tests, which cannot run
HIP kernels; the defaults are the HIP implementations and there is no CPU fallback.
"""
from typing import Optional

import numpy as np

from .index import BatchedSearchResults, BaseIndex, SearchResults, MI355XFlatIndex, METRIC_L2


def shard_bounds(n_total: int, world: int, rank: int):
    """Contiguous row range [lo, hi) of ``rank``: ceil(N/world) rows per shard, rounded up to a
    multiple of 64 (the panel height) so every shard but the last holds whole panels."""
    per = -(-n_total // world)
    per = -(-per // 64) * 64
    lo = min(rank * per, n_total)
    hi = min(lo + per, n_total)
    return lo, hi


def _hip_merge(Ds, Is, metric):
    """[W,nq,k] CUDA tensors -> merged (D, I) through mq_topk_merge_f32."""
    import torch
    from . import _lib
    lib = _lib.load()
    W, nq, k = Ds.shape
    D = torch.empty((nq, k), dtype=torch.float32, device=Ds.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=Ds.device)
    with torch.cuda.device(Ds.device):
        _lib.check(lib.mq_topk_merge_f32(Ds.data_ptr(), Is.data_ptr(), W, nq, k, int(metric), D.data_ptr(),
                                         I.data_ptr(), torch.cuda.current_stream(Ds.device).cuda_stream),
                   "mq_topk_merge_f32")
    return D, I


class ShardedFlatIndex(BaseIndex):
    def __init__(self, string_factory: Optional[str] = None, metric_type: Optional[int] = None, group=None,
                 local_index=None, merge_fn=None, device=None):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.metric_type = METRIC_L2 if metric_type is None else int(metric_type)
        self.string_factory = string_factory
        self.local = local_index if local_index is not None else MI355XFlatIndex(
            device=device, string_factory=string_factory, metric_type=metric_type)
        self.merge_fn = merge_fn or _hip_merge
        self.ntotal = 0

    # ---------------------------------------------------------------- construction
    def add_global(self, vectors):
        """Every rank is handed the same [N,d] matrix (e.g. a memory-mapped Arrow column) and keeps
        rows shard_bounds(N, world, rank)."""
        n = len(vectors)
        lo, hi = shard_bounds(n, self.world, self.rank)
        self.local.id_offset = lo
        if hi > lo:
            self.local.add(np.asarray(vectors[lo:hi], dtype=np.float32))
        self.ntotal = n

    def add_local(self, rows, id_offset: int, n_total: int):
        """This rank's shard, already selected by the caller (rows numpy or device tensor)."""
        self.local.id_offset = int(id_offset)
        self.local.add(rows)
        self.ntotal = int(n_total)

    def add_vectors(self, vectors, column: Optional[str] = None, **kwargs):
        if column is not None:
            from .index import iter_arrow_column
            n = len(vectors)
            lo, hi = shard_bounds(n, self.world, self.rank)
            self.local.id_offset = lo
            seen = 0
            pend = None
            for block in iter_arrow_column(vectors, column):
                b_lo, b_hi = seen, seen + block.shape[0]
                seen = b_hi
                s, e = max(lo, b_lo), min(hi, b_hi)
                if s < e:
                    part = block[s - b_lo:e - b_lo]
                    pend = part if pend is None else np.concatenate([pend, part])
                    full = (pend.shape[0] // 64) * 64
                    if full:
                        self.local.add(pend[:full], total_hint=hi - lo)
                        pend = pend[full:]
            if pend is not None and pend.shape[0]:
                self.local.add(pend, total_hint=hi - lo)
            self.ntotal = n
        else:
            self.add_global(vectors)

    # ---------------------------------------------------------------- search
    def search_device(self, queries, k):
        """queries replicated on every rank (same tensor) -> merged (D, I) on every rank."""
        import torch
        import torch.distributed as dist
        Dl, Il = self.local.search_device(queries, k)
        if self.world == 1:
            return Dl, Il
        nq = Dl.shape[0]
        # rank-major concatenation along dim 0 ([world*nq, k]); viewed as [world, nq, k] for the merge
        Ds = torch.empty((self.world * nq, k), dtype=Dl.dtype, device=Dl.device)
        Is = torch.empty((self.world * nq, k), dtype=Il.dtype, device=Il.device)
        dist.all_gather_into_tensor(Ds, Dl.contiguous(), group=self.group)
        dist.all_gather_into_tensor(Is, Il.contiguous(), group=self.group)
        return self.merge_fn(Ds.view(self.world, nq, k), Is.view(self.world, nq, k), self.metric_type)

    def search_batch(self, queries, k: int = 10, **kwargs) -> BatchedSearchResults:
        import torch
        queries = np.asarray(queries)
        if len(queries.shape) != 2:
            raise ValueError("Shape of query must be 2D")
        dev = getattr(self.local, "_torch_device", None) or "cpu"
        q = torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32)).to(dev)
        D, I = self.search_device(q, k)
        return BatchedSearchResults(D.cpu().numpy(), I.cpu().numpy().astype(int))

    def search(self, query, k: int = 10, **kwargs) -> SearchResults:
        query = np.asarray(query)
        if len(query.shape) != 1 and (len(query.shape) != 2 or query.shape[0] != 1):
            raise ValueError("Shape of query is incorrect, it has to be either a 1D array or 2D (1, N)")
        scores, indices = self.search_batch(query.reshape(1, -1), k)
        return SearchResults(scores[0], indices[0].astype(int))
