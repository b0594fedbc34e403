"""ctypes front-end of oracle/knn_oracle.c (TEST INFRASTRUCTURE, see oracle/__init__.py).

Also holds ``knn_numpy_f64``: an independent float64 brute force used (a) to validate the C
restatement and (b) to judge free-form fp32 results by re-scoring ("hard
parts": two correct fp32 implementations may swap near-tied neighbours).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libknn_oracle.so")
_lib = None


def build(force=False):
    """Compile knn_oracle.c with gcc (a few seconds)."""
    src = os.path.join(_HERE, "knn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.c_void_p
        L.oracle_knn_f32.argtypes = [fp, ctypes.c_int64, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int64, fp, fp]
        L.oracle_knn_f32.restype = ctypes.c_int
        L.oracle_knn_f32_ex.argtypes = [fp, ctypes.c_int64, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int64, fp, fp]
        L.oracle_knn_f32_ex.restype = ctypes.c_int
        L.oracle_knn_f32_ex2.argtypes = [fp, ctypes.c_int64, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, fp, fp]
        L.oracle_knn_f32_ex2.restype = ctypes.c_int
        L.oracle_topk_merge_ex.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp]
        L.oracle_topk_merge_ex.restype = ctypes.c_int
        L.oracle_heap_init.argtypes = [ctypes.c_int, ctypes.c_int, fp, fp]
        L.oracle_heap_init.restype = None
        L.oracle_heap_add_block.argtypes = [fp, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, fp, fp]
        L.oracle_heap_add_block.restype = None
        L.oracle_heap_finish.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp, fp, fp]
        L.oracle_heap_finish.restype = None
        L.oracle_sgemm_mr.restype = ctypes.c_int
        L.oracle_pack_queries.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp]
        L.oracle_pack_queries.restype = None
        L.oracle_sgemm_nt.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, fp, ctypes.c_int64]
        L.oracle_sgemm_nt.restype = None
        L.oracle_heap_add_block_panels.argtypes = [fp, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, fp, fp]
        L.oracle_heap_add_block_panels.restype = None
        L.oracle_l2norm_rows_f32.argtypes = [fp, ctypes.c_int64, ctypes.c_int]
        L.oracle_l2norm_rows_f32.restype = None
        L.oracle_l2norm_rows_faiss_f32.argtypes = [fp, ctypes.c_int64, ctypes.c_int]
        L.oracle_l2norm_rows_faiss_f32.restype = None
        L.oracle_sqnorm_rows_f32.argtypes = [fp, ctypes.c_int64, ctypes.c_int, fp]
        L.oracle_sqnorm_rows_f32.restype = None
        L.oracle_topk_merge.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp]
        L.oracle_topk_merge.restype = ctypes.c_int
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_set_num_threads.argtypes = [ctypes.c_int]
        L.oracle_set_num_threads.restype = None
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    """OpenMP threads of the following oracle calls (0 / None: leave as is)."""
    if n:
        lib().oracle_set_num_threads(int(n))


def l2norm_rows(x, form="numpy"):
    """Row L2-normalisation with the oracle's fixed arithmetic (returns a new array).  ``form``: "numpy" = the reference's
    ``L2norm()`` (x / sqrt(sum x^2), a zero row becomes NaN), "faiss" = FAISS's ``fvec_renorm_L2`` (x * float(1.0 / sqrt(sum x^2)),
    rows whose squared norm is not > 0 left untouched) -- see oracle/knn_oracle.c."""
    x = _f32(x).copy()
    if x.ndim != 2:
        raise ValueError("expected a 2-D array")
    if form == "numpy":
        lib().oracle_l2norm_rows_f32(x.ctypes.data, x.shape[0], x.shape[1])
    elif form == "faiss":
        lib().oracle_l2norm_rows_faiss_f32(x.ctypes.data, x.shape[0], x.shape[1])
    else:
        raise ValueError(f"l2norm form must be 'numpy' or 'faiss', got {form!r}")
    return x


def sqnorm_rows(x):
    x = _f32(x)
    out = np.empty(x.shape[0], dtype=np.float32)
    lib().oracle_sqnorm_rows_f32(x.ctypes.data, x.shape[0], x.shape[1], out.ctypes.data)
    return out


FLT_MAX = float(np.finfo(np.float32).max)  # FAISS's heap neutral value: what an unfilled slot reports (+ for L2, - for IP)
L2_FORMS = {"auto": 0, "expanded": 1, "direct": 2}
TIE_ORDERS = {"id_asc": 0, "id_desc": 1}  # knn_oracle.c header: this library's documented tie policy, not FAISS's


def knn(X, Q, k, metric=0, id_offset=0, l2norm=False, l2_form="auto", tie_order="id_asc", l2norm_form="numpy"):
    """Brute-force top-k of Q against X; returns (D f32 [nq,k], I i64 [nq,k]).

    ``l2norm=True`` applies the "L2norm," transform to both sides first (FAISS
    NormalizationTransform on add and on search) in the arithmetic ``l2norm_form`` names ("numpy" | "faiss", see l2norm_rows).
    ``l2_form`` (metric 1): "auto" = FAISS's rule
    (fewer than 20 queries: direct sum of (q-x)^2; otherwise ||q||^2+||x||^2-2<q,x> clamped at 0),
    or force "expanded" / "direct".  ``tie_order``: "id_asc" (default: among exactly equal scores the lower id
    is better) or "id_desc" (the higher id is better), for membership at the k-th boundary and output order alike."""
    X, Q = _f32(X), _f32(Q)
    if Q.ndim != 2 or X.ndim != 2 or Q.shape[1] != X.shape[1]:
        raise ValueError("shape mismatch")
    if l2norm:
        X, Q = l2norm_rows(X, l2norm_form), l2norm_rows(Q, l2norm_form)
    nq, d = Q.shape
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    rc = lib().oracle_knn_f32_ex2(X.ctypes.data, X.shape[0], d, Q.ctypes.data, nq, k, int(metric), L2_FORMS[l2_form],
                                  TIE_ORDERS[tie_order], int(id_offset), D.ctypes.data, I.ctypes.data)
    if rc != 0:
        raise ValueError(f"oracle_knn_f32_ex2 failed: {rc}")
    return D, I


def topk_merge(Ds, Is, metric=0, tie_order="id_asc"):
    """Merge [nshards,nq,k] per-shard lists (global ids) into the k best per query."""
    Ds = _f32(Ds)
    Is = np.ascontiguousarray(Is, dtype=np.int64)
    ns, nq, k = Ds.shape
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    rc = lib().oracle_topk_merge_ex(Ds.ctypes.data, Is.ctypes.data, ns, nq, k, int(metric), TIE_ORDERS[tie_order],
                                    D.ctypes.data, I.ctypes.data)
    if rc != 0:
        raise ValueError(f"oracle_topk_merge failed: {rc}")
    return D, I


def knn_blas(X, Q, k, metric=0, block=1024, threads=None, query_block=4096, backend="torch"):
    """The same search ORGANISED the way FAISS's IndexFlat runs it for 20 or more queries
    (faiss/utils/distances.cpp, exhaustive_inner_product_blas / exhaustive_L2sqr_blas): blocks of ``query_block`` (4096 =
    distance_compute_blas_query_bs) queries x ``block`` (1024 = distance_compute_blas_database_bs) database rows go through
    the host BLAS's sgemm (torch.mm = MKL here, all cores, into one reused buffer), then FAISS's result-handler rule in
    C / OpenMP over the queries (knn_oracle.c::oracle_heap_add_block): a score enters a query's heap only if it beats the
    heap's current k-th best strictly.  Scores carry the BLAS library's summation order, not the fmaf chain of knn(): this
    is bench.py's fast CPU leg, compared with knn() only up to float64 near-ties (tests/test_oracle_cpu.py).
    ``backend``: what runs the sgemm -- "torch" (MKL in this image), "numpy" (its bundled OpenBLAS) or "c" (the register-blocked
    AVX-512 / AVX2 kernel of knn_oracle.c, oracle_sgemm_nt: MKL takes a slow code path on AMD hosts and OpenBLAS stops at 64
    threads, so bench.py calibrates the three and keeps the fastest)."""
    import torch
    use_numpy = backend == "numpy"
    use_c = backend == "c"
    if threads and use_c:
        set_num_threads(threads)
    elif threads:
        torch.set_num_threads(int(threads))
    X, Q = _f32(X), _f32(Q)
    Xt, Qt = torch.from_numpy(X), torch.from_numpy(Q)
    nq, N = Qt.shape[0], Xt.shape[0]
    L = lib()
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    S = torch.empty((min(query_block, max(nq, 1)), block), dtype=torch.float32)
    for q0 in range(0, nq, query_block):
        qb = Qt[q0:q0 + query_block]
        n = qb.shape[0]
        hv = np.empty((n, k), dtype=np.float32)
        hi = np.empty((n, k), dtype=np.int64)
        L.oracle_heap_init(n, k, hv.ctypes.data, hi.ctypes.data)
        qn = (qb * qb).sum(1) if metric == 1 else None
        Sp = None
        if use_c:
            MR = int(L.oracle_sgemm_mr())
            Qp = np.empty(((n + MR - 1) // MR) * Qt.shape[1] * MR, dtype=np.float32)
            qc = np.ascontiguousarray(qb.numpy())
            L.oracle_pack_queries(qc.ctypes.data, n, Qt.shape[1], Qp.ctypes.data)
        for s in range(0, N, block):
            xb = Xt[s:s + block]
            nb = xb.shape[0]
            Sb = S[:n, :nb] if nb == block else torch.empty((n, nb), dtype=torch.float32)
            if use_c and metric == 0:
                # inner product: panel-major scores (contiguous per thread of the sgemm), read as they are by the heap pass
                xc = np.ascontiguousarray(xb.numpy())
                npan = (nb + 31) // 32
                if Sp is None or Sp.size < npan * n * 32:
                    Sp = np.empty(npan * n * 32, dtype=np.float32)
                L.oracle_sgemm_nt(Qp.ctypes.data, n, Qt.shape[1], xc.ctypes.data, nb, Sp.ctypes.data, -1)
                L.oracle_heap_add_block_panels(Sp.ctypes.data, n, s, nb, k, hv.ctypes.data, hi.ctypes.data)
                continue
            if use_c:
                xc = np.ascontiguousarray(xb.numpy())
                L.oracle_sgemm_nt(Qp.ctypes.data, n, Qt.shape[1], xc.ctypes.data, nb, Sb.data_ptr(), Sb.stride(0))
            elif use_numpy:
                np.matmul(qb.numpy(), xb.numpy().T, out=Sb.numpy())
            else:
                torch.mm(qb, xb.T, out=Sb)
            if metric == 1:  # FAISS: ||q||^2 + ||x||^2 - 2 <q, x>, clamped at 0; the heap keeps the goodness -distance
                Sb.mul_(-2.0).add_(qn[:, None]).add_((xb * xb).sum(1)[None, :]).clamp_min_(0.0).neg_()
            L.oracle_heap_add_block(Sb.data_ptr(), n, Sb.stride(0), s, nb, k, hv.ctypes.data, hi.ctypes.data)
        L.oracle_heap_finish(n, k, int(metric), hv.ctypes.data, hi.ctypes.data, D[q0:q0 + n].ctypes.data, I[q0:q0 + n].ctypes.data)
    return D, I


# ----------------------------------------------------------------------------------------------
# independent cross-checks (slow; small cases only)
# ----------------------------------------------------------------------------------------------
def knn_numpy_f64(X, Q, k, metric=0):
    """float64 scores + (score, id) lexicographic order. Independent of knn_oracle.c."""
    X64, Q64 = np.asarray(X, np.float64), np.asarray(Q, np.float64)
    if metric == 1:
        S = ((Q64[:, None, :] - X64[None, :, :]) ** 2).sum(2) if Q64.shape[0] * X64.shape[0] * X64.shape[1] <= 1 << 24 else \
            np.maximum((Q64 ** 2).sum(1)[:, None] + (X64 ** 2).sum(1)[None, :] - 2 * (Q64 @ X64.T), 0)
        order = np.lexsort((np.broadcast_to(np.arange(S.shape[1]), S.shape), S), axis=1)
    else:
        S = Q64 @ X64.T
        order = np.lexsort((np.broadcast_to(np.arange(S.shape[1]), S.shape), -S), axis=1)
    I = order[:, :k]
    D = np.take_along_axis(S, I, axis=1)
    if I.shape[1] < k:
        pad = k - I.shape[1]
        I = np.concatenate([I, -np.ones((I.shape[0], pad), np.int64)], 1)
        D = np.concatenate([D, np.full((D.shape[0], pad), FLT_MAX if metric == 1 else -FLT_MAX)], 1)
    return D, I.astype(np.int64)


def chain_scores_python(X, Q):
    """Pure-Python/numpy-scalar statement of the fmaf chain for TINY inputs: validates that the C
    file's vectorised loops really are the k-ordered fp32 fma chain. fma is emulated exactly in
    float64 (a*b of two fp32 is exact in fp64; one fp64 add then one rounding to fp32 is a correctly
    rounded fp32 fma unless the fp64 add itself is inexact AND lands on an fp32 rounding boundary --
    double rounding; inputs for this check are kept to small-exponent-range values where that cannot
    change the result measurably, and any mismatch is reported by the caller)."""
    X = np.asarray(X, np.float32)
    Q = np.asarray(Q, np.float32)
    out = np.zeros((Q.shape[0], X.shape[0]), np.float32)
    for i in range(Q.shape[0]):
        for j in range(X.shape[0]):
            acc = np.float32(0)
            for kk in range(X.shape[1]):
                acc = np.float32(np.float64(Q[i, kk]) * np.float64(X[j, kk]) + np.float64(acc))
            out[i, j] = acc
    return out
