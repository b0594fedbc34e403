"""CPU suite: the oracle against the committed golden vectors (minted by running the reference's own
KnowledgeBase, tools/make_golden.py) and against independent computations."""
import os

import numpy as np

FLT_MAX = np.finfo(np.float32).max  # FAISS heap neutral value reported by unfilled slots
import pytest

from oracle import knn as ok

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    z = np.load(os.path.join(GOLDEN, f"knn_{name}.npz"))
    return {k: z[k] for k in z.files}


def _cases(g):
    for key in g:
        if key.startswith("I_") and not key.startswith("I_none"):
            tag = key[2:]
            l2 = tag.startswith("l2norm_")
            parts = tag.replace("l2norm_", "").split("_")
            yield tag, l2, int(parts[0][1:]), int(parts[1][1:])


@pytest.mark.parametrize("name", ["random", "small_nq", "lattice", "ties", "lattice_small_nq", "ties_small_nq"])
def test_oracle_reproduces_golden(name):
    """SELF-MINTED goldens: tests/golden/knn_*.npz were produced by the REFERENCE's KnowledgeBase.search_batch run in the
    build container over tools/ref_import.py's stand-in for `faiss` -- and that stand-in computes with this very oracle (scores, tie
    rule, the FAISS-form "L2norm,").  What the files pin is therefore the reference's PLUMBING (list -> ndarray, host-side L2norm, the
    call into the index, None handling) and that the oracle has not drifted since they were minted; the arithmetic is pinned against
    independent float64 / torch.mm brute force on the order-independent files below, and against a real FAISS only where `import faiss`
    works (tests/test_faiss_parity_cpu.py, tools/compare_with_faiss.py: skipped in this image)."""
    g = _load(name)
    X, Q = g["X"].astype(np.float32), g["Q"].astype(np.float32)
    n = 0
    for tag, l2, metric, k in _cases(g):
        Qin = Q
        if l2:
            # the reference normalises queries on the host (ir/search.py:144-145) before the index does it again
            Qin = (Q / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float32)
        D, I = ok.knn(X, Qin, k, metric=metric, l2norm=l2, l2norm_form="faiss")  # "L2norm,Flat" with `device: null`
        assert np.array_equal(I, g[f"I_{tag}"]), tag
        assert np.array_equal(D, g[f"D_{tag}"]), tag
        n += 1
    assert n >= 2


@pytest.mark.parametrize("name", ["lattice", "ties"])
@pytest.mark.parametrize("metric", [0, 1])
def test_golden_lattice_is_order_independent(name, metric):
    """On integer data every fp32 summation order is exact: the golden must equal a float64 brute force
    and a torch.mm (BLAS order) brute force -- i.e. what FAISS IndexFlat returns with lower-id tie-break."""
    import torch
    g = _load(name)
    X, Q = g["X"].astype(np.float32), g["Q"].astype(np.float32)
    k = 100
    D64, I64 = ok.knn_numpy_f64(X, Q, k, metric)
    assert np.array_equal(g[f"I_m{metric}_k{k}"], I64)
    S = (torch.from_numpy(Q) @ torch.from_numpy(X).T).numpy()
    if metric == 1:
        S = (Q ** 2).sum(1)[:, None] + (X ** 2).sum(1)[None, :] - 2 * S
        order = np.lexsort((np.broadcast_to(np.arange(S.shape[1]), S.shape), S), axis=1)[:, :k]
    else:
        order = np.lexsort((np.broadcast_to(np.arange(S.shape[1]), S.shape), -S), axis=1)[:, :k]
    assert np.array_equal(order, g[f"I_m{metric}_k{k}"])
    assert np.array_equal(np.take_along_axis(S, order, 1), g[f"D_m{metric}_k{k}"])


@pytest.mark.parametrize("name", ["lattice_small_nq", "ties_small_nq"])
@pytest.mark.parametrize("metric", [0, 1])
def test_small_batch_goldens_are_order_independent(name, metric):
    """Fewer than 20 queries: FAISS's sequential path (direct sum of (q-x)^2 for L2).  On these integer fixtures every
    partial sum is exact in fp32, so the golden must equal an independent float64 brute force of that form."""
    g = _load(name)
    X, Q = g["X"].astype(np.float32), g["Q"].astype(np.float32)
    assert Q.shape[0] < 20
    D64, I64 = ok.knn_numpy_f64(X, Q, 100, metric)
    assert np.array_equal(g[f"I_m{metric}_k100"], I64)
    assert np.array_equal(g[f"D_m{metric}_k100"].astype(np.float64), D64)


def test_l2_form_follows_faiss_blas_threshold():
    """19 queries -> direct form, 20 -> expanded form (faiss::distance_compute_blas_threshold = 20).  A KB row equal to
    the query separates the forms: the direct sum is exactly 0, and distances to near-duplicates keep their relative
    accuracy instead of the cancellation error of ||q||^2 + ||x||^2 - 2<q,x>."""
    rng = np.random.default_rng(5)
    X = rng.standard_normal((500, 96)).astype(np.float32) * 7
    Q = X[:20].copy()
    X[100:120] = Q * np.float32(1 + 2 ** -12)            # near-duplicates of the queries
    Dd, Id = ok.knn(X, Q[:19], 3, metric=1)
    Da, Ia = ok.knn(X, Q[:19], 3, metric=1, l2_form="direct")
    assert np.array_equal(Dd, Da) and np.array_equal(Id, Ia)
    assert (Dd[:, 0] == 0).all() and (Id[:, 0] == np.arange(19)).all()
    De, Ie = ok.knn(X, Q, 3, metric=1)
    Df, If = ok.knn(X, Q, 3, metric=1, l2_form="expanded")
    assert np.array_equal(De, Df) and np.array_equal(Ie, If)
    D64, _ = ok.knn_numpy_f64(X, Q[:19], 3, 1)
    rel_direct = np.abs(Dd[:, 1] - D64[:, 1]) / D64[:, 1]
    rel_expanded = np.abs(Df[:19, 1] - D64[:, 1]) / D64[:, 1]
    assert rel_direct.max() < 1e-5 and rel_expanded.max() > 1e-3   # the forms are observably different


@pytest.mark.parametrize("metric", [0, 1])
def test_blas_leg_agrees_with_the_chain_oracle_up_to_near_ties(metric):
    """bench.py's fast CPU leg (sgemm blocks + top-k, FAISS's own organisation for >= 20 queries) against the fmaf-chain
    oracle: same neighbours except float64 near-ties, scores within fp32 summation error."""
    rng = np.random.default_rng(12)
    X = rng.standard_normal((5000, 96), dtype=np.float32)
    Q = rng.standard_normal((40, 96), dtype=np.float32)
    D, I = ok.knn(X, Q, 50, metric=metric)
    Db, Ib = ok.knn_blas(X, Q, 50, metric=metric, block=1024)
    assert np.allclose(D, Db, rtol=1e-4, atol=1e-3)
    D64, _ = ok.knn_numpy_f64(X, Q, 5000, metric)
    for q in range(len(Q)):
        for i in set(I[q]) ^ set(Ib[q]):
            s64 = dict(zip(ok.knn_numpy_f64(X, Q[q:q + 1], 5000, metric)[1][0], D64[q]))
            assert abs(s64[i] - D64[q][49]) < 1e-3


def test_free_form_golden_matches_float64_within_ties():
    """Free-form fp32 data: two correct fp32 implementations may swap near-ties; every index mismatch vs
    float64 must be such a near-tie (|score diff| tiny)."""
    g = _load("random")
    X, Q = g["X"], g["Q"]
    D64, I64 = ok.knn_numpy_f64(X, Q, 100, 0)
    I = g["I_m0_k100"]
    S = Q.astype(np.float64) @ X.astype(np.float64).T
    for q in np.nonzero((I != I64).any(1))[0]:
        for a, b in zip(I[q], I64[q]):
            if a != b:
                assert abs(S[q, a] - S[q, b]) <= 2 ** -18 * max(1.0, abs(S[q, a]))


def test_fma_chain_definition():
    rng = np.random.default_rng(5)
    X = rng.standard_normal((6, 41)).astype(np.float32)
    Q = rng.standard_normal((4, 41)).astype(np.float32)
    D, I = ok.knn(X, Q, 6)
    S = ok.chain_scores_python(X, Q)
    assert np.array_equal(np.take_along_axis(S, I, 1), D)


def test_fewer_rows_than_k_and_empty():
    X = np.eye(3, 8, dtype=np.float32)
    Q = np.ones((2, 8), np.float32)
    D, I = ok.knn(X, Q, 5)
    assert (I[:, 3:] == -1).all() and (D[:, 3:] == -FLT_MAX).all() and (I[:, :3] == [0, 1, 2]).all()
    D, I = ok.knn(X, Q, 5, metric=1)
    assert (I[:, 3:] == -1).all() and (D[:, 3:] == FLT_MAX).all()
    D, I = ok.knn(np.zeros((0, 8), np.float32), Q, 2)
    assert (I == -1).all()


def test_nan_never_enters():
    rng = np.random.default_rng(1)
    X = rng.standard_normal((50, 8)).astype(np.float32)
    X[7] = np.nan
    Q = rng.standard_normal((3, 8)).astype(np.float32)
    D, I = ok.knn(X, Q, 50)
    assert 7 not in I and (I[:, -1] == -1).all()


def test_merge_equals_unsharded():
    rng = np.random.default_rng(2)
    X = rng.integers(-3, 4, (900, 8)).astype(np.float32)
    Q = rng.integers(-3, 4, (11, 8)).astype(np.float32)
    for metric in (0, 1):
        parts = [ok.knn(X[s:s + 300], Q, 40, metric=metric, id_offset=s) for s in range(0, 900, 300)]
        Dm, Im = ok.topk_merge(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), metric)
        D, I = ok.knn(X, Q, 40, metric=metric)
        assert np.array_equal(Im, I) and np.array_equal(Dm, D)


def test_l2norm_rows_close_to_numpy():
    rng = np.random.default_rng(3)
    X = rng.standard_normal((20, 33)).astype(np.float32)
    a = ok.l2norm_rows(X)
    b = X / np.linalg.norm(X, axis=1, keepdims=True)
    assert np.allclose(a, b, rtol=0, atol=2e-7)
    assert np.isnan(ok.l2norm_rows(np.zeros((1, 4), np.float32))).all()  # no epsilon, like the reference


def test_l2norm_rows_faiss_form():
    """FAISS's NormalizationTransform (fvec_renorm_L2, as published): one float reciprocal per row taken from a DOUBLE division,
    one multiplication per element, rows whose squared norm is not > 0 left exactly as they are."""
    rng = np.random.default_rng(4)
    X = rng.standard_normal((50, 33)).astype(np.float32)
    X[3] = 0.0                    # zero row: stays zero (numpy form: NaN)
    X[5] = 1e-30                  # squares underflow to 0: stays as it is
    X[7] = 1e-20                  # squared norm is a denormal > 0: normalised
    X[9, 0] = np.nan              # nr is NaN, not > 0: untouched
    a = ok.l2norm_rows(X, form="faiss")
    nr = ok.sqnorm_rows(X)
    for i in range(50):
        if nr[i] > 0:
            inv = np.float32(1.0 / np.float64(np.sqrt(np.float32(nr[i]))))
            assert np.array_equal(a[i], X[i] * inv), i
        else:
            assert np.array_equal(a[i], X[i], equal_nan=True), i
    assert not a[3].any() and np.array_equal(a[5], X[5]) and abs(float(np.linalg.norm(a[7].astype(np.float64))) - 1) < 1e-4  # (a denormal squared norm keeps ~17 bits)
    b = ok.l2norm_rows(X)         # the numpy form on the same rows
    assert np.isnan(b[3]).all() and not np.isfinite(b[5]).any()
    ok_rows = [i for i in range(50) if i not in (3, 5, 9)]
    assert np.allclose(a[ok_rows], b[ok_rows], rtol=3e-7, atol=0) and not np.array_equal(a[ok_rows], b[ok_rows])
    with pytest.raises(ValueError):
        ok.l2norm_rows(X, form="blas")
    # through knn(): a zero KB row is retrievable (score 0) in FAISS's form and never in numpy's
    Q = rng.standard_normal((4, 33)).astype(np.float32)
    Df, If = ok.knn(X[:8], Q, 8, l2norm=True, l2norm_form="faiss")
    Dn, In = ok.knn(X[:8], Q, 8, l2norm=True, l2norm_form="numpy")
    assert all(3 in row for row in If) and all(3 not in row for row in In) and (In[:, -1] == -1).all()


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("k", [1, 7, 100, 300])
def test_tie_order_is_a_total_order_parameter(metric, k):
    """tie_order = id_asc / id_desc on tie-heavy integer data (every fp32 summation order is exact there): the result must be
    the k first rows of the full lexicographic sort by (score best first, id ascending | descending) -- membership at the
    k-th boundary and the order inside equal-score runs alike.  This is THIS LIBRARY'S documented policy (knn_oracle.c
    header), not a statement about FAISS."""
    rng = np.random.default_rng(5)
    X = rng.integers(-2, 3, (700, 16)).astype(np.float32)
    Q = rng.integers(-2, 3, (23, 16)).astype(np.float32)
    S = Q.astype(np.float64) @ X.astype(np.float64).T
    if metric == 1:
        S = -((Q.astype(np.float64)[:, None, :] - X.astype(np.float64)[None, :, :]) ** 2).sum(2)
    ids = np.broadcast_to(np.arange(X.shape[0]), S.shape)
    for tie, id_key in (("id_asc", ids), ("id_desc", -ids)):
        order = np.lexsort((id_key, -S), axis=1)[:, :k]
        D, I = ok.knn(X, Q, k, metric=metric, tie_order=tie)
        assert np.array_equal(I, order), tie
        want = np.take_along_axis(S, order, 1)
        assert np.array_equal(D, (-want if metric == 1 else want).astype(np.float32))
    # the two orders agree as SETS wherever the k-th and (k+1)-th scores differ
    Da, Ia = ok.knn(X, Q, k, metric=metric, tie_order="id_asc")
    Dd, Id = ok.knn(X, Q, k, metric=metric, tie_order="id_desc")
    assert np.array_equal(Da, Dd)
    full = np.sort(S, axis=1)[:, ::-1]
    for q in range(len(Q)):
        if k < X.shape[0] and full[q, k - 1] != full[q, k]:
            assert set(Ia[q]) == set(Id[q])


def test_shard_merge_follows_the_tie_order():
    rng = np.random.default_rng(6)
    X = rng.integers(-1, 2, (512, 8)).astype(np.float32)
    Q = rng.integers(-1, 2, (9, 8)).astype(np.float32)
    for tie in ("id_asc", "id_desc"):
        parts = [ok.knn(X[s:s + 128], Q, 20, metric=0, id_offset=s, tie_order=tie) for s in range(0, 512, 128)]
        D, I = ok.topk_merge(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), 0, tie_order=tie)
        Dw, Iw = ok.knn(X, Q, 20, metric=0, tie_order=tie)
        assert np.array_equal(D, Dw) and np.array_equal(I, Iw)


@pytest.mark.parametrize("metric", [0, 1])
def test_blas_leg_equals_the_chain_oracle_on_lattice_data(metric):
    """FAISS's organisation (sgemm blocks + the strict-'>' heap rule in C) on integer data, where the BLAS summation order
    cannot matter: scores AND ids equal the fmaf-chain oracle, ties at the k-th boundary included; ragged last block,
    k larger than one block, more than one query block."""
    rng = np.random.default_rng(7)
    X = rng.integers(-3, 4, (3000, 32)).astype(np.float32)
    Q = rng.integers(-3, 4, (70, 32)).astype(np.float32)
    for k, block, qb in ((100, 1024, 4096), (10, 700, 32), (1200, 1024, 64)):
        D, I = ok.knn(X, Q, k, metric=metric, l2_form="expanded")
        for backend in ("torch", "numpy", "c"):   # MKL, OpenBLAS, the register-blocked kernel of knn_oracle.c
            Db, Ib = ok.knn_blas(X, Q, k, metric=metric, block=block, query_block=qb, backend=backend)
            assert np.array_equal(D, Db) and np.array_equal(I, Ib), (k, block, qb, backend)
    Db, Ib = ok.knn_blas(X[:40], Q, 64, metric=metric)  # k > ntotal: neutral values and -1
    D, I = ok.knn(X[:40], Q, 64, metric=metric, l2_form="expanded")
    assert np.array_equal(D, Db) and np.array_equal(I, Ib)
