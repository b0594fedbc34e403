"""GPU parity of FAISS's small-batch L2 path (fewer than 20 queries: direct sums of (q-x)^2, csrc/knn_direct.inc)
through the C ABI, against the oracle's "direct" form (oracle/knn_oracle.c) -- bit-exact scores and indices -- and
of the rule that picks the form from the size of the WHOLE batch (include/meerqat_hip.h, MQ_KNN_L2_DIRECT_BELOW)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FLT_MAX = np.finfo(np.float32).max


def _index(X, factory="Flat", screen=None):
    from viquae_amd.index import MI355XFlatIndex
    idx = MI355XFlatIndex(string_factory=factory, metric_type=1, screen=screen)
    idx.add_vectors(X)
    return idx


def _same(a, b, what):
    assert np.array_equal(a[1], b[1]), f"{what}: indices differ"
    assert np.array_equal(a[0], b[0]), f"{what}: distances differ (max {np.nanmax(np.abs(a[0] - b[0]))})"


@pytest.mark.parametrize("screen", [True, False])
@pytest.mark.parametrize("n,d,nq,k", [
    (1000, 768, 5, 100), (777, 100, 3, 1), (5000, 64, 19, 128), (300, 48, 1, 10), (70000, 96, 7, 100),
    (64, 16, 19, 100),     # k > N: unfilled slots
    (200000, 768, 16, 100),
])
def test_direct_form_matches_oracle(n, d, nq, k, screen):
    from oracle import knn as ok
    rng = np.random.default_rng(n + d + nq)
    X = rng.standard_normal((n, d), dtype=np.float32)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    Q[0] = X[n // 3]                                   # an exact copy: distance exactly 0 in the direct form
    idx = _index(X, screen=screen)
    got = idx.search_batch(Q, k)
    _same(got, ok.knn(X, Q, k, metric=1, l2_form="direct"), "direct")
    assert got[0][0, 0] == 0.0 and got[1][0, 0] == n // 3
    if k > n:
        assert (got[1][:, n:] == -1).all() and (got[0][:, n:] == FLT_MAX).all()


def test_l2norm_factory_and_ties_and_non_finite_rows():
    from oracle import knn as ok
    rng = np.random.default_rng(3)
    X = rng.integers(-2, 3, (3000, 16)).astype(np.float32)      # massive ties: lower id must win
    Q = rng.integers(-2, 3, (19, 16)).astype(np.float32)
    _same(_index(X).search_batch(Q, 100), ok.knn(X, Q, 100, metric=1), "ties")
    X = rng.standard_normal((2000, 40), dtype=np.float32)
    Q = rng.standard_normal((11, 40), dtype=np.float32)
    _same(_index(X, "L2norm,Flat").search_batch(Q, 50), ok.knn(X, Q, 50, metric=1, l2norm=True), "l2norm")
    X = X[:120].copy()
    X[10, 0], X[20, 5], X[30, 7] = np.inf, -np.inf, np.nan      # never enter a result heap
    got = _index(X).search_batch(Q, 128)
    _same(got, ok.knn(X, Q, 128, metric=1), "non-finite")
    assert not np.isin(got[1], [10, 20, 30]).any() and (got[1][:, 117:] == -1).all() and (got[0][:, 117:] == FLT_MAX).all()


def test_form_is_chosen_from_the_whole_batch():
    """19 queries -> direct, 20 -> BLAS form; a 4100-query batch (cut into several C-ABI calls) is BLAS form throughout,
    including its last few queries."""
    from oracle import knn as ok
    rng = np.random.default_rng(9)
    X = rng.standard_normal((3000, 128), dtype=np.float32) * 5
    Q = rng.standard_normal((4100, 128), dtype=np.float32) * 5
    Q[:20] = X[:20]
    X[100:120] = Q[:20] * np.float32(1 + 2 ** -12)
    idx = _index(X)
    d19 = idx.search_batch(Q[:19], 10)
    _same(d19, ok.knn(X, Q[:19], 10, metric=1, l2_form="direct"), "19 queries")
    d20 = idx.search_batch(Q[:20], 10)
    _same(d20, ok.knn(X, Q[:20], 10, metric=1, l2_form="expanded"), "20 queries")
    assert not np.array_equal(d19[0], d20[0][:19])       # the two forms are observably different here
    big = idx.search_batch(Q, 10)
    _same(big, ok.knn(X, Q, 10, metric=1, l2_form="expanded"), "4100 queries")


def test_single_query_over_a_large_shard_streams_the_kb_once():
    """interact/system.py's use: one query.  500k x 768 rows (1.5 GB): checked against the oracle bit for bit."""
    import torch
    from oracle import knn as ok
    from viquae_amd.index import MI355XFlatIndex
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=1)
    blocks = []
    for s in range(0, 500_000, 1 << 16):
        x = torch.randn((min(1 << 16, 500_000 - s), 768), generator=g, device="cuda")
        idx.add(x, total_hint=500_000)
        blocks.append(x.cpu().numpy())
    X = np.concatenate(blocks)
    q = torch.randn((1, 768), generator=g, device="cuda")
    D, I = idx.search_device(q, 100)
    Do, Io = ok.knn(X, q.cpu().numpy(), 100, metric=1)
    assert np.array_equal(I.cpu().numpy(), Io) and np.array_equal(D.cpu().numpy(), Do)
