"""GPU: a screened index keeps no panel-layout copy of the shard (1.5x the matrix instead of 2.5x).  Everything that used to
read the panel copy -- the exact-scan fallback of overflowing query tiles, FAISS's small-batch L2 form, reconstruct / save --
then reads the row-major copy and must give the same bits as an index that keeps the panel, as the exact index and as the
oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _three(X, metric, factory="Flat", pieces=(1000, 37, 4096 + 5)):
    from viquae_amd.index import MI355XFlatIndex
    a = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=True)                     # no panel copy
    b = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=True, keep_panel=True)
    c = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=False)
    assert not a.keep_panel and b.keep_panel and c.keep_panel
    s = 0
    for i, p in enumerate(list(pieces) + [len(X)]):      # ragged appends: nothing is aligned to 64 rows
        a.add(X[s:s + p])
        s += p
        if s >= len(X):
            break
    b.add_vectors(X)
    c.add_vectors(X)
    assert a._packed is None and a.ntotal == b.ntotal == len(X)
    return a, b, c


def _tie_heavy(rng, n, d, kind):
    if kind == "binary":                       # integer scores: thousands of rows tie at the k-th
        return rng.integers(0, 2, (n, d)).astype(np.float32)
    base = rng.standard_normal((7, d), dtype=np.float32) * 3     # a few distinct free-form rows repeated: exact duplicates,
    return base[rng.integers(0, 7, n)]                            # non-trivial fp32 chains


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("n,d,kind", [(40000, 16, "binary"), (33333, 50, "dup"), (70001, 130, "dup"), (20011, 768, "dup")])
def test_rowmajor_fallback_equals_panel_fallback_exact_scan_and_oracle(metric, n, d, kind):
    from oracle import knn as ok
    rng = np.random.default_rng(n + metric)
    X = _tie_heavy(rng, n, d, kind)
    Q = (rng.integers(0, 2, (300, d)).astype(np.float32) if kind == "binary" else rng.standard_normal((300, d), dtype=np.float32))
    if kind == "binary":
        Q[0] = Q[299] = 1.0
    a, b, c = _three(X, metric)
    k = 100
    Da, Ia = a.search_batch(Q, k)
    flagged = a.screen_stats(300, k)[0]             # query tiles that overflowed and were recomputed from the row-major copy
    assert flagged == 2 if kind == "dup" else flagged >= (1 if metric == 0 else 0)   # thousands of exact duplicates always overflow
    Db, Ib = b.search_batch(Q, k)
    Dc, Ic = c.search_batch(Q, k)
    assert np.array_equal(Ia, Ib) and np.array_equal(Da, Db)
    assert np.array_equal(Ia, Ic) and np.array_equal(Da, Dc)
    Do, Io = ok.knn(X, Q[:40], k, metric=metric)
    assert np.array_equal(Ia[:40], Io) and np.array_equal(Da[:40], Do)


@pytest.mark.parametrize("factory", ["Flat", "L2norm,Flat"])
@pytest.mark.parametrize("metric", [0, 1])
def test_no_panel_index_on_free_form_data(metric, factory):
    """No fallback here: stored rows, norms, screened results and the small-batch L2 form all come from the row-major path."""
    from oracle import knn as ok
    rng = np.random.default_rng(17 + metric)
    X = rng.standard_normal((30007, 203), dtype=np.float32) * 2 + 0.5
    Q = rng.standard_normal((150, 203), dtype=np.float32)
    a, b, c = _three(X, metric, factory)
    assert np.array_equal(a.reconstruct_n(), c.reconstruct_n())                      # same "L2norm," arithmetic
    assert np.array_equal(a.reconstruct_n(64 * 11 + 3, 200), c.reconstruct_n(64 * 11 + 3, 200))
    assert torch.equal(a._sqnorm[:a.ntotal], c._sqnorm[:c.ntotal])
    for nq in (150, 20, 19, 7, 1):                                                    # 19 and below: FAISS's direct L2 form
        Da, Ia = a.search_batch(Q[:nq], 100)
        Dc, Ic = c.search_batch(Q[:nq], 100)
        assert np.array_equal(Ia, Ic) and np.array_equal(Da, Dc), nq
        Qo = Q[:nq]
        if factory != "Flat":
            Qo = (Qo / np.linalg.norm(Qo, axis=1, keepdims=True)).astype(np.float32)   # the reference normalises on the host too
            Da, Ia = a.search_batch(Qo, 100)
        Do, Io = ok.knn(X, Qo, 100, metric=metric, l2norm=factory != "Flat")
        assert np.array_equal(Ia, Io) and np.array_equal(Da, Do), nq


def test_no_panel_index_saves_and_loads(tmp_path):
    from viquae_amd.index import MI355XFlatIndex
    rng = np.random.default_rng(4)
    X = rng.standard_normal((5003, 96), dtype=np.float32)
    Q = rng.standard_normal((33, 96), dtype=np.float32)
    a = MI355XFlatIndex(string_factory="L2norm,Flat", metric_type=0, screen=True)
    a.add_vectors(X)
    f = tmp_path / "idx.mq"
    a.save(f)
    # a loaded "L2norm," file defaults to FAISS's query arithmetic (index.py load()); ask for this index's own to compare bits
    b = MI355XFlatIndex.load(f, l2norm_form=a.l2norm_form)
    assert b._packed is None and b.ntotal == 5003
    Da, Ia = a.search_batch(Q, 10)
    Db, Ib = b.search_batch(Q, 10)
    assert np.array_equal(Ia, Ib) and np.array_equal(Da, Db)


def test_hbm_footprint_of_a_screened_index():
    """1.5x the fp32 matrix (row-major fp32 + bf16), not 2.5x."""
    from viquae_amd.index import MI355XFlatIndex
    n, d = 65536, 768
    a = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
    a.add(torch.randn((n, d), device="cuda"), total_hint=n)
    held = sum(t.numel() * t.element_size() for t in (a._rowmajor, a._bf16, a._sqnorm) if t is not None)
    assert a._packed is None and held <= 1.51 * n * d * 4
