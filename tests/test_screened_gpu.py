"""GPU: the screened search (bf16 screening scan + exact fp32 re-scoring, csrc/knn_screen.inc) must return
exactly what the exact fp32 scan and the CPU oracle return -- scores bit-for-bit, ids, tie order -- on
friendly data (no fallback) and on data built to defeat the screen (fallback to the exact scan)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(X, factory="Flat"):
    from viquae_amd.index import MI355XFlatIndex
    a = MI355XFlatIndex(string_factory=factory, metric_type=0, screen=True)
    b = MI355XFlatIndex(string_factory=factory, metric_type=0, screen=False)
    a.add_vectors(X)
    b.add_vectors(X)
    assert a.screen and not b.screen
    return a, b


@pytest.mark.parametrize("n,d,nq,k", [(20000, 768, 300, 100), (5000, 64, 37, 10), (3000, 100, 5, 128), (70000, 128, 513, 100)])
def test_screened_equals_exact_scan_and_oracle(n, d, nq, k):
    from oracle import knn as ok
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, d), dtype=np.float32)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    a, b = _pair(X)
    Da, Ia = a.search_batch(Q, k)
    Db, Ib = b.search_batch(Q, k)
    assert np.array_equal(Ia, Ib) and np.array_equal(Da, Db)
    Do, Io = ok.knn(X, Q, k)
    assert np.array_equal(Ia, Io) and np.array_equal(Da, Do)
    flagged, rescored, mx = a.screen_stats(nq, k)[:3]
    assert flagged == 0 and rescored >= nq * min(k, n) and mx <= 1024  # the screen did the work, nothing fell back


def test_screen_is_lossless_on_adversarial_scales():
    """Rows with wildly different norms: the margin uses max ||x||, so the screen must still keep every true
    neighbour (or fall back); results stay exact."""
    from oracle import knn as ok
    rng = np.random.default_rng(5)
    X = rng.standard_normal((12000, 96), dtype=np.float32)
    X[::7] *= 50.0
    X[1::13] *= 1e-3
    Q = rng.standard_normal((64, 96), dtype=np.float32)
    a, _ = _pair(X)
    D, I = a.search_batch(Q, 100)
    Do, Io = ok.knn(X, Q, 100)
    assert np.array_equal(I, Io) and np.array_equal(D, Do)


def test_tie_heavy_data_falls_back_and_stays_exact():
    """Small-integer data: thousands of rows share the k-th score, the candidate buffers overflow, the tile is
    flagged and recomputed by the exact scan (lower id wins ties)."""
    from oracle import knn as ok
    rng = np.random.default_rng(11)
    X = rng.integers(0, 2, (40000, 16)).astype(np.float32)  # scores are integers 0..16: thousands of rows tie at the k-th
    Q = rng.integers(0, 2, (21, 16)).astype(np.float32)
    Q[0] = 1.0
    a, _ = _pair(X)
    D, I = a.search_batch(Q, 100)
    Do, Io = ok.knn(X, Q, 100)
    assert np.array_equal(I, Io) and np.array_equal(D, Do)
    assert a.screen_stats(21, 100)[0] == 1  # the single query tile was recomputed exactly


def test_screened_l2norm_factory_and_incremental_add():
    from oracle import knn as ok
    from viquae_amd.index import MI355XFlatIndex
    rng = np.random.default_rng(3)
    X = rng.standard_normal((9000, 80), dtype=np.float32)
    Q = rng.standard_normal((33, 80), dtype=np.float32)
    a = MI355XFlatIndex(string_factory="L2norm,Flat", metric_type=0, screen=True)
    a.add(X[:6400])
    a.add(X[6400:])
    D, I = a.search_batch(Q, 50)
    Do, Io = ok.knn(X, Q, 50, l2norm=True)
    assert np.array_equal(I, Io) and np.array_equal(D, Do)


@pytest.mark.parametrize("n,d,nq,k", [(3000, 64, 70, 10), (20000, 768, 300, 100), (70001, 100, 513, 128), (5000, 30, 1, 1)])
def test_l2_screen_equals_exact_scan_and_oracle(n, d, nq, k):
    """metric_type 1 (FAISS's default): the screen ranks by q.x - ||x||^2/2 through two extra bf16 columns; distances and
    ids must equal the exact fp32 L2 scan and the oracle bit for bit."""
    import torch
    from oracle import knn as ok
    from viquae_amd.index import MI355XFlatIndex
    rng = np.random.default_rng(n + d)
    X = rng.standard_normal((n, d), dtype=np.float32)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    Q[0] = X[7]          # an exact hit: distance 0 (the clamp)
    X[11] = X[7]         # ... twice: a clamped tie, lowest id first
    scr = MI355XFlatIndex(string_factory="Flat", metric_type=1, screen=True)
    assert scr.screen
    scr.add_vectors(X)
    ex = MI355XFlatIndex(string_factory="Flat", metric_type=1, screen=False)
    ex.add_vectors(X)
    D, I = scr.search_batch(Q, k)
    D0, I0 = ex.search_batch(Q, k)
    Do, Io = ok.knn(X, Q, k, metric=1)
    assert np.array_equal(I, I0) and np.array_equal(D, D0)
    assert np.array_equal(I, Io) and np.array_equal(D, Do)
    assert I[0, 0] == 7 and D[0, 0] == 0.0 and (k == 1 or I[0, 1] == 11)
    if nq >= 20:  # fewer than 20 L2 queries take FAISS's sequential form (csrc/knn_direct.inc), not the screen
        assert scr.screen_stats(nq, k)[0] == 0  # no tile needed the exact fallback


def test_l2_screen_with_many_exact_duplicates_of_the_query():
    """More than k rows at distance exactly 0: the exact path returns the k lowest ids; so must the screen."""
    from viquae_amd.index import MI355XFlatIndex
    rng = np.random.default_rng(5)
    X = rng.standard_normal((4000, 96), dtype=np.float32)
    dup = rng.choice(4000, 300, replace=False)
    X[dup] = X[dup[0]]
    Q = np.stack([X[dup[0]], rng.standard_normal(96).astype(np.float32)])
    a = MI355XFlatIndex(string_factory="Flat", metric_type=1, screen=True)
    a.add_vectors(X)
    b = MI355XFlatIndex(string_factory="Flat", metric_type=1, screen=False)
    b.add_vectors(X)
    D, I = a.search_batch(Q, 100)
    D0, I0 = b.search_batch(Q, 100)
    assert np.array_equal(I, I0) and np.array_equal(D, D0)
    assert np.array_equal(I[0], np.sort(dup)[:100]) and np.all(D[0] == 0.0)


def test_l2_margin_dominates_the_measured_screening_error():
    """|S~' - (qn - d_f)/2| <= margin'/2 with S~' recomputed in float64 from the bf16 operands incl. the (h, l) columns."""
    import torch
    from viquae_amd.index import MI355XFlatIndex
    g = torch.Generator(device="cuda").manual_seed(8)
    n, d, nq, k = 5000, 200, 200, 10
    X = torch.randn((n, d), generator=g, device="cuda") * torch.exp(0.3 * torch.randn((n, 1), generator=g, device="cuda"))
    Q = torch.randn((nq, d), generator=g, device="cuda")
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=1, screen=True)
    idx.add(X)
    idx.search_device(Q, k)
    kernel_max_margin = idx.screen_stats(nq, k)[5] * 1e-6
    xn = idx._sqnorm[:n].double()
    v = -0.5 * idx._sqnorm[:n]
    h = v.to(torch.bfloat16)
    l = (v - h.float()).to(torch.bfloat16)
    c = idx._center if idx._center is not None else torch.zeros(d, device="cuda")   # the screen rounds x - c
    screen = Q.to(torch.bfloat16).double() @ (X - c).to(torch.bfloat16).double().T + (h.double() + l.double())[None]
    qn = (Q.double() ** 2).sum(1)
    d_exact = (qn[:, None] + xn[None]) - 2 * (Q.double() @ X.double().T)
    target = (qn[:, None] - d_exact) / 2 - (Q.double() @ c.double())[:, None]       # the same ranking, shifted by q.c
    dev = (screen - target).abs().max().item()
    assert kernel_max_margin >= 2 * dev, (kernel_max_margin, dev)
    assert kernel_max_margin <= 40 * dev   # and it is not vacuous


def test_nan_inf_rows_with_screen():
    from oracle import knn as ok
    rng = np.random.default_rng(9)
    X = rng.standard_normal((3000, 32), dtype=np.float32)
    X[17, 3] = np.nan
    X[100, 0] = np.inf
    X[200, 5] = -np.inf
    Q = rng.standard_normal((6, 32), dtype=np.float32)
    a, b = _pair(X)
    Da, Ia = a.search_batch(Q, 20)
    Do, Io = ok.knn(X, Q, 20)
    assert np.array_equal(Ia, Io) and np.array_equal(Da, Do, equal_nan=True)


@pytest.mark.parametrize("kind", ["gauss", "midpoints", "wide_range"])
def test_margin_dominates_the_measured_screening_error(kind):
    """The lossless argument needs |S~ - S| <= margin / 2 for every (query, row).  S~ is recomputed here as the exact
    (float64) product of the bf16-rounded operands and S as the exact float64 product; the kernel's largest margin
    (screen_stats) must dominate twice the largest deviation, and must itself equal the documented formula
    2 (||q|| max||xc - xc~|| + ||q - q~|| max||xc~|| + dp 2.98e-7 ||q|| max||x||), xc = x - centre, up to its slack factors."""
    import torch
    from viquae_amd.index import MI355XFlatIndex
    g = torch.Generator(device="cuda").manual_seed(5)
    n, d, nq, k = 6000, 200, 300, 10
    if kind == "gauss":
        X = torch.randn((n, d), generator=g, device="cuda")
        Q = torch.randn((nq, d), generator=g, device="cuda")
    elif kind == "midpoints":
        # every element sits exactly half way between two bf16 values: the worst case of the rounding
        m = torch.randint(128, 256, (n, d), generator=g, device="cuda").float()
        X = (m + 0.5) / 128.0 * torch.where(torch.rand((n, d), generator=g, device="cuda") < 0.5, -1.0, 1.0)
        mq = torch.randint(128, 256, (nq, d), generator=g, device="cuda").float()
        Q = (mq + 0.5) / 128.0
    else:
        X = torch.randn((n, d), generator=g, device="cuda") * torch.exp(4 * torch.randn((n, 1), generator=g, device="cuda"))
        Q = torch.randn((nq, d), generator=g, device="cuda") * torch.exp(4 * torch.randn((nq, 1), generator=g, device="cuda"))
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
    idx.add(X)
    D, I = idx.search_device(Q, k)
    ex = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=False)
    ex.add(X)
    D2, I2 = ex.search_device(Q, k)
    assert torch.equal(D, D2) and torch.equal(I, I2)
    stats = idx.screen_stats(nq, k)
    # the screen rounds the CENTRED rows xc = x - c (c = mean of the first rows added; any fixed c ranks alike)
    c = idx._center if idx._center is not None else torch.zeros(d, device="cuda")
    Xc = X - c                                                                   # fp32, as the kernel forms it
    Xb, Qb = Xc.to(torch.bfloat16).double(), Q.to(torch.bfloat16).double()
    dev = ((Q.double() @ Xc.double().T) - (Qb @ Xb.T)).abs().amax(dim=1)         # per query, over all rows
    qn, dqn = Q.double().norm(dim=1), (Q.double() - Qb).norm(dim=1)
    xn = torch.maximum(X.double().norm(dim=1).max(), Xc.double().norm(dim=1).max())
    dxn = (Xc.double() - Xb).norm(dim=1).max()
    dp = (d + 63) // 64 * 64
    eps = qn * dxn + dqn * Xb.norm(dim=1).max() + dp * 2.98e-7 * qn * xn
    assert torch.all(dev <= eps), float((dev / eps).max())
    kernel_max_margin = stats[5] * 1e-6
    assert kernel_max_margin >= 2 * float(dev.max()) * 0.999
    assert 2 * float(eps.max()) * 0.99 <= kernel_max_margin * (1 + 1e-3) + 1e-6
    assert kernel_max_margin <= 2 * float(eps.max()) * 1.02 + 2e-6   # slack factors stay within 2 %


def _anisotropic(n, d, nq, seed, shift=9.0, noise=0.25):
    g = torch.Generator(device="cuda").manual_seed(seed)
    mu = torch.randn((1, d), generator=g, device="cuda")
    mu = shift * mu / mu.norm()
    X = mu + noise * torch.randn((n, d), generator=g, device="cuda")
    Q = mu + noise * torch.randn((nq, d), generator=g, device="cuda")
    return X, Q



@pytest.mark.parametrize("factory,metric", [("Flat", 0), ("L2norm,Flat", 0), ("Flat", 1)])
def test_centred_screen_on_embeddings_with_a_shared_component(factory, metric, monkeypatch):
    """DPR / CLIP-like data: every vector = a large common direction + small isotropic noise.  The index centres its bf16
    screening copy on the mean of the first rows (q.(x - c) ranks like q.x): results stay bit-identical to the exact
    scan and the candidate sets shrink against the uncentred screen."""
    from viquae_amd.index import MI355XFlatIndex
    X, Q = _anisotropic(40000, 256, 300, 3)

    def run(center):
        monkeypatch.setenv("MQ_KNN_CENTER", "1" if center else "0")
        idx = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=True)
        idx.add(X[:25600])
        idx.add(X[25600:])          # the centre chosen at the first add is kept for later rows
        assert (idx._center is not None) == center
        D, I = idx.search_device(Q, 100)
        return D, I, idx.screen_stats(300, 100)

    Dc, Ic, sc = run(True)
    Du, Iu, su = run(False)
    ex = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=False)
    ex.add(X)
    De, Ie = ex.search_device(Q, 100)
    assert torch.equal(Dc, De) and torch.equal(Ic, Ie) and torch.equal(Du, De) and torch.equal(Iu, Ie)
    assert sc[0] == 0 and sc[1] < 0.8 * su[1], (sc[:3], su[:3])   # fewer rows re-scored, nothing fell back


def test_centred_margin_dominates_the_measured_screening_error(monkeypatch):
    """|S~c - Sc| <= margin / 2 with S~c = bf16(q) . bf16(x - c) and Sc = q . (x - c) in float64, for every (query, row);
    and the centred margin is well below the uncentred one on such data.  (The rows-only centring: MQ_KNN_CENTER_QUERIES=0;
    with the queries centred as well -- the default at this width, where the two columns are free -- see the next test.)"""
    from viquae_amd.index import MI355XFlatIndex
    monkeypatch.setenv("MQ_KNN_CENTER_QUERIES", "0")
    X, Q = _anisotropic(6000, 200, 200, 5)
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
    idx.add(X)
    idx.search_device(Q, 10)
    margin = idx.screen_stats(200, 10)[5] * 1e-6
    c = idx._center.double()
    Xc32 = (X - idx._center)                       # fp32, as the kernel forms it
    screen = Q.to(torch.bfloat16).double() @ Xc32.to(torch.bfloat16).double().T
    exact = Q.double() @ (X.double() - c).T
    dev = (screen - exact).abs().max().item()
    assert margin >= 2 * dev, (margin, dev)
    qn, dqn = Q.double().norm(dim=1), (Q.double() - Q.to(torch.bfloat16).double()).norm(dim=1)
    unc = 2 * (qn * (X.double() - X.to(torch.bfloat16).double()).norm(dim=1).max() + dqn * X.double().norm(dim=1).max()).max().item()
    assert margin < 0.8 * unc, (margin, unc)
    s = idx._xmax2.cpu().numpy()
    assert abs(s[0] - float((X.double() ** 2).sum(1).max())) < 1e-3 * s[0] and s[2] < 0.6 * s[0]


@pytest.mark.parametrize("d,shift", [(200, 9.0), (1000, 20.0), (2048, 40.0), (128, 9.0), (127, 9.0), (63, 6.0)])
def test_centred_query_margin_dominates_the_measured_screening_error(d, shift, monkeypatch):
    """MQ_METRIC_IP_CENTRED (round 5): S~ = bf16(q - c) . bf16(x - c) + (h + l), (h, l) the bf16 pair of v = c . (x - c), against
    S = q . x - q . c in float64 -- |S~ - S| <= margin / 2 for every (query, row), the margin is not vacuous, it sits well below
    the rows-only margin on data with a large shared component, and the search is the exact scan's bit for bit."""
    from viquae_amd.index import METRIC_IP_CENTRED, MI355XFlatIndex
    n, nq, k = 6000, 200, 10
    X, Q = _anisotropic(n, d, nq, 5 + d, shift=shift, noise=0.25)
    if d in (128, 127, 63):
        # d % 64 == 0 and d <= 768: not chosen by default (a 3rd K block); d % 64 == 63: the two row-term columns fall into
        # DIFFERENT K blocks of the tile layout (to_bf16_rows_centred_kernel writes them apart)
        monkeypatch.setenv("MQ_KNN_CENTER_QUERIES", "1")
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
    idx.add(X[:2560])
    idx.add(X[2560:])
    assert idx._screen_metric == METRIC_IP_CENTRED
    D, I = idx.search_device(Q, k)
    stats = idx.screen_stats(nq, k)
    margin = stats[5] * 1e-6
    ex = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=False)
    ex.add(X)
    De, Ie = ex.search_device(Q, k)
    assert torch.equal(D, De) and torch.equal(I, Ie) and stats[0] == 0
    c = idx._center
    Xc32, Qc32 = X - c, Q - c                                                    # fp32, as the kernels form them
    v = torch.zeros(n, device="cuda")
    for t in range(d):                                                           # the k-ordered fp32 fma chain of the pack kernel
        v = torch.addcmul(v, Xc32[:, t], c[t])                                   # (fused or not: inside the 2^-24 d terms of the margin)
    h = v.to(torch.bfloat16)
    l = (v - h.float()).to(torch.bfloat16)
    screen = Qc32.to(torch.bfloat16).double() @ Xc32.to(torch.bfloat16).double().T + (h.double() + l.double())[None]
    target = Q.double() @ X.double().T - (Q.double() @ c.double())[:, None]
    dev = (screen - target).abs().max().item()
    assert margin >= 2 * dev, (margin, dev)
    assert margin <= 60 * dev, (margin, dev)                                      # not vacuous
    monkeypatch.setenv("MQ_KNN_CENTER_QUERIES", "0")
    rows_only = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
    rows_only.add(X)
    assert rows_only._screen_metric == 0
    D0, I0 = rows_only.search_device(Q, k)
    st0 = rows_only.screen_stats(nq, k)
    assert torch.equal(D0, De) and torch.equal(I0, Ie)
    assert margin < 0.5 * st0[5] * 1e-6, (margin, st0[5] * 1e-6)                  # the margin now follows ||q - c||, not ||q||
    assert st0[0] > 0 or stats[1] <= st0[1]                                       # never more rows re-scored (unless the rows-only screen gave up)


def test_centred_queries_policy():
    """When the index centres its queries as well: inner product only, a centre that carries a quarter of the squared norms, and
    either free columns (d % 64 in 1 ... 62) or a width beyond the streaming kernel's 768 columns (where the columns cost a K block:
    three quarters); never d = 767; d = 768 like the other multiples of 64 since round 6."""
    from viquae_amd.index import METRIC_IP_CENTRED, MI355XFlatIndex

    def metric_of(d, shift, metric=0, factory="Flat"):
        X, _ = _anisotropic(3000, d, 4, d, shift=shift, noise=0.25)
        idx = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=True)
        idx.add(X)
        return idx._screen_metric

    assert metric_of(200, 9.0) == METRIC_IP_CENTRED and metric_of(200, 0.0) == 0          # no shared component: plain
    assert metric_of(1024, 20.0) == METRIC_IP_CENTRED and metric_of(2048, 30.0, factory="L2norm,Flat") == METRIC_IP_CENTRED
    assert metric_of(767, 20.0) == 0                                                       # a 13th K block would cost the streaming kernel
    assert metric_of(768, 20.0) == METRIC_IP_CENTRED and metric_of(768, 5.0) == 0             # round 6: alone in the 13th block, the two columns
                                                                                           # travel as fp32 beside twelve (when the centre dominates)
    assert metric_of(512, 20.0) == METRIC_IP_CENTRED and metric_of(512, 5.0) == 0             # a K block more: only when it dominates
    assert metric_of(200, 9.0, metric=1) == 1                                              # the L2 screen keeps its own row term
    X, _ = _anisotropic(3000, 200, 4, 1)
    few = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
    few.add(X[:10])                                                                        # ten rows are not evidence of a common component
    few.add(X[10:])
    assert few._screen_metric == 0


def test_centred_queries_on_every_index_path(monkeypatch, tmp_path):
    """MQ_METRIC_IP_CENTRED through the paths an index has: a kept panel with ragged appends (the open-panel re-pack), row shards on
    one device (every shard decides for itself), k beyond the screen (row ranges), several query chunks with the post-scan half
    on a second stream, save -> load.  Always the exact scan's answer."""
    from viquae_amd.index import METRIC_IP_CENTRED, MI355XFlatIndex
    from viquae_amd.sharded import LocalShardsFlatIndex
    X, Q = _anisotropic(41000, 200, 4096 + 300, 17, shift=12.0, noise=0.25)
    ex = MI355XFlatIndex(string_factory="L2norm,Flat", metric_type=0, screen=False, l2norm_form="faiss")
    ex.add(X)
    De, Ie = ex.search_device(Q, 100)
    # kept panel, ragged appends
    a = MI355XFlatIndex(string_factory="L2norm,Flat", metric_type=0, screen=True, keep_panel=True, l2norm_form="faiss")
    for lo, hi in ((0, 1000), (1000, 1037), (1037, 30001), (30001, 41000)):
        a.add(X[lo:hi])
    assert a._screen_metric == METRIC_IP_CENTRED
    D, I = a.search_device(Q, 100)
    assert torch.equal(D, De) and torch.equal(I, Ie)
    # chunks pipelined over two streams
    monkeypatch.setenv("MQ_KNN_TAIL_OVERLAP", "1")
    D, I = a.search_device(Q, 100)
    assert torch.equal(D, De) and torch.equal(I, Ie)
    monkeypatch.delenv("MQ_KNN_TAIL_OVERLAP")
    # k beyond the screen: row ranges, merged and proved
    D3, I3 = a.search_device(Q[:300], 300)
    De3, Ie3 = ex.search_device(Q[:300], 300)
    assert torch.equal(D3, De3) and torch.equal(I3, Ie3)
    # row shards on one device
    sh = LocalShardsFlatIndex([0] * 4, string_factory="L2norm,Flat", metric_type=0, allow_repeated_devices=True, l2norm_form="faiss")
    sh.add_vectors(X.cpu().numpy())
    assert all(s._screen_metric == METRIC_IP_CENTRED for s in sh.shards)
    Ds, Is = sh.search_batch(Q[:500].cpu().numpy(), 100)
    assert np.array_equal(Is, Ie[:500].cpu().numpy()) and np.array_equal(Ds, De[:500].cpu().numpy())
    # save -> load rebuilds the screen (the file holds the fp32 rows)
    a.save(tmp_path / "kb.index")
    b = MI355XFlatIndex.load(tmp_path / "kb.index")
    assert b._screen_metric == METRIC_IP_CENTRED
    Db, Ib = b.search_device(Q[:256], 100)
    assert torch.equal(Db, De[:256]) and torch.equal(Ib, Ie[:256])
