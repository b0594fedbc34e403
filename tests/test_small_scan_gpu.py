"""GPU: the streaming screening scan for ONE query tile (csrc/knn_small.inc: queries in registers, the whole LDS a ring of 32-row
tiles, one workgroup per CU) must give exactly what the 256 x 256 tile kernel, the exact fp32 scan and the CPU oracle give --
scores bit for bit, ids, tie order.  It serves searches of at most 256 queries (the reference's ``Dataset.map`` batch,
experiments/ir/viquae/dpr/search/config.json:25 -> meerqat/ir/search.py:146) over shards of >= 65,536 rows with at most 768
bf16 columns (one workgroup per CU = 256 slabs of at least eight 32-row tiles)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _default_switches():
    # the A/B switches are set through the C ABI (mq_knn_set_option), never through the environment of a running process
    from viquae_amd import _lib
    lib = _lib.load()
    lib.mq_knn_set_option(_lib.KNN_OPT_SMALL_SCAN, 1)
    lib.mq_knn_set_option(_lib.KNN_OPT_SMALL_MIN_TILES, 0)
    lib.mq_knn_set_option(_lib.KNN_OPT_SMALL_WAVES, 8)
    yield
    lib.mq_knn_set_option(_lib.KNN_OPT_SMALL_SCAN, 1)
    lib.mq_knn_set_option(_lib.KNN_OPT_SMALL_MIN_TILES, 0)
    lib.mq_knn_set_option(_lib.KNN_OPT_SMALL_WAVES, 8)


def _index(X, metric=0, factory="Flat", screen=True, tie_order=None):
    from viquae_amd.index import MI355XFlatIndex
    idx = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=screen, tie_order=tie_order)
    idx.add_vectors(X)
    return idx


def _both_scans(idx, Q, k):
    """(D, I) through the streaming kernel and through the tile kernel of the same index.  The streaming kernel has two
    instances -- two waves per SIMD (round 5, csrc/knn_small8.inc: the default) and one (round 4, csrc/knn_small.inc) -- which
    must agree bit for bit; the first one's results and statistics are returned."""
    from viquae_amd import _lib
    assert idx.scan_kind(len(Q), k) == "stream"
    assert _lib.load().mq_knn_get_option(_lib.KNN_OPT_SMALL_WAVES) == 8
    D1, I1 = idx.search_batch(Q, k)
    stats = idx.screen_stats(len(Q), k)
    with _lib.knn_option(_lib.KNN_OPT_SMALL_WAVES, 4):
        D4, I4 = idx.search_batch(Q, k)
        stats4 = idx.screen_stats(len(Q), k)
    assert np.array_equal(I4, I1) and np.array_equal(D4, D1) and stats4[0] == stats[0]
    from viquae_amd import _lib
    with _lib.knn_option(_lib.KNN_OPT_SMALL_SCAN, 0):
        assert idx.scan_kind(len(Q), k) == "tile"
        D0, I0 = idx.search_batch(Q, k)
    return (D1, I1), (D0, I0), stats


@pytest.mark.parametrize("n,d,nq,k,metric", [
    (66000, 768, 256, 100, 0),    # the reference's shape (DPR, inner product)
    (65549, 768, 199, 100, 0),    # ragged: N not a multiple of 32, a partial query tile
    (70000, 512, 64, 100, 0),     # CLIP width: the 8-K-block instance (ring of four tiles); one live wave
    (90000, 64, 21, 10, 0),       # one K block, ring of six
    (70001, 32, 200, 100, 0),     # one K block, all four waves live, both query halves (a bug of the first version: the second
    (65536, 3, 256, 1, 1),        #   half's first fragments were read from the NEXT ring slot) -- found by tools/stress_small_scan.py
    (72345, 320, 130, 128, 0),    # five K blocks, k at the limit of the fused selection
    (66666, 700, 37, 1, 0),       # d not a multiple of 64 (zero-padded columns)
    (68000, 510, 100, 100, 1),    # L2 through two extra columns: dp = 512
    (66000, 766, 256, 50, 1),     # L2 at the widest the registers take (dp = 768)
    (200000, 128, 256, 100, 0),   # 24 tiles per slab
])
def test_streaming_scan_equals_tile_scan_and_oracle(n, d, nq, k, metric):
    from oracle import knn as ok
    rng = np.random.default_rng(n + d)
    X = rng.standard_normal((n, d), dtype=np.float32)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    idx = _index(X, metric)
    (D1, I1), (D0, I0), stats = _both_scans(idx, Q, k)
    assert np.array_equal(I1, I0) and np.array_equal(D1, D0)
    Do, Io = ok.knn(X, Q, k, metric=metric)
    assert np.array_equal(I1, Io) and np.array_equal(D1, Do)
    assert stats[0] == 0 and stats[1] >= nq * k  # the screen did the work: nothing fell back to the exact scan


def test_not_served_cases_keep_the_tile_kernel():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((66000, 768), dtype=np.float32)
    ip, l2 = _index(X, 0), _index(X, 1)
    assert ip.scan_kind(256, 100) == "stream" and ip.scan_kind(257, 100) == "tile"    # more than one query tile
    assert ip.scan_kind(256, 129) == "tile"                                            # k beyond the fused selection
    assert ip.scan_kind(10, 300) == "tile" and ip.scan_kind(10, 1800) == "none"        # k beyond the screen: row ranges (round 4) / exact rounds
    assert l2.scan_kind(256, 100) == "stream"                                          # d = 768 + the two L2 columns = 13 K blocks: the row term travels as fp32 (round 6)
    from viquae_amd import _lib
    with _lib.knn_option(_lib.KNN_OPT_SMALL_WAVES, 4):
        assert l2.scan_kind(256, 100) == "tile"                                        # ... by the two-waves-per-SIMD instance only
    assert l2.scan_kind(19, 100) == "none"                                             # FAISS's small-batch L2 form
    small = _index(X[:60000], 0)
    assert small.scan_kind(256, 100) == "tile"                                         # fewer than eight 32-row tiles per workgroup


def test_l2norm_factory_and_both_tie_orders():
    from oracle import knn as ok
    rng = np.random.default_rng(7)
    X = rng.standard_normal((66500, 96), dtype=np.float32)
    X[100:140] = X[50:90]  # exact duplicates: tied scores inside the top-k
    Q = rng.standard_normal((77, 96), dtype=np.float32)
    for tie in ("id_asc", "id_desc"):
        idx = _index(X, 0, factory="L2norm,Flat", tie_order=tie)
        (D1, I1), (D0, I0), stats = _both_scans(idx, Q, 100)
        assert np.array_equal(I1, I0) and np.array_equal(D1, D0)
        Do, Io = ok.knn(X, Q, 100, l2norm=True, tie_order=tie)
        assert np.array_equal(I1, Io) and np.array_equal(D1, Do)


def test_tie_heavy_data_overflows_the_pool_halves_and_falls_back():
    """Small-integer data: thousands of rows tie at the k-th score, every lane's pool half fills up, the tile is flagged and the
    exact scan recomputes it (lower id wins)."""
    from oracle import knn as ok
    rng = np.random.default_rng(11)
    X = rng.integers(0, 2, (70000, 16)).astype(np.float32)
    Q = rng.integers(0, 2, (21, 16)).astype(np.float32)
    Q[0] = 1.0
    idx = _index(X, 0)
    assert idx.scan_kind(21, 100) == "stream"
    D, I = idx.search_batch(Q, 100)
    Do, Io = ok.knn(X, Q, 100)
    assert np.array_equal(I, Io) and np.array_equal(D, Do)
    assert idx.screen_stats(21, 100)[0] == 1


@pytest.mark.parametrize("copies", [40, 300])
def test_runs_of_equal_rows_fill_pool_halves_beyond_their_first_line(copies):
    """Every row repeated `copies` times in a run: the slabs that hold a query's best rows take dozens (40 copies) to more than a
    hundred (300 copies: a run is longer than a slab, and the k-th score is tied 300 ways) keys per lane half.  The two-waves
    kernel fills a half front to back and cand_select_kernel reads its first 64-byte line directly, the lines behind it from a
    list in LDS (POOL_LAYOUT_HALVES); the answer must stay the exact scan's and the oracle's, without the fallback."""
    from oracle import knn as ok
    rng = np.random.default_rng(copies)
    n, d, nq = 70000, 48, 200
    X = np.repeat(rng.standard_normal((n // copies + 1, d), dtype=np.float32), copies, axis=0)[:n]
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    for tie in ("id_asc", "id_desc"):
        idx = _index(X, 0, tie_order=tie)
        (D1, I1), (D0, I0), stats = _both_scans(idx, Q, 100)
        assert np.array_equal(I1, I0) and np.array_equal(D1, D0)
        Do, Io = ok.knn(X, Q, 100, tie_order=tie)
        assert np.array_equal(I1, Io) and np.array_equal(D1, Do)
        assert stats[0] == 0 and stats[3] > 16, stats  # no fallback; some pool held more than two first lines' worth


def test_identical_rows_overrun_the_line_list_and_fall_back():
    """All rows equal: every lane half takes every row it sees (~137 keys = 17 lines, 8.7 k lines per query against a list of
    1024), the walk finishes the slow way, finds more ties than the selection can hold and hands the tile to the exact scan --
    lower (or higher) ids first."""
    from oracle import knn as ok
    X = np.ones((70000, 16), dtype=np.float32)
    Q = np.random.default_rng(1).standard_normal((30, 16)).astype(np.float32)
    for tie in ("id_asc", "id_desc"):
        idx = _index(X, 0, tie_order=tie)
        assert idx.scan_kind(30, 100) == "stream"
        D, I = idx.search_batch(Q, 100)
        Do, Io = ok.knn(X, Q, 100, tie_order=tie)
        assert np.array_equal(I, Io) and np.array_equal(D, Do)
        assert idx.screen_stats(30, 100)[0] == 1


def test_heavy_tailed_and_clustered_rows():
    """Rows of wildly different norms and a shared direction (what the centred screen is for): thresholds stay valid."""
    from oracle import knn as ok
    rng = np.random.default_rng(5)
    X = rng.standard_normal((80000, 128), dtype=np.float32) + 3.0 * rng.standard_normal((1, 128), dtype=np.float32)
    X[::7] *= 20.0
    X[1::13] *= 1e-3
    Q = rng.standard_normal((200, 128), dtype=np.float32) + 2.0
    idx = _index(X, 0)
    (D1, I1), (D0, I0), _ = _both_scans(idx, Q, 100)
    assert np.array_equal(I1, I0) and np.array_equal(D1, D0)
    Do, Io = ok.knn(X, Q, 100)
    assert np.array_equal(I1, Io) and np.array_equal(D1, Do)


def test_planted_neighbours_in_every_slab_position():
    """One planted best row per query, spread over the shard so that every slab boundary, the warm-up tiles (scored twice) and the
    last, ragged tile hold some: each must come back first, once."""
    rng = np.random.default_rng(9)
    n, d, nq = 66000 + 17, 256, 256
    X = 0.05 * rng.standard_normal((n, d), dtype=np.float32)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    rows = np.unique(np.concatenate([np.linspace(0, n - 1, nq - 8).astype(np.int64), np.arange(n - 8, n)]))[:nq]
    rows = np.resize(rows, nq)
    for q, r in enumerate(rows):
        X[r] = Q[q]
    idx = _index(X, 0)
    assert idx.scan_kind(nq, 100) == "stream"
    D, I = idx.search_batch(Q, 100)
    owner = {int(r): q for q, r in enumerate(rows)}  # a row planted twice keeps its last query
    for q, r in enumerate(rows):
        if owner[int(r)] == q:
            assert I[q, 0] == r, (q, r, I[q, :3])
        assert len(set(I[q].tolist())) == 100
    exact = _index(X, 0, screen=False)
    De, Ie = exact.search_batch(Q, 100)
    assert np.array_equal(I, Ie) and np.array_equal(D, De)


def test_consecutive_searches_reuse_the_workspace():
    """The slot / bound words are reset by every call: a second search with other queries must not see the first one's bounds."""
    from oracle import knn as ok
    rng = np.random.default_rng(21)
    X = rng.standard_normal((66000, 200), dtype=np.float32)
    idx = _index(X, 0)
    for scale in (5.0, 0.01, 1.0):
        Q = scale * rng.standard_normal((256, 200), dtype=np.float32)
        D, I = idx.search_batch(Q, 100)
        Do, Io = ok.knn(X, Q, 100)
        assert np.array_equal(I, Io) and np.array_equal(D, Do)


@pytest.mark.parametrize("d,shift", [(200, 9.0), (512, 20.0), (700, 12.0)])
def test_centred_queries_through_the_streaming_kernel(d, shift):
    """MQ_METRIC_IP_CENTRED (the index centres its queries too: a large shared component) on shards the streaming kernel serves:
    the two row-term columns ride through the MFMAs like any other column -- 4, 9 (one K block more than d = 512 alone) and 11 K
    blocks -- and the result is the tile kernel's, the exact scan's and the oracle's."""
    from oracle import knn as ok
    from viquae_amd.index import METRIC_IP_CENTRED
    g = torch.Generator(device="cuda").manual_seed(d)
    mu = torch.randn((1, d), generator=g, device="cuda")
    mu = shift * mu / mu.norm()
    X = (mu + 0.25 * torch.randn((70000, d), generator=g, device="cuda")).cpu().numpy()
    Q = (mu + 0.25 * torch.randn((256, d), generator=g, device="cuda")).cpu().numpy()
    idx = _index(X, 0)
    assert idx._screen_metric == METRIC_IP_CENTRED
    (D1, I1), (D0, I0), stats = _both_scans(idx, Q, 100)
    assert np.array_equal(I1, I0) and np.array_equal(D1, D0)
    Do, Io = ok.knn(X, Q, 100)
    assert np.array_equal(I1, Io) and np.array_equal(D1, Do)
    assert stats[0] == 0


# ---------------------------------------------------------------------------------------------------------------------
# round 6: the L2 metric at d = 768 (13 K blocks of bf16 columns) through the streaming kernel -- the row term -||x||^2 / 2 enters
# the accumulation chain as fp32 (csrc/knn_small8.inc, ROWTERM) instead of through two more columns
# ---------------------------------------------------------------------------------------------------------------------
def _stream_and_tile(idx, Q, k):
    from viquae_amd import _lib
    assert idx.scan_kind(len(Q), k) == "stream"
    D1, I1 = idx.search_batch(Q, k)
    stats = idx.screen_stats(len(Q), k)
    with _lib.knn_option(_lib.KNN_OPT_SMALL_SCAN, 0):
        assert idx.scan_kind(len(Q), k) == "tile"
        D0, I0 = idx.search_batch(Q, k)
    return (D1, I1), (D0, I0), stats


@pytest.mark.parametrize("n,nq,k", [(66000, 256, 100), (65549, 199, 100), (131072, 20, 1), (70017, 64, 128), (200000, 256, 10)])
def test_l2_at_768_columns_streams_and_equals_tile_scan_and_oracle(n, nq, k):
    from oracle import knn as ok
    rng = np.random.default_rng(n + nq)
    X = rng.standard_normal((n, 768), dtype=np.float32)
    Q = rng.standard_normal((nq, 768), dtype=np.float32)
    Q[0] = X[n // 3]                      # an exact copy: the clamp at distance 0
    X[n - 1] = Q[1] * np.float32(1.0001)  # the best row of a query is the shard's LAST row (ragged last tile when n % 32 != 0)
    idx = _index(X, 1)
    (D1, I1), (D0, I0), stats = _stream_and_tile(idx, Q, k)
    assert np.array_equal(I1, I0) and np.array_equal(D1, D0)
    Do, Io = ok.knn(X, Q, k, metric=1)
    assert np.array_equal(I1, Io) and np.array_equal(D1, Do)
    assert I1[0, 0] == n // 3 and I1[1, 0] == n - 1
    assert stats[0] == 0 and stats[1] >= nq * k


def test_l2_at_768_columns_ties_norm_spread_and_l2norm_factory():
    from oracle import knn as ok
    rng = np.random.default_rng(5)
    # integer lattice with duplicated rows: massive exact ties, the lower id must win; rows of very different norms (the row term
    # decides the order, not the inner product)
    X = rng.integers(-3, 4, (66000, 768)).astype(np.float32)
    X[1000:1100] = X[2000:2100]
    X[40000:41000] *= np.float32(0.25)
    X[50000:50500] *= np.float32(4.0)
    Q = rng.integers(-3, 4, (64, 768)).astype(np.float32)
    Q[:8] = X[1000:1008]
    for tie in ("id_asc", "id_desc"):
        idx = _index(X, 1, tie_order=tie)
        assert idx.scan_kind(64, 100) == "stream"
        D, I = idx.search_batch(Q, 100)
        Do, Io = ok.knn(X, Q, 100, metric=1, tie_order=tie)
        assert np.array_equal(I, Io) and np.array_equal(D, Do), tie
    Xn = rng.standard_normal((66000, 768), dtype=np.float32) * rng.uniform(0.1, 30, (66000, 1)).astype(np.float32)
    Qn = rng.standard_normal((100, 768), dtype=np.float32)
    idx = _index(Xn, 1, factory="L2norm,Flat")
    assert idx.scan_kind(100, 50) == "stream"
    D, I = idx.search_batch(Qn, 50)
    Do, Io = ok.knn(Xn, Qn, 50, metric=1, l2norm=True)
    assert np.array_equal(I, Io) and np.array_equal(D, Do)
    idx = _index(Xn, 1)                   # norms from 3 to 800: -||x||^2 / 2 spans five orders of magnitude
    D, I = idx.search_batch(Qn, 50)
    Do, Io = ok.knn(Xn, Qn, 50, metric=1)
    assert np.array_equal(I, Io) and np.array_equal(D, Do)


def test_centred_queries_at_768_columns_stream():
    """MQ_METRIC_IP_CENTRED at d = 768: the two row-term columns are alone in a 13th K block; the streaming kernel reads twelve and
    takes h + l of each row's pair as the start of its accumulation chain.  A shared component that carries more than three quarters
    of the squared norms switches the centred-query screen on by itself."""
    from oracle import knn as ok
    from viquae_amd.index import METRIC_IP_CENTRED
    g = torch.Generator(device="cuda").manual_seed(768)
    mu = torch.randn((1, 768), generator=g, device="cuda")
    mu = 14.0 * mu / mu.norm()
    X = (mu + 0.25 * torch.randn((70000, 768), generator=g, device="cuda")).cpu().numpy()
    Q = (mu + 0.25 * torch.randn((256, 768), generator=g, device="cuda")).cpu().numpy()
    idx = _index(X, 0)
    assert idx._screen_metric == METRIC_IP_CENTRED
    (D1, I1), (D0, I0), stats = _stream_and_tile(idx, Q, 100)
    assert np.array_equal(I1, I0) and np.array_equal(D1, D0)
    Do, Io = ok.knn(X, Q, 100)
    assert np.array_equal(I1, Io) and np.array_equal(D1, Do)
    assert stats[0] == 0
