"""GPU parity tests of the two round-3 boundary parameters, through the C ABI, against oracle/knn_oracle.c:

* ``tie_order`` in {"id_asc", "id_desc"} (MQ_KNN_FLAG_TIE_ID_DESC / MQ_MERGE_TIE_ID_DESC): which of several EXACTLY tied rows
  ranks first -- membership at the k-th boundary and output order alike -- on every search path (exact fp32 scan, screened
  search and its exact fallback, FAISS's small-batch L2 form) and in the shard merges.  These are THIS LIBRARY'S documented
  policies (oracle/knn_oracle.c header); nothing here is a claim about FAISS's behaviour on exact ties.
* k above 128, up to MQ_KNN_MAX_K = 2048 (the reference's ``--k`` is a user option, meerqat/ir/search.py:12,135; FAISS
  IndexFlat takes any k): ceil(k / 128) exact scans with per-query key ceilings.

Bar: BIT-EXACT scores and indices.  Plus the ADVICE-r2 fix: a multi-GPU single-process index loaded from an "L2norm,Flat"
file normalises its queries on every shard."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FLT_MAX = np.finfo(np.float32).max


def _mk(n, d, nq, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "ties":
        return rng.integers(-2, 3, (n, d)).astype(np.float32), rng.integers(-2, 3, (nq, d)).astype(np.float32)
    if kind == "lattice":
        return rng.integers(-128, 129, (n, d)).astype(np.float32), rng.integers(-128, 129, (nq, d)).astype(np.float32)
    return rng.standard_normal((n, d), dtype=np.float32), rng.standard_normal((nq, d), dtype=np.float32)


def _search(X, Q, k, metric, tie, screen, factory="Flat", keep_panel=None):
    from viquae_amd.index import MI355XFlatIndex
    idx = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=screen, tie_order=tie, keep_panel=keep_panel)
    idx.add_vectors(X)
    return idx.search_batch(Q, k)


def _same(got, want, what=""):
    D, I = got
    Do, Io = want
    bad = np.nonzero((I != Io).any(axis=1))[0]
    assert bad.size == 0, f"{what}: index mismatch in {bad.size} queries, first {bad[:4]}: {I[bad[0]][:12]} vs {Io[bad[0]][:12]}"
    assert np.array_equal(D, Do), f"{what}: score mismatch"


@pytest.mark.parametrize("tie", ["id_asc", "id_desc"])
@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("screen", [False, True])
@pytest.mark.parametrize("nq,k", [(37, 100), (21, 1), (7, 100), (260, 128)])
def test_tie_order_on_tie_heavy_data(tie, metric, screen, nq, k):
    """ints in [-2, 2], d = 16: hundreds of exactly equal scores around every k-th boundary.  nq = 7 with the L2 metric takes
    FAISS's small-batch form; the screened index floods its candidate buffers here and falls back to the exact scan."""
    from oracle import knn as ok
    X, Q = _mk(3000, 16, nq, 11 + nq, "ties")
    _same(_search(X, Q, k, metric, tie, screen), ok.knn(X, Q, k, metric=metric, tie_order=tie), f"{tie} m{metric} s{screen}")


@pytest.mark.parametrize("tie", ["id_asc", "id_desc"])
@pytest.mark.parametrize("metric", [0, 1])
def test_tie_order_with_planted_duplicates_on_the_screened_path(tie, metric):
    """Free-form rows with exact duplicates planted (real KBs hold duplicate passages / images): few enough that the screen
    keeps them all as candidates and does NOT fall back -- the order inside each equal-score pair comes from the re-scored
    keys.  "L2norm,Flat" as well: duplicates stay duplicates after the transform."""
    from oracle import knn as ok
    from viquae_amd.index import MI355XFlatIndex
    rng = np.random.default_rng(3)
    X = rng.standard_normal((20000, 64), dtype=np.float32)
    Q = rng.standard_normal((64, 64), dtype=np.float32)
    top = ok.knn(X, Q, 30, metric=metric)[1]
    for q in range(0, 64, 2):                      # duplicate some of each query's best rows elsewhere in the KB
        for j, src in enumerate(top[q][:5]):
            X[15000 + q * 8 + j] = X[src]
    for factory in ("Flat", "L2norm,Flat"):
        idx = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=True, tie_order=tie)
        idx.add_vectors(X)
        got = idx.search_batch(Q, 100)
        assert idx.screen_stats(64, 100)[0] == 0   # no query tile was recomputed by the exact scan
        _same(got, ok.knn(X, Q, 100, metric=metric, l2norm=factory != "Flat", tie_order=tie), f"{tie} {factory}")
        Da, Ia = ok.knn(X, Q, 100, metric=metric, l2norm=factory != "Flat", tie_order="id_asc")
        if tie == "id_desc" and factory == "Flat":
            assert not np.array_equal(Ia, got[1])  # the knob is observable on this data


@pytest.mark.parametrize("tie", ["id_asc", "id_desc"])
@pytest.mark.parametrize("metric", [0, 1])
def test_shard_merges_follow_the_tie_order(tie, metric):
    """8 shards on one device (records + mq_topk_merge_records_f32) and the dense mq_topk_merge_f32 == one index."""
    import torch
    from oracle import knn as ok
    from viquae_amd.sharded import LocalShardsFlatIndex, _hip_merge
    X, Q = _mk(4000, 16, 50, 5, "ties")
    want = ok.knn(X, Q, 100, metric=metric, tie_order=tie)
    sh = LocalShardsFlatIndex([0] * 8, string_factory="Flat", metric_type=metric, allow_repeated_devices=True, tie_order=tie)
    sh.add_vectors(X)
    _same(sh.search_batch(Q, 100), want, "records merge")
    parts = [ok.knn(X[s:s + 1000], Q, 100, metric=metric, id_offset=s, tie_order=tie) for s in range(0, 4000, 1000)]
    Ds = torch.from_numpy(np.stack([p[0] for p in parts])).cuda()
    Is = torch.from_numpy(np.stack([p[1] for p in parts])).cuda()
    D, I = _hip_merge(Ds, Is, metric, tie)
    _same((D.cpu().numpy(), I.cpu().numpy()), want, "dense merge")


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("screen", [False, True])
@pytest.mark.parametrize("k", [129, 200, 224, 225, 256, 1000])
@pytest.mark.parametrize("kind", ["normal", "ties"])
def test_k_above_128(kind, k, screen, metric):
    """ceil(k / 128) scans with key ceilings == the oracle's single pass, for both index kinds (a screened index without a
    panel copy scans its row-major rows), both metrics, free-form and tie-heavy data (equal-score runs that straddle the
    round boundaries), more than one query tile.  A screened index serves k <= 224 through its screen (a 256-key final
    sort; the tie-heavy data overflows its pools and exercises the fallback rounds under the screen), the rounds beyond."""
    from oracle import knn as ok
    n, d, nq = (5000, 96, 300) if kind == "normal" else (3000, 16, 40)
    X, Q = _mk(n, d, nq, 7 + k, kind)
    _same(_search(X, Q, k, metric, "id_asc", screen), ok.knn(X, Q, k, metric=metric), f"k={k}")


@pytest.mark.parametrize("tie", ["id_asc", "id_desc"])
def test_k_129_to_224_through_the_screen_on_a_large_shard(tie):
    """Enough rows for several slabs and chunks per slab, so that the screen's own machinery (stripe bound, pool compactions,
    candidate selection, 256-key final sort) decides -- against the exact fp32 scan of the same index."""
    import torch
    from viquae_amd.index import MI355XFlatIndex
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.randn((200_000, 128), generator=g, device="cuda")
    X[1000:1040] = X[0:40]            # duplicates: equal scores across the cut
    Q = torch.randn((700, 128), generator=g, device="cuda")
    res = {}
    for screen in (True, False):
        idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=screen, tie_order=tie)
        idx.add(X)
        res[screen] = [tuple(t.cpu().numpy() for t in idx.search_device(Q, k)) for k in (129, 200, 224)]
        if screen:
            st = idx.screen_stats(Q.shape[0], 224)
            assert st[0] == 0, "the 224-neighbour search was meant to stay inside the screen (no tile recomputed)"
    for a, b in zip(res[True], res[False]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_k_above_128_other_shapes():
    from oracle import knn as ok
    # k > ntotal: the tail is (neutral value, -1), whole rounds of it included
    X, Q = _mk(200, 32, 30, 1, "normal")
    for metric in (0, 1):
        D, I = _search(X, Q, 700, metric, "id_asc", True)
        _same((D, I), ok.knn(X, Q, 700, metric=metric), "k > ntotal")
        assert (I[:, 200:] == -1).all() and (np.abs(D[:, 200:]) == FLT_MAX).all()
    # FAISS's small-batch L2 form (fewer than 20 queries) with rounds, both tie orders, "L2norm,Flat"
    X, Q = _mk(4000, 16, 9, 2, "ties")
    for tie in ("id_asc", "id_desc"):
        _same(_search(X, Q, 300, 1, tie, True), ok.knn(X, Q, 300, metric=1, tie_order=tie), f"direct {tie}")
        _same(_search(X, Q, 300, 0, tie, False), ok.knn(X, Q, 300, metric=0, tie_order=tie), f"exact {tie}")
    X, Q = _mk(3000, 48, 25, 3, "normal")
    _same(_search(X, Q, 2048, 0, "id_asc", True, factory="L2norm,Flat"), ok.knn(X, Q, 2048, metric=0, l2norm=True), "k = 2048")
    from viquae_amd.index import MI355XFlatIndex
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0)
    idx.add_vectors(X)
    with pytest.raises(NotImplementedError):
        idx.search_batch(Q, 2049)


@pytest.mark.parametrize("metric", [0, 1])
def test_k_above_128_through_shards(metric):
    from oracle import knn as ok
    from viquae_amd.sharded import LocalShardsFlatIndex
    X, Q = _mk(6000, 32, 70, 9, "lattice")
    sh = LocalShardsFlatIndex([0] * 4, string_factory="Flat", metric_type=metric, allow_repeated_devices=True)
    sh.add_vectors(X)
    _same(sh.search_batch(Q, 500), ok.knn(X, Q, 500, metric=metric), "4 shards, k = 500")


def test_k_1000_at_a_larger_size_equals_torch_on_lattice_data():
    """100k x 128 integer rows (every fp32 summation order is exact), 512 queries, k = 1000: scores must equal a plain
    torch matmul's top-k values, and ids must be a valid ordering of them (ascending inside equal-score runs)."""
    import torch
    from viquae_amd.index import MI355XFlatIndex
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randint(-8, 9, (100_000, 128), generator=g, device="cuda").float()
    Q = torch.randint(-8, 9, (512, 128), generator=g, device="cuda").float()
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
    idx.add(X)
    D, I = idx.search_device(Q, 1000)
    S = Q @ X.T
    want = torch.topk(S, 1000, dim=1).values
    assert torch.equal(D, want)
    assert torch.equal(torch.gather(S, 1, I), D)
    same = D[:, 1:] == D[:, :-1]
    assert bool(((I[:, 1:] > I[:, :-1]) | ~same).all())
    # membership at the boundary: every row strictly better than the k-th is present, ties go to the lowest ids
    kth = D[:, -1:]
    assert bool(((S > kth).sum(1) == (D > kth).sum(1)).all())
    tied_ids = torch.where(S == kth, torch.arange(S.shape[1], device="cuda")[None, :], S.shape[1])
    n_tied = (D == kth).sum(1)
    for q in range(0, 512, 37):
        want_ids = torch.sort(tied_ids[q]).values[: int(n_tied[q])]
        assert torch.equal(torch.sort(I[q][D[q] == kth[q]]).values, want_ids)


@pytest.mark.parametrize("dp", [64, 768, 4096])
@pytest.mark.parametrize("pattern", ["gauss", "all_positive_equal", "alternating_cancellation", "wide_range"])
def test_mfma_accumulation_term_of_the_screening_margin(dp, pattern):
    """The one term of the screening margin that rests on the matrix instruction's undocumented internal summation
    (csrc/knn_screen.inc, screen_margin_kernel: dp * 2.98e-7 * ||q|| * max||x||, of which dp * 2^-22 * ||q~|| ||x~|| is
    allowed for the bf16 MFMA's fp32 accumulation): mq_diag_mfma_bf16_dot returns what v_mfma_f32_32x32x16_bf16 ACTUALLY
    accumulates, chained over dp / 16 steps like the scan does, and it must lie within dp * 2^-22 * ||a|| ||b|| of the float64
    sum of the same bf16 operands -- on Gaussian data, on all-positive equal-magnitude products (the accumulator grows
    monotonically: worst case for absorbed low bits), on alternating-sign cancellation (large partial sums, tiny result) and
    on rows whose scales span 2^-12 .. 2^12.  The measured ratio is reported on failure."""
    import torch
    from viquae_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(dp + len(pattern))
    if pattern == "gauss":
        A = torch.randn((32, dp), generator=g, device="cuda")
        B = torch.randn((32, dp), generator=g, device="cuda")
    elif pattern == "all_positive_equal":
        A = torch.full((32, dp), 1.0, device="cuda") * (1 + torch.arange(32, device="cuda")[:, None] / 128.0)
        B = torch.full((32, dp), 1.0, device="cuda") * (1 + torch.arange(32, device="cuda")[:, None] / 64.0)
    elif pattern == "alternating_cancellation":
        sign = torch.where(torch.arange(dp, device="cuda") % 2 == 0, 1.0, -1.0)
        A = (1.0 + torch.rand((32, dp), generator=g, device="cuda") * 2 ** -6) * sign[None, :] * 100.0
        B = 1.0 + torch.rand((32, dp), generator=g, device="cuda") * 2 ** -6
    else:
        A = torch.randn((32, dp), generator=g, device="cuda") * torch.exp2(torch.randint(-12, 13, (32, 1), generator=g, device="cuda").float())
        B = torch.randn((32, dp), generator=g, device="cuda") * torch.exp2(torch.randint(-12, 13, (1, dp), generator=g, device="cuda").float() / 4)
    Ab, Bb = A.to(torch.bfloat16).contiguous(), B.to(torch.bfloat16).contiguous()
    out = torch.empty((32, 32), dtype=torch.float32, device="cuda")
    _lib.check(lib.mq_diag_mfma_bf16_dot(Ab.data_ptr(), Bb.data_ptr(), dp, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    exact = Ab.double() @ Bb.double().T
    bound = dp * 2.0 ** -22 * Ab.double().norm(dim=1)[:, None] * Bb.double().norm(dim=1)[None, :]
    ratio = ((out.double() - exact).abs() / bound).max()
    assert float(ratio) <= 1.0, f"MFMA accumulation error is {float(ratio):.3f} x the allowed dp * 2^-22 * ||a|| ||b||"
    # and well inside it: the margin takes 4x the textbook bound
    assert float(ratio) <= 0.5, float(ratio)


def test_local_shards_loaded_from_an_l2norm_file_normalise_their_queries(tmp_path):
    """KnowledgeBase(load=True, device=[...]) of an "L2norm,Flat" file built a LocalShardsFlatIndex whose
    shards searched with un-normalised queries.  Must equal MI355XFlatIndex.load on the same file (scores too)."""
    import datasets
    from viquae_amd.index import MI355XFlatIndex
    from viquae_amd.ir.search import KnowledgeBase
    from viquae_amd import sharded
    rng = np.random.default_rng(0)
    X = rng.standard_normal((3000, 48), dtype=np.float32) * 3.0
    Q = rng.standard_normal((33, 48), dtype=np.float32) * 5.0
    for metric in (0, 1):
        src = MI355XFlatIndex(string_factory="L2norm,Flat", metric_type=metric)
        src.add_vectors(X)
        path = str(tmp_path / f"kb_m{metric}.mqflat")
        src.save(path)
        want = MI355XFlatIndex.load(path).search_batch(Q, 50)
        loc = sharded.LocalShardsFlatIndex([0, 0, 0], allow_repeated_devices=True).load_rows(path)
        assert loc.metric_type == metric and loc.do_l2norm and all(s.do_l2norm and s.metric_type == metric for s in loc.shards)
        _same(loc.search_batch(Q, 50), want, f"LocalShards.load_rows m{metric}")
        # through the reference's surface: a load config that names neither factory nor metric
        ds = datasets.Dataset.from_dict({"passage": [str(i) for i in range(len(X))]})
        orig = sharded.visible_gpus
        sharded.visible_gpus = lambda: [0, 0]
        try:
            import viquae_amd.sharded as shmod
            real = shmod.LocalShardsFlatIndex
            shmod.LocalShardsFlatIndex = lambda devs, **kw: real(devs, allow_repeated_devices=True, **kw)
            kb = KnowledgeBase(dataset=ds, index_kwargs={"idx": {"column": "vec", "load": True, "file": path, "device": -1}})
        finally:
            sharded.visible_gpus = orig
            shmod.LocalShardsFlatIndex = real
        D, I = kb.search_batch("idx", Q, k=50)
        assert np.array_equal(I, want[1]) and np.array_equal(D, want[0])


def test_sharded_load_rows_takes_metric_and_transform_from_the_file(tmp_path):
    """ShardedFlatIndex.load_rows with a config that omits string_factory / metric_type."""
    from viquae_amd.index import MI355XFlatIndex
    from viquae_amd.sharded import ShardedFlatIndex
    rng = np.random.default_rng(1)
    X = rng.standard_normal((2000, 32), dtype=np.float32) * 2.0
    Q = rng.standard_normal((25, 32), dtype=np.float32)
    src = MI355XFlatIndex(string_factory="L2norm,Flat", metric_type=0)
    src.add_vectors(X)
    path = str(tmp_path / "kb.mqflat")
    src.save(path)
    want = MI355XFlatIndex.load(path).search_batch(Q, 20)
    sh = ShardedFlatIndex().load_rows(path)   # defaults: "Flat", L2 -- the file says "L2norm,Flat", inner product
    assert sh.metric_type == 0 and sh.local.metric_type == 0 and sh.local.do_l2norm and sh.do_l2norm
    _same(sh.search_batch(Q, 20), want, "ShardedFlatIndex.load_rows")


@pytest.mark.parametrize("metric,factory", [(0, "Flat"), (1, "Flat"), (0, "L2norm,Flat")])
def test_save_in_faiss_format_and_load_back(tmp_path, metric, factory):
    """the reference's save_path (meerqat/ir/search.py:247-248) produces a file FAISS reads back; `save`
    of a "*.faiss" path writes that layout (IxFI / IxF2, IxPT + VNrm for "L2norm,Flat") and `load` returns the same index."""
    import datasets
    from viquae_amd.index import MI355XFlatIndex, read_index_file_header
    from viquae_amd.ir.search import KnowledgeBase
    from viquae_amd.sharded import LocalShardsFlatIndex
    rng = np.random.default_rng(4)
    X = rng.standard_normal((1500, 40), dtype=np.float32) * 2
    Q = rng.standard_normal((30, 40), dtype=np.float32)
    ds = datasets.Dataset.from_dict({"vec": [r for r in X]})
    path = str(tmp_path / "kb.faiss")
    kb = KnowledgeBase(dataset=ds, index_kwargs={"idx": {"column": "vec", "string_factory": factory, "metric_type": metric,
                                                          "save_path": path}})
    want = kb.search_batch("idx", Q, k=60)
    with open(path, "rb") as f:
        assert f.read(4) == (b"IxPT" if "L2norm" in factory else (b"IxFI" if metric == 0 else b"IxF2"))
    n, d, m, l2, off = read_index_file_header(path)
    assert (n, d, m, l2) == (1500, 40, metric, "L2norm" in factory)
    back = MI355XFlatIndex.load(path)
    assert back.do_l2norm == ("L2norm" in factory) and back.metric_type == metric
    # KnowledgeBase.search_batch normalises the queries on the host as well (idempotent): hand the index the same input
    from viquae_amd.ir.search import L2norm
    q_in = L2norm(Q) if "L2norm" in factory else Q
    got = back.search_batch(q_in, 60)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])
    kb2 = KnowledgeBase(dataset=ds, index_kwargs={"idx": {"column": "vec", "load": True, "file": path, "string_factory": factory}})
    got2 = kb2.search_batch("idx", Q, k=60)
    assert np.array_equal(got2[1], want[1]) and np.array_equal(got2[0], want[0])
    # a multi-shard index writes the same file
    sh = LocalShardsFlatIndex([0, 0, 0], string_factory=factory, metric_type=metric, allow_repeated_devices=True,
                              l2norm_form="faiss")  # what KnowledgeBase asked for above (`device` absent = null)
    sh.add_vectors(X)
    sh.save(str(tmp_path / "kb_shards.faiss"))
    assert open(str(tmp_path / "kb_shards.faiss"), "rb").read() == open(path, "rb").read()
