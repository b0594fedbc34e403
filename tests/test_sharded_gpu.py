"""The N > 1 search path on ONE MI355X: (a) through real RCCL -- init_process_group("nccl"), world size 1,
ShardedFlatIndex with the collective forced, shard records, mq_topk_merge_records_f32; (b) the 8-shard shape of BASELINE
configs[4] as eight shards on one device (LocalShardsFlatIndex: records copied device-to-device, merged); (c) the
reference's `device` convention from KnowledgeBase (device=-1 / list).  Everything must equal one unsharded index."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def rccl_world1():
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("metric,factory", [(0, "Flat"), (1, "Flat"), (0, "L2norm,Flat")])
def test_rccl_world1_records_gather_and_merge_equal_unsharded(rccl_world1, metric, factory):
    import torch
    from viquae_amd.index import MI355XFlatIndex
    from viquae_amd.sharded import ShardedFlatIndex
    assert rccl_world1.get_backend() == "nccl"
    rng = np.random.default_rng(metric + len(factory))
    X = rng.standard_normal((30000, 256), dtype=np.float32)
    Q = rng.standard_normal((9000, 256), dtype=np.float32)     # three pipelined chunks (4096 + 4096 + 808)
    sharded = ShardedFlatIndex(string_factory=factory, metric_type=metric, always_gather=True)
    sharded.add_vectors(X)
    assert sharded.world == 1 and sharded.ntotal == 30000 and sharded.always_gather
    D, I = sharded.search_batch(Q, 100)
    single = MI355XFlatIndex(string_factory=factory, metric_type=metric)
    single.add_vectors(X)
    Ds, Is = single.search_batch(Q, 100)
    assert np.array_equal(I, Is) and np.array_equal(D, Ds)
    # a rank with an EMPTY shard still takes part (it used to raise while its peers sat in the collective)
    empty = ShardedFlatIndex(string_factory=factory, metric_type=metric, always_gather=True)
    empty.d, empty.ntotal = 256, 0
    De, Ie = empty.search_device(torch.from_numpy(Q[:100]).cuda(), 10)
    assert (Ie == -1).all() and (De.abs() == torch.finfo(torch.float32).max).all()


@pytest.mark.parametrize("metric", [0, 1])
def test_second_halves_gather_and_merge_on_the_second_stream(rccl_world1, monkeypatch, metric):
    """Several chunks over a screened shard: only the scans stay on the caller's stream (sharded.py search_device);
    bit-identical to the serial chunk loop (MQ_KNN_TAIL_OVERLAP=0) and to the exact scan, also twice in a row."""
    import torch
    from viquae_amd.index import MI355XFlatIndex
    from viquae_amd.sharded import ShardedFlatIndex
    g = torch.Generator(device="cuda").manual_seed(11 + metric)
    X = torch.randn((40000, 64), generator=g, device="cuda")
    Q = torch.randn((4096 * 3 + 77, 64), generator=g, device="cuda")
    sh = ShardedFlatIndex(string_factory="Flat", metric_type=metric, always_gather=True)
    sh.add_vectors(X.cpu().numpy())
    exact = MI355XFlatIndex(string_factory="Flat", metric_type=metric, screen=False)
    exact.add(X)
    De, Ie = exact.search_device(Q, 30)
    monkeypatch.setenv("MQ_KNN_TAIL_OVERLAP", "0")
    D0, I0 = sh.search_device(Q, 30)
    monkeypatch.setenv("MQ_KNN_TAIL_OVERLAP", "1")
    D1, I1 = sh.search_device(Q, 30)
    D2, I2 = sh.search_device(Q, 30)
    for D, I in ((D0, I0), (D1, I1), (D2, I2)):
        assert torch.equal(I, Ie) and torch.equal(D, De)


def test_pipelined_chunks_beyond_the_screen_keep_their_records_until_gathered(rccl_world1, monkeypatch):
    """for k beyond the screen (> 224) the FRONT half writes D / I straight into the chunk's record; with two records
    alternating, chunk i may only be scanned once chunk i-2's all-gather has finished reading the same record (the wait is on an
    event recorded after finish(i-2), not after its second half).  Five chunks, k = 300, collective forced, several times over."""
    import torch
    from viquae_amd.index import MI355XFlatIndex
    from viquae_amd.sharded import ShardedFlatIndex
    g = torch.Generator(device="cuda").manual_seed(21)
    X = torch.randn((20000, 48), generator=g, device="cuda")
    Q = torch.randn((4096 * 4 + 300, 48), generator=g, device="cuda")
    sh = ShardedFlatIndex(string_factory="Flat", metric_type=0, always_gather=True)
    sh.add_vectors(X.cpu().numpy())
    exact = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=False)
    exact.add(X)
    De, Ie = exact.search_device(Q, 300)
    monkeypatch.setenv("MQ_KNN_TAIL_OVERLAP", "1")
    for _ in range(3):
        D, I = sh.search_device(Q, 300)
        assert torch.equal(I, Ie) and torch.equal(D, De)


def test_knowledge_base_builds_the_sharded_index_under_torch_distributed(rccl_world1, monkeypatch, tmp_path):
    """b2: inside a torch.distributed job add_or_load_faiss_index must build this rank's shard, not a full index."""
    from viquae_amd import sharded
    monkeypatch.setattr(rccl_world1, "get_world_size", lambda group=None: 2)  # what a rank of a 2-GPU job sees
    made = sharded.make_flat_index(device=None, string_factory="Flat", metric_type=0)
    assert isinstance(made, sharded.ShardedFlatIndex) and made.world == 2 and made.rank == 0
    assert made.local._torch_device is not None          # resolved at construction (an empty shard needs it)


@pytest.mark.parametrize("metric", [0, 1])
def test_eight_shards_on_one_device_equal_unsharded(metric):
    """BASELINE configs[4]'s shape (8 row shards, per-shard top-100, merge) with every shard on cuda:0."""
    import torch
    from viquae_amd.index import MI355XFlatIndex
    from viquae_amd.sharded import LocalShardsFlatIndex, shard_bounds
    rng = np.random.default_rng(8 + metric)
    n = 8 * 20000 - 37
    X = rng.standard_normal((n, 768), dtype=np.float32)
    Q = rng.standard_normal((300, 768), dtype=np.float32)
    idx = LocalShardsFlatIndex([0] * 8, string_factory="Flat", metric_type=metric, allow_repeated_devices=True)
    idx.add_vectors(X)
    assert [s.ntotal for s in idx.shards] == [hi - lo for lo, hi in (shard_bounds(n, 8, r) for r in range(8))]
    D, I = idx.search_batch(Q, 100)
    single = MI355XFlatIndex(string_factory="Flat", metric_type=metric)
    single.add_vectors(X)
    Ds, Is = single.search_batch(Q, 100)
    assert np.array_equal(I, Is) and np.array_equal(D, Ds)
    # tiny KB: trailing shards are empty (N < 64 * shards)
    small = LocalShardsFlatIndex([0] * 8, string_factory="Flat", metric_type=metric, allow_repeated_devices=True)
    small.add_vectors(X[:200])
    assert [s.ntotal for s in small.shards] == [64, 64, 64, 8, 0, 0, 0, 0]
    one = MI355XFlatIndex(string_factory="Flat", metric_type=metric)
    one.add_vectors(X[:200])
    a, b = small.search_batch(Q[:30], 100), one.search_batch(Q[:30], 100)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    # save (one whole-matrix file) -> single-index load
    import tempfile
    with tempfile.TemporaryDirectory() as t:
        idx.save(os.path.join(t, "kb.index"))
        back = MI355XFlatIndex.load(os.path.join(t, "kb.index"))
        c = back.search_batch(Q[:64], 100)
        assert np.array_equal(c[1], Is[:64]) and np.array_equal(c[0], Ds[:64])


def test_knowledge_base_device_minus_one_and_device_list():
    """The reference's `device` key (datasets/search.py:315-347): -1 = all GPUs of the process, a list = those GPUs."""
    import datasets
    from viquae_amd.ir.search import KnowledgeBase
    rng = np.random.default_rng(2)
    X = rng.standard_normal((3000, 64), dtype=np.float32)
    Q = rng.standard_normal((40, 64), dtype=np.float32)
    ds = datasets.Dataset.from_dict({"vec": [r for r in X]})
    res = []
    for device in (None, 0, -1, [0]):
        kb = KnowledgeBase(dataset=ds, index_kwargs={"idx": {"column": "vec", "string_factory": "L2norm,Flat", "device": device,
                                                             "metric_type": 0}})
        res.append(kb.search_batch("idx", Q, k=100))
    # 0, -1 and [0] name the same device: the reference's GPU work-around (numpy's L2norm on the column) -- equal bit for bit.
    for D, I in res[2:]:
        assert np.array_equal(I, res[1][1]) and np.array_equal(D, res[1][0])
    # `device: null` is FAISS's own NormalizationTransform (x * float(1.0 / sqrt(nr))): the same neighbours, scores within the
    # last bits of the other arithmetic (tests/test_l2norm_forms_gpu.py pins each form against its oracle)
    assert np.array_equal(res[0][1], res[1][1]) and np.allclose(res[0][0], res[1][0], rtol=0, atol=3e-7)
    assert not np.array_equal(res[0][0], res[1][0])


def test_faiss_file_written_by_the_reference_loads(tmp_path):
    """`load: true` of the shipped configs: a faiss.write_index file of IndexPreTransform(L2norm) + IndexFlatIP."""
    import struct
    from oracle import knn as ok
    from viquae_amd.index import MI355XFlatIndex
    rng = np.random.default_rng(6)
    X = ok.l2norm_rows(rng.standard_normal((1000, 48), dtype=np.float32), form="faiss")   # what the FAISS file stores: normalised rows
    Q = rng.standard_normal((25, 48), dtype=np.float32)

    def hdr(d, n, metric):
        return struct.pack("<iqqq?i", d, n, 1 << 20, 1 << 20, True, metric)

    p = tmp_path / "kb.faiss"
    p.write_bytes(b"IxPT" + hdr(48, 1000, 0) + struct.pack("<i", 1) + b"VNrm" + struct.pack("<f", 2.0) + struct.pack("<ii?", 48, 48, True)
                  + b"IxFI" + hdr(48, 1000, 0) + struct.pack("<Q", 48000) + X.tobytes())
    idx = MI355XFlatIndex.load(str(p))
    assert idx.do_l2norm and idx.metric_type == 0 and idx.ntotal == 1000
    D, I = idx.search_batch(Q, 20)
    assert idx.l2norm_form == "faiss"  # a stored IndexPreTransform is FAISS's object: its queries take FAISS's NormalizationTransform
    Do, Io = ok.knn(X, ok.l2norm_rows(Q, form="faiss"), 20, metric=0)
    assert np.array_equal(I, Io) and np.array_equal(D, Do)


def test_config4_full_size_12M_rows_in_eight_shards_16k_queries():
    """BASELINE configs[4] at full size on ONE MI355X (288 GB of HBM hold it): 12M x 768 fp32 rows in eight 1.5M-row shards
    (screened search per shard, shard records, merge) against ONE unsharded 12M-row index searched by the exact fp32 scan --
    16,384 queries, top-100, scores and ids bit for bit.  Only the transport differs from the 8-GPU job (device-to-device
    copies instead of the RCCL all-gather, which tests/test_sharded_gpu.py runs at world size 1)."""
    import torch
    from viquae_amd.index import MI355XFlatIndex
    from viquae_amd.sharded import LocalShardsFlatIndex
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if free < 150 * (1 << 30):
        pytest.skip(f"needs ~135 GB of free HBM, {free >> 30} GB available")
    from viquae_amd.sharded import shard_bounds
    n_total, shards, d, nq, k = 12_000_000, 8, 768, 16384, 100
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    sharded = LocalShardsFlatIndex([0] * shards, string_factory="Flat", metric_type=0, allow_repeated_devices=True, screen=True)
    whole = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=False)
    bounds = [shard_bounds(n_total, shards, r) for r in range(shards)]     # 1,500,032 rows per shard (whole panels), the last 1,499,776
    for sh, (lo, _) in zip(sharded.shards, bounds):
        sh.id_offset = lo
    planted = {}
    for b0 in range(0, n_total, 1 << 16):
        x = torch.randn((min(1 << 16, n_total - b0), d), generator=g, device=dev)
        whole.add(x, total_hint=n_total)
        for r, (lo, hi) in enumerate(bounds):
            s0, e0 = max(lo, b0), min(hi, b0 + x.shape[0])
            if s0 < e0:
                sharded.shards[r].add(x[s0 - b0:e0 - b0], total_hint=hi - lo)
                if s0 == lo:
                    planted[lo + 5] = x[lo + 5 - b0].clone()   # one known row per shard
    sharded.ntotal, sharded.d = n_total, d
    assert whole.ntotal == n_total and [sh.ntotal for sh in sharded.shards] == [hi - lo for lo, hi in bounds]
    Q = torch.randn((nq, d), generator=g, device=dev)
    for i, (gid, x) in enumerate(planted.items()):
        Q[i * 1000] = 2.0 * x                              # its own row must be its top-1, whatever shard holds it
    D, I = sharded.search_device(Q, k)
    D0, I0 = whole.search_device(Q, k)
    torch.cuda.synchronize()
    assert torch.equal(I, I0) and torch.equal(D, D0)
    for i, gid in enumerate(planted):
        assert int(I[i * 1000, 0]) == gid
    assert int(I.max()) >= bounds[7][0] and int(I.min()) >= 0   # neighbours come from every part of the id range
    del sharded, whole
    torch.cuda.empty_cache()
