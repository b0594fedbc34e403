"""CPU suite: host-side mirror of meerqat.ir.search (no GPU: a test-only index object backed by the
oracle is registered in place of the HIP index, exactly where MI355XFlatIndex would sit)."""
import os
import sys

import numpy as np
import pytest

from oracle import knn as ok

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _oracle_index(X, metric, l2norm=False):
    from datasets.search import BaseIndex, BatchedSearchResults

    class OracleIndex(BaseIndex):
        def search_batch(self, queries, k=10, **kw):
            if len(np.shape(queries)) != 2:
                raise ValueError("Shape of query must be 2D")
            D, I = ok.knn(X, queries, k, metric=metric, l2norm=l2norm, l2norm_form="faiss")  # FAISS's NormalizationTransform
            return BatchedSearchResults(D, I.astype(int))

    return OracleIndex()


def _kb(X, metric=0, l2norm=False):
    import datasets
    from viquae_amd.ir.search import Index, KnowledgeBase, register_index
    ds = datasets.Dataset.from_dict({"vec": [r for r in X]})
    kb = KnowledgeBase(dataset=ds)
    register_index(kb.dataset, "idx", _oracle_index(X, metric, l2norm))
    kb.indexes["idx"] = Index(key="q", do_L2norm=l2norm)
    return kb


def test_L2norm_matches_reference_definition():
    from viquae_amd.ir.search import L2norm
    rng = np.random.default_rng(0)
    q = rng.standard_normal((4, 9)).astype(np.float32)
    assert np.array_equal(L2norm(q), q / np.linalg.norm(q, axis=1, keepdims=True))
    with np.errstate(invalid="ignore"):
        assert np.isnan(L2norm(np.zeros((1, 3), np.float32))).all()


def test_search_batch_golden_plumbing():
    z = np.load(os.path.join(GOLDEN, "knn_random.npz"))
    X, Q = z["X"], z["Q"]
    kb = _kb(X)
    D, I = kb.search_batch("idx", [list(map(float, q)) for q in Q], k=10)
    assert D.dtype == np.float32 and np.array_equal(I, z["I_m0_k10"]) and np.array_equal(D, z["D_m0_k10"])
    kb = _kb(X, l2norm=True)  # host L2norm on queries (do_L2norm) + the index's own transform
    D, I = kb.search_batch("idx", Q, k=10)
    assert np.array_equal(I, z["I_l2norm_m0_k10"]) and np.array_equal(D, z["D_l2norm_m0_k10"])


def test_search_batch_if_not_None_golden():
    z = np.load(os.path.join(GOLDEN, "knn_random.npz"))
    X, Q, mask = z["X"], z["Q"], z["none_mask"]
    kb = _kb(X)
    queries = [Q[i] if mask[i] else None for i in range(len(Q))]
    S, Ix = kb.search_batch_if_not_None("idx", queries, k=10)
    assert len(S) == len(Q)
    got_D = np.stack([s for s, m in zip(S, mask) if m])
    got_I = np.stack([s for s, m in zip(Ix, mask) if m])
    assert np.array_equal(got_I, z["I_none_m0_k10"]) and np.array_equal(got_D, z["D_none_m0_k10"])
    assert all(len(s) == 0 and len(i) == 0 for s, i, m in zip(S, Ix, mask) if not m)
    S, Ix = kb.search_batch_if_not_None("idx", [None, None], k=3)
    assert S == [[], []] and Ix == [[], []]


def test_dataset_api_sees_registered_index():
    X = np.random.default_rng(1).standard_normal((50, 8)).astype(np.float32)
    kb = _kb(X, metric=1)
    assert kb.dataset.list_indexes() == ["idx"]
    res = kb.dataset.search_batch("idx", X[:2], k=3)
    assert np.array_equal(np.asarray(res.total_indices)[:, 0], [0, 1])
    from datasets.search import MissingIndex
    with pytest.raises(MissingIndex):
        kb.dataset.search_batch("nope", X[:2], k=3)


def test_string_factory_and_metric_parsing():
    from viquae_amd.index import MI355XFlatIndex, parse_string_factory
    assert parse_string_factory(None) is False and parse_string_factory("Flat") is False
    assert parse_string_factory("L2norm,Flat") is True
    for bad in ("IVF4096,Flat", "HNSW32", "L2norm,PQ16", "PCA64,Flat"):
        with pytest.raises(ValueError):
            parse_string_factory(bad)
    assert MI355XFlatIndex(string_factory="Flat").metric_type == 1  # FAISS default metric is L2
    assert MI355XFlatIndex(metric_type=0).metric_type == 0
    with pytest.raises(ValueError):
        MI355XFlatIndex(metric_type=7)
    with pytest.raises(ValueError):
        MI355XFlatIndex(custom_index=object())


def test_index_kwargs_accept_legacy_keys_and_reject_sparse_kinds(monkeypatch):
    """Every shipped experiments/ir/**/config.json carries es/kind_str/normalization/interpolation_weight."""
    import datasets
    from viquae_amd.ir import search as S
    made = {}

    class FakeIndex:
        def __init__(self, device=None, string_factory=None, metric_type=None, screen=None):
            made.update(device=device, string_factory=string_factory, metric_type=metric_type)

        def add_vectors(self, ds, column=None, **kw):
            made["column"] = column

    from viquae_amd import sharded
    monkeypatch.setattr(sharded, "MI355XFlatIndex", FakeIndex)  # what make_flat_index builds for one GPU
    ds = datasets.Dataset.from_dict({"DPR_few_shot": [[0.0, 1.0]]})
    kw = {"column": "DPR_few_shot", "es": False, "kind_str": "TEXT", "key": "DPR_few_shot",
          "normalization": {"method": "normalize", "mean": 71.3295, "std": 2.16671}, "string_factory": "Flat",
          "load": False, "device": None, "metric_type": 0}  # experiments/ir/viquae/dpr/search/config.json:5-19
    kb = S.KnowledgeBase(dataset=ds, index_kwargs={"DPR_few_shot_dp": kw})
    assert made == {"device": None, "string_factory": "Flat", "metric_type": 0, "column": "DPR_few_shot"}
    assert kb.indexes["DPR_few_shot_dp"].key == "DPR_few_shot" and not kb.indexes["DPR_few_shot_dp"].do_L2norm
    assert "DPR_few_shot_dp" in kb.dataset.list_indexes()
    with pytest.raises(NotImplementedError):
        kb.add_or_load_index(column="x", kind="ES")
    with pytest.raises(KeyError):
        kb.add_or_load_index(column="x", kind="NOPE")


def test_make_flat_index_follows_the_reference_device_convention(monkeypatch):
    """datasets/search.py:315-347: int >= 0 -> that GPU, int < 0 -> all GPUs, list -> those GPUs (b2)."""
    from viquae_amd import sharded

    class One:
        def __init__(self, device=None, **kw):
            self.device = device

    class Many:
        def __init__(self, devices, **kw):
            self.devices = list(devices)

    monkeypatch.setattr(sharded, "MI355XFlatIndex", One)
    monkeypatch.setattr(sharded, "LocalShardsFlatIndex", Many)
    monkeypatch.setattr(sharded, "visible_gpus", lambda: [0, 1, 2, 3])
    assert sharded.make_flat_index(device=None).device is None
    assert sharded.make_flat_index(device=2).device == 2
    assert sharded.make_flat_index(device=[3]).device == [3]
    assert sharded.make_flat_index(device=-1).devices == [0, 1, 2, 3]
    assert sharded.make_flat_index(device=[1, 3]).devices == [1, 3]
    monkeypatch.setattr(sharded, "visible_gpus", lambda: [0])
    assert sharded.make_flat_index(device=-1).device == 0   # "all GPUs" of a one-GPU box


def test_shard_record_layout_matches_the_c_abi():
    from viquae_amd import _lib
    from viquae_amd.sharded import record_layout, _record_views, _gathered_views
    import torch
    lib = _lib.load()
    for nq, k in ((1, 1), (3, 5), (4096, 100), (7, 128), (255, 33)):
        rec, ids = record_layout(nq, k)
        assert rec == lib.mq_shard_record_bytes(nq, k) and ids == lib.mq_shard_record_ids_offset(nq, k)
        assert rec % 16 == 0 and ids % 8 == 0 and ids >= nq * k * 4 and rec >= ids + nq * k * 8
        buf = torch.zeros(2 * rec, dtype=torch.uint8)
        for r in range(2):
            D, I = _record_views(buf[r * rec:(r + 1) * rec], nq, k)
            D.fill_(r + 0.5)
            I.fill_(r + 7)
        Ds, Is = _gathered_views(buf, 2, nq, k)
        assert Ds.shape == (2, nq, k) and float(Ds[1, -1, -1]) == 1.5 and int(Is[0, 0, 0]) == 7 and int(Is[1, -1, -1]) == 8


def test_faiss_flat_files_of_the_reference_can_be_read(tmp_path):
    """save_path / save_faiss_index of the reference write faiss.write_index files: IndexFlat ("Flat") or
    IndexPreTransform + NormalizationTransform + IndexFlat ("L2norm,Flat").  Layout per FAISS's published
    index_write.cpp (files assembled by hand: no FAISS binary in this image)."""
    import struct
    from viquae_amd.index import read_index_file_header
    X = np.arange(12, dtype=np.float32).reshape(4, 3)

    def hdr(d, n, metric):
        return struct.pack("<iqqq?i", d, n, 1 << 20, 1 << 20, True, metric)

    flat = b"IxF2" + hdr(3, 4, 1) + struct.pack("<Q", 12) + X.tobytes()
    p = tmp_path / "flat.faiss"
    p.write_bytes(flat)
    n, d, metric, l2norm, off = read_index_file_header(str(p))
    assert (n, d, metric, l2norm) == (4, 3, 1, False)
    assert np.array_equal(np.fromfile(str(p), dtype=np.float32, count=12, offset=off).reshape(4, 3), X)
    pre = (b"IxPT" + hdr(3, 4, 0) + struct.pack("<i", 1) + b"VNrm" + struct.pack("<f", 2.0) + struct.pack("<ii?", 3, 3, True)
           + b"IxFI" + hdr(3, 4, 0) + struct.pack("<Q", 12) + X.tobytes())
    p2 = tmp_path / "pre.faiss"
    p2.write_bytes(pre)
    n, d, metric, l2norm, off = read_index_file_header(str(p2))
    assert (n, d, metric, l2norm) == (4, 3, 0, True)
    assert np.array_equal(np.fromfile(str(p2), dtype=np.float32, count=12, offset=off).reshape(4, 3), X)
    bad = tmp_path / "ivf.faiss"
    bad.write_bytes(b"IwFl" + hdr(3, 4, 1))
    with pytest.raises(ValueError):
        read_index_file_header(str(bad))


def test_shard_bounds_cover_rows_once():
    from viquae_amd.sharded import shard_bounds
    for n in (0, 1, 63, 64, 1000, 1_500_000, 12_000_001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(lo % 64 == 0 for lo, hi in spans if hi > lo)


def test_iter_arrow_column_zero_copy_blocks():
    import datasets
    from viquae_amd.index import iter_arrow_column
    X = np.random.default_rng(2).standard_normal((1000, 12)).astype(np.float32)
    ds = datasets.Dataset.from_dict({"v": [r for r in X], "t": [str(i) for i in range(1000)]})
    got = np.concatenate(list(iter_arrow_column(ds, "v")))
    assert got.dtype == np.float32 and np.array_equal(got, X)
    sel = ds.select([5, 3, 999])
    got = np.concatenate(list(iter_arrow_column(sel, "v")))
    assert np.array_equal(got, X[[5, 3, 999]])


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from viquae_amd._lib import MeerqatHipError
    from viquae_amd.index import MI355XFlatIndex
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0)
    with pytest.raises(MeerqatHipError):
        idx.add_vectors(np.zeros((4, 8), np.float32))


def test_length_partition_is_optimal_and_covers_everything():
    """_partition_by_length (padding-aware encoder forward): contiguous groups of the ascending lengths, every sequence in
    exactly one, total padded tokens no worse than equal-count groups and equal to brute force on a small case."""
    import itertools
    from viquae_amd.encoders import _partition_by_length
    rng = np.random.default_rng(0)
    lens = np.sort(np.clip(rng.normal(130, 30, 2048).astype(int), 8, 256))
    for nb in (1, 2, 4, 8):
        parts = _partition_by_length(lens, nb)
        assert len(parts) <= nb and parts[0].start == 0 and parts[-1].stop == len(lens)
        assert all(a.stop == b.start for a, b in zip(parts, parts[1:]))
        cost = sum((p.stop - p.start) * lens[p].max() for p in parts)
        equal = sum(len(c) * c.max() for c in np.array_split(lens, nb))
        assert cost <= equal
    small = np.sort(rng.integers(1, 30, 12))
    best = min(sum((b - a) * small[a:b].max() for a, b in zip((0,) + cut, cut + (12,)))
               for r in range(0, 3) for cut in itertools.combinations(range(1, 12), r))
    got = _partition_by_length(small, 3)
    assert sum((p.stop - p.start) * small[p].max() for p in got) == best
    assert _partition_by_length(np.array([5, 5, 5]), 4) == [slice(0, 3)]


def test_padding_plan_host_logic_on_cpu_tensors():
    """_length_buckets is plain tensor logic: right-padded masks with something to cut get a plan that covers every
    sequence once, each group cut at its longest length; everything else declines."""
    import torch
    from viquae_amd.encoders import _length_buckets
    rng = np.random.default_rng(0)
    lens = rng.integers(8, 200, 4000)
    mask = torch.from_numpy((np.arange(256)[None] < lens[:, None]).astype(np.int64))
    plan = _length_buckets(mask)
    assert plan is not None and 1 <= len(plan) <= 8
    seen = np.concatenate([idx.numpy() for idx, _ in plan])
    assert sorted(seen.tolist()) == list(range(4000))
    for idx, L in plan:
        assert L == lens[idx.numpy()].max()
    assert _length_buckets(torch.ones((4000, 256), dtype=torch.int64)) is None           # nothing to cut
    assert _length_buckets(mask.flip(1)) is None                                         # left padding
    assert _length_buckets(None) is None
    few = _length_buckets(mask[:50])                                                      # small batch: one group at its longest
    assert few is not None and len(few) == 1 and few[0][1] == lens[:50].max()
    os.environ["MQ_ENC_PAD_SKIP"] = "0"
    try:
        assert _length_buckets(mask) is None
    finally:
        del os.environ["MQ_ENC_PAD_SKIP"]


def test_bench_gpus_n_starts_its_own_ranks_before_touching_a_gpu(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment must launch N ranks itself (a
    torch.distributed.run child), and must do so before anything initialises HIP in the parent."""
    import importlib
    import subprocess
    import torch
    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("parent touched the GPU")))
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # more ranks than GPUs: refused before anything is launched
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    seen.clear()
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "only 2 GPU" in str(e.value.code) and not seen


def test_faiss_file_writer_round_trips_through_the_reader_and_matches_the_hand_assembled_layout(tmp_path):
    """save() of a "*.faiss" / "*.index" path (or format="faiss") writes what faiss.write_index writes for IndexFlat /
    IndexPreTransform + NormalizationTransform + IndexFlat (meerqat/ir/search.py:247-248: save_faiss_index): byte-identical
    to the hand-assembled files of the reader test above, and read back by read_index_file_header."""
    import struct
    from viquae_amd.index import index_file_format, index_file_header, read_index_file_header
    X = np.arange(12, dtype=np.float32).reshape(4, 3)

    def hdr(d, n, metric):
        return struct.pack("<iqqq?i", d, n, 1 << 20, 1 << 20, True, metric)

    assert index_file_header(4, 3, 1, False, "faiss") == b"IxF2" + hdr(3, 4, 1) + struct.pack("<Q", 12)
    assert index_file_header(4, 3, 0, True, "faiss") == (b"IxPT" + hdr(3, 4, 0) + struct.pack("<i", 1) + b"VNrm" + struct.pack("<f", 2.0)
                                                         + struct.pack("<ii?", 3, 3, True) + b"IxFI" + hdr(3, 4, 0) + struct.pack("<Q", 12))
    assert index_file_format("kb/dpr.faiss") == "faiss" and index_file_format("kb/dpr.index") == "faiss"
    assert index_file_format("kb/dpr.bin") == "mqflat" and index_file_format("kb/dpr.bin", "faiss") == "faiss"
    with pytest.raises(ValueError):
        index_file_format("x", "hnsw")
    for metric, l2norm, fmt in ((0, False, "faiss"), (1, False, "faiss"), (0, True, "faiss"), (1, True, "faiss"), (0, True, "mqflat")):
        p = tmp_path / f"m{metric}{int(l2norm)}.{fmt}"
        head = index_file_header(4, 3, metric, l2norm, fmt)
        p.write_bytes(head + X.tobytes())
        n, d, m, l2, off = read_index_file_header(str(p))
        assert (n, d, m, l2, off) == (4, 3, metric, l2norm, len(head))
    # an empty index still has a readable header
    p = tmp_path / "empty.faiss"
    p.write_bytes(index_file_header(0, 3, 0, False, "faiss"))
    assert read_index_file_header(str(p))[:4] == (0, 3, 0, False)
