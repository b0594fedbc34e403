"""GPU parity against the committed golden vectors, called the way the reference is called:
KnowledgeBase(index_kwargs=...) -> add_or_load_faiss_index -> search_batch / search_batch_if_not_None
(meerqat/ir/search.py:102-249).  The goldens were minted by the reference's own KnowledgeBase
(tools/make_golden.py).  Bar: bit-exact scores and indices."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _kb(X, factory, metric):
    import datasets
    from viquae_amd.ir.search import KnowledgeBase
    ds = datasets.Dataset.from_dict({"vec": [r for r in X.astype(np.float32)], "passage": [str(i) for i in range(len(X))]})
    # index kwargs exactly as experiments/ir/viquae/dpr/search/config.json writes them (legacy keys included)
    kw = {"column": "vec", "es": False, "kind_str": "TEXT", "key": "q", "normalization": {"method": "normalize"},
          "string_factory": factory, "load": False, "device": None, "metric_type": metric}
    return KnowledgeBase(dataset=ds, index_kwargs={"idx": kw})


def _tags(z):
    for key in z.files:
        if key.startswith("I_") and not key.startswith("I_none"):
            tag = key[2:]
            l2 = tag.startswith("l2norm_")
            m, k = tag.replace("l2norm_", "").split("_")
            yield tag, l2, int(m[1:]), int(k[1:])


@pytest.mark.parametrize("name", ["random", "small_nq", "lattice", "ties", "lattice_small_nq", "ties_small_nq"])
def test_knowledge_base_reproduces_reference_goldens(name):
    z = np.load(os.path.join(GOLDEN, f"knn_{name}.npz"))
    X, Q = z["X"].astype(np.float32), z["Q"].astype(np.float32)
    kbs = {}
    for tag, l2, metric, k in _tags(z):
        key = (l2, metric)
        if key not in kbs:
            kbs[key] = _kb(X, "L2norm,Flat" if l2 else "Flat", metric)
            assert kbs[key].indexes["idx"].do_L2norm == l2
        D, I = kbs[key].search_batch("idx", [list(map(float, q)) for q in Q], k=k)
        assert isinstance(D, np.ndarray) and D.dtype == np.float32 and I.dtype.kind == "i"
        assert np.array_equal(I, z[f"I_{tag}"]), f"{name}/{tag}: indices differ"
        assert np.array_equal(D, z[f"D_{tag}"]), f"{name}/{tag}: scores differ"


def test_none_queries_golden():
    z = np.load(os.path.join(GOLDEN, "knn_random.npz"))
    X, Q, mask = z["X"], z["Q"], z["none_mask"]
    kb = _kb(X, "Flat", 0)
    S, Ix = kb.search_batch_if_not_None("idx", [Q[i] if mask[i] else None for i in range(len(Q))], k=10)
    assert np.array_equal(np.stack([s for s, m in zip(Ix, mask) if m]), z["I_none_m0_k10"])
    assert np.array_equal(np.stack([s for s, m in zip(S, mask) if m]), z["D_none_m0_k10"])
    assert all(len(s) == 0 for s, m in zip(S, mask) if not m)


def test_dataset_level_api_and_errors():
    from datasets.search import MissingIndex
    z = np.load(os.path.join(GOLDEN, "knn_random.npz"))
    kb = _kb(z["X"], "Flat", 1)
    ds = kb.dataset
    scores, examples = ds.get_nearest_examples("idx", z["X"][5], k=3)
    assert examples["passage"][0] == "5" and scores[0] == 0.0
    res = ds.get_nearest_examples_batch("idx", z["X"][:4], k=2)
    assert [e["passage"][0] for e in res.total_examples] == ["0", "1", "2", "3"]
    with pytest.raises(ValueError):
        ds.search_batch("idx", z["X"][0], k=3)  # not 2-D
    with pytest.raises(MissingIndex):
        ds.search_batch("nope", z["X"][:2], k=3)
    with pytest.raises(NotImplementedError):
        ds.search_batch("idx", z["X"][:2], k=2049)  # MQ_KNN_MAX_K = 2048 (k = 129 .. 2048: tests/test_tie_order_bigk_gpu.py)


def test_save_path_then_load(tmp_path):
    z = np.load(os.path.join(GOLDEN, "knn_random.npz"))
    import datasets
    from viquae_amd.ir.search import KnowledgeBase
    ds = datasets.Dataset.from_dict({"vec": [r for r in z["X"]]})
    p = str(tmp_path / "flat.mq")
    kb = KnowledgeBase(dataset=ds, index_kwargs={"a": {"column": "vec", "string_factory": "L2norm,Flat", "metric_type": 0,
                                                     "save_path": p}})
    kb2 = KnowledgeBase(dataset=ds, index_kwargs={"a": {"column": "vec", "string_factory": "L2norm,Flat", "metric_type": 0,
                                                      "load": True, "file": p}})
    Da, Ia = kb.search_batch("a", z["Q"], k=10)
    Db, Ib = kb2.search_batch("a", z["Q"], k=10)
    assert np.array_equal(Ia, z["I_l2norm_m0_k10"]) and np.array_equal(Ib, Ia) and np.array_equal(Da, Db)
