"""GPU: the two arithmetics of the "L2norm," prefix (include/meerqat_hip.h MQ_L2NORM_NUMPY / MQ_L2NORM_FAISS) on every ingest
and search path, against the oracle's restatement of each (oracle/knn_oracle.c) -- including the rows on which they differ in
kind, not only in the last bit: a zero row (FAISS: stays zero, retrievable with score 0; numpy: NaN, never retrieved), a row
whose squares underflow, a row whose squared norm is a denormal.  And the rule that picks the form: the reference's ``device``
key (meerqat/ir/search.py:230-245)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rows(n, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d), dtype=np.float32)
    X[3] = 0.0       # zero row
    X[5] = 1e-30     # x * x underflows to 0: squared norm 0
    X[7] = 1e-20     # squared norm d * 1e-40: a denormal > 0
    X[11] *= 1e18    # a huge row (finite squared norm)
    return X


@pytest.mark.parametrize("form", ["numpy", "faiss"])
@pytest.mark.parametrize("screen", [False, True])
@pytest.mark.parametrize("metric", [0, 1])
def test_both_forms_on_both_index_kinds(form, screen, metric):
    from oracle import knn as ok
    from viquae_amd.index import MI355XFlatIndex
    n, d = 1500, 40
    X = _rows(n, d, 1)
    Q = np.random.default_rng(2).standard_normal((23, d)).astype(np.float32)
    Q[4] = 0.0  # a zero query: NaN scores everywhere in numpy's form, zeros in FAISS's
    idx = MI355XFlatIndex(string_factory="L2norm,Flat", metric_type=metric, screen=screen, l2norm_form=form)
    idx.add(X[:700])
    idx.add(X[700:])  # ragged second add: the row-major / panel paths of both index kinds
    assert idx.l2norm_form == form
    stored = idx.reconstruct_n(0, n)
    want = ok.l2norm_rows(X, form=form)
    assert np.array_equal(stored, want, equal_nan=True)
    for k in (10, n):  # k = n: every retrievable row comes back
        D, I = idx.search_batch(Q, k)
        Do, Io = ok.knn(X, Q, k, metric=metric, l2norm=True, l2norm_form=form)
        assert np.array_equal(I, Io) and np.array_equal(D, Do, equal_nan=True), (form, screen, metric, k)
    D, I = idx.search_batch(Q, n)
    real = [q for q in range(len(Q)) if q != 4]
    if form == "faiss":
        assert all(3 in I[q] for q in real)                      # the zero row is an ordinary row of score 0 / distance ||q||^2
    else:
        assert all(3 not in I[q] for q in real) and (I[real, -1] == -1).all()   # NaN rows never enter a result
        assert (I[4] == -1).all()                                # the zero query became NaN


def test_small_batch_l2_form_and_queries_only():
    """Fewer than 20 L2 queries (FAISS's direct form) with each arithmetic, and mq_l2norm_rows_form_f32 on its own."""
    import ctypes
    import torch
    from oracle import knn as ok
    from viquae_amd import _lib
    from viquae_amd.index import L2NORM_FORMS, MI355XFlatIndex
    X = _rows(900, 24, 3)
    Q = np.random.default_rng(4).standard_normal((7, 24)).astype(np.float32)
    for form in ("numpy", "faiss"):
        idx = MI355XFlatIndex(string_factory="L2norm,Flat", metric_type=1, l2norm_form=form)
        idx.add(X)
        D, I = idx.search_batch(Q, 50)
        Do, Io = ok.knn(X, Q, 50, metric=1, l2norm=True, l2norm_form=form)
        assert np.array_equal(I, Io) and np.array_equal(D, Do, equal_nan=True)
        t = torch.from_numpy(X.copy()).cuda()
        lib = _lib.load()
        _lib.check(lib.mq_l2norm_rows_form_f32(t.data_ptr(), t.shape[0], t.shape[1], L2NORM_FORMS[form],
                                               torch.cuda.current_stream().cuda_stream), "mq_l2norm_rows_form_f32")
        torch.cuda.synchronize()
        assert np.array_equal(t.cpu().numpy(), ok.l2norm_rows(X, form=form), equal_nan=True)
    assert lib.mq_l2norm_rows_form_f32(t.data_ptr(), 1, 1, 3, None) < 0  # unknown form
    # both kernels behind the entry (8 rows per workgroup up to 4096 rows -- a search's queries --, 64 beyond), ragged widths
    for n, d in ((4096, 2048), (4097, 2048), (5000, 100), (256, 1000), (3, 1)):
        Y = _rows(max(n, 12), d, n + d)[:n] if d >= 2 else np.random.default_rng(1).standard_normal((n, d)).astype(np.float32)
        for form in ("numpy", "faiss"):
            t = torch.from_numpy(Y.copy()).cuda()
            _lib.check(lib.mq_l2norm_rows_form_f32(t.data_ptr(), n, d, L2NORM_FORMS[form], torch.cuda.current_stream().cuda_stream), "l2norm")
            torch.cuda.synchronize()
            assert np.array_equal(t.cpu().numpy(), ok.l2norm_rows(Y, form=form), equal_nan=True), (n, d, form)


def test_knowledge_base_picks_the_form_from_the_device_key():
    """``device: null`` (every shipped config): FAISS's own transform.  A ``device``: the reference's numpy work-around.
    ``l2norm_form`` overrides; a directly built index defaults to "numpy" (MQ_KNN_L2NORM_FORM)."""
    import datasets
    from viquae_amd.index import MI355XFlatIndex
    from viquae_amd.ir.search import KnowledgeBase
    X = _rows(300, 16, 5)
    ds = lambda: datasets.Dataset.from_dict({"vec": [r for r in X]})  # noqa: E731
    base = {"column": "vec", "key": "q", "string_factory": "L2norm,Flat", "metric_type": 0}
    got = {}
    for name, extra in (("none", {"device": None}), ("zero", {"device": 0}), ("forced", {"device": 0, "l2norm_form": "faiss"})):
        kb = KnowledgeBase(dataset=ds(), index_kwargs={"idx": dict(base, **extra)})
        got[name] = kb.dataset._indexes["idx"].l2norm_form
        D, I = kb.search_batch("idx", X[20:25], k=3)
        assert (I[:, 0] == np.arange(20, 25)).all()
    assert got == {"none": "faiss", "zero": "numpy", "forced": "faiss"}
    # a LOADED "L2norm,Flat" file is FAISS's IndexPreTransform whatever device it is loaded onto (the reference's work-around
    # only strips the transform when it BUILDS, meerqat/ir/search.py:238-244): the device key does not choose the form then
    import tempfile, os
    path = os.path.join(tempfile.mkdtemp(), "idx.faiss")
    kb = KnowledgeBase(dataset=ds(), index_kwargs={"idx": dict(base, device=None, save_path=path)})
    for extra, want in (({"device": 0}, "faiss"), ({"device": None}, "faiss"), ({"device": 0, "l2norm_form": "numpy"}, "numpy")):
        kb2 = KnowledgeBase(dataset=ds(), index_kwargs={"idx": dict(base, load=True, file=path, **extra)})
        assert kb2.dataset._indexes["idx"].l2norm_form == want, extra
        D2, I2 = kb2.search_batch("idx", X[20:25], k=3)
        assert (I2[:, 0] == np.arange(20, 25)).all()
    assert MI355XFlatIndex(string_factory="L2norm,Flat").l2norm_form == "numpy"
    with pytest.raises(ValueError):
        MI355XFlatIndex(string_factory="L2norm,Flat", l2norm_form="blas")
