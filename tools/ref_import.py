"""Imports the REFERENCE (``/root/reference``, read-only) in this container so that its own plumbing
can be run to mint golden vectors (tools/make_golden.py).  Never used on the GPU box and never
imported by the product or the tests: only the arrays it produces travel (tests/golden/*.npz).

The reference imports several packages that are not installed here (docopt, ranx, spacy,
jsonargparse, numba ...).  None of them is on the hot path; they are replaced by inert stub modules
so that ``meerqat.ir.search`` / ``meerqat.ir.embedding`` import.  ``faiss`` (the third-party module
that owns the kNN arithmetic, requirements.txt:14) is absent too: a stand-in whose ``IndexFlat`` calls
the CPU oracle is registered so that the reference's KnowledgeBase -> datasets.FaissIndex call chain
runs end to end.  That pins the PLUMBING (argument handling, L2norm, None filtering, shapes, dtypes,
batching in add_vectors); it does not pin FAISS's own arithmetic -- see oracle/knn_oracle.c header.
"""
import sys
import types

import numpy as np

REFERENCE = "/root/reference"


class _Any:
    """Inert placeholder: callable, attribute-able, usable as a decorator or base class."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        return _Any()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Any


def _module(name, **attrs):
    m = _StubModule(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _FlatStandIn:
    """faiss.IndexFlat stand-in: storage + oracle search (metric 0 = IP, 1 = L2)."""

    def __init__(self, d, metric=1):
        self.d, self.metric_type, self.ntotal = d, metric, 0
        self._rows = []
        self.verbose = False

    def add(self, x):
        x = np.asarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[1] == self.d
        self._rows.append(x.copy())
        self.ntotal += x.shape[0]

    def train(self, x):
        pass

    def _matrix(self):
        return np.concatenate(self._rows) if self._rows else np.zeros((0, self.d), np.float32)

    def search(self, q, k):
        from oracle import knn as ok
        return ok.knn(self._matrix(), np.asarray(q, np.float32), k, metric=self.metric_type)


class _PreTransformStandIn:
    """faiss.IndexPreTransform(NormalizationTransform(d, 2.0), IndexFlat) stand-in ("L2norm,Flat").  It normalises with
    oracle.knn.l2norm_rows(form="faiss") -- the function the goldens are later compared with: files minted through this class are
    SELF-MINTED for the transform's arithmetic (they pin the reference's plumbing, not FAISS's last bit; tests/test_oracle_cpu.py)."""

    def __init__(self, d, metric):
        self.index = _FlatStandIn(d, metric)
        self.d, self.verbose = d, False

    @property
    def ntotal(self):
        return self.index.ntotal

    def train(self, x):
        pass

    def add(self, x):
        from oracle import knn as ok
        self.index.add(ok.l2norm_rows(np.asarray(x, np.float32), form="faiss"))  # NormalizationTransform = fvec_renorm_L2

    def search(self, q, k):
        from oracle import knn as ok
        return self.index.search(ok.l2norm_rows(np.asarray(q, np.float32), form="faiss"), k)


class _RunStandIn:
    """ranx.Run stand-in: just the two attributes meerqat/ir/fuse.py touches (``run``, ``name``)."""

    def __init__(self, run=None, name=None):
        self.run = {} if run is None else run
        self.name = name


class _TypedDictStandIn(dict):
    """numba.typed.Dict stand-in (``Dict.empty(key_type, value_type)`` -> plain dict)."""

    @classmethod
    def empty(cls, key_type=None, value_type=None):
        return cls()


def _index_factory(d, description, metric=1):
    parts = description.split(",")
    if parts == ["Flat"]:
        return _FlatStandIn(d, metric)
    if parts == ["L2norm", "Flat"]:
        return _PreTransformStandIn(d, metric)
    raise ValueError(f"stand-in index_factory: unsupported '{description}'")


def install_stubs():
    import datasets  # must be imported BEFORE the stubs (its pickler probes spacy)
    import transformers  # noqa: F401  (probes torchvision availability: import before stubbing it)
    import transformers.models.dpr, transformers.models.clip  # noqa: F401,E401
    import datasets.search as dsearch

    if "docopt" not in sys.modules:
        _module("docopt", docopt=lambda *a, **k: {})
    if "ranx" not in sys.modules:
        _module("ranx", Run=_RunStandIn, Qrels=_Any, compare=_Any(), fuse=_Any(), optimize_fusion=_Any())
    if "spacy" not in sys.modules:
        sp = _module("spacy", Language=type("Language", (), {}), load=_Any())
        _module("spacy.lang")
        _module("spacy.lang.en", English=_Any)
        sp.lang = sys.modules["spacy.lang"]
    if "jsonargparse" not in sys.modules:
        _module("jsonargparse", CLI=_Any())
    if "numba" not in sys.modules:
        ident = lambda *a, **k: (a[0] if a and callable(a[0]) and not k else (lambda f: f))  # noqa: E731
        nb = _module("numba", njit=ident, jit=ident, prange=range, types=_Any(), config=_Any())
        _module("numba.typed", List=list, Dict=_TypedDictStandIn)
        nb.typed = sys.modules["numba.typed"]
    if "torchvision" not in sys.modules:
        tv = _module("torchvision")
        _module("torchvision.transforms", Compose=_Any, Resize=_Any, CenterCrop=_Any, ToTensor=_Any, Normalize=_Any)
        _module("torchvision.models")
        tv.transforms, tv.models = sys.modules["torchvision.transforms"], sys.modules["torchvision.models"]
    if not hasattr(datasets, "set_caching_enabled"):
        datasets.set_caching_enabled = lambda b: None
    if "faiss" not in sys.modules:
        _module("faiss", IndexFlat=_FlatStandIn, index_factory=_index_factory, METRIC_INNER_PRODUCT=0, METRIC_L2=1)
        dsearch._has_faiss = True
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)


def import_reference_search():
    install_stubs()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import meerqat.ir.search as ref_search
    return ref_search


def import_reference_fuse():
    install_stubs()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import meerqat.ir.fuse as ref_fuse
    return ref_fuse


def import_reference_embedding():
    install_stubs()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import meerqat.ir.embedding as ref_embedding
    return ref_embedding
