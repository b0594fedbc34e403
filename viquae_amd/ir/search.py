"""Host-side mirror of ``meerqat.ir.search`` for the dense (FAISS-kind) indexes.

Same names, arguments and error behaviour as the reference for the hot-path surface
(SURVEY.md section 8 a1-a4, b2):

* ``L2norm``                                   meerqat/ir/search.py:43-46
* ``IndexKind`` / ``Index``                    meerqat/ir/search.py:49-75
* ``KnowledgeBase``                            meerqat/ir/search.py:81-249
    ``search_batch``                             :135-146
    ``search_batch_if_not_None``                 :148-171
    ``add_or_load_index``                        :173-205
    ``add_or_load_faiss_index``                  :207-249

What differs underneath: ``add_or_load_faiss_index`` builds a :class:`viquae_amd.index.MI355XFlatIndex`
(HBM-resident, hand-written HIP scan) instead of a FAISS index and registers it in
``dataset._indexes`` so that ``Dataset.search_batch`` / ``get_nearest_examples_batch`` keep working.
Sparse kinds (ES / PYSERINI) are outside this build's scope and raise.
"""
import enum
import json
import warnings

import numpy as np

from ..index import MI355XFlatIndex

# keys that every shipped experiments/ir/**/config.json still carries but that the current
# reference code path no longer consumes: accepted and ignored
LEGACY_INDEX_KEYS = ("es", "kind_str", "normalization", "interpolation_weight")


def L2norm(queries):
    """Unit-normalise each row of a batch of same-dimension vectors (no epsilon: a zero vector
    yields NaN, as in the reference)."""
    queries = np.asarray(queries)
    return queries / np.linalg.norm(queries, axis=1, keepdims=True)


class IndexKind(enum.Enum):
    FAISS = 0
    ES = 1
    PYSERINI = 2


class Index:
    """Book-keeping about one index of a KB: which dataset column holds the queries (``key``),
    the kind of index, and whether queries are L2-normalised before searching."""

    def __init__(self, key, kind=IndexKind.FAISS, do_L2norm=False):
        self.key = key
        self.kind = kind
        self.do_L2norm = do_L2norm


def _int_keys(d):
    out = {}
    for k, v in d.items():
        try:
            out[int(k)] = v
        except (TypeError, ValueError):
            out[k] = v
    return out


class KnowledgeBase:
    """A KB (a ``datasets.Dataset``) searchable through several named indexes.

    Parameters mirror the reference: ``kb_path``, ``index_mapping_path``, ``many2one``,
    ``index_kwargs`` (name -> kwargs of ``add_or_load_index``), ``es_client``, ``load_dataset``.
    ``dataset`` (extra) lets a caller hand over an in-memory Dataset instead of a path.
    """

    def __init__(self, kb_path=None, index_mapping_path=None, many2one=None, index_kwargs={},
                 es_client=None, load_dataset=True, dataset=None):
        if dataset is not None:
            self.dataset = dataset
        elif load_dataset:
            from datasets import load_from_disk
            self.dataset = load_from_disk(kb_path)
        else:
            self.dataset = None
        self.es_client = es_client
        self.indexes = {}
        if index_mapping_path is None:
            self.index_mapping = None
        else:
            with open(index_mapping_path, "rt") as file:
                self.index_mapping = json.load(file, object_hook=_int_keys)
        self.many2one = many2one
        for index_name, index_kwarg in index_kwargs.items():
            self.add_or_load_index(index_name=index_name, **index_kwarg)

    # ------------------------------------------------------------------ search
    def search_batch(self, index_name, queries, k=100):
        """Pre-process the queries as the index requires, then ``self.dataset.search_batch``.
        Returns (scores [nq,k] float32, indices [nq,k] int), rows best-first."""
        index = self.indexes[index_name]
        if index.kind != IndexKind.FAISS:
            raise NotImplementedError(f"{index.kind} indexes are outside the MI355X build (dense FAISS-kind only)")
        queries = np.array(queries, dtype=np.float32)
        if index.do_L2norm:
            queries = L2norm(queries)
        return self.dataset.search_batch(index_name, queries, k=k)

    def search_batch_if_not_None(self, index_name, queries, k=100):
        """Searches only the queries that are not None; a None query gets empty results ([])."""
        kept = [i for i, q in enumerate(queries) if q is not None]
        scores_batch = [[] for _ in queries]
        indices_batch = [[] for _ in queries]
        if not kept:
            return scores_batch, indices_batch
        found_scores, found_indices = self.search_batch(index_name, [queries[i] for i in kept], k=k)
        for row, i in enumerate(kept):
            scores_batch[i] = found_scores[row]
            indices_batch[i] = found_indices[row]
        return scores_batch, indices_batch

    # ------------------------------------------------------------------ construction
    def add_or_load_index(self, column=None, index_name=None, kind=None, key=None, **index_kwarg):
        """Dispatch on ``kind`` (None -> FAISS). ``index_name`` defaults to ``column``."""
        kind = IndexKind.FAISS if kind is None else IndexKind[kind]
        if index_name is None:
            index_name = column
        if kind != IndexKind.FAISS:
            raise NotImplementedError(f"{kind} indexes (sparse retrieval) are outside the MI355X build")
        if index_kwarg.get("es"):
            # the shipped BM25 configs (experiments/ir/viquae/bm25/config.json, bm25+arcface+clip+imagenet/config_*.json) still
            # say `"es": true`, the spelling `kind: "ES"` replaced: an Elasticsearch index over a TEXT column, not a dense one
            raise NotImplementedError(f"index '{index_name}' is an Elasticsearch (BM25) index ('es': true): sparse retrieval is "
                                      "outside the MI355X build (dense FAISS-kind only)")
        for legacy in LEGACY_INDEX_KEYS:
            index_kwarg.pop(legacy, None)
        do_L2norm = self.add_or_load_faiss_index(column, index_name=index_name, **index_kwarg)
        self.indexes[index_name] = Index(key=key, kind=kind, do_L2norm=do_L2norm)

    def add_or_load_faiss_index(self, column, index_name=None, load=False, save_path=None, string_factory=None,
                                device=None, metric_type=None, batch_size=1000, train_size=None, file=None,
                                faiss_verbose=None, tie_order=None, l2norm_form=None, **kwargs):
        """Builds (or loads from ``file``) the exact index over ``column`` and registers it as
        ``index_name``.  Returns ``do_L2norm`` (inferred from 'L2norm' in ``string_factory``).

        The "L2norm," transform is applied on device while packing (csrc/knn.hip), in the arithmetic the reference would have
        used for the same ``device`` key: with ``device: null`` (every shipped config) FAISS's own NormalizationTransform
        normalises the KB rows on add and the queries on search -- ``x * float(1.0 / sqrt(sum x^2))``, a zero row stays a
        zero row (``l2norm_form="faiss"``); with a ``device``, the reference's GPU work-around (meerqat/ir/search.py:238-244)
        normalises the column with numpy's ``L2norm`` instead -- ``x / sqrt(sum x^2)``, a zero row becomes NaN
        (``l2norm_form="numpy"``).  ``l2norm_form`` (extra key) overrides the choice.  ``tie_order`` (extra key, "id_asc" |
        "id_desc"): which of several EXACTLY tied rows ranks first -- this library's policy knob, see
        viquae_amd.index.MI355XFlatIndex and INTEGRATION.md section D."""
        if kwargs:
            warnings.warn(f"add_or_load_faiss_index: ignoring unknown arguments {sorted(kwargs)}")
        if index_name is None:
            index_name = column
        do_L2norm = string_factory is not None and "L2norm" in string_factory
        # which index class serves `device` (one GPU, all GPUs of this process, or this rank's shard of a
        # torch.distributed job): viquae_amd.sharded.make_flat_index
        from ..sharded import ShardedFlatIndex, make_flat_index
        if l2norm_form is None and do_L2norm and not load:
            # only a BUILD follows the `device` key (the reference strips the transform when it builds for a GPU, :238-244); a
            # stored "L2norm,Flat" file is FAISS's IndexPreTransform whatever it is loaded onto: None lets the loaders choose
            # FAISS's arithmetic for such a file
            l2norm_form = "faiss" if device is None else "numpy"
        if load:
            if file is None:
                raise ValueError("load=True needs `file` (the path passed to save_faiss_index / save_path)")
            index = make_flat_index(device=device, string_factory=string_factory, metric_type=metric_type, tie_order=tie_order,
                                    l2norm_form=l2norm_form)
            if isinstance(index, ShardedFlatIndex):
                index.load_rows(file)  # every rank reads only its own row range
            elif isinstance(index, MI355XFlatIndex):
                index = MI355XFlatIndex.load(file, device=device, tie_order=tie_order, l2norm_form=l2norm_form)
            else:
                index.load_rows(file)  # LocalShardsFlatIndex: metric and "L2norm," from the file, set on EVERY shard
        else:
            index = make_flat_index(device=device, string_factory=string_factory, metric_type=metric_type, tie_order=tie_order,
                                    l2norm_form=l2norm_form)
            index.add_vectors(self.dataset, column=column, batch_size=batch_size, train_size=train_size,
                              faiss_verbose=faiss_verbose)
            if save_path is not None:
                index.save(save_path)
        register_index(self.dataset, index_name, index)
        return do_L2norm


def register_index(dataset, index_name, index):
    """What ``Dataset.add_faiss_index`` does last (datasets/search.py:493): attach the index object
    under ``index_name`` so the IndexableMixin API (search_batch, get_nearest_examples_batch,
    save/drop_index ...) finds it."""
    if not hasattr(dataset, "_indexes"):
        raise TypeError("expected a datasets.Dataset (IndexableMixin)")
    dataset._indexes[index_name] = index
    return index
