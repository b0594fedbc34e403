"""CPU: the Searcher mirror against a golden minted by the REFERENCE's Searcher.__call__
(tools/make_golden_searcher.py): run dicts (ids, scores, order, cut at k), article->passage fan-out with the
per-passage penalty, many2one='max', relevance judgement.  The dense index is served by the oracle here; the GPU
variant (tests/test_searcher_gpu.py) uses the HIP index."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "searcher.json")


def build(golden, case, index_factory):
    import datasets
    from viquae_amd.ir.search import Index, KnowledgeBase, register_index
    from viquae_amd.ir.searcher import Searcher
    art = np.asarray(golden["art_vec"], dtype=np.float32)
    kb = KnowledgeBase(dataset=datasets.Dataset.from_dict({"vec": [v for v in art]}))
    kb.index_mapping = {int(a): v for a, v in golden["mapping"].items()}
    kb.many2one = golden["cases"][case]["many2one"]
    register_index(kb.dataset, "dense", index_factory(art))
    kb.indexes["dense"] = Index(key="vec_q")
    ref_kb = datasets.Dataset.from_dict({"passage": golden["passages"]})
    s = Searcher(kb_kwargs={"articles": {}}, k=golden["cases"][case]["k"], kbs={"articles": kb}, reference_kb=ref_kb)
    b = golden["batch"]
    s({"id": b["id"], "vec_q": [np.asarray(q, np.float32) for q in b["vec_q"]], "output": b["output"]})
    return s


def check(s, want):
    assert s.runs.keys() == want["runs"].keys()
    for q, run in want["runs"]["dense"].items():
        got = s.runs["dense"][q]
        assert list(got.keys()) == list(run.keys()), q  # same documents in the same insertion order
        assert np.allclose([got[d] for d in run], [run[d] for d in run], rtol=0, atol=1e-6), q
    assert {q: sorted(v) for q, v in s.qrels.items()} == {q: sorted(v) for q, v in want["qrels"].items()}
    assert {q: sorted(v) for q, v in s.qnonrels.items()} == {q: sorted(v) for q, v in want["qnonrels"].items()}


def oracle_index(art):
    from datasets.search import BaseIndex, BatchedSearchResults
    from oracle import knn as ok

    class OracleIndex(BaseIndex):
        def search_batch(self, queries, k=10, **kw):
            D, I = ok.knn(art, queries, k, metric=0)
            return BatchedSearchResults(D, I.astype(int))
    return OracleIndex()


@pytest.mark.parametrize("case", ["one2many", "many2one_max", "cut"])
def test_searcher_matches_reference_golden(case):
    golden = json.load(open(GOLDEN))
    check(build(golden, case, oracle_index), golden["cases"][case])


def test_find_relevant_whole_word_matching():
    from viquae_amd.ir.searcher import find_relevant
    kb = [{"passage": "The Eiffel Tower is in Paris, France."}, {"passage": "Parisian cafes"}, {"passage": "A tower."}]
    orig, rel = find_relevant([0, 1, 2], "Paris", ["the Tower"], kb)
    assert orig == [0] and rel == [0, 2]
    with pytest.raises(NotImplementedError):
        find_relevant([0], "1", ["1"], kb, question_type="Numerical")


def test_cut_at_k_with_an_article_that_maps_to_no_passage_ranked_first():
    """the cut-at-k test never fired when the first hit had an empty index_mapping entry, so the run grew past
    k.  Expected = the reference's loop (meerqat/ir/search.py:413-440): insert every passage of a hit, then test the cut."""
    from viquae_amd.ir.search import KnowledgeBase
    from viquae_amd.ir.searcher import Searcher
    import datasets
    kb = KnowledgeBase(dataset=datasets.Dataset.from_dict({"vec": [[0.0]] * 5}))
    kb.index_mapping = {0: [], 1: [10, 11], 2: [], 3: [12, 13, 14], 4: [15]}
    s = Searcher(kb_kwargs={"a": {}}, k=3, kbs={"a": kb}, reference_kb=datasets.Dataset.from_dict({"passage": ["x"]}))
    for many2one in (None, "max"):
        kb.many2one = many2one
        for indices in ([0, 1, 2, 3, 4], [0, 2, 1, 3, 4], [1, 0, 3, 2, 4], [0, 2], [3, 0, 1]):
            scores = [5.0 - 0.5 * r for r in range(len(indices))]
            want = {}
            for score, i in zip(scores, indices):           # the reference's loop
                for n, j in enumerate(kb.index_mapping[i]):
                    if many2one is None:
                        want[str(j)] = float(np.float32(score) - n * 1e-8)   # float32 arithmetic, like the reference under NumPy 2
                    elif str(j) not in want or want[str(j)] < score:
                        want[str(j)] = score
                if len(want) >= s.k:
                    break
            got = {}
            s._fill_run(kb, got, scores, indices)
            assert list(got) == list(want) and np.allclose(list(got.values()), list(want.values()), rtol=0, atol=1e-9), (many2one, indices)
