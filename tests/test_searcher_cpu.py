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


def oracle_index(art, metric=0):
    from datasets.search import BaseIndex, BatchedSearchResults
    from oracle import knn as ok

    class OracleIndex(BaseIndex):
        metric_type = metric

        def search_batch(self, queries, k=10, **kw):
            D, I = ok.knn(art, queries, k, metric=metric)
            return BatchedSearchResults(D, I.astype(int))
    return OracleIndex()


@pytest.mark.parametrize("case", ["one2many", "many2one_max", "cut"])
def test_searcher_matches_reference_golden(case):
    golden = json.load(open(GOLDEN))
    check(build(golden, case, oracle_index), golden["cases"][case])


def _reference_answer_preprocess(answer):
    """meerqat/data/loading.py:152-164, as written there (character loop, re.sub, split / join)."""
    import re
    import string
    exclude = set(string.punctuation)
    text = "".join(ch for ch in answer.lower() if ch not in exclude)
    text = re.sub(r"\b(a|an|the)\b", " ", text)
    return " ".join(text.split())


def test_answer_preprocess_and_passage_cache_equal_the_reference_statement():
    """The relevance judgement's string work (f1, meerqat/ir/search.py:442-457 -> metrics.py:79-124): the translate-based
    preprocessing gives the reference's strings; judging through the passage cache (one Arrow `take` per request, every passage
    preprocessed once) gives the reference loop's lists."""
    import re
    import string
    import datasets
    from viquae_amd.ir.searcher import PassageTexts, answer_preprocess, find_relevant
    rng = np.random.default_rng(0)
    alphabet = list(string.ascii_letters + string.digits + string.punctuation + "    \t\n" + "éÉßøŒ«»’“”–—…¿¡ñçüİı́") + [" the ", " a ", " an ", "The", "AN", " a", "the"]
    texts = ["".join(rng.choice(alphabet, rng.integers(0, 120))) for _ in range(3000)] + ["", "the", "a an the", "An  apple, a day!", "THE-END"]
    for t in texts:
        assert answer_preprocess(t) == _reference_answer_preprocess(t), repr(t)
    words = ["paris", "tower", "eiffel", "lyon", "seine", "river", "the", "of", "1889", "gustave"]
    passages = [" ".join(rng.choice(words, rng.integers(3, 30))).capitalize() + rng.choice([".", "!", "", " (x)"]) for _ in range(500)]
    kb = datasets.Dataset.from_dict({"passage": passages})

    def reference_loop(retrieved, original, aliases):
        orig, rel = [], []
        for i in retrieved:
            passage = _reference_answer_preprocess(kb[int(i)]["passage"])
            if re.search(rf"\b{_reference_answer_preprocess(original)}\b", passage) is not None:
                orig.append(int(i))
                rel.append(int(i))
                continue
            for a in aliases:
                if re.search(rf"\b{_reference_answer_preprocess(a)}\b", passage) is not None:
                    rel.append(int(i))
                    break
        return orig, rel

    cache = PassageTexts(kb, "passage")
    assert cache._column is not None
    for trial in range(40):
        retrieved = [str(i) for i in rng.choice(500, 60, replace=False)]
        original = str(rng.choice(["Paris", "The Eiffel Tower", "Gustave", "1889!", "nowhere"]))
        aliases = [str(a) for a in rng.choice(["Lyon", "the Seine", "river of", "", "tower."], rng.integers(0, 4), replace=False)]
        want = reference_loop(retrieved, original, aliases)
        assert find_relevant(retrieved, original, aliases, kb) == want
        assert find_relevant(retrieved, original, aliases, kb, passages=cache) == want
    assert 0 < len(cache.texts) <= 500
    assert find_relevant([], "x", [], kb, passages=cache) == ([], [])
    small = PassageTexts(kb, "passage", capacity=50)             # the cache empties itself instead of growing without bound
    small.get_many(list(range(40)))
    small.get_many(list(range(40, 80)))
    assert len(small.texts) == 40 and small.get_many([41, 3]) == [answer_preprocess(passages[41]), answer_preprocess(passages[3])]
    shuffled = kb.shuffle(seed=0)                                # an indices mapping: rows through the dataset, not the raw column
    assert PassageTexts(shuffled, "passage")._column is None
    assert PassageTexts(shuffled, "passage").get_many([0, 1]) == [answer_preprocess(shuffled[0]["passage"]), answer_preprocess(shuffled[1]["passage"])]


@pytest.mark.parametrize("mapped", [False, True])
def test_judged_job_keeps_its_runs_as_arrays(mapped):
    """Round 5: with a reference KB (every shipped search config) the relevance judgement needs the SET of documents a question
    retrieved, not its dict -- the runs stay rows of the result arrays, and runs / qrels / qnonrels equal the reference's loop."""
    import re
    import datasets
    from datasets.search import BaseIndex, BatchedSearchResults
    from viquae_amd.ir.runs import ArrayRun
    from viquae_amd.ir.search import Index, KnowledgeBase, register_index
    from viquae_amd.ir.searcher import Searcher
    rng = np.random.default_rng(5)
    words = ["paris", "tower", "eiffel", "lyon", "seine", "river", "the", "of", "1889", "gustave"]
    n_pass = 600
    passages = [" ".join(rng.choice(words, rng.integers(3, 25))).capitalize() + "." for _ in range(n_pass)]
    ref = datasets.Dataset.from_dict({"passage": passages})
    n_art = 150
    mapping = None
    if mapped:
        perm = rng.permutation(n_pass)
        mapping = {a: [int(x) for x in perm[4 * a:4 * a + int(rng.integers(0, 5))]] for a in range(n_art)}

    class Canned(BaseIndex):
        calls = []

        def search_batch(self, queries, k=10, **kw):
            n = len(queries)
            I = np.argsort(rng.random((n, n_art if mapped else n_pass)), axis=1)[:, :k]
            D = -np.sort(-rng.standard_normal((n, k)).astype(np.float32), axis=1)
            self.calls.append((D, I))
            return BatchedSearchResults(D, I)

    kb = KnowledgeBase(dataset=datasets.Dataset.from_dict({"vec": [[0.0]]}))
    idx = Canned()
    idx.calls = []
    register_index(kb.dataset, "dense", idx)
    kb.indexes["dense"] = Index(key="q")
    if mapped:
        kb.index_mapping, kb.many2one = mapping, None
    s = Searcher(kb_kwargs={"kb": {}}, k=12, kbs={"kb": kb}, reference_kb=ref)
    answers = [{"original_answer": "Paris", "answer": ["Paris", "the Seine"]}, {"original_answer": "Gustave Eiffel", "answer": ["Eiffel"]},
               {"original_answer": "nowhere", "answer": []}]
    batches = [[f"q{b}_{i}" for i in range(n)] for b, n in enumerate((4, 7))]
    outs = []
    for ids in batches:
        out = [answers[int(rng.integers(0, 3))] for _ in ids]
        outs.append(out)
        s({"id": ids, "q": [np.zeros(1, np.float32)] * len(ids), "output": out})
    got = s.runs["dense"]
    assert isinstance(got, ArrayRun) and got.lazy_questions() == 11
    want = _reference_loop(12, mapping, None, [(ids, D, I) for ids, (D, I) in zip(batches, idx.calls)])
    assert list(got) == list(want) and all(list(got[q].items()) == list(want[q].items()) for q in want)
    for ids, out in zip(batches, outs):
        for q, gt in zip(ids, out):
            rel = set()
            for doc in want[q]:
                p = _reference_answer_preprocess(passages[int(doc)])
                if any(re.search(rf"\b{_reference_answer_preprocess(a)}\b", p) for a in [gt["original_answer"]] + gt["answer"]):
                    rel.add(doc)
            assert set(s.qrels[q]) == rel and set(s.qnonrels[q]) == set(want[q]) - rel, q


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


def test_arrow_query_transport_changes_the_transport_not_the_runs(tmp_path):
    """dataset_search reads the query vectors of a batch straight from the Arrow table (~0.05 ms per 256-query batch instead
    of ~55 ms of Python-list decoding under the shipped "format": {}); columns with None queries, datasets with an indices
    mapping and user-chosen formats take the ordinary path; the runs are the same either way."""
    import datasets
    from viquae_amd.ir import searcher as S
    from viquae_amd.ir.search import Index, KnowledgeBase, register_index
    datasets.disable_progress_bars()
    rng = np.random.default_rng(0)
    art = rng.standard_normal((300, 24)).astype(np.float32)
    Q = rng.standard_normal((37, 24)).astype(np.float32)
    faces = [None if i % 3 == 0 else list(map(float, Q[i])) for i in range(37)]
    qs = datasets.Dataset.from_dict({"id": [str(i) for i in range(37)], "vec_q": [q for q in Q], "face_q": faces,
                                     "vec2_q": [q for q in Q[::-1]], "output": [{"answer": ["a"], "original_answer": "a"}] * 37})
    (tmp_path / "qrels.json").write_text("{}")

    def kb_kwargs():
        kb = KnowledgeBase(dataset=datasets.Dataset.from_dict({"vec": [v for v in art]}))
        register_index(kb.dataset, "dense", oracle_index(art))
        register_index(kb.dataset, "face", oracle_index(art))
        register_index(kb.dataset, "dense2", oracle_index(art))
        kb.indexes["dense"], kb.indexes["face"], kb.indexes["dense2"] = Index(key="vec_q"), Index(key="face_q"), Index(key="vec2_q")
        return dict(kb_kwargs={"kb": {}}, k=10, kbs={"kb": kb}, qrels=str(tmp_path / "qrels.json"), do_fusion=False)

    with pytest.warns(UserWarning):
        s_py = S.Searcher(**kb_kwargs())
    qs.map(s_py, batched=True, batch_size=16, load_from_cache_file=False)           # the reference's own way
    fast = S.ArrowQueryColumns(qs, s_py)
    assert list(fast.columns) == ["vec_q", "vec2_q"]                                # face_q holds None: ordinary path
    assert np.array_equal(fast.batch("vec_q", list(range(5, 21))), Q[5:21])
    assert np.array_equal(fast.batch("vec_q", [3, 9, 4]), Q[[3, 9, 4]])
    with pytest.warns(UserWarning):
        s_fast = S.dataset_search(qs, report=False, map_kwargs={"batch_size": 16, "load_from_cache_file": False}, **kb_kwargs())
    assert s_fast.runs == s_py.runs and len(s_fast.runs["dense"]["5"]) == 10 and s_fast.runs["face"]["0"] == {}
    assert not S.ArrowQueryColumns(qs.with_format("numpy"), s_py)                   # the user's format stands
    assert not S.ArrowQueryColumns(qs.select([3, 1, 2]), s_py)                      # indices mapping: ordinary path


def _reference_loop(k, mapping, many2one, batches):
    """meerqat/ir/search.py:413-440 restated: runs[q][str(doc)] = score, fan-out, penalty, cut at k after each hit."""
    run = {}
    for q_ids, scores_batch, indices_batch in batches:
        for q_id, scores, indices in zip(q_ids, scores_batch, indices_batch):
            r = run.setdefault(q_id, {})
            for score, i in zip(scores, indices):
                penalty = np.float32(0.0)
                if mapping is not None:
                    for n, j in enumerate(mapping[int(i)]):
                        j = str(j)
                        if many2one is None:
                            r[j] = float(np.float32(score) - np.float32(1e-8 * n))
                        elif j not in r or r[j] < score:
                            r[j] = float(score)
                else:
                    r[str(int(i))] = float(score)
                if len(r) >= k:
                    break
    return run


@pytest.mark.parametrize("mapping,many2one", [(None, None), ("csr", None), ("csr", "max")])
def test_runs_kept_as_arrays_equal_the_reference_loop(mapping, many2one):
    """Round 3: without on-the-fly relevance the searcher keeps each batch's [nq, k] arrays and builds
    the dicts once, when `runs` is read.  Same dicts, same insertion order, as the reference's per-hit loop -- with repeated
    question ids across batches (inserted into the existing run), duplicate ids inside a batch, -1 fillers (k > ntotal),
    a cut k smaller than what the index returns, and index_mapping fan-out in both many2one modes."""
    import datasets
    from datasets.search import BaseIndex, BatchedSearchResults
    from viquae_amd.ir.search import Index, KnowledgeBase, register_index
    from viquae_amd.ir.searcher import Searcher
    rng = np.random.default_rng(1)
    n_art = 40

    class Canned(BaseIndex):
        def __init__(self):
            self.calls = []

        def search_batch(self, queries, k=10, **kw):
            nq = len(queries)
            I = np.stack([rng.permutation(n_art)[:k] for _ in range(nq)]).astype(int)
            D = np.sort(rng.standard_normal((nq, k)).astype(np.float32), axis=1)[:, ::-1].copy()
            if len(self.calls) == 1:          # second batch: the index ran out of rows for some queries
                I[0, 5:] = -1
                D[0, 5:] = -np.finfo(np.float32).max
            self.calls.append((D, I))
            return BatchedSearchResults(D, I)

    kb = KnowledgeBase(dataset=datasets.Dataset.from_dict({"vec": [[0.0]] * n_art}))
    idx = Canned()
    register_index(kb.dataset, "dense", idx)
    kb.indexes["dense"] = Index(key="q")
    if mapping is not None:
        kb.index_mapping = {a: [int(x) for x in rng.integers(0, 200, rng.integers(0, 4))] for a in range(n_art)}
        kb.index_mapping[-1] = []
        kb.many2one = many2one
    import tempfile
    qrels = os.path.join(tempfile.mkdtemp(), "qrels.json")
    open(qrels, "wt").write("{}")
    with pytest.warns(UserWarning):
        s = Searcher(kb_kwargs={"kb": {}}, k=8, kbs={"kb": kb}, qrels=qrels)
    batches_ids = [["a", "b", "c"], ["d", "e"], ["a", "f"], ["g", "g", "h"]]   # "a" comes back; "g" twice in one batch
    for ids in batches_ids:
        s({"id": ids, "q": [np.zeros(1, np.float32)] * len(ids)})
    assert s._pending and not any(s._runs["dense"].values())                     # nothing was turned into dicts yet
    want = _reference_loop(8, kb.index_mapping, many2one, [(ids, D, I) for ids, (D, I) in zip(batches_ids, idx.calls)])
    got = s.runs["dense"]
    assert not s._pending
    assert list(got) == list(want)
    for q in want:
        assert list(got[q]) == list(want[q]), q
        assert np.allclose(list(got[q].values()), list(want[q].values()), rtol=0, atol=1e-9), q


@pytest.mark.parametrize("many2one", [None, "max"])
def test_block_fan_out_through_a_disjoint_mapping_equals_the_reference_loop(many2one, tmp_path):
    """Round 5: the article -> passage fan-out of the fusion configs' image indexes (index_mapping_path, many2one) for a WHOLE
    block of hits at once when no run entry can be written twice (a disjoint mapping, distinct hits per query): same dicts, same
    order, same float32 penalties, same cut behind the hit that fills the run -- incl. articles without passages, a cut that
    falls on such an article, runs that never fill, and k larger than what one article brings.  A repeated hit or a passage shared
    by two articles takes the per-query path; so does a question id seen twice.  (And the CSR form of the mapping is built once
    per KB: round 4 rebuilt it for every query -- 0.2 s per question at 200 k articles.)"""
    import datasets
    from datasets.search import BaseIndex, BatchedSearchResults
    from viquae_amd.ir import searcher as S
    from viquae_amd.ir.runs import ArrayRun
    from viquae_amd.ir.search import Index, KnowledgeBase, register_index
    rng = np.random.default_rng(7)
    n_art = 400
    lens = rng.integers(0, 5, n_art)                      # 0 ... 4 passages per article
    lens[:3] = 0
    off = np.concatenate([[0], np.cumsum(lens)])
    passages = rng.permutation(off[-1] + 50)[:off[-1]]    # disjoint, in no particular order
    mapping = {int(a): [int(x) for x in passages[off[a]:off[a + 1]]] for a in range(n_art)}

    class Canned(BaseIndex):
        calls = []

        def search_batch(self, queries, k=10, **kw):
            n = len(queries)
            I = np.argsort(rng.random((n, n_art)), axis=1)[:, :k]
            rest = [a for a in I[0].tolist() if a > 2]     # a query that starts with the three articles without passages
            I[0] = np.array(([0, 1, 2] + rest + [3 + j for j in range(k)])[:k])   # (distinct; padding only when k > len(rest) + 3)
            if len(set(I[0].tolist())) < k:
                I[0] = np.arange(k)
            D = -np.sort(-rng.standard_normal((n, k)).astype(np.float32), axis=1)
            self.calls.append((D, I))
            return BatchedSearchResults(D, I)

    def searcher(k):
        kb = KnowledgeBase(dataset=datasets.Dataset.from_dict({"vec": [[0.0]] * n_art}))
        idx = Canned()
        idx.calls = []
        register_index(kb.dataset, "img", idx)
        kb.indexes["img"] = Index(key="q")
        kb.index_mapping, kb.many2one = mapping, many2one
        (tmp_path / "qrels.json").write_text("{}")
        with pytest.warns(UserWarning):
            return S.Searcher(kb_kwargs={"kb": {}}, k=k, kbs={"kb": kb}, qrels=str(tmp_path / "qrels.json")), idx, kb

    built = []
    plain_init = S._Mapping.__init__
    S._Mapping.__init__ = lambda self, m: (built.append(1), plain_init(self, m))[1]
    try:
        for k in (1, 7, 40, 300):                         # 300: no run ever fills (40 hits x <= 4 passages)
            s, idx, kb = searcher(k)
            batches = [[f"q{b}_{i}" for i in range(n)] for b, n in enumerate((5, 3, 9))]
            for ids in batches:
                s({"id": ids, "q": [np.zeros(1, np.float32)] * len(ids)})
            want = _reference_loop(k, mapping, many2one, [(ids, D, I) for ids, (D, I) in zip(batches, idx.calls)])
            got = s.runs["img"]
            assert isinstance(got, ArrayRun) and got.lazy_questions() == 17          # every block went through as a block
            assert list(got) == list(want)
            for q in want:
                assert list(got[q]) == list(want[q]), (k, q)
                assert list(got[q].values()) == list(want[q].values()), (k, q)        # float32 penalties, bit for bit
        assert len(built) == 4                                                       # one CSR per searcher's KB, not one per query
        # a repeated article among a query's hits: the dict semantics matter again -> the per-query path, same answer
        s, idx, kb = searcher(7)
        plain = idx.search_batch

        def with_repeat(queries, k=10, **kw):
            r = plain(queries, k=k)
            r.total_indices[1, 4] = r.total_indices[1, 2]
            return r
        idx.search_batch = with_repeat
        s({"id": ["a", "b", "c"], "q": [np.zeros(1, np.float32)] * 3})
        want = _reference_loop(7, mapping, many2one, [(["a", "b", "c"], *idx.calls[-1])])
        got = s.runs["img"]
        assert got.lazy_questions() == 0 and got == want and [list(got[q]) for q in want] == [list(want[q]) for q in want]
        # a passage shared by two articles: never the block path
        shared = dict(mapping)
        shared[399] = list(mapping[399]) + [int(passages[0])]
        assert not S._Mapping(shared).disjoint and S._Mapping(mapping).disjoint
    finally:
        S._Mapping.__init__ = plain_init


@pytest.mark.parametrize("window", [0, 16, 20, 64, 4096])
def test_search_ahead_windows_serve_every_batch_the_arrays_of_its_own_search(tmp_path, monkeypatch, window):
    """ArrowQueryColumns.search: one search per WINDOW of consecutive rows instead of one per `Dataset.map` batch
    (MQ_SEARCH_WINDOW; a query's exact top-k does not depend on its batch) -- same runs, fewer trips to the index."""
    import datasets
    from viquae_amd.ir import searcher as S
    from viquae_amd.ir.search import Index, KnowledgeBase, register_index
    datasets.disable_progress_bars()
    rng = np.random.default_rng(1)
    art = rng.standard_normal((200, 16)).astype(np.float32)
    Q = rng.standard_normal((100, 16)).astype(np.float32)
    qs = datasets.Dataset.from_dict({"id": [str(i) for i in range(100)], "vec_q": [q for q in Q],
                                     "output": [{"answer": ["a"], "original_answer": "a"}] * 100})
    (tmp_path / "qrels.json").write_text("{}")
    calls = []

    def kb_kwargs():
        kb = KnowledgeBase(dataset=datasets.Dataset.from_dict({"vec": [v for v in art]}))
        register_index(kb.dataset, "dense", oracle_index(art))
        kb.indexes["dense"] = Index(key="vec_q")
        plain = kb.search_batch

        def counted(index_name, queries, k=100):
            calls.append(len(queries))
            return plain(index_name, queries, k=k)
        kb.search_batch = counted
        return dict(kb_kwargs={"kb": {}}, k=7, kbs={"kb": kb}, qrels=str(tmp_path / "qrels.json"), do_fusion=False)

    monkeypatch.setenv("MQ_SEARCH_WINDOW", "0")
    with pytest.warns(UserWarning):
        want = S.dataset_search(qs, report=False, map_kwargs={"batch_size": 16, "load_from_cache_file": False}, **kb_kwargs()).runs
    assert calls == [16] * 6 + [4]
    del calls[:]
    monkeypatch.setenv("MQ_SEARCH_WINDOW", str(window))
    with pytest.warns(UserWarning):
        got = S.dataset_search(qs, report=False, map_kwargs={"batch_size": 16, "load_from_cache_file": False}, **kb_kwargs()).runs
    assert got == want and list(got["dense"]) == [str(i) for i in range(100)]
    # windows are whole batches (20 rows under 16-row batches -> 16 = the batch itself: no window)
    expect = {0: [16] * 6 + [4], 16: [16] * 6 + [4], 20: [16] * 6 + [4], 64: [64, 36], 4096: [100]}[window]
    assert calls == expect


@pytest.mark.parametrize("batch_size,expect", [(24, [110, 14]), (16, [16] * 6 + [14]), (32, [110, 14])])
def test_search_ahead_keeps_the_l2_form_of_every_batch(tmp_path, monkeypatch, batch_size, expect):
    """under the L2 metric the arithmetic of a score depends on the SIZE of the call (fewer than 20 queries: FAISS's
    direct sums; 20 or more: the BLAS form), so a batch of fewer than 20 rows must not be served from a window's arrays, and a
    window is only searched ahead when the batch and the window both have at least 20 rows.  110 questions: the last batch has
    14 rows (24- and 32-row batches) -- runs bit-equal to MQ_SEARCH_WINDOW=0, with the short batch searched by its own call."""
    import datasets
    from viquae_amd.ir import searcher as S
    from viquae_amd.ir.search import Index, KnowledgeBase, register_index
    datasets.disable_progress_bars()
    rng = np.random.default_rng(3)
    art = rng.standard_normal((300, 24)).astype(np.float32)
    Q = rng.standard_normal((110, 24)).astype(np.float32)
    qs = datasets.Dataset.from_dict({"id": [str(i) for i in range(110)], "vec_q": [q for q in Q],
                                     "output": [{"answer": ["a"], "original_answer": "a"}] * 110})
    (tmp_path / "qrels.json").write_text("{}")
    calls = []

    def kb_kwargs():
        kb = KnowledgeBase(dataset=datasets.Dataset.from_dict({"vec": [v for v in art]}))
        register_index(kb.dataset, "dense", oracle_index(art, metric=1))
        kb.indexes["dense"] = Index(key="vec_q")
        plain = kb.search_batch

        def counted(index_name, queries, k=100):
            calls.append(len(queries))
            return plain(index_name, queries, k=k)
        kb.search_batch = counted
        return dict(kb_kwargs={"kb": {}}, k=9, kbs={"kb": kb}, qrels=str(tmp_path / "qrels.json"), do_fusion=False)

    from oracle import knn as ok
    tail = Q[110 - 14:]
    assert not np.array_equal(ok.knn(art, tail, 9, metric=1)[0], ok.knn(art, Q, 9, metric=1)[0][110 - 14:])  # the forms differ
    monkeypatch.setenv("MQ_SEARCH_WINDOW", "0")
    with pytest.warns(UserWarning):
        want = S.dataset_search(qs, report=False, map_kwargs={"batch_size": batch_size, "load_from_cache_file": False}, **kb_kwargs()).runs
    del calls[:]
    monkeypatch.setenv("MQ_SEARCH_WINDOW", "4096")
    with pytest.warns(UserWarning):
        got = S.dataset_search(qs, report=False, map_kwargs={"batch_size": batch_size, "load_from_cache_file": False}, **kb_kwargs()).runs
    assert got == want
    assert calls == expect


def test_passage_cache_eviction_refetches_the_whole_request():
    """with a small capacity the eviction dropped ids of the CURRENT request that were cached -> KeyError."""
    from viquae_amd.ir.searcher import PassageTexts, answer_preprocess
    kb = [{"passage": f"The passage, number {i}!"} for i in range(12)]
    cache = PassageTexts(kb, capacity=4)
    assert cache.get_many([0, 1, 2]) == [answer_preprocess(kb[i]["passage"]) for i in (0, 1, 2)]
    assert cache.get_many([0, 1, 3, 4, 5]) == [answer_preprocess(kb[i]["passage"]) for i in (0, 1, 3, 4, 5)]
    assert cache.get_many([5, 5, 0, 11]) == [answer_preprocess(kb[i]["passage"]) for i in (5, 5, 0, 11)]
