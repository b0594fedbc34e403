"""CPU: runs kept as arrays to the end of a search job (viquae_amd/ir/runs.py) -- the mapping behaves like the
reference's ``runs[index_name]`` dict (meerqat/ir/search.py:413-440), and the run file written straight from the arrays by
the library's host formatter (``mq_format_run_json``) is BYTE FOR BYTE what ``json.dump`` of those dicts writes (:485-498).
Loading the library and calling its host-only entries needs no GPU."""
import io
import json
import os

import numpy as np
import pytest


def _dict_of(q_ids, ids, scores):
    out = {}
    for q, i, s in zip(q_ids, ids.tolist(), scores.tolist()):
        d = {}
        for j, v in zip(i, s):
            if j < 0:
                break
            d[str(j)] = v
        out[q] = d
    return out


def test_float_repr_is_cpythons(hip_lib):
    from viquae_amd.ir.runs import ArrayRun
    rng = np.random.default_rng(0)
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.1, 1e-4, 1e-5, 9.999e-5, 1e15, 1e16, 123456789012345678.0, 1e22, 1e-310, 5e-324,
                        1.7976931348623157e308, np.inf, -np.inf, np.nan, 0.5, 100.0, 1e3, 123456.789, 2.5e-7, 1 / 3,
                        np.finfo(np.float32).max, -np.finfo(np.float32).max, 16777216.0, 0.30000000000000004], dtype=np.float64)
    mags = 10.0 ** rng.uniform(-30, 30, 4000)
    vals = np.concatenate([special, rng.standard_normal(4000) * mags, np.round(rng.standard_normal(500) * 1e6),
                           rng.integers(-10 ** 17, 10 ** 17, 500).astype(np.float64)])
    vals = np.resize(vals, (len(vals) // 7) * 7).reshape(-1, 7)
    for scores in (vals, vals.astype(np.float32)):   # f64 tables (fused runs) and f32 tables (search results)
        with np.errstate(over="ignore"):
            scores = np.ascontiguousarray(scores)
        ids = rng.integers(0, 2 ** 62, scores.shape)
        q_ids = [f"q{i}" for i in range(len(scores))]
        run = ArrayRun()
        run.add_block(q_ids, ids, scores)
        want = json.dumps(_dict_of(q_ids, ids, scores)).encode()
        assert run.json_bytes() == want
        assert run.json_bytes(n_threads=1) == want and run.json_bytes(n_threads=5) == want
        assert run.lazy_questions() == len(q_ids)            # nothing was turned into dicts on the way
        back = json.loads(want)
        assert list(back) == q_ids


def test_many_rows_many_threads(hip_lib):
    from viquae_amd.ir.runs import ArrayRun
    rng = np.random.default_rng(1)
    nq, k = 3000, 100
    scores = rng.standard_normal((nq, k)).astype(np.float32)
    ids = np.argsort(rng.random((nq, 5000)), axis=1)[:, :k]
    ids[5, 40:] = -1            # a short row (fewer rows in the KB than k: FAISS's -1 padding)
    ids[6, :] = -1              # an empty row
    q_ids = [str(i) for i in range(nq)]
    run = ArrayRun()
    run.add_block(q_ids[:1000], ids[:1000], scores[:1000])
    run.add_block(q_ids[1000:], ids[1000:], scores[1000:])
    want = json.dumps(_dict_of(q_ids, ids, scores)).encode()
    assert run.json_bytes() == want
    f = io.StringIO()
    run.dump_json(f)
    assert f.getvalue().encode() == want
    assert run["6"] == {} and len(run["5"]) == 40 and not run.is_filled("6") and run.is_filled("5")


def test_mapping_semantics_and_mixed_entries(hip_lib, tmp_path):
    from viquae_amd.ir.runs import ArrayRun, dump_run
    rng = np.random.default_rng(2)
    ids = rng.integers(0, 1000, (4, 3))
    scores = rng.standard_normal((4, 3)).astype(np.float32)
    run = ArrayRun()
    run["first"] = {"x": 1.5, "y": float("inf")}                       # entries filled by the reference's loop are plain dicts
    run.add_block(['a "quoted" id', "é", "c", "d"], ids, scores)
    run.setdefault("last", {})["7"] = 0.25
    want = {"first": {"x": 1.5, "y": float("inf")}, **_dict_of(['a "quoted" id', "é", "c", "d"], ids, scores), "last": {"7": 0.25}}
    assert list(run) == list(want) and len(run) == 6 and "c" in run and "zz" not in run
    assert run.lazy_questions() == 4
    assert run.json_bytes() == json.dumps(want).encode()                 # ensure_ascii escapes, Infinity, mixed spans
    got = run["c"]
    assert got == want["c"] and run.lazy_questions() == 3
    got["new"] = 9.0                                                    # a caller may mutate what it was handed
    assert run["c"]["new"] == 9.0
    want["c"]["new"] = 9.0
    assert run == want and dict(run.items()) == want and run.to_dict() == want
    assert run.json_bytes() == json.dumps(want).encode()                 # the span of rows is now cut in two around "c"
    dump_run(run, tmp_path / "r.json")
    dump_run(want, tmp_path / "d.json")
    assert (tmp_path / "r.json").read_bytes() == (tmp_path / "d.json").read_bytes()
    with pytest.raises(ValueError):
        run.add_block(["c"], ids[:1], scores[:1])                       # a question already in the run is not a new row
    with pytest.raises(KeyError):
        run["nope"]
    del run["first"]
    assert "first" not in run and len(run) == 5
    assert ArrayRun().json_bytes() == b"{}"


def test_tables_for_the_late_fusion_equal_the_dict_path(hip_lib):
    from viquae_amd.ir import fuse as hfuse
    from viquae_amd.ir.runs import ArrayRun
    rng = np.random.default_rng(3)
    q_ids = [f"q{i}" for i in range(50)]
    runs_a, runs_d = [], []
    for k in (7, 12):
        ids = np.argsort(rng.random((50, 400)), axis=1)[:, :k]
        ids[3, 4:] = -1
        scores = rng.standard_normal((50, k)).astype(np.float32)
        r = ArrayRun()
        r.add_block(q_ids[:20], ids[:20], scores[:20])
        r.add_block(q_ids[20:], ids[20:], scores[20:])
        runs_a.append(r)
        runs_d.append(_dict_of(q_ids, ids, scores))
    qa, na, ia, sa = hfuse.runs_to_tables(runs_a, device="cpu")
    qd, nd, idd, sd = hfuse.runs_to_tables(runs_d, device="cpu")
    assert qa == qd and na is None and nd is None
    assert np.array_equal(ia.numpy(), idd.numpy()) and np.array_equal(sa.numpy(), sd.numpy())
    assert all(r.lazy_questions() == 50 for r in runs_a)
    runs_a[0]["q9"]                                                      # one entry became a dict: the dict path serves the run
    qm, nm, im, sm = hfuse.runs_to_tables(runs_a, device="cpu")
    assert np.array_equal(im.numpy(), idd.numpy()) and np.array_equal(sm.numpy(), sd.numpy())


def test_dataset_search_writes_the_run_file_json_dump_would(tmp_path, hip_lib):
    """The whole job on the CPU (an index object answering from numpy): runs stay arrays through ``Dataset.map``, the run file is
    the bytes ``json.dump`` of the reference-style dicts gives, and reading ``searcher.runs`` afterwards still yields those dicts."""
    import datasets
    from viquae_amd.ir import searcher as S
    from viquae_amd.ir.runs import ArrayRun
    from viquae_amd.ir.search import Index, KnowledgeBase, register_index
    from datasets.search import BaseIndex, BatchedSearchResults
    datasets.disable_progress_bars()
    rng = np.random.default_rng(4)
    art = rng.standard_normal((300, 16)).astype(np.float32)
    Q = rng.standard_normal((70, 16)).astype(np.float32)

    class Numpy(BaseIndex):
        metric_type = 0

        def search_batch(self, queries, k=10, **kw):
            S_ = np.asarray(queries, np.float32) @ art.T
            I = np.argsort(-S_, axis=1, kind="stable")[:, :k]
            return BatchedSearchResults(np.take_along_axis(S_, I, 1), I)

    qs = datasets.Dataset.from_dict({"id": [f"question {i}" for i in range(70)], "vec_q": [q for q in Q],
                                     "output": [{"answer": ["a"], "original_answer": "a"}] * 70})
    (tmp_path / "qrels.json").write_text("{}")
    kb = KnowledgeBase(dataset=datasets.Dataset.from_dict({"vec": [v for v in art]}))
    register_index(kb.dataset, "dense", Numpy())
    kb.indexes["dense"] = Index(key="vec_q")
    out = tmp_path / "metrics"
    with pytest.warns(UserWarning):
        s = S.dataset_search(qs, k=9, metric_save_path=out, report=False, map_kwargs={"batch_size": 16, "load_from_cache_file": False},
                             kb_kwargs={"kb": {}}, kbs={"kb": kb}, qrels=str(tmp_path / "qrels.json"), do_fusion=False)
    run = s.runs["dense"]
    assert isinstance(run, ArrayRun) and run.lazy_questions() == 70       # the job built no per-hit object
    D, I = Numpy().search_batch(Q, 9)
    want = _dict_of([f"question {i}" for i in range(70)], I, D)
    assert (out / "dense.json").read_bytes() == json.dumps(want).encode()
    assert run == want and list(run["question 3"]) == [str(i) for i in I[3]]
    assert os.path.exists(out / "qrels.json")


def test_run_file_with_question_ids_that_are_not_strings():
    """json.dump coerces int / float / bool / None dict keys to strings; the array writer must write the same bytes."""
    import io
    import json
    import numpy as np
    from viquae_amd.ir.runs import ArrayRun
    ids = np.array([[3, 1, -1], [7, 8, 9], [2, -1, -1], [4, 5, 6], [1, 2, 3]], dtype=np.int64)
    scores = np.array([[1.5, 0.25, 0], [3, 2, 1], [9.75, 0, 0], [1, 0.5, 0.1], [2, 1, 0]], dtype=np.float32)
    q_ids = [5, "six", 7.5, True, None]
    run = ArrayRun()
    run.add_block(q_ids, ids, scores)
    run[12] = {"4": 1.0}                       # a dict entry with an int id as well
    want = json.dumps(run.to_dict()).encode()
    assert run.json_bytes() == want
    assert json.loads(run.json_bytes()) == {"5": {"3": 1.5, "1": 0.25}, "six": {"7": 3.0, "8": 2.0, "9": 1.0}, "7.5": {"2": 9.75},
                                            "true": {"4": 1.0, "5": 0.5, "6": float(np.float32(0.1))}, "null": {"1": 2.0, "2": 1.0, "3": 0.0},
                                            "12": {"4": 1.0}}
    buf = io.BytesIO()
    run.dump_json(buf)
    assert buf.getvalue() == want
