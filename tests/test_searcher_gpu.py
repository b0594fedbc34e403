"""GPU: Searcher mirror end to end over the HIP index against the reference-minted golden."""
import json

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["one2many", "many2one_max", "cut"])
def test_searcher_hip_index_matches_reference_golden(case):
    from tests.test_searcher_cpu import GOLDEN, build, check
    from viquae_amd.index import MI355XFlatIndex

    def hip_index(art):
        idx = MI355XFlatIndex(string_factory="Flat", metric_type=0)
        idx.add_vectors(art)
        return idx
    golden = json.load(open(GOLDEN))
    check(build(golden, case, hip_index), golden["cases"][case])


def test_dataset_search_two_indexes_then_fusion(tmp_path):
    """dataset_search over a KB with a DPR-like and a CLIP-like index, config shaped like
    experiments/ir/viquae/dpr+clip/config.json: per-index runs from the HIP search, then gzmuv/defmin/wsum fusion on
    the device, checked against the oracle chain (oracle kNN -> oracle fusion)."""
    import datasets
    import numpy as np
    from oracle import fuse as ofuse, knn as ok
    from viquae_amd.ir.searcher import dataset_search
    rng = np.random.default_rng(11)
    n, nq, k = 3000, 24, 20
    kb_cols = {"dpr": rng.standard_normal((n, 48)).astype(np.float32), "clip": rng.standard_normal((n, 32)).astype(np.float32),
               "passage": [f"passage number {i}" for i in range(n)]}
    kb_path = str(tmp_path / "kb")
    datasets.Dataset.from_dict({c: list(v) if c == "passage" else [r for r in v] for c, v in kb_cols.items()}).save_to_disk(kb_path)
    q_dpr = rng.standard_normal((nq, 48)).astype(np.float32)
    q_clip = rng.standard_normal((nq, 32)).astype(np.float32)
    q_clip_list = [None if i % 5 == 0 else q for i, q in enumerate(q_clip)]   # some questions have no image vector
    questions = datasets.Dataset.from_dict({
        "id": [f"q{i}" for i in range(nq)], "dpr_q": [q for q in q_dpr], "clip_q": q_clip_list,
        "output": [{"original_answer": "number 7", "answer": ["number 7"]}] * nq})
    config = {
        "kb_kwargs": {kb_path: {"index_kwargs": {
            "dpr": {"column": "dpr", "key": "dpr_q", "string_factory": "Flat", "metric_type": 0, "device": 0,
                    "es": False, "kind_str": "TEXT", "normalization": {"method": "zscore"}, "interpolation_weight": 0.5},
            "clip": {"column": "clip", "key": "clip_q", "string_factory": "L2norm,Flat", "metric_type": 0, "device": 0}}}},
        "reference_kb_path": kb_path, "reference_key": "passage",
        "fusion_kwargs": {"subcommand": "test", "norm": "gzmuv", "defmin": True,
                          "subcommand_kwargs": {"best_params": {"weights": [0.6, 0.4]}}},
    }
    out = tmp_path / "metrics"
    searcher = dataset_search(questions, k=k, metric_save_path=out, **config)
    assert searcher.do_fusion and set(searcher.runs) == {"dpr", "clip"}
    # oracle chain
    D1, I1 = ok.knn(kb_cols["dpr"], q_dpr, k, metric=0)
    keep = [i for i in range(nq) if i % 5 != 0]
    D2, I2 = ok.knn(ok.l2norm_rows(kb_cols["clip"]), ok.l2norm_rows(q_clip[keep]), k, metric=0)
    want_dpr = {f"q{i}": {str(int(j)): float(s) for s, j in zip(D1[i], I1[i])} for i in range(nq)}
    want_clip = {f"q{i}": {} for i in range(nq)}
    for row, i in enumerate(keep):
        want_clip[f"q{i}"] = {str(int(j)): float(s) for s, j in zip(D2[row], I2[row])}
    assert {q: list(r) for q, r in searcher.runs["dpr"].items()} == {q: list(r) for q, r in want_dpr.items()}
    assert {q: list(r) for q, r in searcher.runs["clip"].items()} == {q: list(r) for q, r in want_clip.items()}
    want = ofuse.fusion_test([searcher.runs["dpr"], searcher.runs["clip"]], [0.6, 0.4], norm="gzmuv", defmin=True)
    fused = searcher.fusion if isinstance(searcher.fusion, dict) else searcher.fusion.to_dict()
    from tests.test_fuse_gpu import assert_same_run
    assert_same_run(fused, want)
    assert_same_run(json.load(open(out / "test_run.json")), want)
    assert (out / "dpr.json").exists() and (out / "clip.json").exists() and (out / "qrels.json").exists()


def test_search_cli_main(tmp_path):
    """python -m viquae_amd.ir.searcher <dataset> <config> --k --metrics: the reference's CLI wiring (search.py:527-543)."""
    import datasets
    import numpy as np
    from oracle import knn as ok
    from viquae_amd.ir import searcher
    rng = np.random.default_rng(2)
    X = rng.standard_normal((500, 32)).astype(np.float32)
    kb_path, q_path = str(tmp_path / "kb"), str(tmp_path / "questions")
    datasets.Dataset.from_dict({"vec": [r for r in X], "passage": [f"p {i}" for i in range(500)]}).save_to_disk(kb_path)
    Q = rng.standard_normal((9, 32)).astype(np.float32)
    datasets.Dataset.from_dict({"id": [f"q{i}" for i in range(9)], "vec_q": [r for r in Q],
                                "output": [{"original_answer": "p 3", "answer": ["p 3"]}] * 9}).save_to_disk(q_path)
    config = {"kb_kwargs": {kb_path: {"index_kwargs": {"dense": {"column": "vec", "key": "vec_q", "string_factory": "Flat",
                                                                  "metric_type": 0}}}},
              "reference_kb_path": kb_path, "reference_key": "passage", "format": {"type": "numpy", "columns": ["vec_q"],
                                                                                   "output_all_columns": True}}
    cfg_path = tmp_path / "config.json"
    cfg_path.write_text(json.dumps(config))
    s = searcher.main(q_path, str(cfg_path), k=5, metrics=str(tmp_path / "m"))
    D, I = ok.knn(X, Q, 5, metric=0)
    assert {q: list(r) for q, r in s.runs["dense"].items()} == {f"q{i}": [str(int(j)) for j in I[i]] for i in range(9)}
    assert (tmp_path / "m" / "dense.json").exists()
