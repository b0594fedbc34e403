"""Late fusion, CPU side: the oracle restatement against the outputs of the reference's own functions
(tests/golden/fuse.json, minted by tools/make_golden_fuse.py), and the host-side run <-> table conversion."""
import json
import os

import numpy as np
import pytest

from oracle import fuse as ofuse

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "fuse.json")


def cases():
    with open(GOLDEN) as file:
        return json.load(file)["cases"]


@pytest.mark.parametrize("case", cases(), ids=lambda c: c["name"])
def test_oracle_matches_reference_default_minimum_and_gzmuv(case):
    runs = case["runs"]
    defmin = ofuse.default_minimum(runs)
    assert defmin == case["reference_default_minimum"]
    assert [ofuse.gzmuv_norm(r) for r in defmin] == case["reference_gzmuv_after_defmin"]
    assert [ofuse.gzmuv_norm(r) for r in runs] == case["reference_gzmuv_no_defmin"]
    # inputs are not modified (the reference mutates its Run objects in place; the oracle copies)
    assert runs == case["runs"]


@pytest.mark.parametrize("case", cases(), ids=lambda c: c["name"])
def test_oracle_fusion_pipeline(case):
    fused = ofuse.fusion_test(case["runs"], case["weights"], norm="gzmuv", defmin=True)
    assert fused == case["unpinned_wsum_gzmuv_defmin"]
    assert list(fused) == list(case["runs"][0])
    for q, results in fused.items():
        scores = list(results.values())
        assert scores == sorted(scores, reverse=True)
        union = set().union(*[run[q].keys() for run in case["runs"]])
        assert results.keys() == union


def test_oracle_wsum_properties():
    runs = [{"q": {"1": 1.0, "2": 3.0}}, {"q": {"2": 1.0, "3": 5.0}}]
    fused = ofuse.wsum(runs, [0.5, 2.0])
    assert fused == {"q": {"3": 10.0, "2": 3.5, "1": 0.5}}
    # ties -> ascending integer id
    assert list(ofuse.wsum([{"q": {"10": 1.0, "9": 1.0, "100": 1.0}}], [1.0])["q"]) == ["9", "10", "100"]
    # default minimum: a run without results for a query stays empty
    dm = ofuse.default_minimum([{"q": {"1": 2.0, "2": 1.0}}, {"q": {}}, {"q": {"3": 7.0}}])
    assert dm == [{"q": {"1": 2.0, "2": 1.0, "3": 1.0}}, {"q": {}}, {"q": {"3": 7.0, "1": 7.0, "2": 7.0}}]


def test_runs_to_tables_roundtrip_cpu():
    torch = pytest.importorskip("torch")
    from viquae_amd.ir import fuse as hfuse
    runs = [{"a": {"5": 1.5, "7": 0.5}, "b": {}}, {"a": {"7": 2.0}, "b": {"1": 1.0, "2": 0.0, "3": -1.0}}]
    q_ids, names, ids, scores = hfuse.runs_to_tables(runs, device="cpu")
    assert q_ids == ["a", "b"] and names is None
    assert ids.shape == (2, 2, 3) and ids.dtype == torch.int64 and scores.dtype == torch.float64
    assert ids[0, 0].tolist() == [5, 7, -1] and ids[0, 1].tolist() == [-1, -1, -1]
    assert scores[1, 1].tolist() == [1.0, 0.0, -1.0]
    back = hfuse.tables_to_run(q_ids, names, ids[1], scores[1], torch.tensor([1, 3], dtype=torch.int32))
    assert back == runs[1]
    # non-numeric document ids are numbered in sorted order
    q_ids, names, ids, _ = hfuse.runs_to_tables([{"q": {"x": 1.0, "b": 2.0}}], device="cpu")
    assert names == ["b", "x"] and ids[0, 0].tolist() == [1, 0]
    with pytest.raises(ValueError):
        hfuse.runs_to_tables([{"q": {}}, {"other": {}}], device="cpu")


def test_fusion_refuses_what_it_does_not_implement():
    pytest.importorskip("torch")
    from viquae_amd.ir import fuse as hfuse
    f = hfuse.Fusion(runs=[{"q": {"1": 1.0}}, {"q": {"1": 2.0}}], norm="gzmuv", method="rrf")
    with pytest.raises(NotImplementedError):
        f.test({"weights": [0.5, 0.5]})
    with pytest.raises(ValueError):          # the weight search needs relevance judgements
        f.fit()
    f = hfuse.Fusion(qrels={"q": {"1": 1}}, runs=[{"q": {"1": 1.0}}, {"q": {"1": 2.0}}], norm="gzmuv", method="rrf")
    with pytest.raises(NotImplementedError):  # only ranx's `wsum` grid is restated
        f.fit()


def test_wsum_trial_set_is_the_restatement_of_ranx_grid():
    """Host logic of Fusion.fit: the trial set (ranx fusion/wsum.py + common.py as published, float-sum quirk included) equals
    the oracle's, and so do the parsed metrics and the CSR form of the qrels the device takes."""
    pytest.importorskip("torch")
    import numpy as np
    from oracle import fuse as ofuse
    from viquae_amd.ir import fuse as hfuse, metrics as M
    for n, count in ((2, 11), (3, 62), (4, 256)):
        trials = hfuse.wsum_trials(n)
        assert trials == ofuse.wsum_trials(n) and len(trials) == count
        assert all(sum(t) == 1.0 for t in trials) and trials[0][-1] == 1.0 and trials[-1][0] == 1.0
    assert M.parse_metric("mrr@100") == (0, 100) and M.parse_metric("precision") == (1, 0) and M.parse_metric("hit_rate@20") == (2, 20)
    with pytest.raises(NotImplementedError):
        M.parse_metric("ndcg@10")
    ptr, rel = M.qrels_to_csr({"a": {"7": 1, "3": 1, "9": 0}, "c": {"x": 1}}, ["a", "b", "c"])
    assert ptr.tolist() == [0, 2, 2, 3] and rel[:2].tolist() == [3, 7] and rel[2] > 1 << 61   # a relevant document no run can hold
    q_ids, ids, lookup = M.run_tables({"a": {"7": 2.0, "1": 1.0}, "b": {}})
    assert q_ids == ["a", "b"] and ids.tolist() == [[7, 1], [-1, -1]] and lookup is None
