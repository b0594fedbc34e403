#!/usr/bin/env python3
"""Golden for the Searcher mirror: runs the REFERENCE's Searcher.__call__ (meerqat/ir/search.py:401-459) in this
container (Elasticsearch stubbed, FAISS served by the oracle stand-in) over a small KB with an article->passage
index_mapping, with and without many2one='max', and stores batch + runs + qrels as JSON."""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ref_import  # noqa: E402


def main():
    import datasets
    ref_search = ref_import.import_reference_search()
    ref_search.Elasticsearch = lambda *a, **k: None  # the reference constructs a client even when unused
    rng = np.random.default_rng(3)
    words = ["paris", "rome", "berlin", "madrid", "lisbon", "vienna", "oslo", "cairo", "delhi", "tokyo", "lima", "quito"]
    n_art, d = 40, 16
    art_vec = rng.standard_normal((n_art, d)).astype(np.float32)
    mapping, passages = {}, []
    for a in range(n_art):
        m = int(rng.integers(1, 4))
        mapping[str(a)] = list(range(len(passages), len(passages) + m))
        for _ in range(m):
            passages.append("The city of " + " and ".join(rng.choice(words, 3, replace=False)) + ".")
    out = {"passages": passages, "mapping": mapping, "art_vec": art_vec.tolist(), "cases": {}}
    with tempfile.TemporaryDirectory() as tmp:
        datasets.Dataset.from_dict({"vec": [v for v in art_vec]}).save_to_disk(os.path.join(tmp, "articles"))
        datasets.Dataset.from_dict({"passage": passages}).save_to_disk(os.path.join(tmp, "passages"))
        json.dump(mapping, open(os.path.join(tmp, "map.json"), "w"))
        nq = 7
        Q = rng.standard_normal((nq, d)).astype(np.float32)
        batch = {"id": [f"q{i}" for i in range(nq)], "vec_q": [q for q in Q],
                 "output": [{"original_answer": str(rng.choice(words)).title(), "answer": [str(w) for w in rng.choice(words, 2)]}
                            for _ in range(nq)]}
        out["batch"] = {"id": batch["id"], "vec_q": Q.tolist(), "output": batch["output"]}
        for name, many2one, k in (("one2many", None, 10), ("many2one_max", "max", 10), ("cut", None, 3)):
            s = ref_search.Searcher(
                kb_kwargs={os.path.join(tmp, "articles"): {
                    "index_mapping_path": os.path.join(tmp, "map.json"), "many2one": many2one,
                    "index_kwargs": {"dense": {"column": "vec", "key": "vec_q", "string_factory": "Flat", "metric_type": 0}}}},
                k=k, reference_kb_path=os.path.join(tmp, "passages"), reference_key="passage")
            s(dict(batch))
            runs = {n: {q: {doc: float(sc) for doc, sc in r.items()} for q, r in run.items()} for n, run in s.runs.items()}
            out["cases"][name] = {"k": k, "many2one": many2one, "runs": runs, "qrels": s.qrels,
                                  "qnonrels": {q: sorted(v) for q, v in s.qnonrels.items()}}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "searcher.json"), "w"))
    print("wrote tests/golden/searcher.json", {k: len(v["runs"]["dense"]) for k, v in out["cases"].items()})


if __name__ == "__main__":
    main()
