#!/usr/bin/env python3
"""Mints tests/golden/*.npz by RUNNING THE REFERENCE in this container (never on the GPU box).

kNN fixtures (``knn_*.npz``): a ``datasets.Dataset`` with a vector column is saved to disk and
searched through the reference's own ``meerqat.ir.search.KnowledgeBase`` (``search_batch`` and
``search_batch_if_not_None``, meerqat/ir/search.py:135-171), whose FAISS dependency is served by the
stand-in of tools/ref_import.py (the CPU oracle).  Integer-lattice and tie-heavy cases are
additionally asserted against an independent float64 computation at generation time: on such data
every summation order gives the same fp32 SCORES, so the score arrays are also what any correct
IndexFlat returns; which of several exactly tied rows is listed (and in which order) is THIS
LIBRARY'S documented policy -- lower id first, "id_asc" -- not a statement about FAISS, whose
tie behaviour depends on its version and on k (oracle/knn_oracle.c header).

Encoder fixtures (``dpr_*.npz``, ``clip_*.npz``): Hugging Face ``DPRContextEncoder`` /
``CLIPModel.get_image_features`` (the code the reference calls, meerqat/ir/embedding.py:226,
meerqat/image/embedding.py:156-161) run on CPU fp32 with seeded weights from
``oracle.encoders.seeded_state`` -- only the small outputs are stored; weights are regenerated
from the seed on both sides.

Usage: python tools/make_golden.py [knn] [encoders]
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from tools import ref_import  # noqa: E402


def _kb(ref_search, X, factory, metric, tmp):
    import datasets
    ds = datasets.Dataset.from_dict({"vec": [r for r in X.astype(np.float32)], "passage": [str(i) for i in range(len(X))]})
    path = os.path.join(tmp, f"kb_{abs(hash((factory, metric, X.shape))) % 10**8}")
    ds.save_to_disk(path)
    # index kwargs as the shipped configs write them, legacy keys removed (they make the current
    # reference raise TypeError)
    kw = {"column": "vec", "key": "q", "string_factory": factory, "load": False, "device": None, "metric_type": metric}
    return ref_search.KnowledgeBase(path, index_kwargs={"idx": kw})


def _f64_check(X, Q, D, I, metric, k):
    from oracle.knn import knn_numpy_f64
    D64, I64 = knn_numpy_f64(X, Q, k, metric)
    assert np.array_equal(I, I64), "lattice fixture disagrees with the independent float64 computation"
    assert np.array_equal(D.astype(np.float64), D64)


def make_knn():
    ref_search = ref_import.import_reference_search()
    rng = np.random.default_rng(20240601)
    cases = {
        # name: (X, Q, ks, exact-in-any-order?)
        "random": (rng.standard_normal((2048, 64), dtype=np.float32), rng.standard_normal((37, 64), dtype=np.float32),
                   (1, 10, 100), False),
        "small_nq": (rng.standard_normal((1500, 96), dtype=np.float32), rng.standard_normal((5, 96), dtype=np.float32),
                     (100,), False),
        "lattice": (rng.integers(-128, 129, (1024, 768)).astype(np.float32),
                    rng.integers(-128, 129, (37, 768)).astype(np.float32), (100,), True),
        "ties": (rng.integers(-2, 3, (3000, 16)).astype(np.float32), rng.integers(-2, 3, (21, 16)).astype(np.float32),
                 (100,), True),
    }
    # fewer than 20 queries: FAISS's sequential L2 path (direct sum of (q-x)^2).  Values in [-64, 64] keep every
    # partial sum of squares below 2^24, so the fp32 result is exact in any order = what FAISS itself returns.
    # (drawn after the cases above so that their data, hence their fixtures, stay what they were)
    cases["lattice_small_nq"] = (rng.integers(-64, 65, (1024, 768)).astype(np.float32),
                                 rng.integers(-64, 65, (7, 768)).astype(np.float32), (100,), True)
    cases["ties_small_nq"] = (rng.integers(-2, 3, (3000, 16)).astype(np.float32), rng.integers(-2, 3, (19, 16)).astype(np.float32),
                              (100,), True)
    with tempfile.TemporaryDirectory() as tmp:
        for name, (X, Q, ks, exact) in cases.items():
            out = {"X": X.astype(np.int16) if exact else X, "Q": Q.astype(np.int16) if exact else Q}
            for factory in ("Flat", "L2norm,Flat"):
                if exact and factory != "Flat":
                    continue
                for metric in (0, 1):
                    kb = _kb(ref_search, X, factory, metric, tmp)
                    assert kb.indexes["idx"].do_L2norm == ("L2norm" in factory)
                    for k in ks:
                        D, I = kb.search_batch("idx", [list(map(float, q)) for q in Q], k=k)  # list-of-lists in, as HF hands it
                        assert D.dtype == np.float32 and I.shape == (len(Q), k)
                        if exact:
                            _f64_check(X, Q, D, I, metric, k)
                        tag = f"{'l2norm_' if 'L2norm' in factory else ''}m{metric}_k{k}"
                        out[f"D_{tag}"], out[f"I_{tag}"] = D, I.astype(np.int64)
            # None-query pattern through search_batch_if_not_None
            if name == "random":
                mask = np.array([i % 3 != 1 for i in range(len(Q))])
                kb = _kb(ref_search, X, "Flat", 0, tmp)
                queries = [Q[i] if mask[i] else None for i in range(len(Q))]
                S, Ix = kb.search_batch_if_not_None("idx", queries, k=10)
                assert all((len(s) == 0) == (not m) for s, m in zip(S, mask))
                out["none_mask"] = mask
                out["D_none_m0_k10"] = np.stack([s for s, m in zip(S, mask) if m])
                out["I_none_m0_k10"] = np.stack([s for s, m in zip(Ix, mask) if m]).astype(np.int64)
            np.savez_compressed(os.path.join(GOLDEN, f"knn_{name}.npz"), **out)
            print("wrote", f"knn_{name}.npz", {k: v.shape for k, v in out.items() if k.startswith("I_")})


if __name__ == "__main__":
    what = sys.argv[1:] or ["knn", "encoders"]
    os.makedirs(GOLDEN, exist_ok=True)
    if "knn" in what:
        make_knn()
    if "encoders" in what:
        from tools import make_golden_encoders
        make_golden_encoders.main()
