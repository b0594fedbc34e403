"""Parity against FAISS ITSELF -- runs only where ``import faiss`` works (not in the build container, not on the GPU box: there the
kNN oracle stays "parity unpinned" against FAISS and these tests are skipped; a user holding FAISS runs them, or
``python tools/compare_with_faiss.py [--hip]``).  Call sites pinned: meerqat/ir/search.py:146 (search), :245 (add)."""
import glob
import os

import numpy as np
import pytest

from tools import compare_with_faiss as cmp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_the_comparison_logic_on_the_oracle_stand_in():
    """No FAISS needed: the tool against the stand-in that minted the goldens must find nothing (and must find planted damage)."""
    mod = cmp._standin_faiss()
    rows = []
    assert cmp.run(mod, out=rows.append) == 0
    assert {r["case"] for r in rows} >= {"knn_random.npz", "knn_lattice.npz", "knn_ties_small_nq.npz", "neutral+zero-row"}
    assert all(r.get("max_ulp", 0) == 0 for r in rows)
    D = np.array([5, 4, 4, 3], np.float32)
    assert cmp.compare_lists(D, np.array([1, 2, 3, 4]), D, np.array([1, 3, 2, 4]), True, 0.0)[2] == 0      # order inside a tie
    assert cmp.compare_lists(D, np.array([1, 2, 3, 4]), D, np.array([1, 2, 3, 9]), True, 0.0)[2] == 0      # another row tied at the k-th
    assert cmp.compare_lists(D, np.array([1, 2, 3, 4]), D, np.array([7, 2, 3, 4]), True, 0.0)[2] == 1      # a different best row
    assert cmp.ulp_distance(np.float32(1.0), np.nextafter(np.float32(1.0), np.float32(2.0))) == 1


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "knn_*.npz"))), ids=os.path.basename)
def test_faiss_reproduces_the_goldens(path):
    faiss = pytest.importorskip("faiss")
    for row in cmp.compare_golden(path, faiss):
        assert row["unexplained_ids"] == 0, row
        if row["exact_data"]:
            assert row["max_ulp"] == 0, row          # order-independent data: bit for bit, in both L2 forms
        else:
            assert row["max_abs"] <= 4e-6 * 1e3, row  # free-form data: summation order only


def test_faiss_neutral_values_and_zero_row():
    faiss = pytest.importorskip("faiss")
    for row in cmp.compare_neutral_and_zero_row(faiss, cmp.oracle_search):
        assert row["neutral_values_agree"] and row["zero_row_agrees"], row
