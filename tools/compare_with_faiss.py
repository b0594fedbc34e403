#!/usr/bin/env python3
"""Pins this library's kNN arithmetic against a REAL FAISS build, when one is importable (none is in the build container: the
oracle is "parity unpinned" against FAISS itself).  For a user who holds FAISS:

    python tools/compare_with_faiss.py            # FAISS vs the committed goldens (tests/golden/knn_*.npz = this library's answers)
    python tools/compare_with_faiss.py --hip      # additionally FAISS vs the HIP library on this machine's GPU, free-form data
    python tools/compare_with_faiss.py --selftest # the comparison logic itself, against the oracle stand-in (no FAISS needed)

What is compared, per golden case and per (factory, metric, k) -- the reference's call sites are meerqat/ir/search.py:146 (search)
and :245 (add), through datasets' FaissIndex (index_factory(d, string_factory, metric), add, search):
  * scores: maximum distance in units in the last place (ulp) and in absolute terms.  On the integer-lattice and tie-heavy
    goldens every fp32 summation order is exact, so FAISS must reproduce the golden scores bit for bit, in BOTH L2 forms (the
    goldens hold 37 / 21 queries = BLAS form and 7 / 19 queries = direct form); on free-form data FAISS's sgemm / SIMD order
    differs from the k-ordered fma chain by a few ulp of the largest partial sum;
  * ids: rows of equal score may be listed in any order and, at the k-th boundary, any of the tied rows may be kept -- FAISS's
    choice depends on its version and on k (oracle/knn_oracle.c header) -- so ids are compared as SETS inside runs of equal score,
    and a disagreement is only counted when it is not explained by a tie (exact scores) or a near-tie (free-form data: the two
    score lists agree within the tolerance once sorted);
  * the neutral values of unfilled slots (k > ntotal): id -1, score -FLT_MAX (IP) / +FLT_MAX (L2);
  * "L2norm,Flat": FAISS's NormalizationTransform against MQ_L2NORM_FAISS (a planted zero row included).
Exit status 0 = nothing unexplained."""
import argparse
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
FLT_MAX = np.finfo(np.float32).max


def ulp_distance(a, b):
    """Elementwise distance of two float32 arrays in units in the last place (monotone integer map of the bit patterns)."""
    def key(x):
        u = np.ascontiguousarray(x, np.float32).view(np.int32).astype(np.int64)
        return np.where(u < 0, -(u & 0x7FFFFFFF), u)
    return np.abs(key(a) - key(b))


def compare_lists(D_ref, I_ref, D_got, I_got, exact_scores, tol):
    """One query's two result lists.  Returns (max ulp, max abs, unexplained id differences)."""
    ulp = int(ulp_distance(D_ref, D_got).max()) if len(D_ref) else 0
    dabs = float(np.max(np.abs(D_ref.astype(np.float64) - D_got.astype(np.float64)))) if len(D_ref) else 0.0
    if np.array_equal(I_ref, I_got):
        return ulp, dabs, 0
    bad = 0
    # group the positions by score (exact) or by proximity (tolerance): inside a group any order / any tied member is fine
    edges = [0]
    for j in range(1, len(D_ref)):
        same = (D_ref[j] == D_ref[j - 1]) if exact_scores else abs(float(D_ref[j]) - float(D_ref[j - 1])) <= tol
        if not same:
            edges.append(j)
    edges.append(len(D_ref))
    for a, b in zip(edges[:-1], edges[1:]):
        ra, ga = set(I_ref[a:b].tolist()), set(I_got[a:b].tolist())
        if ra != ga:
            last_group = b == len(D_ref)  # at the k-th boundary other tied rows may have been kept: scores decide
            scores_agree = np.array_equal(np.sort(D_ref[a:b]), np.sort(D_got[a:b])) if exact_scores else \
                np.allclose(np.sort(D_ref[a:b]), np.sort(D_got[a:b]), rtol=0, atol=tol)
            if not (last_group and scores_agree) and not (not exact_scores and scores_agree):
                bad += len(ra ^ ga) // 2
    return ulp, dabs, bad


def faiss_search(faiss, X, Q, k, factory, metric):
    index = faiss.index_factory(int(X.shape[1]), factory, faiss.METRIC_INNER_PRODUCT if metric == 0 else faiss.METRIC_L2)
    index.add(np.ascontiguousarray(X, np.float32))
    D, I = index.search(np.ascontiguousarray(Q, np.float32), k)
    return np.asarray(D, np.float32), np.asarray(I, np.int64)


def compare_golden(path, faiss, host_l2norm=None):
    """FAISS against every (factory, metric, k) result of one golden file.  Yields report rows."""
    from viquae_amd.ir.search import L2norm
    z = np.load(path)
    exact = z["X"].dtype.kind == "i"  # integer-lattice / tie-heavy cases: every summation order is exact
    X, Q = z["X"].astype(np.float32), z["Q"].astype(np.float32)
    for key in z.files:
        if not key.startswith("I_") or key.startswith("I_none"):
            continue
        tag = key[2:]
        l2n = tag.startswith("l2norm_")
        m, kk = tag.replace("l2norm_", "").split("_")
        metric, k = int(m[1:]), int(kk[1:])
        Qin = (host_l2norm or L2norm)(Q) if l2n else Q  # KnowledgeBase.search_batch normalises on the host first (:144-145)
        D, I = faiss_search(faiss, X, Qin, k, "L2norm,Flat" if l2n else "Flat", metric)
        tol = 4e-6 * float(max(np.abs(z[f"D_{tag}"]).max(), 1.0))
        worst = [compare_lists(z[f"D_{tag}"][q], z[f"I_{tag}"][q], D[q], I[q], exact, tol) for q in range(len(Q))]
        yield {"case": os.path.basename(path), "tag": tag, "queries": len(Q), "l2_form": ("direct" if len(Q) < 20 else "blas") if metric else "-",
               "exact_data": exact, "max_ulp": max(w[0] for w in worst), "max_abs": max(w[1] for w in worst),
               "unexplained_ids": sum(w[2] for w in worst)}


def compare_neutral_and_zero_row(faiss, search):
    """k > ntotal (neutral fill) and a zero KB row under "L2norm,Flat".  `search(X, Q, k, factory, metric)` = the other side."""
    rng = np.random.default_rng(7)
    X = rng.standard_normal((50, 16)).astype(np.float32)
    X[3] = 0.0
    Q = rng.standard_normal((21, 16)).astype(np.float32)
    rows = []
    for factory in ("Flat", "L2norm,Flat"):
        for metric in (0, 1):
            Df, If = faiss_search(faiss, X, Q, 64, factory, metric)
            Do, Io = search(X, Q, 64, factory, metric)
            fill = -FLT_MAX if metric == 0 else FLT_MAX
            ok = bool((If[:, 50:] == -1).all() and (Io[:, 50:] == -1).all() and (Df[:, 50:] == fill).all() and (Do[:, 50:] == fill).all())
            zero_in = bool(all(3 in r for r in If[:, :50]) == all(3 in r for r in Io[:, :50]))
            rows.append({"case": "neutral+zero-row", "tag": f"{factory} m{metric}", "neutral_values_agree": ok, "zero_row_agrees": zero_in})
    return rows


def oracle_search(X, Q, k, factory, metric):
    from oracle import knn as ok
    return ok.knn(X, Q, k, metric=metric, l2norm="L2norm" in factory, l2norm_form="faiss")


def hip_search(X, Q, k, factory, metric):
    from viquae_amd.index import MI355XFlatIndex
    idx = MI355XFlatIndex(string_factory=factory, metric_type=metric, l2norm_form="faiss")
    idx.add_vectors(X)
    D, I = idx.search_batch(Q, k)
    return np.asarray(D, np.float32), np.asarray(I, np.int64)


def _standin_faiss():
    """A module-shaped stand-in serving FAISS's API from the oracle (tools/ref_import.py): for --selftest only."""
    import types
    from tools import ref_import
    m = types.SimpleNamespace(METRIC_INNER_PRODUCT=0, METRIC_L2=1)

    def index_factory(d, factory, metric):
        return ref_import._PreTransformStandIn(d, metric) if "L2norm" in factory else ref_import._FlatStandIn(d, metric)
    m.index_factory = index_factory
    return m


def run(faiss, hip=False, out=print):
    bad = 0
    for path in sorted(glob.glob(os.path.join(GOLDEN, "knn_*.npz"))):
        for row in compare_golden(path, faiss):
            out(row)
            bad += row["unexplained_ids"] + (row["max_ulp"] if row["exact_data"] else 0)
    for row in compare_neutral_and_zero_row(faiss, hip_search if hip else oracle_search):
        out(row)
        bad += (not row["neutral_values_agree"]) + (not row["zero_row_agrees"])
    if hip:
        rng = np.random.default_rng(11)
        X = rng.standard_normal((100_000, 128)).astype(np.float32)
        for nq in (7, 256):
            Q = rng.standard_normal((nq, 128)).astype(np.float32)
            for metric in (0, 1):
                Df, If = faiss_search(faiss, X, Q, 100, "Flat", metric)
                Dh, Ih = hip_search(X, Q, 100, "Flat", metric)
                tol = 4e-6 * float(np.abs(Df).max())
                w = [compare_lists(Dh[q], Ih[q], Df[q], If[q], False, tol) for q in range(nq)]
                row = {"case": "hip-vs-faiss 100k x 128", "tag": f"m{metric} nq{nq}", "max_ulp": max(x[0] for x in w), "max_abs": max(x[1] for x in w),
                       "unexplained_ids": sum(x[2] for x in w)}
                out(row)
                bad += row["unexplained_ids"]
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--hip", action="store_true", help="also compare FAISS with the HIP library on this machine's GPU")
    ap.add_argument("--selftest", action="store_true", help="run the comparison against the oracle stand-in instead of FAISS")
    args = ap.parse_args()
    if args.selftest:
        mod = _standin_faiss()
    else:
        try:
            import faiss as mod
        except ImportError:
            sys.exit("faiss is not importable here: nothing to compare with (python tools/compare_with_faiss.py --selftest checks the tool)")
    unexplained = run(mod, hip=args.hip)
    print("unexplained differences:", unexplained)
    sys.exit(1 if unexplained else 0)
