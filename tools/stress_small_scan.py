"""Randomised stress of the one-query-tile streaming scan (csrc/knn_small.inc) against the exact fp32 scan on the GPU: bit-equal
scores and ids over shards of 65,536 ... 400,000 rows, 1 ... 256 queries, any d up to 768, k up to 128, both metrics, both
"L2norm," arithmetics, both tie orders, clustered / duplicated / heavy-tailed / low-rank / sorted data.
usage: python tools/stress_small_scan.py [n_cases] [first_seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viquae_amd.index import MI355XFlatIndex


def case(seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    pick = lambda xs: xs[int(torch.randint(0, len(xs), (1,), generator=g, device="cuda"))]  # noqa: E731
    n = pick([65536, 65537, 70001, 100000, 131072, 200003, 400000])
    d = pick([3, 32, 64, 100, 200, 256, 320, 512, 700, 766, 768])
    nq = pick([1, 7, 19, 20, 21, 63, 64, 65, 100, 200, 255, 256])
    k = pick([1, 10, 100, 128])
    regime = pick(["normal", "normal", "clustered", "dups", "scaled", "lowrank", "sorted", "l2norm_numpy", "l2norm_faiss", "shared", "shared"])
    metric = pick([0, 0, 1])
    tie = pick(["id_asc", "id_asc", "id_desc"])
    X = torch.randn((n, d), generator=g, device="cuda")
    Q = torch.randn((nq, d), generator=g, device="cuda")
    factory, form = "Flat", None
    if regime == "clustered":
        c = torch.randn((50, d), generator=g, device="cuda") * 3
        X = c[torch.randint(0, 50, (n,), generator=g, device="cuda")] + 0.05 * X
        Q = c[torch.randint(0, 50, (nq,), generator=g, device="cuda")] + 0.05 * Q
    elif regime == "dups":
        src = torch.randint(0, n, (n // 2,), generator=g, device="cuda")
        dst = torch.randint(0, n, (n // 2,), generator=g, device="cuda")
        X[dst] = X[src]
    elif regime == "scaled":
        X = X * torch.exp(2 * torch.randn((n, 1), generator=g, device="cuda"))
        Q = Q * torch.exp(2 * torch.randn((nq, 1), generator=g, device="cuda"))
    elif regime == "lowrank":
        r = torch.randn((4, d), generator=g, device="cuda")
        X = torch.randn((n, 4), generator=g, device="cuda") @ r + 1e-3 * X
        Q = torch.randn((nq, 4), generator=g, device="cuda") @ r
    elif regime == "sorted":
        X = X[torch.argsort(X @ Q[0])]
    elif regime.startswith("l2norm"):
        factory, form = "L2norm,Flat", regime.split("_")[1]
    elif regime == "shared":  # a common component 0.5 ... 6 x the isotropic part: the centred-query screen where the index picks it
        ratio = float(pick([0.5, 1.0, 2.0, 4.0, 6.0]))
        mu = torch.randn((1, d), generator=g, device="cuda")
        mu = ratio * mu / mu.norm() * d ** 0.5
        X, Q = X + mu, Q + mu
    return X, Q, k, regime, factory, form, metric, tie


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = streamed = flagged = 0
    by_regime = {}
    t0 = time.time()
    for seed in range(first, first + n_cases):
        X, Q, k, regime, factory, form, metric, tie = case(seed)
        a = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=True, tie_order=tie, l2norm_form=form)
        a.add(X)
        b = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=False, tie_order=tie, l2norm_form=form)
        b.add(X)
        kind = a.scan_kind(Q.shape[0], k)
        D, I = a.search_device(Q, k)
        D0, I0 = b.search_device(Q, k)
        torch.cuda.synchronize()
        streamed += kind == "stream"
        if kind != "none":
            f = a.screen_stats(Q.shape[0], k)[0]
            flagged += f
            if kind == "stream":
                r = by_regime.setdefault(regime, [0, 0])
                r[0] += 1
                r[1] += f
        same = torch.equal(I, I0) and torch.equal(D.view(torch.int32), D0.view(torch.int32))
        if not same:
            bad += 1
            print(f"MISMATCH seed {seed}: n {X.shape[0]} d {X.shape[1]} nq {Q.shape[0]} k {k} {regime} metric {metric} {tie} scan {kind}", flush=True)
        del a, b
    print(f"{n_cases} cases from seed {first}: {streamed} through the streaming scan, {flagged} query tiles recomputed exactly, "
          f"{bad} mismatches ({time.time() - t0:.0f} s)")
    print("streamed cases / of which recomputed exactly, by data regime:", {k_: tuple(v) for k_, v in sorted(by_regime.items())})
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
