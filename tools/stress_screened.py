"""Randomised stress of the screened search against the exact fp32 scan on the GPU (bit-equal scores and ids), over
larger shapes than the pytest fuzz: many slabs / few slabs, several query tiles, clustered and duplicated data,
adversarial scales.  usage: python tools/stress_screened.py [n_cases] [first_seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viquae_amd.index import MI355XFlatIndex


def case(seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    pick = lambda xs: xs[int(torch.randint(0, len(xs), (1,), generator=g, device="cuda"))]  # noqa: E731
    n = pick([5000, 40000, 70000, 200000, 333333, 600000])
    d = pick([32, 64, 100, 200, 256, 512, 700, 768, 1000, 1024])
    nq = pick([1, 100, 256, 300, 1024, 2500, 4096, 5000])
    k = pick([1, 10, 100, 128, 129, 160, 200, 224, 225, 256, 400, 512, 1000, 1792])  # beyond 224: row ranges, merged and proved
    regime = pick(["normal", "clustered", "dups", "scaled", "lowrank", "sorted", "l2norm", "shared", "shared", "shared_l2norm"])
    metric = pick([0, 0, 1])
    X = torch.randn((n, d), generator=g, device="cuda")
    Q = torch.randn((nq, d), generator=g, device="cuda")
    factory = "Flat"
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
    elif regime == "l2norm":
        factory = "L2norm,Flat"
    elif regime in ("shared", "shared_l2norm"):  # a common component 0.5 ... 6 x the isotropic part: the centred-query screen (round 5)
        ratio = float(pick([0.5, 1.0, 2.0, 4.0, 6.0]))
        mu = torch.randn((1, d), generator=g, device="cuda")
        mu = ratio * mu / mu.norm() * d ** 0.5
        X, Q = X + mu, Q + mu
        factory = "L2norm,Flat" if regime == "shared_l2norm" else "Flat"
    return X, Q, k, regime, factory, metric


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = big = 0
    t0 = time.time()
    for seed in range(first, first + n_cases):
        X, Q, k, regime, factory, metric = case(seed)
        a = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=True)
        a.add(X)
        b = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=False)
        b.add(X)
        D, I = a.search_device(Q, k)
        D0, I0 = b.search_device(Q, k)
        ok = torch.equal(I, I0) and torch.equal(D, D0)
        st = a.screen_stats(Q.shape[0], k) if k <= 224 else (-1, 0)  # beyond 224 the workspace holds the last row range's search
        big += k > 224 and a.scan_kind(Q.shape[0], k) != "none"
        print(f"seed {seed:4d} {'L2' if metric else 'IPc' if a._screen_metric == 2 else 'IP'} {regime:13s} N={X.shape[0]:6d} d={X.shape[1]:3d} nq={Q.shape[0]:4d} k={k:3d} "
              f"{'ok ' if ok else 'MISMATCH'} exact-recomputed tiles {st[0]} cand/query {st[1] / max(1, min(Q.shape[0], 4096)):.0f}", flush=True)
        bad += not ok
        del a, b
        torch.cuda.empty_cache()
    print(f"{n_cases} cases ({big} with k > 224 served over row ranges), {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
