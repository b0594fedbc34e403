"""The one-query-tile streaming scan (csrc/knn_small.inc) against the 256 x 256 tile kernel, same process, same index:
results must be equal bit for bit; ms per search for both.
python3 tools/small_scan_check.py [rows] [d] [metric] [nq list, comma separated] [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viquae_amd.index import MI355XFlatIndex

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_500_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
metric = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nqs = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "256,200,64,20").split(",")]
k = int(sys.argv[5]) if len(sys.argv) > 5 else 100
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
idx = MI355XFlatIndex(string_factory="Flat", metric_type=metric)
for s in range(0, rows, 1 << 16):
    idx.add(torch.randn((min(1 << 16, rows - s), d), generator=g, device=dev), total_hint=rows)


def run(Q, small, steps):
    from viquae_amd import _lib
    _lib.load().mq_knn_set_option(_lib.KNN_OPT_SMALL_SCAN, 1 if small else 0)
    D, I = idx.search_device(Q, k)
    torch.cuda.synchronize()
    st = idx.screen_stats(Q.shape[0], k)
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.search_device(Q, k)
    torch.cuda.synchronize()
    return D.clone(), I.clone(), (time.perf_counter() - t0) / max(steps, 1), st


bad = 0
for nq in nqs:
    Q = torch.randn((nq, d), generator=g, device=dev)
    for rep in range(2):
        D1, I1, t1, st1 = run(Q, True, 30)
        D0, I0, t0, st0 = run(Q, False, 30)
        same = bool(torch.equal(D0, D1) and torch.equal(I0, I1))
        bad += not same
        print(f"nq={nq} rows={rows} d={d} metric={metric}: small {t1 * 1e3:.3f} ms (flagged {st1[0]}, cand {st1[1]}, maxpool {st1[3]}, keys {st1[4]}) | "
              f"tile {t0 * 1e3:.3f} ms (flagged {st0[0]}, cand {st0[1]}, maxpool {st0[3]}, keys {st0[4]}) | equal {same}", flush=True)
sys.exit(1 if bad else 0)
