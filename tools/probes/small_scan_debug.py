import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
sys.path.insert(0, "tools")
import stress_small_scan as st
from viquae_amd import _lib
from viquae_amd.index import MI355XFlatIndex
for seed in [int(x) for x in sys.argv[1:]]:
    X, Q, k, regime, factory, form, metric, tie = st.case(seed)
    a = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=True, tie_order=tie, l2norm_form=form); a.add(X)
    b = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=False, tie_order=tie, l2norm_form=form); b.add(X)
    D, I = a.search_device(Q, k); torch.cuda.synchronize(); s1 = a.screen_stats(Q.shape[0], k)
    with _lib.knn_option(_lib.KNN_OPT_SMALL_SCAN, 0):   # the tile kernel (the switches are read through the C ABI, not the environment)
        Dt, It = a.search_device(Q, k); torch.cuda.synchronize(); s0 = a.screen_stats(Q.shape[0], k)
    D0, I0 = b.search_device(Q, k); torch.cuda.synchronize()
    badq = (~((I == I0).all(1) & (D.view(torch.int32) == D0.view(torch.int32)).all(1))).nonzero().flatten().tolist()
    print(seed, X.shape, Q.shape, k, regime, metric, tie, "stream stats", s1[:5], "tile stats", s0[:5], "tile ok", bool(torch.equal(It, I0)), "bad queries", badq[:20], len(badq))
    for q in badq[:2]:
        miss = [int(x) for x in I0[q].tolist() if x not in set(I[q].tolist())]
        print("  q", q, "missing ids", miss[:8], "rows mod 32:", [m % 32 for m in miss[:8]], "tile idx:", [m // 32 for m in miss[:8]], "slab of tile:", [(m // 32) * 256 // ((X.shape[0] + 31) // 32) for m in miss[:8]])
        print("   got", I[q][:6].tolist(), D[q][:4].tolist(), "want", I0[q][:6].tolist(), D0[q][:4].tolist())
