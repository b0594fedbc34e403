"""Cycle accounting of knn_scan_kernel / screen_scan_kernel (SCREEN=1) (needs a -DMQ_TIMING build: MEERQAT_HIP_LIB=ab/lib_timing.so;
NQ, D from the environment; with NQ <= 256 and D <= 768 export MQ_KNN_SMALL=0 BEFORE the library loads -- the switches are read once -- to keep the tile kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viquae_amd import _lib
from viquae_amd.index import MI355XFlatIndex

N, d, nq, k = 1_500_000, int(os.environ.get("D", 768)), int(os.environ.get("NQ", 4096)), 100
dev = torch.device("cuda")
idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=os.environ.get("SCREEN", "0") == "1")
g = torch.Generator(device=dev); g.manual_seed(0)
for s in range(0, N, 1 << 16):
    idx.add(torch.randn((min(1 << 16, N - s), d), generator=g, device=dev), total_hint=N)
Q = torch.randn((nq, d), generator=g, device=dev)
SCREEN = os.environ.get("SCREEN", "0") == "1"
NS = 8 if SCREEN else 4
dbg = torch.zeros(32768 + 256 * 8, dtype=torch.int64, device=dev)  # (+ cand_select's stamps behind the scan's slots in timing builds)
os.environ["MQ_DBG_PTR"] = str(dbg.data_ptr())
idx.search_device(Q, k); torch.cuda.synchronize()
dbg.zero_()
idx.search_device(Q, k); torch.cuda.synchronize()
t = dbg[:256 * 16 * NS].view(256, 16, NS).double()
tot = t.sum(-1)
print("per-wave total cycles: mean %.3e min %.3e max %.3e" % (tot.mean(), tot.min(), tot.max()))
names = (["K loop issue", "MFMA tail wait", "refresh", "sub-tile tests+appends", "(unused)", "counters+flags", "flag check+compaction", "warm poll+refresh"] if SCREEN
         else ["K loop", "scan+append", "barrier", "compaction"])
for i, name in enumerate(names):
    print(f"{name:12s} mean {t[..., i].mean():.3e} ({100 * t[..., i].mean() / tot.mean():.2f} %)  max-wave {t[..., i].max():.3e}")
