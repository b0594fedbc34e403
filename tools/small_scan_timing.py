"""Cycle accounting of screen_small_kernel / screen_small8_kernel (needs a -DMQ_TIMING build: MEERQAT_HIP_LIB=ab/lib_timing.so; NQ, D,
MQ_KNN_SMALL_WAVES = 4 | 8 from the environment)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viquae_amd.index import MI355XFlatIndex

N, d, nq, k = 1_500_000, int(os.environ.get("D", 768)), int(os.environ.get("NQ", 256)), 100
dev = torch.device("cuda")
idx = MI355XFlatIndex(string_factory="Flat", metric_type=0)
g = torch.Generator(device=dev); g.manual_seed(0)
for s in range(0, N, 1 << 16):
    idx.add(torch.randn((min(1 << 16, N - s), d), generator=g, device=dev), total_hint=N)
Q = torch.randn((nq, d), generator=g, device=dev)
NW = int(os.environ.get('MQ_KNN_SMALL_WAVES', '8'))
dbg = torch.zeros(32768 + 256 * 8, dtype=torch.int64, device=dev)  # (+ cand_select's stamps behind the scan's slots in timing builds)
os.environ["MQ_DBG_PTR"] = str(dbg.data_ptr())
for _ in range(3):
    idx.search_device(Q, k)
torch.cuda.synchronize()
dbg.zero_()
idx.search_device(Q, k); torch.cuda.synchronize()
t = dbg[:256 * NW * 8].view(256, NW, 8).double()
cyc = t[..., :6].sum(-1)
rt = t[..., 6]
print("nq %d variant %s: per-wave cycles mean %.3e max %.3e; kernel wall %.1f us (100 MHz ticks) -> clock %.2f GHz" %
      (nq, os.environ.get("MQ_SMALL_VARIANT", "0"), cyc.mean(), cyc.max(), rt.mean() / 100.0, cyc.mean() / (rt.mean() * 10.0)))
names = ["wait item", "barrier", "bounds", "selection", "end of warm-up" if NW == 4 else "DMA issue", "MFMA phases"]
items = 1_500_000 / 32 / 256 + 2
print(" cycles per item and wave: %.0f" % (cyc.mean() / items))
if NW == 8:
    we = t[:, 0, 7]
    print(" warm-up items per slab: mean %.1f min %d max %d" % (we.mean(), we.min(), we.max()))
for w in range(NW):
    print(" wave %d: " % w + "  ".join(f"{n} {t[:, w, i].mean() / items:.0f}" for i, n in enumerate(names)))
