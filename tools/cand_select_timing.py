"""Where cand_select_kernel's time goes at one query tile (needs a -DMQ_TIMING build: MEERQAT_HIP_LIB=ab/lib_timing.so)."""
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
dbg = torch.zeros(32768 + 256 * 8, dtype=torch.int64, device=dev)  # the scan kernels' slots, then cand_select's (CS_DBG_AT)
os.environ["MQ_DBG_PTR"] = str(dbg.data_ptr())
for _ in range(3):
    idx.search_device(Q, k)
torch.cuda.synchronize()
dbg.zero_()
idx.search_device(Q, k); torch.cuda.synchronize()
t = dbg[32768:32768 + nq * 8].view(nq, 8).double()
names = ["stripe-maxima bound", "pool walk (collect)", "bisection + second walk", "k-th of the block", "compaction + rows out"]
print("cand_select, %d queries: first start -> last end %.1f us; per workgroup %.1f us mean, %.1f max; keys collected %.0f mean %.0f max" % (
    nq, (t[:, 5].max() - t[:, 0].min()) / 100, (t[:, 5] - t[:, 0]).mean() / 100, (t[:, 5] - t[:, 0]).max() / 100, t[:, 6].mean(), t[:, 6].max()))
for i, n in enumerate(names):
    print("  %-26s %.2f us" % (n, (t[:, i + 1] - t[:, i]).mean() / 100))
