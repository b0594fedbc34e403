"""When do the workgroups of the eight-wave GEMM reach their C-store epilogue?  Needs a -DXW_ABL=512 build (tools/ab_build.sh xw_stamp
"-DXW_ABL=512"; MEERQAT_HIP_LIB=ab/lib_xw_stamp.so): an EPI_NONE launch then takes `bias` as a debug buffer and stamps s_memtime at the
start and the end of every tile's epilogue.  Prints, per tile number, the spread of the start times over the 256 workgroups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from viquae_amd import _lib, encoders as E

M, K, N = 204800, 768, 2304
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn((M, K), generator=g, device="cuda") * 0.5
w = torch.randn((N, K), generator=g, device="cuda") * 0.05
asp, wsp = E.SplitAct(*E.split_bf16(a)), E.split_bf16_tiled(w)
for stagger in [int(x) for x in sys.argv[1:]] or [0]:
    _lib.load().mq_gemm_set_option(_lib.GEMM_OPT_STAGGER, stagger)
    dbg = torch.zeros((256 * 64 * 2,), dtype=torch.int64, device="cuda")
    for _ in range(2):
        E.gemm_nt(asp, w, dbg.view(torch.float32), None, E.EPI_NONE, wsplit=wsp)
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(256, 64, 2).astype(np.float64)
    t0 = t[:, 0, 0].min()
    print(f"stagger {stagger}: ticks relative to the earliest first epilogue; per tile number: min / median / max start, median duration")
    for it in (0, 1, 2, 5, 10, 20, 27):
        s, d = t[:, it, 0] - t0, t[:, it, 1] - t[:, it, 0]
        ok = t[:, it, 0] > 0
        print(f"  tile {it:2d}: start {s[ok].min():9.0f} {np.median(s[ok]):9.0f} {s[ok].max():9.0f}   spread {s[ok].max() - s[ok].min():8.0f}   epilogue {np.median(d[ok]):7.0f} (max {d[ok].max():7.0f})   period {np.median(t[ok, it, 0] - t[ok, max(it - 1, 0), 0]):8.0f}")
