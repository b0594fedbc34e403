"""Randomised check of the implicit-GEMM convolution kernels (csrc/conv.hip) on the GPU: random batch / image sizes (odd, tiny, wider
than a tile), channel counts, strides, every tile shape that divides the channel count and the patch kernel, both epilogues --
against mq_im2col_split_f32 + mq_gemm_nt_bf16x3s_f32 on the same pairs.  Tap-major K order: bit for bit; channel-major (and the patch
kernel, which shares its order): bit for bit among themselves and within fp32 rounding of the explicit path.
usage: python tools/stress_conv.py [n_cases] [first_seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viquae_amd import _lib
from viquae_amd.arcface import ArcFaceR50
from viquae_amd.encoders import EPI_BIAS, EPI_BIAS_RESIDUAL, SplitAct, gemm_nt, split_bf16_tiled

TILES = {1: 256, 2: 128, 3: 128, 4: 64}


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib = _lib.load()
    zeros = torch.zeros(64, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    bad, t0, patch_cases = 0, time.time(), 0
    for seed in range(first, first + n_cases):
        g = torch.Generator(device="cuda").manual_seed(seed)
        pick = lambda xs: xs[int(torch.randint(0, len(xs), (1,), generator=g, device="cuda"))]  # noqa: E731
        B, H, W = pick([1, 2, 3, 5, 9]), pick([1, 2, 5, 7, 14, 23, 28, 56]), pick([1, 3, 6, 7, 14, 28, 57, 112, 130])
        C, N, stride = pick([32, 64, 96, 128, 256]), pick([64, 128, 192, 256, 512]), pick([1, 1, 2])
        x = torch.randn((B, H, W, C), generator=g, device="cuda")
        w2 = torch.randn((N, 9 * C), generator=g, device="cuda") * 0.05
        bias, slope = torch.randn(N, generator=g, device="cuda"), torch.rand(N, generator=g, device="cuda") * 0.4
        scale, shift = torch.rand(N, generator=g, device="cuda") + 0.5, torch.randn(N, generator=g, device="cuda")
        ws = split_bf16_tiled(w2)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        M = B * Ho * Wo
        res = torch.randn((M, N), generator=g, device="cuda")
        A = ArcFaceR50._im2col(x, B, H, W, C, False, 3, 3, stride, 1, 9 * C, None, None, None)
        y_plain = gemm_nt(A, w2, bias=bias, epilogue=EPI_BIAS, wsplit=ws)
        y_res = gemm_nt(A, w2, bias=bias, residual=res, epilogue=EPI_BIAS_RESIDUAL, wsplit=ws)
        want_prelu = ArcFaceR50._im2col(y_plain.view(B, Ho, Wo, N), B, Ho, Wo, N, False, 1, 1, 1, 0, N, slope, None, None).rowmajor()
        want_aff = ArcFaceR50._im2col(y_res.view(B, Ho, Wo, N), B, Ho, Wo, N, False, 1, 1, 1, 0, N, None, scale, shift).rowmajor()
        xin = ArcFaceR50._im2col(x, B, H, W, C, False, 1, 1, 1, 0, C, None, None, None)

        def run(tile, prelu):
            P = SplitAct.empty(M, N, x.device)
            Y = None if prelu else torch.zeros((M, N), device="cuda")
            _lib.check(lib.mq_conv3x3_pair_f32(xin.hi.data_ptr(), xin.lo.data_ptr(), B, H, W, C, stride, ws[0].data_ptr(), ws[1].data_ptr(), N,
                                               bias.data_ptr(), slope.data_ptr() if prelu else None, None if prelu else res.data_ptr(),
                                               None if prelu else scale.data_ptr(), None if prelu else shift.data_ptr(),
                                               None if prelu else Y.data_ptr(), P.hi.data_ptr(), P.lo.data_ptr(), zeros.data_ptr(), tile, st),
                       "mq_conv3x3_pair_f32")
            return Y, P.rowmajor()

        ok = True
        tiles = [t for t, nt in TILES.items() if N % nt == 0] + [0]
        ref_c = None
        for tile in tiles:
            _, p = run(tile, True)
            y, p2 = run(tile, False)
            ok &= all(torch.equal(a, b) for a, b in zip(p, want_prelu)) and torch.equal(y, y_res)
            ok &= all(torch.equal(a, b) for a, b in zip(p2, want_aff))
            yc, _ = run(tile | 0x100, False)
            ok &= bool((yc - y_res).abs().max() <= 1e-5 * float(y_res.abs().max()))
            ok &= ref_c is None or torch.equal(yc, ref_c)
            ref_c = yc
        if stride == 1 and W <= 127:
            patch_cases += 1
            yp, pp = run(5 | 0x100, False)
            _, pc = run(tiles[0] | 0x100, False)
            ok &= torch.equal(yp, ref_c) and all(torch.equal(a, b) for a, b in zip(pp, pc))
            _, pq = run(5 | 0x100, True)
            _, pr = run(tiles[0] | 0x100, True)
            ok &= all(torch.equal(a, b) for a, b in zip(pq, pr))
        print(f"seed {seed} B={B} {H}x{W} C={C} N={N} stride={stride} tiles={tiles} {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
    print(f"{n_cases} cases ({patch_cases} also through the patch kernel), {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
