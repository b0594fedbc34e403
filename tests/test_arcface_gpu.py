"""GPU: ArcFace r50 (viquae_amd/arcface.py: convolutions as im2col + split-bf16 GEMMs, csrc/conv.hip) and the face alignment
against oracle/arcface.py, the numpy restatement of the PUBLISHED IResNet-50 / Umeyama / cv2.warpAffine -- parity unpinned against
the un-vendored originals (arcface_torch, scikit-image, OpenCV).  north_star's 1e-3 on the embeddings."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_im2col_pairs_against_numpy():
    """mq_im2col_split_f32: patches, PReLU and the BatchNorm affine on in-bounds elements only, zero padding, NCHW input, padded K."""
    from viquae_amd.arcface import ArcFaceR50
    rng = np.random.default_rng(0)
    for (B, H, W, C, k, stride, pad, nchw) in [(2, 9, 7, 16, 3, 1, 1, False), (3, 8, 8, 8, 3, 2, 1, False), (2, 6, 6, 24, 1, 2, 0, False),
                                               (2, 10, 10, 3, 3, 1, 1, True), (4, 7, 7, 32, 7, 1, 0, False)]:
        x = rng.standard_normal((B, C, H, W) if nchw else (B, H, W, C)).astype(np.float32)
        slope, scale, shift = (rng.uniform(0.1, 0.4, C).astype(np.float32), rng.uniform(0.5, 1.5, C).astype(np.float32),
                               rng.standard_normal(C).astype(np.float32))
        kpad = (k * k * C + 31) // 32 * 32
        dev = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
        A = ArcFaceR50._im2col(dev(x), B, H, W, C, nchw, k, k, stride, pad, kpad, dev(slope), dev(scale), dev(shift))
        hi, lo = A.rowmajor()
        got = (hi.view(torch.bfloat16).float() + lo.view(torch.bfloat16).float()).cpu().numpy()
        xh = x.transpose(0, 2, 3, 1) if nchw else x
        t = np.where(xh >= 0, xh, xh * slope) * scale + shift
        tp = np.pad(t, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        want = np.zeros((B, Ho, Wo, kpad), np.float32)
        for kh in range(k):
            for kw in range(k):
                want[..., (kh * k + kw) * C:(kh * k + kw + 1) * C] = tp[:, kh:kh + Ho * stride:stride, kw:kw + Wo * stride:stride]
        want = want.reshape(B * Ho * Wo, kpad)
        assert got.shape == want.shape
        assert np.allclose(got, want, rtol=2e-5, atol=1e-6), (B, H, W, C, k, stride, pad, nchw)  # hi + lo keeps 16 bits; fma vs mul+add
        assert not got[:, k * k * C:].any()


@pytest.mark.parametrize("tile", [1, 2, 3, 4])   # MQ_CONV_TILE_256x256, 512x128, 256x128, 512x64
@pytest.mark.parametrize("B,H,W,C,N,stride", [(2, 9, 7, 32, 256, 1), (3, 8, 8, 64, 256, 2), (1, 23, 17, 96, 256, 1),
                                               (5, 14, 14, 128, 512, 2), (2, 56, 56, 64, 256, 1)])
def test_implicit_conv3x3_equals_im2col_plus_gemm_bit_for_bit(tile, B, H, W, C, N, stride):
    """mq_conv3x3_pair_f32 (the GEMM's LDS-DMA gathers the patches; PReLU / residual / the next BatchNorm in the epilogue) against
    mq_im2col_split_f32 + mq_gemm_nt_bf16x3s_f32 on the same pairs: same products, same order -> the same bits, for every tile
    shape, ragged row counts, stride 2, odd image sizes, borders."""
    from viquae_amd import _lib
    from viquae_amd.arcface import ArcFaceR50
    from viquae_amd.encoders import EPI_BIAS, EPI_BIAS_RESIDUAL, SplitAct, gemm_nt, split_bf16_tiled
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + H * 10 + C + tile)
    x = torch.randn((B, H, W, C), generator=g, device="cuda")
    w2 = torch.randn((N, 9 * C), generator=g, device="cuda") * 0.05
    bias = torch.randn(N, generator=g, device="cuda")
    slope = torch.rand(N, generator=g, device="cuda") * 0.4
    scale = torch.rand(N, generator=g, device="cuda") + 0.5
    shift = torch.randn(N, generator=g, device="cuda")
    ws = split_bf16_tiled(w2)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    M = B * Ho * Wo
    res = torch.randn((M, N), generator=g, device="cuda")
    zeros = torch.zeros(64, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    # the explicit path
    A = ArcFaceR50._im2col(x, B, H, W, C, False, 3, 3, stride, 1, 9 * C, None, None, None)
    y_plain = gemm_nt(A, w2, bias=bias, epilogue=EPI_BIAS, wsplit=ws)
    y_res = gemm_nt(A, w2, bias=bias, residual=res, epilogue=EPI_BIAS_RESIDUAL, wsplit=ws)
    want_prelu = ArcFaceR50._im2col(y_plain.view(B, Ho, Wo, N), B, Ho, Wo, N, False, 1, 1, 1, 0, N, slope, None, None)
    want_aff = ArcFaceR50._im2col(y_res.view(B, Ho, Wo, N), B, Ho, Wo, N, False, 1, 1, 1, 0, N, None, scale, shift)
    # the implicit path, from the pair of x
    xin = ArcFaceR50._im2col(x, B, H, W, C, False, 1, 1, 1, 0, C, None, None, None)
    P = SplitAct.empty(M, N, x.device)
    P.hi.zero_(), P.lo.zero_()
    _lib.check(lib.mq_conv3x3_pair_f32(xin.hi.data_ptr(), xin.lo.data_ptr(), B, H, W, C, stride, ws[0].data_ptr(), ws[1].data_ptr(), N,
                                       bias.data_ptr(), slope.data_ptr(), None, None, None, None, P.hi.data_ptr(), P.lo.data_ptr(),
                                       zeros.data_ptr(), tile, st), "mq_conv3x3_pair_f32")
    for got, want in zip(P.rowmajor(), want_prelu.rowmajor()):
        assert torch.equal(got, want)
    Y = torch.zeros((M, N), device="cuda")
    P2 = SplitAct.empty(M, N, x.device)
    _lib.check(lib.mq_conv3x3_pair_f32(xin.hi.data_ptr(), xin.lo.data_ptr(), B, H, W, C, stride, ws[0].data_ptr(), ws[1].data_ptr(), N,
                                       bias.data_ptr(), None, res.data_ptr(), scale.data_ptr(), shift.data_ptr(), Y.data_ptr(),
                                       P2.hi.data_ptr(), P2.lo.data_ptr(), zeros.data_ptr(), tile, st), "mq_conv3x3_pair_f32")
    assert torch.equal(Y, y_res)
    for got, want in zip(P2.rowmajor(), want_aff.rowmajor()):
        assert torch.equal(got, want)
    # K walked as (channel block, tap): the same products, another order
    Y4 = torch.zeros((M, N), device="cuda")
    _lib.check(lib.mq_conv3x3_pair_f32(xin.hi.data_ptr(), xin.lo.data_ptr(), B, H, W, C, stride, ws[0].data_ptr(), ws[1].data_ptr(), N,
                                       bias.data_ptr(), None, res.data_ptr(), None, None, Y4.data_ptr(), None, None, zeros.data_ptr(),
                                       tile | 0x100, st), "mq_conv3x3_pair_f32")
    assert (Y4 - y_res).abs().max() <= 1e-5 * float(y_res.abs().max())
    # the patch kernel (stride 1): the taps of a channel block read one LDS-resident patch -- the bits of the channel-major gather
    if stride == 1 and W <= 127:
        Y5 = torch.zeros((M, N), device="cuda")
        P5 = SplitAct.empty(M, N, x.device)
        _lib.check(lib.mq_conv3x3_pair_f32(xin.hi.data_ptr(), xin.lo.data_ptr(), B, H, W, C, stride, ws[0].data_ptr(), ws[1].data_ptr(), N,
                                           bias.data_ptr(), None, res.data_ptr(), scale.data_ptr(), shift.data_ptr(), Y5.data_ptr(),
                                           P5.hi.data_ptr(), P5.lo.data_ptr(), zeros.data_ptr(), 5 | 0x100, st), "mq_conv3x3_pair_f32")
        assert torch.equal(Y5, Y4)
        P6 = SplitAct.empty(M, N, x.device)
        _lib.check(lib.mq_conv3x3_pair_f32(xin.hi.data_ptr(), xin.lo.data_ptr(), B, H, W, C, stride, ws[0].data_ptr(), ws[1].data_ptr(), N,
                                           bias.data_ptr(), slope.data_ptr(), None, None, None, None, P6.hi.data_ptr(), P6.lo.data_ptr(),
                                           zeros.data_ptr(), 5 | 0x100, st), "mq_conv3x3_pair_f32")
        P7 = SplitAct.empty(M, N, x.device)
        _lib.check(lib.mq_conv3x3_pair_f32(xin.hi.data_ptr(), xin.lo.data_ptr(), B, H, W, C, stride, ws[0].data_ptr(), ws[1].data_ptr(), N,
                                           bias.data_ptr(), slope.data_ptr(), None, None, None, None, P7.hi.data_ptr(), P7.lo.data_ptr(),
                                           zeros.data_ptr(), tile | 0x100, st), "mq_conv3x3_pair_f32")
        for got, want in zip(P6.rowmajor(), P7.rowmajor()):
            assert torch.equal(got, want)
    # no pair output requested: Y alone
    Y3 = torch.zeros((M, N), device="cuda")
    _lib.check(lib.mq_conv3x3_pair_f32(xin.hi.data_ptr(), xin.lo.data_ptr(), B, H, W, C, stride, ws[0].data_ptr(), ws[1].data_ptr(), N,
                                       bias.data_ptr(), None, res.data_ptr(), None, None, Y3.data_ptr(), None, None, zeros.data_ptr(), 0, st),
               "mq_conv3x3_pair_f32")
    assert torch.equal(Y3, y_res)


@pytest.mark.parametrize("M,N,K,nsplit", [(256, 512, 25088, 49), (37, 512, 25088, 7), (300, 96, 1024, 5), (8, 512, 25088, 1000)])
def test_split_k_gemm_against_the_one_pass_gemm(M, N, K, nsplit):
    """mq_gemm_nt_bf16x3s_splitk_f32: same products, partial sums combined in split order -- equal to the one-pass GEMM within fp32
    rounding of a K-long sum (and exactly equal with one split)."""
    from viquae_amd import _lib
    from viquae_amd.encoders import EPI_BIAS, SplitAct, gemm_nt, split_bf16, split_bf16_tiled
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn((M, K), generator=g, device="cuda")
    w = torch.randn((N, K), generator=g, device="cuda") * 0.02
    bias = torch.randn(N, generator=g, device="cuda")
    A = SplitAct(*split_bf16(a))
    ws = split_bf16_tiled(w)
    want = gemm_nt(A, w, bias=bias, epilogue=EPI_BIAS, wsplit=ws)
    st = torch.cuda.current_stream().cuda_stream
    for ns in (nsplit, 1):
        out = torch.zeros((M, N), device="cuda")
        part = torch.empty((min(ns, K // 32), M, N), device="cuda")
        _lib.check(lib.mq_gemm_nt_bf16x3s_splitk_f32(A.hi.data_ptr(), A.lo.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), bias.data_ptr(),
                                                     out.data_ptr(), M, N, K, 1, ns, part.data_ptr(), st), "mq_gemm_nt_bf16x3s_splitk_f32")
        if ns == 1:
            assert torch.equal(out, want)
        else:
            ref = a.double() @ w.double().T + bias.double()
            assert (out.double() - ref).abs().max() <= 2 * (want.double() - ref).abs().max() + 1e-6
            assert torch.allclose(out, want, rtol=0, atol=2e-4 * float(want.abs().max()))


@pytest.mark.parametrize("B,H,W", [(2, 12, 8), (3, 112, 112), (1, 6, 20)])
def test_stem_kernel_against_float64(B, H, W):
    """mq_stem_conv3x3_f32: conv (3 -> 64, 3 x 3, pad 1) + bias + PReLU, then the two pairs it writes: the affine of every pixel
    and the plain activation at even coordinates; fp32 fused multiply-adds against a float64 convolution."""
    import torch.nn.functional as F
    from viquae_amd import _lib
    from viquae_amd.encoders import SplitAct
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(B + H + W)
    x = torch.rand((B, 3, H, W), generator=g, device="cuda") * 2 - 1
    w = torch.randn((64, 3, 3, 3), generator=g, device="cuda") * 0.3
    bias, slope = torch.randn(64, generator=g, device="cuda"), torch.rand(64, generator=g, device="cuda") * 0.4
    scale, shift = torch.rand(64, generator=g, device="cuda") + 0.5, torch.randn(64, generator=g, device="cuda")
    wt = w.permute(2, 3, 1, 0).reshape(27, 64).contiguous()   # row (kh * 3 + kw) * 3 + c
    P, D = SplitAct.empty(B * H * W, 64, x.device), SplitAct.empty(B * (H // 2) * (W // 2), 64, x.device)
    _lib.check(lib.mq_stem_conv3x3_f32(x.data_ptr(), B, H, W, wt.data_ptr(), bias.data_ptr(), slope.data_ptr(), scale.data_ptr(),
                                       shift.data_ptr(), P.hi.data_ptr(), P.lo.data_ptr(), D.hi.data_ptr(), D.lo.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "mq_stem_conv3x3_f32")
    v = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    v = torch.where(v >= 0, v, v * slope.double()[None, :, None, None]).permute(0, 2, 3, 1)      # NHWC
    val = lambda sa: sum(t.view(torch.bfloat16).double() for t in sa.rowmajor())  # noqa: E731
    want_p = (v * scale.double() + shift.double()).reshape(-1, 64)
    want_d = v[:, ::2, ::2].reshape(-1, 64)
    tol = 2e-5 * float(v.abs().max())   # hi + lo keeps 16 mantissa bits
    assert (val(P) - want_p).abs().max() <= tol * float(scale.max()) and (val(D) - want_d).abs().max() <= tol
    # without the downsample operand
    P2 = SplitAct.empty(B * H * W, 64, x.device)
    _lib.check(lib.mq_stem_conv3x3_f32(x.data_ptr(), B, H, W, wt.data_ptr(), bias.data_ptr(), slope.data_ptr(), scale.data_ptr(),
                                       shift.data_ptr(), P2.hi.data_ptr(), P2.lo.data_ptr(), None, None,
                                       torch.cuda.current_stream().cuda_stream), "mq_stem_conv3x3_f32")
    assert all(torch.equal(a, b) for a, b in zip(P.rowmajor(), P2.rowmajor()))
    assert lib.mq_stem_conv3x3_f32(x.data_ptr(), B, H, 6, wt.data_ptr(), bias.data_ptr(), slope.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                   P.hi.data_ptr(), P.lo.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream) != 0  # W % 4


def test_conv3x3_argument_checks():
    from viquae_amd import _lib
    lib = _lib.load()
    t = torch.zeros(1 << 18, dtype=torch.int16, device="cuda")
    f = torch.zeros(1 << 14, device="cuda")
    p, q = t.data_ptr(), f.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    ok = dict(B=1, H=4, W=4, C=32, stride=1, N=64)
    call = lambda C=32, N=64, stride=1, slope=q, res=None, Y=None, tile=0: lib.mq_conv3x3_pair_f32(  # noqa: E731
        p, p, 1, 4, 4, C, stride, p, p, N, q, slope, res, None, None, Y, p, p, p, tile, st)
    assert call() == 0
    assert call(C=24) != 0 and call(N=96) != 0 and call(stride=3) != 0
    assert call(slope=None) != 0                 # neither PReLU nor residual + Y
    assert call(res=q, Y=q) != 0                 # PReLU form takes no residual
    assert call(tile=1) != 0 and call(tile=9) != 0  # 256-column tile on 64 channels; unknown tile
    torch.cuda.synchronize()
    assert ok


def test_implicit_and_im2col_forwards_agree(monkeypatch):
    from oracle import arcface as oa
    from viquae_amd.arcface import ArcFaceR50
    model = ArcFaceR50.from_state_dict(oa.seeded_state(2)).cuda()
    x = torch.from_numpy(np.random.default_rng(3).uniform(-1, 1, (5, 3, 112, 112)).astype(np.float32)).cuda()
    monkeypatch.setenv("MQ_ARCFACE_CONV", "im2col")
    a = model(x)
    # the default forward differs from the all-im2col one in the ORDER of the same fp32-class products (K walked by channel block,
    # split-K head) and in the stem (direct fp32 convolution instead of three bf16 products): equal within fp32 rounding
    monkeypatch.setenv("MQ_ARCFACE_CONV", "implicit")
    for order in ("tap", None):
        if order:
            monkeypatch.setenv("MQ_CONV_KORDER", order)
        else:
            monkeypatch.delenv("MQ_CONV_KORDER")
        c = model(x)
        assert (a - c).abs().max() <= 3e-5 * a.abs().max()


def test_arcface_r50_matches_the_oracle():
    from oracle import arcface as oa
    from viquae_amd.arcface import ArcFaceR50
    st = oa.seeded_state(0)
    x = np.random.default_rng(1).uniform(-1, 1, (8, 3, 112, 112)).astype(np.float32)
    want = oa.iresnet_forward(st, x)
    model = ArcFaceR50.from_state_dict(st).cuda()
    got = model(torch.from_numpy(x).cuda()).cpu().numpy()
    assert got.shape == (8, 512) and np.isfinite(got).all()
    assert np.abs(got - want).max() <= 1e-3, np.abs(got - want).max()
    # batch chunking gives the same rows
    model.chunk = 3
    again = model(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(again, got)
    with pytest.raises(ValueError):
        model(torch.zeros((1, 3, 64, 64), device="cuda"))


def test_arcface_r50_matches_the_torch_statement_golden():
    """The HIP forward against tests/golden/arcface_r50_8.npz -- IResNet-50 stated with torch-CPU's own conv2d / batch_norm /
    prelu / linear (tools/make_golden_arcface.py), NOT the numpy oracle: within 1e-3 (north_star's encoder tolerance) of the
    fp32 statement and of its float64 twin, and an order of magnitude closer to them than the reference's own fp16-autocast
    forward is (meerqat/image/face_recognition.py:55-56)."""
    import os
    from oracle import arcface as oa
    from viquae_amd.arcface import ArcFaceR50
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "arcface_r50_8.npz"))
    x = np.stack([oa.preprocess(f) for f in g["faces_u8"]])
    model = ArcFaceR50.from_state_dict(oa.seeded_state(int(g["seed_weights"]))).cuda()
    got = model(torch.from_numpy(x).cuda()).cpu().numpy()
    err32, err64 = np.abs(got - g["embeddings"]).max(), np.abs(got - g["embeddings_f64"]).max()
    assert err32 <= 1e-3 and err64 <= 1e-3, (err32, err64)
    assert err64 < 0.1 * float(g["fp16_autocast_max_abs_dev"]), (err64, float(g["fp16_autocast_max_abs_dev"]))


def test_face_alignment_kernel_is_the_oracles_warp_bit_for_bit():
    """mq_warp_affine_faces_f32 = cv2.warpAffine's published fixed-point bilinear arithmetic + ToTensor + Normalize: against
    oracle.arcface.warp_affine / preprocess on random images, faces near and across the image border (constant border 0)."""
    from oracle import arcface as oa
    from viquae_amd.image.face_recognition import SRC, SimilarityTransform, align_faces_device
    rng = np.random.default_rng(3)
    images = [rng.integers(0, 256, (h, w, 3)).astype(np.uint8) for h, w in ((240, 320), (375, 500), (97, 61))]
    faces, want = [], []
    tform = SimilarityTransform()
    for i, im in enumerate(images):
        h, w = im.shape[:2]
        for scale, (cx, cy), rot in ((1.6, (w * 0.5, h * 0.5), 0.2), (0.7, (w * 0.1, h * 0.2), -0.5), (3.0, (w * 0.9, h * 0.95), 1.0)):
            c, s_ = np.cos(rot) * scale, np.sin(rot) * scale
            lm = (SRC - 56.0) @ np.array([[c, s_], [-s_, c]], np.float32) + np.array([cx, cy], np.float32)
            lm = (lm + rng.normal(0, 0.5, lm.shape)).astype(np.float32)
            assert tform.estimate(lm, SRC)
            assert np.allclose(tform.params, oa.umeyama(lm, SRC))
            faces.append((i, tform.params[0:2, :].copy()))
            want.append(oa.preprocess(oa.warp_affine(im, tform.params[0:2, :])))
    got = align_faces_device(images, faces, "cuda").cpu().numpy()
    assert got.shape == (9, 3, 112, 112)
    assert np.array_equal(got, np.stack(want))


def test_compute_face_embedding_like_the_reference(tmp_path, monkeypatch):
    """meerqat/image/face_recognition.py:72-102 end to end on a small batch: None landmarks -> None, max_n_faces, regrouping per
    image; embeddings within 1e-3 of the oracle pipeline (align -> ToTensor / Normalize -> IResNet-50)."""
    from PIL import Image
    from oracle import arcface as oa
    from viquae_amd.arcface import ArcFaceR50
    from viquae_amd.data import loading
    from viquae_amd.image import face_recognition as fr
    rng = np.random.default_rng(5)
    monkeypatch.setattr(loading, "IMAGE_PATH", tmp_path)
    names, lms, imgs = [], [], []
    for i in range(5):
        im = rng.integers(0, 256, (200 + 10 * i, 260, 3)).astype(np.uint8)
        Image.fromarray(im).save(tmp_path / f"f{i}.png")
        names.append(f"f{i}.png")
        imgs.append(im)
        n_faces = (0, 1, 3, 2, 1)[i]
        lms.append(None if n_faces == 0 else [((oa.SRC * (0.8 + 0.3 * f)) + np.array([40.0 + 20 * f, 30.0], np.float32)).tolist() for f in range(n_faces)])
    st = oa.seeded_state(2)
    model = ArcFaceR50.from_state_dict(st).cuda()
    batch = {"image": names, "face_landmarks": lms}
    out = fr.compute_face_embedding(dict(batch), model, fr.get_pil_preprocessor(), fr.SimilarityTransform(), max_n_faces=2)
    emb = out["face_embedding"]
    assert emb[0] is None and [None if e is None else e.shape for e in emb[1:]] == [(1, 512), (2, 512), (2, 512), (1, 512)]
    for i in range(1, 5):
        x = np.stack([oa.preprocess(oa.align_face(imgs[i], np.array(lm, np.float32))) for lm in lms[i][:2]])
        want = oa.iresnet_forward(st, x)
        assert np.abs(emb[i] - want).max() <= 1e-3, (i, np.abs(emb[i] - want).max())
    none = fr.compute_face_embedding({"image": names[:1], "face_landmarks": [None]}, model, None, fr.SimilarityTransform())
    assert none["face_embedding"] == [None]
    # the one-face helper of the reference's signature
    face = fr.similarity_transform(Image.fromarray(imgs[1]), np.array(lms[1][0], np.float32), fr.SRC, fr.SimilarityTransform())
    assert np.array_equal(np.asarray(face), oa.align_face(imgs[1], np.array(lms[1][0], np.float32)))
    from viquae_amd import encoders
    assert encoders.ArcFaceR50 is ArcFaceR50


def test_pipelined_face_job_equals_the_serial_one(tmp_path, monkeypatch):
    """dataset_compute_face_embedding (meerqat/image/face_recognition.py:105-112) software-pipelined -- decode workers, JPEG scans
    finished on the GPU, alignment of batch i + 1 behind the ArcFace forward of batch i -- against the serial
    compute_face_embedding mapped over the same dataset: the same arrays bit for bit, None where the reference has None (no
    landmarks), None for a file that does not decode; `max_n_faces` arriving in map_kwargs like in the shipped config."""
    import datasets
    from PIL import Image
    from oracle import arcface as oa
    from viquae_amd.arcface import ArcFaceR50
    from viquae_amd.data import loading
    from viquae_amd.image import face_recognition as fr
    datasets.disable_progress_bars()
    rng = np.random.default_rng(9)
    img_dir = tmp_path / "img"
    img_dir.mkdir()
    monkeypatch.setattr(loading, "IMAGE_PATH", img_dir)
    yy, xx = np.mgrid[0:240, 0:300]
    names, lms = [], []
    for i in range(23):
        a = np.stack([128 + 90 * np.sin(xx / (7 + c + i)) * np.cos(yy / (5 + c)) + rng.normal(0, 10, xx.shape) for c in range(3)], axis=2)
        im = Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)[: 200 + 3 * (i % 7), : 260 + 5 * (i % 5)])
        ext, kw = (("jpg", dict(quality=88)), ("jpg", dict(quality=70, progressive=True)), ("png", {}), ("jpg", dict(subsampling=0)))[i % 4]
        im.save(img_dir / f"f{i}.{ext}", **kw)
        names.append(f"f{i}.{ext}")
        n_faces = (1, 0, 3, 2, 5, 1, 0)[i % 7]
        lms.append(None if n_faces == 0 else [((oa.SRC * (0.7 + 0.2 * f)) + np.array([30.0 + 25 * f, 20.0 + 3 * i], np.float32)).tolist() for f in range(n_faces)])
    data = open(img_dir / "f4.jpg", "rb").read()
    (img_dir / "cut.jpg").write_bytes(data[: len(data) // 2])     # opens, does not decode
    names[9], lms[9] = "cut.jpg", lms[4]
    names[13] = "missing.jpg"                                      # does not open
    ds_path = tmp_path / "ds"
    datasets.Dataset.from_dict({"image": names, "face_landmarks": lms}).save_to_disk(str(ds_path))
    model = ArcFaceR50.from_state_dict(oa.seeded_state(3)).cuda()
    monkeypatch.setattr(fr, "from_pretrained", lambda **kw: model)
    got = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MQ_EMBED_PIPELINE", flag)
        import shutil
        work = tmp_path / f"ds{flag}"
        shutil.copytree(ds_path, work)
        with pytest.warns(UserWarning):
            out = fr.dataset_compute_face_embedding(str(work), map_kwargs={"max_n_faces": 2, "batch_size": 6})
        assert (fr.dataset_compute_face_embedding.last_pipeline_stats is not None) == (flag == "1")
        got[flag] = out["face_embedding"]
    assert fr.dataset_compute_face_embedding.last_pipeline_stats is None
    for i, (a, b) in enumerate(zip(got["1"], got["0"])):
        if lms[i] is None or i in (9, 13):
            assert a is None and b is None, i
        else:
            a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
            assert a.shape == (min(2, len(lms[i])), 512) and np.array_equal(a, b), i
