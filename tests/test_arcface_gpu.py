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
