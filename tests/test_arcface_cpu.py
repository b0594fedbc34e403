"""CPU: the oracle's restatement of the face path (oracle/arcface.py) and the product's host-side mirror of
``SimilarityTransform.estimate`` -- no GPU, no compute through the HIP library."""
import numpy as np


def test_umeyama_recovers_a_known_similarity_and_mirror_equals_oracle():
    from oracle import arcface as oa
    from viquae_amd.image.face_recognition import SRC, SimilarityTransform, _invert_affine
    rng = np.random.default_rng(0)
    th, sc, t = 0.37, 1.9, np.array([12.5, -7.25])
    R = sc * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    src = rng.uniform(0, 100, (5, 2))
    dst = src @ R.T + t
    T = oa.umeyama(src, dst)
    assert np.allclose(T[:2, :2], R, atol=1e-9) and np.allclose(T[:2, 2], t, atol=1e-8)
    tf = SimilarityTransform()
    assert tf.estimate(src.astype(np.float32), dst.astype(np.float32))
    assert np.allclose(tf.params, oa.umeyama(src.astype(np.float32), dst.astype(np.float32)))
    M = T[:2]
    Mi = _invert_affine(M)
    assert np.allclose(Mi, oa.invert_affine(M))
    assert np.allclose(np.vstack([Mi, [0, 0, 1]]) @ np.vstack([M, [0, 0, 1]]), np.eye(3), atol=1e-9)
    assert SRC.shape == (5, 2) and np.array_equal(SRC, oa.SRC)


def test_warp_affine_identity_and_border():
    from oracle import arcface as oa
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (112, 112, 3)).astype(np.uint8)
    M = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    assert np.array_equal(oa.warp_affine(img, M), img)                 # integer coordinates: weight 32768 on one tap
    shifted = oa.warp_affine(img, np.array([[1.0, 0.0, 10.0], [0.0, 1.0, 5.0]]))
    assert np.array_equal(shifted[5:, 10:], img[:-5, :-10]) and not shifted[:5].any() and not shifted[:, :10].any()   # border 0
    half = oa.warp_affine(img, np.array([[1.0, 0.0, 0.5], [0.0, 1.0, 0.0]]))
    want = ((img[:, :-1].astype(np.int64) + img[:, 1:].astype(np.int64)) * 16384 + 16384) >> 15
    assert np.array_equal(half[:, 1:], want.astype(np.uint8))
    tab = oa._bilinear_table()
    assert (tab.sum(-1) == 32768).all() and tab[0, 0].tolist() == [32768, 0, 0, 0]


def test_iresnet_oracle_shapes_and_determinism():
    from oracle import arcface as oa
    st = oa.seeded_state(0, layers=(1, 1, 1, 1))
    x = np.random.default_rng(2).uniform(-1, 1, (2, 3, 112, 112)).astype(np.float32)
    y = oa.iresnet_forward(st, x, layers=(1, 1, 1, 1))
    assert y.shape == (2, 512) and np.isfinite(y).all() and 0.1 < y.std() < 10
    assert np.array_equal(y, oa.iresnet_forward(oa.seeded_state(0, layers=(1, 1, 1, 1)), x, layers=(1, 1, 1, 1)))
    full = oa.seeded_state(0)
    assert {"conv1.weight", "prelu.weight", "layer3.13.conv2.weight", "layer4.0.downsample.1.running_var", "fc.bias", "features.weight"} <= set(full)
    assert "layer3.14.conv1.weight" not in full and full["fc.weight"].shape == (512, 25088)


def _golden():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "arcface_r50_8.npz"))
    u8 = g["faces_u8"]
    x = ((u8.transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255.0)) - np.float32(0.5)) / np.float32(0.5)
    return g, u8, np.ascontiguousarray(x)


def test_iresnet_oracle_equals_the_torch_statement():
    """tests/golden/arcface_r50_8.npz = IResNet-50 built from torch-CPU's conv2d / batch_norm / prelu / linear
    (tools/make_golden_arcface.py, run in the build container): a second statement of the published network, independent of
    the numpy arithmetic of oracle/arcface.py (im2col + sgemm, broadcast BN).  The restatement must agree with it to fp32
    rounding -- 2e-5 absolute on outputs of magnitude up to 6.8 (3e-6 relative; the two differ in summation order over up to
    25,088-term sums, and the torch statement itself is 8e-6 from its float64 twin)."""
    from oracle import arcface as oa
    g, u8, x = _golden()
    st = oa.seeded_state(int(g["seed_weights"]))
    got = oa.iresnet_forward(st, x)
    assert got.shape == g["embeddings"].shape == (8, 512)
    assert np.abs(got - g["embeddings"]).max() <= 2e-5, np.abs(got - g["embeddings"]).max()
    assert np.abs(got - g["embeddings_f64"]).max() <= 2e-5, np.abs(got - g["embeddings_f64"]).max()
    # the preprocessing of the golden's bytes is the oracle's own ToTensor + Normalize, bit for bit
    assert np.array_equal(np.stack([oa.preprocess(f) for f in u8]), x)
    # what the reference's fp16 autocast (meerqat/image/face_recognition.py:55-56, fp16=True) deviates by on the same faces:
    # two orders of magnitude above the fp32 statements' mutual distance, and above this build's 1e-3 gate
    dev = float(g["fp16_autocast_max_abs_dev"])
    assert 1e-3 < dev < 5e-2 and np.isclose(np.abs(g["embeddings_fp16_autocast"] - g["embeddings"]).max(), dev)
