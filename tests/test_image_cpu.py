"""CPU suite: the image-preprocessing oracle (oracle/image.py) against the golden vectors minted from Pillow +
transformers (tools/make_golden_image.py), against Pillow / transformers themselves when importable, and the host
half of the C ABI (mq_image_plan: pure host arithmetic, needs no GPU)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONFIGS = {
    "default64": dict(size=64, crop=64),
    "bilinear48": dict(size=48, crop=40, kind=2),
    "exact": dict(size=(50, 70), crop=(44, 60)),
    "raw": dict(size=32, crop=32, do_normalize=False),
    "noscale": dict(size=32, crop=32, do_rescale=False),
    "noresize": dict(do_resize=False, crop=30),
}


def _clip_inputs(z, name):
    ims = [z[f"in_{c}"] for c in range(5)]
    return ims if name != "noresize" else [im for im in ims if min(im.shape[:2]) >= 30]


def test_oracle_resize_equals_pillow_goldens():
    from oracle import image as oi
    z = np.load(os.path.join(GOLD, "image_resize.npz"))
    n = 0
    while f"in_{n}" in z:
        im = z[f"in_{n}"]
        for kind in (2, 3):
            want = z[f"out_{n}_k{kind}"]
            got = oi.resize_u8(im, want.shape[0], want.shape[1], kind)
            assert np.array_equal(got, want), (n, kind)
            assert oi.resized_size(im.shape[0], im.shape[1], 32) == want.shape[:2]
        n += 1
    assert n >= 10


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_oracle_clip_preprocess_equals_hf_goldens(name):
    from oracle import image as oi
    z = np.load(os.path.join(GOLD, "image_clip.npz"))
    want = z[f"pixel_values_{name}"]
    got = oi.clip_preprocess(_clip_inputs(z, name), **CONFIGS[name])
    assert got.dtype == want.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got, want)  # bit for bit


def test_oracle_equals_live_pillow_on_random_geometries():
    Image = pytest.importorskip("PIL.Image")
    from oracle import image as oi
    rng = np.random.default_rng(7)
    for _ in range(40):
        h, w = (int(v) for v in rng.integers(1, 400, 2))
        oh, ow = (int(v) for v in rng.integers(1, 300, 2))
        kind = int(rng.integers(2, 4))
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = np.array(Image.fromarray(im).resize((ow, oh), resample=kind))
        assert np.array_equal(oi.resize_u8(im, oh, ow, kind), want), (h, w, oh, ow, kind)


def test_plan_matches_the_oracle_geometry():
    """mq_image_plan: resized sizes (HF rule), crop origin, tap counts, 16-byte aligned packing, workspace totals."""
    import math
    from oracle import image as oi
    from viquae_amd import _lib
    lib = _lib.load()
    sizes = np.array([[300, 200], [224, 224], [225, 1000], [37, 53], [1, 1], [4000, 3000], [223, 500]], dtype=np.int64)
    geom = np.zeros((len(sizes), 12), dtype=np.int64)
    totals = np.zeros(5, dtype=np.int64)
    assert lib.mq_image_plan(sizes.ctypes.data, len(sizes), 1, 224, 0, 224, 224, 3, geom.ctypes.data, totals.ctypes.data) == 0
    off = 0
    for (h, w), g in zip(sizes, geom):
        oh, ow = oi.resized_size(int(h), int(w), 224)
        assert (g[1], g[2], g[3], g[4]) == (h, w, oh, ow)
        assert (g[5], g[6]) == ((oh - 224) // 2, (ow - 224) // 2)
        assert g[0] == off and off % 16 == 0
        off += (h * w * 3 + 15) // 16 * 16
        for in_size, out_size, ks in ((w, ow, g[10]), (h, oh, g[11])):
            assert ks == oi.precompute_coeffs(int(in_size), int(out_size), 3)[1].shape[1]
    assert totals[0] == off and totals[2] == 4000 and totals[4] == 3000 and totals[1] > 0
    # images smaller than the crop window after resizing are refused (HF would zero-pad)
    assert lib.mq_image_plan(sizes.ctypes.data, len(sizes), 1, 100, 0, 224, 224, 3, geom.ctypes.data, totals.ctypes.data) == -4
    assert lib.mq_image_plan(sizes.ctypes.data, len(sizes), 1, 224, 0, 224, 224, 1, geom.ctypes.data, totals.ctypes.data) == -4  # lanczos
    bad = np.array([[0, 5]], dtype=np.int64)
    assert lib.mq_image_plan(bad.ctypes.data, 1, 1, 224, 0, 224, 224, 3, geom.ctypes.data, totals.ctypes.data) == -1
    assert math.isfinite(float(totals[1]))


def test_processor_config_parsing(tmp_path):
    """preprocessor_config.json of the legacy CLIPFeatureExtractor (ints) and of CLIPImageProcessor (dicts)."""
    import json
    from viquae_amd.data.loading import get_class_from_name
    from viquae_amd.image.preprocess import CLIPImageProcessorHIP
    assert get_class_from_name("CLIPFeatureExtractor") is CLIPImageProcessorHIP
    assert get_class_from_name("CLIPImageProcessor") is CLIPImageProcessorHIP
    legacy = {"crop_size": 224, "do_center_crop": True, "do_normalize": True, "do_resize": True, "feature_extractor_type":
              "CLIPFeatureExtractor", "image_mean": [0.48145466, 0.4578275, 0.40821073], "image_std": [0.26862954, 0.26130258,
              0.27577711], "resample": 3, "size": 224}
    (tmp_path / "preprocessor_config.json").write_text(json.dumps(legacy))
    p = CLIPImageProcessorHIP.from_pretrained(tmp_path)
    assert (p.size_mode, p.size_h, p.crop_h, p.crop_w, p.resample) == ("shortest", 224, 224, 224, 3)
    p = CLIPImageProcessorHIP(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336}, resample="bicubic")
    assert (p.size_mode, p.size_h, p.crop_h) == ("shortest", 336, 336)
    with pytest.raises(NotImplementedError):
        CLIPImageProcessorHIP(resample=1)
    with pytest.raises(NotImplementedError):
        CLIPImageProcessorHIP(do_center_crop=False)
    geom, totals = p.plan(np.array([[500, 400]]))
    assert geom[0, 3] == 420 and geom[0, 4] == 336
    if not __import__("torch").cuda.is_available():
        with pytest.raises(RuntimeError):
            p(np.zeros((400, 400, 3), np.uint8))  # no CPU path
