"""GPU: mq_image_preprocess_u8 (csrc/image.hip) through CLIPImageProcessorHIP must reproduce Pillow's 8-bit resize +
transformers' crop / rescale / normalise BIT FOR BIT: against the goldens minted from Pillow + transformers, against
the oracle on seeded geometries (big down-scales, up-scales, identity, extreme aspect ratios, 1-pixel sources), and
through the reference's embed() seam on image files."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HF_CONFIGS = {
    "default64": dict(size={"shortest_edge": 64}, crop_size={"height": 64, "width": 64}),
    "bilinear48": dict(size={"shortest_edge": 48}, crop_size={"height": 40, "width": 40}, resample=2),
    "exact": dict(size={"height": 50, "width": 70}, crop_size={"height": 44, "width": 60}),
    "raw": dict(size={"shortest_edge": 32}, crop_size={"height": 32, "width": 32}, do_normalize=False),
    "noscale": dict(size={"shortest_edge": 32}, crop_size={"height": 32, "width": 32}, do_rescale=False),
    "noresize": dict(do_resize=False, crop_size={"height": 30, "width": 30}),
}


@pytest.mark.parametrize("name", sorted(HF_CONFIGS))
def test_hip_equals_hf_goldens(name):
    from viquae_amd.image.preprocess import CLIPImageProcessorHIP
    z = np.load(os.path.join(GOLD, "image_clip.npz"))
    ims = [z[f"in_{c}"] for c in range(5)]
    if name == "noresize":
        ims = [im for im in ims if min(im.shape[:2]) >= 30]
    got = CLIPImageProcessorHIP(**HF_CONFIGS[name])(ims, return_tensors="pt")["pixel_values"]
    assert got.is_cuda and got.dtype.is_floating_point
    assert np.array_equal(got.cpu().numpy(), z[f"pixel_values_{name}"])


def test_hip_resize_equals_pillow_goldens():
    """do_rescale=False, do_normalize=False, crop = the whole resized image: the float output IS Pillow's uint8 image."""
    from viquae_amd.image.preprocess import CLIPImageProcessorHIP
    z = np.load(os.path.join(GOLD, "image_resize.npz"))
    n = 0
    while f"in_{n}" in z:
        for kind in (2, 3):
            want = z[f"out_{n}_k{kind}"]
            p = CLIPImageProcessorHIP(size={"height": want.shape[0], "width": want.shape[1]}, resample=kind, do_rescale=False,
                                      do_normalize=False, crop_size={"height": want.shape[0], "width": want.shape[1]})
            got = p([z[f"in_{n}"]])["pixel_values"][0].permute(1, 2, 0).cpu().numpy()
            assert np.array_equal(got, want.astype(np.float32)), (n, kind)
        n += 1


SIZES = [(300, 200), (224, 224), (640, 480), (225, 1000), (1000, 225), (2000, 1500), (100, 80), (37, 53), (1, 1), (2, 300),
         (223, 225), (448, 448), (449, 447), (3000, 224), (224, 3000), (500, 375), (375, 500), (60, 9000)]


@pytest.mark.parametrize("kind", [3, 2])
def test_hip_equals_oracle_on_a_ragged_batch(kind):
    from oracle import image as oi
    from viquae_amd.image.preprocess import CLIPImageProcessorHIP
    rng = np.random.default_rng(kind)
    ims = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in SIZES]
    ims[3][:, 300:600] = (255, 0, 255)   # saturated blocks: the cubic overshoot is clipped
    ims[5][500:900] = 0
    got = CLIPImageProcessorHIP(resample=kind)(ims)["pixel_values"].cpu().numpy()
    want = oi.clip_preprocess(ims, kind=kind)
    assert got.shape == (len(SIZES), 3, 224, 224)
    assert np.array_equal(got, want)


def test_crop_windows_wider_than_a_workgroup_and_odd_widths():
    """crop 300 x 333 (more columns than the row pass has threads; 999-byte rows are not dword aligned)."""
    from oracle import image as oi
    from viquae_amd.image.preprocess import CLIPImageProcessorHIP
    rng = np.random.default_rng(12)
    ims = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in [(400, 600), (333, 350), (900, 700)]]
    p = CLIPImageProcessorHIP(size={"shortest_edge": 340}, crop_size={"height": 300, "width": 333})
    got = p(ims)["pixel_values"].cpu().numpy()
    assert np.array_equal(got, oi.clip_preprocess(ims, size=340, crop=(300, 333)))


def test_pil_inputs_modes_and_empty_batch():
    Image = pytest.importorskip("PIL.Image")
    from oracle import image as oi
    from viquae_amd.image.preprocess import CLIPImageProcessorHIP
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    grey = Image.fromarray(rng.integers(0, 256, (70, 60), dtype=np.uint8), mode="L")
    p = CLIPImageProcessorHIP(size=48, crop_size=48)
    got = p([Image.fromarray(rgb), grey, np.ascontiguousarray(rgb[::-1])[::-1]])["pixel_values"].cpu().numpy()
    want = oi.clip_preprocess([rgb, np.asarray(grey.convert("RGB")), rgb], size=48, crop=48)
    assert np.array_equal(got, want)
    assert p([])["pixel_values"].shape == (0, 3, 48, 48)
    with pytest.raises(ValueError):
        p([np.zeros((4, 4), np.uint8)])
    from viquae_amd._lib import MeerqatHipError
    with pytest.raises(MeerqatHipError):
        CLIPImageProcessorHIP(size=20, crop_size=48)([rgb])   # crop larger than the resized image


def test_image_embed_seam_runs_the_device_transform(tmp_path, monkeypatch):
    """meerqat/image/embedding.py:125-166 with class names from vit_config.json: files -> PIL -> device transform ->
    CLIP tower; the embeddings equal the oracle tower applied to the oracle's (= Pillow + HF) pixel values."""
    import json
    from PIL import Image
    from safetensors.torch import save_file
    from oracle import encoders as oe, image as oi
    from viquae_amd.data import loading
    from viquae_amd.image import embedding as IE
    cfg = oe.CLIP_TINY
    state = oe.seeded_state(oe.clip_vision_param_shapes(cfg), 4)
    mdir = tmp_path / "clip"
    mdir.mkdir()
    import torch
    save_file({k: torch.from_numpy(v) for k, v in state.items()}, str(mdir / "model.safetensors"))
    (mdir / "config.json").write_text(json.dumps({"vision_config": dict(cfg), "projection_dim": cfg["projection_dim"]}))
    S = cfg["image_size"]
    (mdir / "preprocessor_config.json").write_text(json.dumps({"feature_extractor_type": "CLIPFeatureExtractor", "size": S, "crop_size": S,
                                                               "resample": 3, "do_resize": True, "do_center_crop": True, "do_normalize": True}))
    rng = np.random.default_rng(3)
    arrays = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in [(50, 70), (S, S), (90, 41)]]
    monkeypatch.setattr(loading, "IMAGE_PATH", tmp_path)
    for i, a in enumerate(arrays):
        Image.fromarray(a).save(tmp_path / f"im{i}.png")
    kw = IE.get_model_and_transform(model_kwargs={"type": "transformers", "class_name": "CLIPModel", "pretrained_model_name_or_path": str(mdir)},
                                    transform_kwargs={"class_name": "CLIPFeatureExtractor", "pretrained_model_name_or_path": str(mdir)})
    assert type(kw["transform"]).__name__ == "CLIPImageProcessorHIP"
    with pytest.warns(UserWarning):
        out = IE.embed({"image": ["im0.png", "nope.png", "im1.png", "im2.png"]}, save_as="clip", call="get_image_features", **kw)
    assert out["clip"][1] is None
    px = oi.clip_preprocess(arrays, size=S, crop=S)
    want = oe.clip_vision_forward(state, cfg, px)
    got = np.stack([out["clip"][i] for i in (0, 2, 3)])
    assert np.abs(got - want).max() < 1e-3


@pytest.mark.parametrize("seed", range(6))
def test_random_geometries_and_configs_fuzz(seed):
    """Random image sizes, shortest-edge / exact resize targets, square and non-square crop windows, both filters, in one
    ragged batch per seed: bit-equal to the Pillow / transformers restatement."""
    from oracle import image as oi
    from viquae_amd.image.preprocess import CLIPImageProcessorHIP
    rng = np.random.default_rng(1000 + seed)
    kind = int(rng.integers(2, 4))
    if seed % 2 == 0:
        size = int(rng.integers(16, 260))
        crop = (int(rng.integers(1, size + 1)), int(rng.integers(1, size + 1)))
        hip_size, o_size = {"shortest_edge": size}, size
    else:
        size = (int(rng.integers(8, 200)), int(rng.integers(8, 300)))
        crop = (int(rng.integers(1, size[0] + 1)), int(rng.integers(1, size[1] + 1)))
        hip_size, o_size = {"height": size[0], "width": size[1]}, size
    ims = [rng.integers(0, 256, (int(h), int(w), 3), dtype=np.uint8) for h, w in rng.integers(1, 700, (9, 2))]
    ims.append(np.full((33, 47, 3), 255, np.uint8))   # saturated: overshoot clipping on every tap
    ims.append(np.zeros((5, 900, 3), np.uint8))
    p = CLIPImageProcessorHIP(size=hip_size, crop_size={"height": crop[0], "width": crop[1]}, resample=kind)
    got = p(ims)["pixel_values"].cpu().numpy()
    want = oi.clip_preprocess(ims, size=o_size, crop=crop, kind=kind)
    assert got.shape == want.shape and np.array_equal(got, want)
