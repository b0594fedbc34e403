"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the image preprocessing in front of the
CLIP tower -- what ``transform(images, return_tensors="pt")`` computes in ``meerqat/image/embedding.py:141-152`` when
the transform is the ``CLIPFeatureExtractor`` of experiments/image_embedding/clip/vit_config.json:13-17.

The arithmetic lives in two un-vendored dependencies:
  * Pillow (reference requirement ``Pillow``; 12.2.0 installed here) -- ``Image.resize(size, BICUBIC)`` on an 8-bit
    RGB image = ``ImagingResample`` (src/libImaging/Resample.c): per axis a table of double-precision filter weights
    (``precompute_coeffs``), normalised, rounded to 22-bit fixed point (``normalize_coeffs_8bpc``), then a horizontal
    and a vertical pass of integer multiply-adds, each rounded and clipped to uint8;
  * transformers 5.15 ``CLIPImageProcessorPil`` (``image_processing_backends.PilBackend._preprocess``,
    ``image_transforms.{get_resize_output_image_size, center_crop, rescale, normalize}``): shortest edge -> ``size``
    with the long edge truncated, centre crop, ``float32(float64(u8) * rescale_factor)``, ``(x - mean) / std`` in float32.

Pinned: ``tests/test_image_cpu.py`` checks this file bit for bit against Pillow itself and against the HF processor,
both run in this container, and against ``tests/golden/image_*.npz`` (made by ``tools/make_golden_image.py`` from
Pillow + HF).  Plain numpy; scalar Python loops only over output coordinates (<= a few hundred per axis)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c: 8 bits of sample, 2 bits of head room for the filter overshoot
BILINEAR, BICUBIC = 2, 3     # PIL.Image.Resampling values
_SUPPORT = {BILINEAR: 1.0, BICUBIC: 2.0}


def _filter(kind, x):
    x = -x if x < 0.0 else x
    if kind == BICUBIC:  # Keys' cubic convolution, a = -0.5 (Resample.c bicubic_filter)
        if x < 1.0:
            return ((-0.5 + 2.0) * x - (-0.5 + 3.0)) * x * x + 1
        if x < 2.0:
            return (((x - 5) * x + 8) * x - 4) * -0.5
        return 0.0
    if kind == BILINEAR:
        return 1.0 - x if x < 1.0 else 0.0
    raise ValueError(f"filter {kind}: only BILINEAR (2) and BICUBIC (3) are restated")


def precompute_coeffs(in_size, out_size, kind=BICUBIC):
    """-> (bounds int32 [out_size, 2] = (first source index, tap count), coefficients int32 [out_size, ksize])
    for the full-image box (0, in_size), as ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` compute them."""
    scale = float(np.float32(in_size) - np.float32(0)) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = _SUPPORT[kind] * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = [_filter(kind, (x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk):
    """One resampling pass along axis 0 of ``img`` (uint8 [in, ...]) -> uint8 [out, ...]."""
    out = np.empty((bounds.shape[0],) + img.shape[1:], dtype=np.uint8)
    wide = img.astype(np.int64)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.tensordot(kk[xx, :n].astype(np.int64), wide[x0:x0 + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        acc = acc.astype(np.int32)  # the C accumulators are 32-bit ints (never overflow: sum |k| < 2^23)
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_u8(img, out_h, out_w, kind=BICUBIC):
    """``PIL.Image.fromarray(img).resize((out_w, out_h), kind)`` for uint8 [H, W, C]: horizontal pass first, then
    vertical, the intermediate rounded to uint8 (ImagingResampleInner); same-size requests copy."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    in_h, in_w = img.shape[:2]
    if (in_h, in_w) == (out_h, out_w):
        return img.copy()
    if in_w != out_w:
        b, k = precompute_coeffs(in_w, out_w, kind)
        img = _pass(img.transpose(1, 0, 2), b, k).transpose(1, 0, 2)
    if in_h != out_h:
        b, k = precompute_coeffs(in_h, out_h, kind)
        img = _pass(img, b, k)
    return np.ascontiguousarray(img)


def resized_size(in_h, in_w, shortest_edge):
    """``get_resize_output_image_size(default_to_square=False)``: (out_h, out_w)."""
    short, long = (in_w, in_h) if in_w <= in_h else (in_h, in_w)
    new_short, new_long = shortest_edge, int(shortest_edge * long / short)
    return (new_long, new_short) if in_w <= in_h else (new_short, new_long)


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess(images, size=224, crop=224, kind=BICUBIC, rescale_factor=1 / 255, mean=CLIP_MEAN, std=CLIP_STD,
                    do_resize=True, do_center_crop=True, do_rescale=True, do_normalize=True):
    """``CLIPImageProcessorPil(**config)(images)["pixel_values"]`` for a list of uint8 [H, W, 3] arrays ->
    float32 [B, 3, crop, crop].  ``size`` is the shortest edge (int) or an (h, w) pair."""
    out = []
    mean32, std32 = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32)
    for img in images:
        img = np.asarray(img, dtype=np.uint8)
        if do_resize:
            oh, ow = resized_size(img.shape[0], img.shape[1], size) if np.isscalar(size) else size
            img = resize_u8(img, oh, ow, kind)
        if do_center_crop:
            ch, cw = (crop, crop) if np.isscalar(crop) else crop
            h, w = img.shape[:2]
            if h < ch or w < cw:
                raise ValueError("centre crop larger than the resized image (zero padding) is not restated")
            top, left = (h - ch) // 2, (w - cw) // 2
            img = img[top:top + ch, left:left + cw]
        x = img.transpose(2, 0, 1)
        if do_rescale:
            x = (x.astype(np.float64) * rescale_factor).astype(np.float32)
        if do_normalize:
            x = x.astype(np.float32) if not np.issubdtype(x.dtype, np.floating) else x
            x = ((x.T - mean32) / std32).T
        out.append(np.ascontiguousarray(x))
    return np.stack(out)
