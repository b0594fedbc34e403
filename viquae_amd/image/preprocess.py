"""Device-side stand-in for the ``CLIPFeatureExtractor`` / ``CLIPImageProcessor`` the reference's image
embedding job uses as its ``transform`` (experiments/image_embedding/clip/vit_config.json:13-17, called at
meerqat/image/embedding.py:141-152).

``CLIPImageProcessorHIP(images, return_tensors="pt")["pixel_values"]`` is bit-identical to Pillow's 8-bit
``Image.resize`` + transformers' centre crop / rescale / normalise, but computed by ``mq_image_preprocess_u8``
(csrc/image.hip) on the GPU: the decoded RGB images of a batch are packed into ONE pinned uint8 buffer, copied to the
device once, and come back as the float32 ``[B, 3, crop, crop]`` tensor the CLIP tower reads -- already in HBM.
There is no CPU path: without the HIP library / a GPU the call raises."""
import ctypes
import json
import os

import numpy as np
import torch

from .. import _lib

_RESAMPLE_NAMES = {"bilinear": 2, "bicubic": 3}


def _pair(v, what):
    """HF size dicts / legacy ints -> (mode, h, w)."""
    if isinstance(v, dict):
        if "shortest_edge" in v and v.get("longest_edge") is None:
            return "shortest", int(v["shortest_edge"]), 0
        if "height" in v and "width" in v:
            return "exact", int(v["height"]), int(v["width"])
        raise NotImplementedError(f"{what}={v}: only shortest_edge or height/width are provided")
    if isinstance(v, (list, tuple)):
        return "exact", int(v[0]), int(v[1])  # legacy feature extractors: (h, w)
    return "shortest", int(v), 0


class CLIPImageProcessorHIP:
    """Same constructor keys as ``CLIPImageProcessor`` / the legacy ``CLIPFeatureExtractor`` config
    (``preprocessor_config.json``): do_resize, size, resample, do_center_crop, crop_size, do_rescale, rescale_factor,
    do_normalize, image_mean, image_std, do_convert_rgb."""
    on_device = True  # embed() must not ship this transform to a multiprocessing pool
    model_input_names = ["pixel_values"]

    def __init__(self, do_resize=True, size=224, resample=3, do_center_crop=True, crop_size=224, do_rescale=True,
                 rescale_factor=1 / 255, do_normalize=True, image_mean=(0.48145466, 0.4578275, 0.40821073),
                 image_std=(0.26862954, 0.26130258, 0.27577711), do_convert_rgb=True, device=None, **ignored):
        if isinstance(resample, str):
            resample = _RESAMPLE_NAMES[resample.lower()]
        resample = int(resample)
        if resample not in (2, 3):
            raise NotImplementedError(f"resample={resample}: the HIP resampler provides PIL BILINEAR (2) and BICUBIC (3)")
        if not do_center_crop:
            raise NotImplementedError("do_center_crop=False: images of different sizes cannot be batched")
        self.do_resize, self.resample = bool(do_resize), resample
        self.size_mode, self.size_h, self.size_w = _pair(size, "size")
        mode, self.crop_h, self.crop_w = _pair(crop_size, "crop_size")
        if mode == "shortest":
            self.crop_w = self.crop_h
        self.do_rescale, self.rescale_factor = bool(do_rescale), float(rescale_factor)
        self.do_normalize = bool(do_normalize)
        self.image_mean = np.asarray(image_mean, dtype=np.float32)
        self.image_std = np.asarray(image_std, dtype=np.float32)
        if self.image_mean.shape != (3,) or self.image_std.shape != (3,):
            raise ValueError("image_mean / image_std must have 3 entries (RGB)")
        self.do_convert_rgb = bool(do_convert_rgb)
        self.device = device

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kwargs):
        path = str(pretrained_model_name_or_path)
        if os.path.isdir(path):
            path = os.path.join(path, "preprocessor_config.json")
        with open(path) as f:
            cfg = json.load(f)
        cfg.update(kwargs)
        for k in ("feature_extractor_type", "image_processor_type", "processor_class"):
            cfg.pop(k, None)
        return cls(**cfg)

    # ---- host side: decoded images -> one packed uint8 buffer + geometry -------------------------------------------
    def _arrays(self, images):
        if not isinstance(images, (list, tuple)):
            images = [images]
        out = []
        for im in images:
            if hasattr(im, "convert"):  # PIL
                if im.mode != "RGB":
                    if not self.do_convert_rgb:
                        raise ValueError(f"image mode {im.mode}: only RGB is supported with do_convert_rgb=False")
                    im = im.convert("RGB")
                im = np.asarray(im)
            im = np.asarray(im)
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise ValueError(f"expected uint8 [H, W, 3] images, got {im.dtype} {im.shape}")
            out.append(im)
        return out

    def plan(self, sizes):
        """sizes int64 [B, 2] (h, w) -> (geom int64 [B, MQ_IMAGE_GEOM], totals int64 [MQ_IMAGE_TOTALS]); host arithmetic only."""
        lib = _lib.load()
        sizes = np.ascontiguousarray(sizes, dtype=np.int64).reshape(-1, 2)
        B = sizes.shape[0]
        geom = np.zeros((B, 12), dtype=np.int64)
        totals = np.zeros(5, dtype=np.int64)
        mode = 0 if not self.do_resize else (1 if self.size_mode == "shortest" else 2)
        _lib.check(lib.mq_image_plan(sizes.ctypes.data, B, mode, self.size_h, self.size_w, self.crop_h, self.crop_w,
                                     self.resample, geom.ctypes.data, totals.ctypes.data), "mq_image_plan")
        return geom, totals

    def preprocess(self, images, return_tensors="pt", **unused):
        if return_tensors not in ("pt", None):
            raise ValueError("the HIP image processor returns device tensors: return_tensors='pt'")
        if not torch.cuda.is_available():
            raise RuntimeError("CLIPImageProcessorHIP needs an MI355X: viquae_amd has no CPU path")
        lib = _lib.load()
        arrays = self._arrays(images)
        B = len(arrays)
        dev = torch.device(self.device if self.device is not None else "cuda")
        out = torch.empty((B, 3, self.crop_h, self.crop_w), dtype=torch.float32, device=dev)
        if B == 0:
            return {"pixel_values": out}
        geom, totals = self.plan(np.array([a.shape[:2] for a in arrays], dtype=np.int64))
        packed = self._staging(int(totals[0]))
        host = packed.numpy()

        def pack(lo, hi):  # numpy's copy releases the GIL: a few threads reach the host memory bandwidth
            for a, g in zip(arrays[lo:hi], geom[lo:hi]):
                n = a.shape[0] * a.shape[1] * 3
                host[g[0]:g[0] + n] = a.reshape(-1)  # C- or non-contiguous views alike
        if B >= 64:
            step = (B + self._PACK_THREADS - 1) // self._PACK_THREADS
            list(self._pool().map(lambda lo: pack(lo, min(B, lo + step)), range(0, B, step)))
        else:
            pack(0, B)
        return self.run_packed(packed, geom, totals, B, out=out)

    def run_packed(self, packed, geom, totals, B, out=None, sync=True, jpeg=None):
        """The device half of :meth:`preprocess`: ``packed`` = a (page-locked) uint8 CPU tensor that already holds the B decoded
        images at the byte offsets ``geom[:, 0]`` of :meth:`plan` (the image pipeline's decode workers write them there
        directly, viquae_amd/image/decode_pool.py) -> {"pixel_values": float32 [B, 3, crop_h, crop_w] on the device}.
        ``sync=False``: nothing is waited for -- the copy and the kernels are only enqueued on the current stream; the caller
        keeps ``packed`` untouched, and the returned ``"_keep"`` tensors alive, until an event recorded after the call fires.
        ``jpeg``: the layout of :func:`viquae_amd.image.jpeg.plan_layout` when some of the images are JPEG files whose scans the
        workers decoded into staging areas -- only the first ``h2d_bytes`` of ``packed`` are copied, and ``mq_jpeg_decode_rgb_u8``
        produces those images' RGB bytes on the device before the resize kernels read them."""
        lib = _lib.load()
        dev = torch.device(self.device if self.device is not None else "cuda")
        if out is None:
            out = torch.empty((B, 3, self.crop_h, self.crop_w), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            keep = ()
            if jpeg is None:
                src = packed[:int(totals[0])].to(dev, non_blocking=True)
            else:
                from . import jpeg as dj
                src = torch.empty(int(totals[0]), dtype=torch.uint8, device=dev)
                src[:jpeg["h2d_bytes"]].copy_(packed[:jpeg["h2d_bytes"]], non_blocking=True)
                keep = (dj.decode_staged(src, jpeg["items"], jpeg["max_blocks"], jpeg["max_strips"]),)
            gdev = torch.from_numpy(geom).to(dev, non_blocking=True)
            ws = torch.empty(max(int(totals[1]), 256), dtype=torch.uint8, device=dev)
            flags = (1 if self.do_rescale else 0) | (2 if self.do_normalize else 0)
            _lib.check(lib.mq_image_preprocess_u8(src.data_ptr(), gdev.data_ptr(), B, totals.ctypes.data, self.crop_h, self.crop_w,
                                                  self.resample, flags, ctypes.c_double(self.rescale_factor),
                                                  self.image_mean.ctypes.data, self.image_std.ctypes.data, out.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream),
                       "mq_image_preprocess_u8")
            if not sync:
                return {"pixel_values": out, "_keep": (src, gdev, ws) + keep}
            # the pinned staging buffer and the workspace may be recycled as soon as this returns
            torch.cuda.current_stream(dev).synchronize()
        return {"pixel_values": out}

    __call__ = preprocess

    # ---- pinned staging buffer (grow-only: pinning 1-2 GB costs more than the whole batch) and packing threads -----
    _PACK_THREADS = 8

    def _staging(self, nbytes):
        buf = getattr(self, "_pinned", None)
        if buf is None or buf.numel() < nbytes:
            buf = self._pinned = torch.empty(max(nbytes, 1 << 20) * 5 // 4, dtype=torch.uint8, pin_memory=True)
        return buf

    def _pool(self):
        pool = getattr(self, "_threads", None)
        if pool is None:
            from concurrent.futures import ThreadPoolExecutor
            pool = self._threads = ThreadPoolExecutor(self._PACK_THREADS)
        return pool
