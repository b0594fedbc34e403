"""Throughput of the device-side CLIP image preprocessing (csrc/image.hip) on one MI355X: the batch the reference's
image-embedding job uses (3072 images, experiments/image_embedding/clip/vit_config.json:3-5), synthetic 500 x 375 RGB
images (the size Wikimedia thumbnails typically have).  Reports the kernels alone (source already in HBM) and the whole
call (packing into pinned memory + H2D + kernels), next to Pillow + transformers on the host cores for a sample."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes

import numpy as np
import torch


def main(B=3072, H=375, W=500, cpu_sample=64):
    from viquae_amd import _lib
    from viquae_amd.image.preprocess import CLIPImageProcessorHIP
    lib = _lib.load()
    rng = np.random.default_rng(0)
    base = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(32)]
    ims = [base[i % 32] if i % 3 else np.ascontiguousarray(base[i % 32].transpose(1, 0, 2)) for i in range(B)]
    p = CLIPImageProcessorHIP()
    p(ims)  # first call: pins the staging buffer, starts the packing threads
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = p(ims)["pixel_values"]
    torch.cuda.synchronize()
    whole = time.perf_counter() - t0
    head = out[:cpu_sample].cpu().numpy()
    # kernels alone
    geom, totals = p.plan(np.array([a.shape[:2] for a in ims], dtype=np.int64))
    src = torch.empty(int(totals[0]), dtype=torch.uint8, device="cuda").random_(0, 256)
    gdev = torch.from_numpy(geom).cuda()
    ws = torch.empty(int(totals[1]), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.check(lib.mq_image_preprocess_u8(src.data_ptr(), gdev.data_ptr(), B, totals.ctypes.data, 224, 224, 3, 3,
                                              ctypes.c_double(1 / 255), p.image_mean.ctypes.data, p.image_std.ctypes.data,
                                              out.data_ptr(), ws.data_ptr(), ws.numel(), st))
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    kern = e0.elapsed_time(e1) / 5 * 1e-3
    alg_bytes = int(totals[0]) + out.numel() * 4  # every source byte read once + the float output written once
    rec = {"images": B, "source": f"{H}x{W} / {W}x{H} RGB uint8", "kernels_ms": round(kern * 1e3, 3),
           "kernels_images_per_s": round(B / kern, 1), "kernels_algorithmic_GBps": round(alg_bytes / kern / 1e9, 1),
           "whole_call_ms": round(whole * 1e3, 1), "whole_call_images_per_s": round(B / whole, 1)}
    try:
        from PIL import Image
        from transformers import CLIPImageProcessor
        hf = CLIPImageProcessor()
        pil = [Image.fromarray(a) for a in ims[:cpu_sample]]
        t0 = time.perf_counter()
        ref = hf(pil, return_tensors="np")["pixel_values"]
        t = time.perf_counter() - t0
        rec["cpu_pillow_transformers_images_per_s_1_thread"] = round(cpu_sample / t, 1)
        rec["identical_to_pillow_transformers"] = bool(np.array_equal(ref, head))
    except Exception as e:  # pragma: no cover
        rec["cpu"] = repr(e)
    return rec


if __name__ == "__main__":
    print(json.dumps(main()))
