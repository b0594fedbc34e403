#!/usr/bin/env python3
"""A SECOND, independent statement of the ArcFace r50 network (the one `meerqat/image/face_recognition.py:55-61` loads through
`arcface_torch.backbones.get_model('r50', fp16=True)`), built from torch-CPU's own operators -- `torch.nn.functional.conv2d`,
`batch_norm`, `prelu`, `linear` -- instead of the numpy im2col / broadcast arithmetic of `oracle/arcface.py`.  It exists so that
the numpy restatement (and through it the HIP path) is no longer checked only against itself ("Next round" 2):
an arithmetic slip in `oracle/arcface.py` (a BN folded with the wrong variance, a stride on the wrong convolution, a flatten in
NHWC order, PReLU slopes applied to the wrong axis) shows up as a difference between the two statements.

Both statements were written from the published definition of IResNet (insightface, recognition/arcface_torch/backbones/
iresnet.py) -- `arcface_torch` itself is un-vendored and not installable here, so the row stays **parity unpinned vs
arcface_torch**; what this script pins is the restatement's arithmetic against torch's operators.

It also measures what the reference's `fp16=True` costs: under `torch.cuda.amp.autocast` every convolution / PReLU of the
backbone runs in half precision with fp32 accumulation, BatchNorm and the residual add take and return half tensors, and the head
(`fc` + `features`) runs in fp32 on `x.float()`.  That is emulated here by rounding the convolution weights and EVERY
intermediate tensor of the backbone to fp16 (computation in fp32 between the roundings) -- max |fp32 - fp16-autocast| over the
same faces is recorded in the golden file and quoted in DESIGN.md section 2.

Runs in the build container only (torch-CPU); commits inputs + outputs only:
    python tools/make_golden_arcface.py        ->  tests/golden/arcface_r50_8.npz
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden", "arcface_r50_8.npz")

STAGES = ((3, 64), (4, 128), (14, 256), (3, 512))   # (blocks, channels) of r50


def faces_from_bytes(u8):
    """uint8 [B, 112, 112, 3] -> fp32 [B, 3, 112, 112]: torchvision's ToTensor (x / 255) then Normalize(0.5, 0.5)."""
    x = torch.from_numpy(u8).permute(0, 3, 1, 2).to(torch.float32).div(255.0)
    return (x - 0.5) / 0.5


def torch_iresnet50(state, x, half=False):
    """IResNet-50 forward in eval mode over a checkpoint-layout state dict of torch tensors.  ``half``: emulate the
    reference's fp16 autocast (see the module docstring)."""
    r = (lambda t: t.to(torch.float16).to(torch.float32)) if half else (lambda t: t)

    def norm(t, name):
        return r(F.batch_norm(t, state[name + ".running_mean"], state[name + ".running_var"], state[name + ".weight"],
                              state[name + ".bias"], training=False, momentum=0.0, eps=1e-5))

    def conv(t, name, stride, padding):
        return r(F.conv2d(t, r(state[name + ".weight"]), None, stride=stride, padding=padding))

    def prelu(t, name):
        return r(F.prelu(t, r(state[name + ".weight"])))

    t = prelu(norm(conv(r(x), "conv1", 1, 1), "bn1"), "prelu")
    for s, (blocks, _) in enumerate(STAGES, start=1):
        for b in range(blocks):
            p = f"layer{s}.{b}"
            stride = 2 if b == 0 else 1
            u = conv(norm(t, p + ".bn1"), p + ".conv1", 1, 1)
            u = prelu(norm(u, p + ".bn2"), p + ".prelu")
            u = norm(conv(u, p + ".conv2", stride, 1), p + ".bn3")
            if b == 0:
                t = norm(conv(t, p + ".downsample.0", stride, 0), p + ".downsample.1")
            t = r(u + t)
    t = norm(t, "bn2")
    t = torch.flatten(t, 1).to(state["fc.weight"].dtype)   # the head runs in full precision on x.float() (dropout p = 0)
    t = F.linear(t, state["fc.weight"], state["fc.bias"])
    return F.batch_norm(t, state["features.running_mean"], state["features.running_var"], state["features.weight"],
                        state["features.bias"], training=False, momentum=0.0, eps=1e-5)


def main():
    from oracle import arcface as oa
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    seed_weights, seed_faces, n = 0, 1, 8
    u8 = np.random.default_rng(seed_faces).integers(0, 256, (n, 112, 112, 3), dtype=np.uint8)
    x = faces_from_bytes(u8)
    state = {k: torch.from_numpy(v) for k, v in oa.seeded_state(seed_weights).items()}
    y32 = torch_iresnet50(state, x).numpy().astype(np.float32)
    y16 = torch_iresnet50(state, x, half=True).numpy().astype(np.float32)
    y64 = torch_iresnet50({k: v.double() for k, v in state.items()}, x.double()).numpy()
    dev16 = float(np.abs(y16 - y32).max())
    print(f"torch fp32 statement: shape {y32.shape}, |y| max {np.abs(y32).max():.3f}, rms {np.sqrt((y32 ** 2).mean()):.3f}")
    print(f"  max |fp32 - float64|        = {np.abs(y32 - y64).max():.3e}")
    print(f"  max |fp16-autocast - fp32|  = {dev16:.3e}   (cosine >= {min(float(a @ b / np.linalg.norm(a) / np.linalg.norm(b)) for a, b in zip(y16, y32)):.6f})")
    want = oa.iresnet_forward(oa.seeded_state(seed_weights), x.numpy())
    print(f"  max |oracle/arcface.py - torch fp32| = {np.abs(want - y32).max():.3e}")
    np.savez_compressed(GOLDEN, faces_u8=u8, embeddings=y32, embeddings_f64=y64.astype(np.float64),
                        embeddings_fp16_autocast=y16, seed_weights=np.int64(seed_weights),
                        fp16_autocast_max_abs_dev=np.float64(dev16))
    print("wrote", GOLDEN, os.path.getsize(GOLDEN), "bytes")


if __name__ == "__main__":
    main()
