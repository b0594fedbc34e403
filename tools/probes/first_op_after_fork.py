import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from viquae_amd.image.decode_pool import DecodePool
mode = sys.argv[1]
def tick(tag, fn):
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); print(f"{mode} {tag}: {time.perf_counter() - t0:.3f} s", flush=True)
torch.zeros(1, device="cuda")
if mode in ("model", "model_pinned", "pinned_only"):
    if mode != "pinned_only":
        from viquae_amd.encoders import CLIPModel
        from bench_encoders import CLIP_VITB32, random_clip_state
        model = CLIPModel.from_state_dict({"vision_config": CLIP_VITB32, "projection_dim": 512}, random_clip_state(CLIP_VITB32, 2)).cuda().eval()
        px = torch.randn(256, 3, 224, 224, device="cuda")
        tick("forward", lambda: model.get_image_features(pixel_values=px))
    if mode in ("model_pinned", "pinned_only"):
        big = torch.empty(2 << 30, dtype=torch.uint8, pin_memory=True)
        tick("h2d from torch-pinned 2 GB", lambda: big.cuda(non_blocking=True))
if mode == "bigalloc":
    keep = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(8)]
t0 = time.perf_counter(); pool = DecodePool(32, 3072 * 768 * 1024, 2); print(f"{mode} pool create {time.perf_counter() - t0:.3f} s", flush=True)
tick("first tiny op after fork", lambda: torch.ones(4).cuda())
tick("h2d slot 0 (1.7 GB)", lambda: pool.tensors[0][: 1700 << 20].cuda(non_blocking=True))
tick("h2d slot 0 again", lambda: pool.tensors[0][: 1700 << 20].cuda(non_blocking=True))
pool.close()
