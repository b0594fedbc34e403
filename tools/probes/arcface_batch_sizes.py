"""ArcFace r50 throughput by faces per forward (`chunk`): the 14 x 14 stage (half the FLOPs) has faces x 196 / 256 row tiles, so
the chunk decides how full its single round of workgroups is.  usage: python tools/probes/arcface_batch_sizes.py [chunk ...]"""
import sys
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import torch
import bench_encoders
from oracle import arcface as oa
from viquae_amd.arcface import ArcFaceR50

sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 320, 328, 512, 640, 656, 1024]
model = ArcFaceR50.from_state_dict(oa.seeded_state(0)).cuda().eval()
for B in sizes:
    model.chunk = B
    px = torch.rand((B, 3, 112, 112), device="cuda") * 2 - 1
    t = bench_encoders.time_it(lambda: model(px), 5)
    print(B, f"{B / t:9.0f} faces/s  {t * 1e3:7.2f} ms  {12.63e9 * B / t / 1e12:6.1f} TFLOP/s algorithmic", flush=True)
