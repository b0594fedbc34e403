"""DPR / CLIP forwards with the weights' bf16 split in tile layout (default) against the row-major split, one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_encoders as be

for rep in range(2):
    for tiled in ("0", "1"):
        os.environ["MQ_ENC_W_TILED"] = tiled
        d = be.dpr_throughput(steps=3)
        c = be.clip_throughput(steps=3) if hasattr(be, "clip_throughput") else None
        print(f"MQ_ENC_W_TILED={tiled}: DPR 2048x100 {d['ms_per_batch']:.2f} ms" + (f", CLIP {c['ms_per_batch']:.2f} ms" if c else ""), flush=True)
