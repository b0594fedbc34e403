import sys, os
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import bench_encoders as be
for mode in ("f32", "pair", "f32", "pair"):
    os.environ["MQ_ENC_RESIDUAL"] = mode
    r = be.dpr_throughput(steps=3)
    p = be.dpr_padded_throughput(steps=2)
    print(mode, f"DPR 2048x100 {r['ms_per_batch']:.2f} ms {r['passages_per_s']:.0f}/s   pad-256 {p['ms_per_batch']:.2f} ms {p['passages_per_s']:.0f}/s", flush=True)
