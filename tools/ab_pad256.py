"""DPR forward on the reference's pad-to-256 workload (packed forward) and at 2048 x 100: python3 tools/ab_pad256.py
(MEERQAT_HIP_LIB selects an A/B build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_encoders as be
d = be.dpr_padded_throughput(steps=3)
e = be.dpr_throughput(steps=3)
print(f"{os.environ.get('MEERQAT_HIP_LIB', 'default')}: pad-to-256 {d['ms_per_batch']:.2f} ms ({d['passages_per_s']:.0f} passages/s), "
      f"2048 x 100 {e['ms_per_batch']:.2f} ms")
