"""DPR bert-base 2048 x 100 dense forward alone (for rocprofv3 --kernel-trace --stats): python3 tools/dpr_profile.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_encoders as b
print(b.dpr_throughput(steps=int(sys.argv[1]) if len(sys.argv) > 1 else 4))
