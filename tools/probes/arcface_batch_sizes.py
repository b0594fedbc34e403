import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import bench_encoders
for B in (64,256,1024):
    print(B, bench_encoders.arcface_throughput(B=B, steps=3))
