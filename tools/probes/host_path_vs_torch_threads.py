import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
n = int(sys.argv[1])
if n: torch.set_num_threads(n)
x = torch.randn(2048, 2048); (x @ x).sum()   # spin the pool up once
import bench_host_path
d = bench_host_path.main()
print(n, d["map_arrow"], d["device"])
