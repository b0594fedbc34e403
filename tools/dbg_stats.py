import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viquae_amd.index import MI355XFlatIndex
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(0)
N = 300000
idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
for s in range(0, N, 1 << 16):
    idx.add(torch.randn((min(1 << 16, N - s), 768), generator=g, device=dev), total_hint=N)
Q = torch.randn((16384, 768), generator=g, device=dev)
for s in (0, 4096, 12288):
    D, I = idx.search_device(Q[s:s + 4096].contiguous(), 100)
    print("contig", s, idx.screen_stats(4096, 100))
    D2, I2 = idx.search_device(Q[s:s + 4096], 100)
    print("view  ", s, idx.screen_stats(4096, 100), torch.equal(D, D2), torch.equal(I, I2))
D, I = idx.search_device(Q, 100)
print("all", idx.screen_stats(16384, 100))
