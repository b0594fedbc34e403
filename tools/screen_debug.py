import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from viquae_amd.index import MI355XFlatIndex
N, d, nq, k = int(os.environ.get("N", 20000)), 768, int(os.environ.get("NQ", 256)), 100
dev = torch.device("cuda")
idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.randn((N, d), generator=g, device=dev); idx.add(X)
Q = torch.randn((nq, d), generator=g, device=dev)
D, I = idx.search_device(Q, k); torch.cuda.synchronize()
print("stats", idx.screen_stats(nq, k))
# compare bf16 scores
Xb = X.to(torch.bfloat16).float(); Qb = Q.to(torch.bfloat16).float()
S = Qb @ Xb.T
print("true bf16 top score q0:", S[0].topk(3).values.tolist(), "exact:", (Q[0] @ X.T).topk(3).values.tolist(), "D0", D[0, :3].tolist())
