"""Times screen_scan_kernel alone via the event pair of mq_knn_search_screened_f32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viquae_amd import _lib
from viquae_amd.index import MI355XFlatIndex
import os as _os
N, d, nq, k = 1_500_000, 768, int(_os.environ.get("NQ", 4096)), 100
dev = torch.device("cuda"); lib = _lib.load()
idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
g = torch.Generator(device=dev); g.manual_seed(0)
for s in range(0, N, 1 << 16):
    idx.add(torch.randn((min(1 << 16, N - s), d), generator=g, device=dev), total_hint=N)
Q = torch.randn((nq, d), generator=g, device=dev)
D, I = idx.search_device(Q, k); torch.cuda.synchronize()
ws = idx._ws; st = torch.cuda.current_stream()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(6)]
for a, b in evs: a.record(st); b.record(st)
torch.cuda.synchronize()
for a, b in evs:
    _lib.check(lib.mq_knn_search_screened_f32(idx._packed.data_ptr() if idx._packed is not None else None, idx._sqnorm.data_ptr(), idx._rowmajor.data_ptr(), idx._bf16.data_ptr(),
        idx._xmax2.data_ptr(), N, d, Q.data_ptr(), nq, k, 0, 0, 0, D.data_ptr(), I.data_ptr(), ws.data_ptr(), ws.numel(), st.cuda_stream, a.cuda_event, b.cuda_event))
torch.cuda.synchronize()
print(os.environ.get("MEERQAT_HIP_LIB", "default").split("/")[-1], "scan ms:", sum(a.elapsed_time(b) for a, b in evs[1:]) / 5)
