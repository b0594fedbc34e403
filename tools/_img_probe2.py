import os, sys, time, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import viquae_amd.image.preprocess as P
orig = P.CLIPImageProcessorHIP.run_packed
def timed_run_packed(self, packed, geom, totals, B, out=None):
    t0 = time.perf_counter()
    dev = torch.device("cuda")
    st = torch.cuda.current_stream(dev)
    src = packed[:int(totals[0])].to(dev, non_blocking=True); t1 = time.perf_counter()
    st.synchronize(); t2 = time.perf_counter()
    r = orig(self, packed, geom, totals, B, out=out); t3 = time.perf_counter()
    print(f"   run_packed: issue h2d {t1-t0:.3f} sync {t2-t1:.3f} full {t3-t2:.3f} stream={st.cuda_stream:#x}", flush=True)
    return r
P.CLIPImageProcessorHIP.run_packed = timed_run_packed
import bench_encode_surface as b
r = b.image_job(n_refs=2 * 3072)
print(json.dumps(r["end_to_end"]))
