import os, sys, time, tempfile, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from viquae_amd.image.decode_pool import DecodePool
mode = sys.argv[1]
torch.zeros(1, device="cuda")
x = torch.randn(8192, 8192, device="cuda")
side = torch.cuda.Stream()
if mode == "busy_before":
    for _ in range(50): y = x @ x
pool = DecodePool(8, 1 << 30, 2)
def copy(tag, stream):
    with torch.cuda.stream(stream):
        t0 = time.perf_counter(); d = pool.tensors[0][: 1 << 30].to("cuda", non_blocking=True); stream.synchronize()
        print(f"{mode} {tag}: {time.perf_counter() - t0:.3f} s", flush=True)
if mode in ("main_default", "busy_before"):
    copy("main thread, default stream", torch.cuda.current_stream())
elif mode == "main_side":
    copy("main thread, side stream", side)
elif mode == "thread_side":
    t = threading.Thread(target=copy, args=("worker thread, side stream", side)); t.start(); t.join()
elif mode == "thread_side_busy":
    def busy():
        for _ in range(200): y = x @ x
        torch.cuda.synchronize()
    t = threading.Thread(target=copy, args=("worker thread, side stream, main busy", side)); 
    for _ in range(20): y = x @ x
    t.start(); busy(); t.join()
copy("second copy", side)
pool.close()
