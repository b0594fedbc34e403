import mmap, time, torch, numpy as np, multiprocessing as mp, os
def bw(t, name, dev_buf):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): dev_buf.copy_(t, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {t.numel() / dt / 1e9:.1f} GB/s  is_pinned={t.is_pinned()}")
n = 256 << 20
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
bw(torch.empty(n, dtype=torch.uint8).pin_memory(), "torch pinned", dev)
m = mmap.mmap(-1, n); t = torch.frombuffer(m, dtype=torch.uint8)
bw(t, "mmap untouched, unregistered", dev)
t.fill_(1)
bw(t, "mmap touched, unregistered", dev)
rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), n, 0); print("register", rc)
bw(t, "mmap touched, registered", dev)
m2 = mmap.mmap(-1, n); t2 = torch.frombuffer(m2, dtype=torch.uint8)
rc = torch.cuda.cudart().cudaHostRegister(t2.data_ptr(), n, 0); print("register untouched", rc)
bw(t2, "mmap registered before first touch", dev)
def child(mm):
    np.frombuffer(mm, dtype=np.uint8)[:] = 7
p = mp.get_context("fork").Process(target=child, args=(m2,)); p.start(); p.join()
bw(t2, "mmap registered, then written by a forked child", dev)
print("value seen on device:", int(dev[12345]))
# shared memory via /dev/shm file
import tempfile
f = open("/dev/shm/mq_probe", "w+b"); f.truncate(n); m3 = mmap.mmap(f.fileno(), n); t3 = torch.frombuffer(m3, dtype=torch.uint8); t3.fill_(3)
rc = torch.cuda.cudart().cudaHostRegister(t3.data_ptr(), n, 0); print("register shm file", rc)
bw(t3, "/dev/shm file mapping, touched, registered", dev)
p = mp.get_context("fork").Process(target=child, args=(m3,)); p.start(); p.join()
bw(t3, "/dev/shm mapping after child write", dev); print("value:", int(dev[777]))
os.unlink("/dev/shm/mq_probe")
