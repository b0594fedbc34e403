#!/usr/bin/env python3
"""Register / scratch usage of the kernels of one HIP source, from hipcc's -Rpass-analysis=kernel-resource-usage (no GPU needed).
usage: python tools/kernel_resources.py viquae_amd/csrc/knn.hip [substring of the kernel name] [extra hipcc flags ...]"""
import re
import subprocess
import sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off",
       "-Wno-unused-result", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
if "error" in out:
    print("\n".join(l for l in out.splitlines() if "error" in l)[:3000])
for name, r in rows.items():
    if pat in name:
        short = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
        print(f"{short:50s} VGPR {r.get('VGPRs', '?'):>4} AGPR {r.get('AGPRs', '?'):>4} scratch {r.get('ScratchSize', '?'):>4} "
              f"vspill {r.get('VGPRs Spill', '?'):>3} sspill {r.get('SGPRs Spill', '?'):>3} occ {r.get('Occupancy', '?')}")
