"""Builds libmeerqat_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m viquae_amd.build` or `viquae_amd.build.build()`.  hipcc cross-compiles without a GPU.
The .so is git-ignored but travels to the GPU box with the source tree.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(CSRC, "libmeerqat_hip.so")
SOURCES = ["knn.hip", "encoder.hip", "conv.hip", "fuse.hip", "image.hip", "diag.hip"]
ARCH = "gfx950"


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build the HIP library")
    return exe


def needs_build():
    if not os.path.exists(SO):
        return True
    so_m = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps.append(os.path.join(HERE, "..", "include", "meerqat_hip.h"))
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".inc"))]
    return any(os.path.getmtime(d) > so_m for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link one shared library. Returns its path."""
    if not force and not needs_build():
        return SO
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-fno-fast-math", "-ffp-contract=off", "-Wno-unused-result"] + srcs + ["-o", SO + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(SO + ".tmp", SO)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
