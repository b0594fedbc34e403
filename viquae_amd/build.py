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
SOURCES = ["knn.hip", "encoder.hip", "conv.hip", "fuse.hip", "image.hip", "jpeg.hip", "diag.hip", "runfmt.cpp"]
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


FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off", "-Wno-unused-result"]
OBJ = os.path.join(CSRC, "_obj")


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


def build(force=False, verbose=False):
    """Compile every source for gfx950 (one object per source, rebuilt only when that source, an include of csrc/ or the C-ABI
    header changed) and link one shared library. Returns its path."""
    if not force and not needs_build():
        return SO
    os.makedirs(OBJ, exist_ok=True)
    shared = [os.path.join(HERE, "..", "include", "meerqat_hip.h")]
    shared += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".inc"))]
    objs, jobs = [], []
    for name in SOURCES:
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJ, name + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + shared):
            cmd = [_hipcc(), f"--offload-arch={ARCH}"] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((obj, subprocess.Popen(cmd)))
    failed = [obj for obj, p in jobs if p.wait() != 0]
    if failed:
        for obj in failed:
            if os.path.exists(obj):
                os.remove(obj)
        raise subprocess.CalledProcessError(1, f"hipcc -c ({', '.join(os.path.basename(o) for o in failed)})")
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs + ["-lpthread", "-o", SO + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(SO + ".tmp", SO)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
