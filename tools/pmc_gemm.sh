#!/bin/bash
# Sustained clock and matrix-pipe utilisation of the split-bf16 encoder GEMMs (tools/bench_gemm_shapes.py):
# pass 1 = rocprofv3 --kernel-trace --stats (durations), pass 2 = --pmc (cycles), no tracing in the counter pass.
#   tools/pmc_gemm.sh [variant [tag]]     (default = the in-tree library, else ab/lib_<variant>.so; tag names the output directory,
#   e.g. `MQ_GEMM_WIDE=0 tools/pmc_gemm.sh default x3s` for the sixteen-wave kernel)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=${1:-default}
O=$R/gpurun_out/pmc_gemm_$V${2:+_$2}
rm -rf $O; mkdir -p $O/trace $O/pmc
if [ "$V" != default ]; then export MEERQAT_HIP_LIB=$R/ab/lib_$V.so; fi
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/tools/bench_gemm_shapes.py > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc --output-format csv -- python3 $R/tools/bench_gemm_shapes.py > $O/pmc.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections, re
O = sys.argv[1]
dur = collections.defaultdict(list)
for f in glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gemm_nt_x3" in row["Kernel_Name"]:
            dur[(re.search(r"gemm_nt_x3[sw]_kernel<[^>]*>", row["Kernel_Name"]) or re.search(r"gemm_nt_x3\w*", row["Kernel_Name"])).group(0)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gemm_nt_x3" in row["Kernel_Name"]:
            acc[(re.search(r"gemm_nt_x3[sw]_kernel<[^>]*>", row["Kernel_Name"]) or re.search(r"gemm_nt_x3\w*", row["Kernel_Name"])).group(0)][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
for k in sorted(dur):
    d = sorted(dur[k]); ms = sum(d) / len(d)
    c = acc.get(k, {})
    gui = c.get("GRBM_GUI_ACTIVE", {}); mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", {})
    if not gui: continue
    cyc = sum(gui.values()) / len(gui) / 8
    busy = sum(mf.values()) / len(mf)
    print(f"{k[-60:]:60s} launches {len(d):3d} mean {ms*1e3:7.3f} ms  (pmc pass: {cyc:.3e} cycles/XCD -> {cyc/ms/1e9:.2f} GHz if same duration)  MFMA busy {busy/(1024*cyc)*100:.1f} %")
PY
