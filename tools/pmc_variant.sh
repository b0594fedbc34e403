#!/bin/bash
# MFMA-busy cycles and GPU-active cycles of screen_scan_kernel for a library variant (A/B of ablation builds):
#   tools/pmc_variant.sh <variant>      (variant "default" = the in-tree library, else ab/lib_<variant>.so)
# Prints kernel name, counter, mean per launch.  One rocprofv3 --pmc pass, no tracing (gpurun's rule).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$1
O=$R/gpurun_out/pmc_$V
rm -rf $O; mkdir -p $O
if [ "$V" != default ]; then export MEERQAT_HIP_LIB=$R/ab/lib_$V.so; fi
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O --output-format csv -- python3 $R/tools/screen_scan_only.py > $O/log.txt 2>&1
tail -1 $O/log.txt
python3 - "$O" <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(f[0])):
    if "screen_scan" in row["Kernel_Name"]:
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
for name, d in acc.items():
    v = list(d.values())
    print(name, "launches", len(v), "mean %.4e" % (sum(v) / len(v)))
PY
