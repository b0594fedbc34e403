#!/bin/bash
# Where attention_x3_kernel's cycles go at the DPR shape (tools/attn_timing.py): one rocprofv3 --pmc pass per counter group (never
# combined with tracing), python3 directly after `--`.   tools/pmc_attention.sh [variant]   (default = the in-tree library)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=${1:-default}
O=$R/gpurun_out/pmc_attn_$V
rm -rf $O; mkdir -p $O
if [ "$V" != default ]; then export MEERQAT_HIP_LIB=$R/ab/lib_$V.so; fi
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $O/a --output-format csv -- python3 $R/tools/attn_timing.py > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/b --output-format csv -- python3 $R/tools/attn_timing.py > $O/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA -d $O/c --output-format csv -- python3 $R/tools/attn_timing.py > $O/c.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "attention_x3_kernel" in row["Kernel_Name"]:
            k = row["Kernel_Name"].split("attention_x3_kernel")[1].split("(")[0]
            acc[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
for k, cs in acc.items():
    print("attention_x3_kernel" + k)
    for name, d in sorted(cs.items()):
        v = list(d.values())
        print(f"   {name:28s} mean per launch {sum(v) / len(v):.4e}  ({len(v)} launches)")
PY
