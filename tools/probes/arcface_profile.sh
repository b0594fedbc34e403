#!/bin/bash
# kernel trace of the ArcFace r50 forward (256 faces per batch): gpurun -- 'bash tools/probes/arcface_profile.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_arcface
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/arc_run.py <<PY
import sys; sys.path.insert(0, "$R/tools"); sys.path.insert(0, "$R")
import bench_encoders
print(bench_encoders.arcface_throughput(B=256, steps=5))
PY
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 /tmp/arc_run.py > $O/kt.log 2>&1
tail -1 $O/kt.log
cd $R
python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/prof_arcface/kt/**/*kernel_stats.csv",recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
