#!/bin/bash
# kernel trace of the one-query-tile search (tools/small_batch_search.py): gpurun -- 'bash tools/profile_small_scan.sh [nq] [tag]'
R=${GRAFT_REPO_ROOT:-$PWD}
NQ=${1:-256}
TAG=${2:-sb}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/tools/small_batch_search.py $NQ 50 > $O/kt.log 2>&1
tail -1 $O/kt.log
cd $R
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_$TAG/kt/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:80], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"])
PY
