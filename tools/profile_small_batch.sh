#!/bin/bash
# kernel profile of search_device at 256 queries (the reference's map batch size): gpurun -- 'bash tools/profile_small_batch.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_sb
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/tools/small_batch_search.py ${1:-256} 50 > $O/kt.log 2>&1
tail -2 $O/kt.log
cd $R
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_sb/kt/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
