#!/bin/bash
# kernel trace of the ArcFace r50 forward (256 faces per batch): gpurun -- 'bash tools/probes/arcface_profile.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_arcface
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/arc_run.py <<PY
import sys; sys.path.insert(0, "$R/tools"); sys.path.insert(0, "$R")
import bench_encoders
print(bench_encoders.arcface_throughput(B=328, steps=5))
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
# the launches of the LAST forward in order (name, workgroups, microseconds)
python3 - <<PY
import csv,glob,re
f=sorted(glob.glob("gpurun_out/prof_arcface/kt/**/*kernel_trace.csv",recursive=True))[-1]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
rows=[r for r in rows if any(x in r["Kernel_Name"] for x in ("im2col","gemm","conv3x3","splitk"))]
wgs=lambda r:(int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"])//max(1,int(r["Workgroup_Size_Y"])))
heads=[i for i,r in enumerate(rows) if "im2col" in r["Kernel_Name"] and wgs(r)[0]==784]
tot={}
for r in rows[heads[-2]+3:heads[-1]+3]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    n=re.sub(r"\(anonymous namespace\)::|void |\(.*","",r["Kernel_Name"])[:40]
    print(f"{n:40s} {wgs(r)[0]:6d} x {wgs(r)[1]:6d}  {d:8.1f} us")
    tot[n]=tot.get(n,0)+d
print({k:round(v/1e3,2) for k,v in tot.items()}, "ms; sum", round(sum(tot.values())/1e3,2))
PY
