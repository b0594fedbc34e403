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
# the launches of the LAST forward in order (name, grid, microseconds)
python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/prof_arcface/kt/**/*kernel_trace.csv",recursive=True))[-1]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
stems=[i for i,r in enumerate(rows) if "im2col" in r["Kernel_Name"] and int(r["Grid_Size_Y"])>=12544*4//64*16]
heads=[i for i,r in enumerate(rows) if "im2col" in r["Kernel_Name"] and int(r["Grid_Size_X"])//256==784]
start=max(i for i in range(len(rows)) if "im2col" in rows[i]["Kernel_Name"] and i < heads[-1] and (heads[-2] if len(heads)>1 else -1) < i and rows[i] is not None and i==min(j for j in range((heads[-2] if len(heads)>1 else -1)+1, heads[-1]+1) if "im2col" in rows[j]["Kernel_Name"] or "gemm" in rows[j]["Kernel_Name"] or "conv3x3" in rows[j]["Kernel_Name"]))
tot={}
for r in rows[start:heads[-1]+2]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    n=r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::","").replace("void ","")[:44]
    print(f"{n:44s} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d} x {int(r['Grid_Size_Y'])//max(1,int(r['Workgroup_Size_Y'])):6d}  {d:8.1f} us")
    tot[n]=tot.get(n,0)+d
print({k:round(v/1e3,2) for k,v in tot.items()}, "ms; sum", round(sum(tot.values())/1e3,2))
PY
