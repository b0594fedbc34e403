cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_attn
mkdir -p $O
for flag in 0 1; do
  export MQ_ENC_QKV_SPLIT=$flag
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt$flag --output-format csv -- python3 $R/tools/dpr_profile.py 4 > $O/kt$flag.log 2>&1
  f=$(find $O/kt$flag -name "*kernel_stats.csv" | head -1)
  echo "== qkv_split=$flag"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(r["Name"][:90].replace("(anonymous namespace)::", ""), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
PY
done
