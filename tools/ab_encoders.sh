#!/bin/bash
# same-box A/B of whole encoder forwards:  tools/ab_encoders.sh "<variant> ..."  ("default" = in-tree library)
for v in $1; do
  if [ "$v" = default ]; then unset MEERQAT_HIP_LIB; else export MEERQAT_HIP_LIB=$PWD/ab/lib_$v.so; fi
  echo "== variant: $v"
  python - <<'PY' 2>/dev/null
import sys, json
sys.path.insert(0, "tools")
import bench_encoders as b
b.dpr_throughput(steps=1)
d = b.dpr_throughput(steps=5); c = b.clip_throughput(); p = b.dpr_padded_throughput(steps=3)
print("dpr %.2f ms  clip %.2f ms  dpr_padded %.2f ms" % (d["ms_per_batch"], c["ms_per_batch"], p["ms_per_batch"]))
PY
done
