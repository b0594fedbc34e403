#!/bin/bash
# same-box A/B of library variants on tools/probe_gemm_m.py:  tools/ab_probe.sh "<variant> ..." [probe args]   ("default" = in-tree)
for v in $1; do
  if [ "$v" = default ]; then unset MEERQAT_HIP_LIB; else export MEERQAT_HIP_LIB=$PWD/ab/lib_$v.so; fi
  echo "== variant: $v"
  python tools/probe_gemm_m.py ${2:---quick --wide-only} 2>/dev/null
done
