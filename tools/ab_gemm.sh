#!/bin/bash
# same-box A/B of the split-bf16 GEMM on the encoder shapes:  tools/ab_gemm.sh "<variant> ..."  ("default" = in-tree library)
for v in $1; do
  if [ "$v" = default ]; then unset MEERQAT_HIP_LIB; else export MEERQAT_HIP_LIB=$PWD/ab/lib_$v.so; fi
  echo "== variant: $v"
  python tools/bench_gemm_shapes.py 2>/dev/null
done
