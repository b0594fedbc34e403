#!/bin/bash
# same-box A/B of library variants:  tools/ab_variants.sh "<name> <name> ..."   ("default" = the in-tree library)
# (each bench bounded by `timeout`; output also appended to gpurun_out/ab_variants.log)
mkdir -p gpurun_out
for v in $1; do
  if [ "$v" = default ]; then unset MEERQAT_HIP_LIB; else export MEERQAT_HIP_LIB=$PWD/ab/lib_$v.so; fi
  echo "== variant: $v" | tee -a gpurun_out/ab_variants.log
  timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-encoders 2>gpurun_out/ab_err.log | python -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
if not t: print('no output (timeout or error)'); sys.exit(0)
j=json.loads(t[-1])
print('value',j['value'],'ms',j['ms_per_step'],'scan_ms',j['roofline']['kernel_ms'],'frac',j['roofline']['frac'],'identical',j.get('other_exact_path',{}).get('results_identical_to_headline_path'))
" | tee -a gpurun_out/ab_variants.log
  tail -2 gpurun_out/ab_err.log >> gpurun_out/ab_variants.log
done
