#!/bin/bash
# same-box A/B of library variants:  tools/ab_variants.sh "<name> <name> ..."   ("default" = the in-tree library)
for v in $1; do
  if [ "$v" = default ]; then unset MEERQAT_HIP_LIB; else export MEERQAT_HIP_LIB=$PWD/ab/lib_$v.so; fi
  echo "== variant: $v"
  python bench.py --steps 8 --warmup 2 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=j.get('secondary',{})
print('value',j['value'],'ms',j['ms_per_step'],'scan_ms',j['roofline']['kernel_ms'],'frac',j['roofline']['frac'],'identical',j.get('other_exact_path',{}).get('results_identical_to_headline_path'), 'resc', s.get('candidates_rescored_per_query'))
"
done
