for v in $1; do export MEERQAT_HIP_LIB=$PWD/ab/lib_$v.so; echo "== $v"; timeout 300 python bench.py --mode exact_f32 --no-encoders --no-other-path --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'])"; done
