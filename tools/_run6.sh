cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/ -q -m gpu 2>&1 | tail -8 > gpurun_out/r3_gpu_tests.log
cat gpurun_out/r3_gpu_tests.log
python bench.py --no-encoders --steps 5 > gpurun_out/r3_bench_small.json 2> gpurun_out/r3_bench_small.err
python -c "
import json
d=json.load(open('gpurun_out/r3_bench_small.json'))
cb=d['cpu_baseline']; print('cpu', cb['value'], cb['cores'], cb.get('threads'), cb['kind']); print(cb['legs'])
"
