cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r3_gpu_tests.log
cat gpurun_out/r3_gpu_tests.log
python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
tail -3 gpurun_out/r3_bench.err
python -c "
import json
d=json.load(open('gpurun_out/r3_bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['kernel_ms'])
print('other', d['other_exact_path'])
cb=d['cpu_baseline']; print('cpu', cb['value'], cb['cores'], cb.get('threads'), cb['kind']); print(cb['legs']); print(cb.get('encoders'))
s=d['secondary']
for k in ('kb_passages_encoded_per_s','images_encoded_per_s','titles_encoded_per_s','dpr_like_data','clip_kb_search'): print(k, s.get(k))
print('dpr', s['dpr']); print('pad', {k:v for k,v in s['dpr_reference_padding'].items() if k in ('passages_per_s','ms_per_batch')}); print('q', {k:v for k,v in s['dpr_questions_reference_padding'].items() if k in ('passages_per_s','ms_per_batch')})
print(json.dumps(s['encode_call_surface'])[:3000])
print(s['reference_call_surface'])
"
