cd $GRAFT_REPO_ROOT
for procs in 32 64 96; do for ext in bmp jpg; do MQ_IMAGE_DECODE_PROCS=$procs timeout 600 python tools/bench_encode_surface.py --image $ext 2>&1 | grep "^{" | python -c "
import sys, json
r = json.loads(sys.stdin.read()); e = r['end_to_end']; print('$procs $ext', e['images_per_s'], e['ms_per_batch'], e['pipeline'])"; done; done
timeout 600 python -m pytest tests/test_pipeline_gpu.py -x -q 2>&1 | tail -2
