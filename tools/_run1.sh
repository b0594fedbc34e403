set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_tie_order_bigk_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r3_t1.log
python -m pytest tests/test_knn_gpu.py tests/test_screened_gpu.py tests/test_golden_gpu.py tests/test_small_batch_l2_gpu.py tests/test_no_panel_gpu.py tests/test_sharded_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r3_t2.log
for i in 1 2; do
MEERQAT_HIP_LIB=$PWD/ab/lib_r02.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-encoders 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r02', d['value'], d['roofline']['kernel_ms'], d['other_exact_path']['kernel_ms'], d['other_exact_path']['results_identical_to_headline_path'])" >> gpurun_out/r3_ab.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-encoders 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['value'], d['roofline']['kernel_ms'], d['other_exact_path']['kernel_ms'], d['other_exact_path']['results_identical_to_headline_path'])" >> gpurun_out/r3_ab.log
done
cat gpurun_out/r3_t1.log gpurun_out/r3_t2.log gpurun_out/r3_ab.log
