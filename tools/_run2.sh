cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_pipeline_gpu.py tests/test_embedding_gpu.py tests/test_image_gpu.py tests/test_searcher_gpu.py tests/test_tie_order_bigk_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r3_t3.log
cat gpurun_out/r3_t3.log
python tools/bench_encode_surface.py 32768 > gpurun_out/r3_encode_surface.json 2> gpurun_out/r3_encode_surface.err
tail -5 gpurun_out/r3_encode_surface.err
cat gpurun_out/r3_encode_surface.json
python tools/bench_host_path.py 2>/dev/null | tail -1 > gpurun_out/r3_host_path.json
cat gpurun_out/r3_host_path.json
nproc; lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)"
