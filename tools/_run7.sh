cd $GRAFT_REPO_ROOT
for v in r02 new r02 new; do
  if [ "$v" = new ]; then unset MEERQAT_HIP_LIB; else export MEERQAT_HIP_LIB=$PWD/ab/lib_$v.so; fi
  echo "== $v"; python3 tools/bench_gemm_shapes.py 2>&1 | grep -v Warn
done
unset MEERQAT_HIP_LIB
python -m pytest tests/test_encoders_gpu.py tests/test_embedding_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -3
bash tools/ab_encoders.sh "r02 default r02 default" 2>&1 | grep -v Warn
