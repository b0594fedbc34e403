cd $GRAFT_REPO_ROOT
python3 tools/attn_timing.py 2>&1 | grep -v Warn
MEERQAT_HIP_LIB=$PWD/ab/lib_r02.so python3 tools/attn_timing.py 2>&1 | grep -v Warn
python3 tools/attn_timing.py 2>&1 | grep -v Warn
python -m pytest tests/test_encoders_gpu.py tests/test_embedding_gpu.py -x -q 2>&1 | tail -3
bash tools/ab_encoders.sh "r02 default" 2>&1 | grep -v Warn
