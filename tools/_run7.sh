cd $GRAFT_REPO_ROOT
python -m pytest tests/test_encoders_gpu.py tests/test_embedding_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -3
bash tools/ab_encoders.sh "r02 default r02 default" 2>&1 | grep -v Warn
