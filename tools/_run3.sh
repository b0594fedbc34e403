cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_encoders_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r3_t4.log
cat gpurun_out/r3_t4.log
python - <<'PY' 2>&1 | grep -v Warn | tee gpurun_out/r3_enc_ab.log
import os, sys, json
sys.path.insert(0, "tools")
import bench_encoders as be
for flag in ("0", "1", "0", "1"):
    os.environ["MQ_ENC_QKV_SPLIT"] = flag
    d = be.dpr_throughput(B=2048, L=100, steps=3)
    p = be.dpr_padded_throughput(steps=3)
    c = be.clip_throughput(B=3072, steps=3)
    print("qkv_split", flag, "dpr100 ms", round(d["ms_per_batch"], 2), "pad256", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in p.items()}, "clip ms", round(c["ms_per_batch"], 2))
PY
