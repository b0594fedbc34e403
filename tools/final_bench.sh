mkdir -p gpurun_out/r06
timeout 1500 python bench.py > gpurun_out/r06/bench.json 2> gpurun_out/r06/bench.err
timeout 600 python bench.py --mode exact_f32 --no-encoders --no-other-path > gpurun_out/r06/bench_exact.json 2>> gpurun_out/r06/bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r06/bench.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
s=d["secondary"]
print("dpr", s["dpr"]["ms_per_batch"], "clip", s["clip"]["ms_per_batch"], "small", s["small_batch"]["ms"], s["small_batch"]["scan_kernel_ms"])
print("l2", json.dumps(s.get("small_batch_l2"))[:700])
print("map_arrow", json.dumps(s["reference_call_surface"].get("map_arrow"))[:300])
PY
