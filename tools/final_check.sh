#!/bin/bash
# Round-end check on a GPU box: the driver's own sequence (GPU tests, smoke) + the default bench line.
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p gpurun_out
python -m pytest tests/ -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -6 > gpurun_out/final_gpu_tests.log
cat gpurun_out/final_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
tail -2 gpurun_out/final_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/final_bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "dtype")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"])
cb = d["cpu_baseline"]; print("cpu", cb["value"], cb["cores"], cb["kind"]); print(cb["sample"][:700]); print(cb.get("encoders"))
s = d["secondary"]
print({k: s[k] for k in ("kb_passages_encoded_per_s", "images_encoded_per_s", "titles_encoded_per_s")})
print("pad256", s["dpr_reference_padding"]["passages_per_s"], "questions", s["dpr_questions_reference_padding"]["passages_per_s"])
e = s["encode_call_surface"]
for k in e:
    print(k, json.dumps(e[k].get("end_to_end", e[k]))[:600], "| forward_only", e[k].get("forward_only"))
print(s["reference_call_surface"].get("map_arrow"), s["dpr_like_data"])
PY
