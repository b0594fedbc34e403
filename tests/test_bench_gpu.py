"""GPU: bench.py keeps its contract -- ONE JSON line with the keys the driver reads -- on a small KB (seconds), single-rank
and through the N > 1 code path (RCCL, world size 1 forced; two ranks sharing the GPU over gloo)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}
ROOFLINE = {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def _run(args, env=None):
    e = dict(os.environ, **(env or {}))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"expected ONE line on stdout, got {len(lines)}: {p.stdout[:500]}"
    rec = json.loads(lines[0])
    assert REQUIRED <= rec.keys() and ROOFLINE <= rec["roofline"].keys()
    assert rec["higher_is_better"] is True and rec["scaling"] == "weak" and rec["vs_baseline"] is None and "workload" in rec["config"]
    assert rec["value"] > 0 and 0 < rec["roofline"]["frac"] < 1
    return rec


def test_single_gpu_line_with_both_exact_paths_and_cpu_baseline():
    rec = _run(["--rows", "120000", "--steps", "3", "--warmup", "1", "--no-encoders", "--cpu-seconds", "1"])
    assert rec["n_gpus"] == 1 and rec["config"]["queries_per_step"] == 4096
    assert rec["other_exact_path"]["results_identical_to_headline_path"] is True
    cpu = rec["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= cpu.keys() and cpu["kind"].startswith("port") and cpu["value"] > 0
    assert set(cpu["legs"]) == {"fmaf_chain_oracle", "faiss_organisation"} and cpu["host"]["physical_cores"] >= 1
    ceil = rec["roofline"]["sustained_mfma_ceiling"]          # the register-only MFMA loop timed on the same box
    assert 1000 < ceil["random_operands_tflops"] <= ceil["zero_operands_tflops"] * 1.02 < 2700
    assert 0 < ceil["frac_of_random_operand_ceiling"] < 1


def test_diag_mfma_loop_through_the_c_abi():
    import torch
    from viquae_amd import _lib
    lib = _lib.load()
    out = torch.full((2 * 1024,), 7.0, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.mq_diag_mfma_bf16_loop(3, 0, 2, out.data_ptr(), s), "mq_diag_mfma_bf16_loop")
    torch.cuda.synchronize()
    assert (out == 0).all()                                   # zero operands
    _lib.check(lib.mq_diag_mfma_bf16_loop(3, 1, 2, out.data_ptr(), s), "mq_diag_mfma_bf16_loop")
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and (out != 0).any()
    assert lib.mq_diag_mfma_bf16_loop(3, 1, 0, out.data_ptr(), s) == -1      # MQ_EINVAL


def test_multi_rank_path_through_rccl_world_1():
    rec = _run(["--rows", "120000", "--steps", "2", "--warmup", "1", "--no-encoders", "--no-cpu-baseline"], {"MQ_BENCH_FORCE_DIST": "1"})
    assert rec["config"]["rccl"] == {"backend": "nccl", "ranks_seen": 1, "launched_by": "torch.distributed.run"}
    assert rec["config"]["queries_per_step"] == 16384 and rec["config"]["step_breakdown_ms"]["chunks_per_step"] == 4


def test_two_ranks_spawned_by_bench_itself():
    """`bench.py --gpus 2` with no WORLD_SIZE: the parent starts the ranks.  Two ranks cannot share one GPU under RCCL, so this
    run uses the developer switches (gloo, both ranks on GPU 0); the driver's multi-GPU run uses neither."""
    env = {"MQ_BENCH_SHARE_GPU": "1", "MQ_BENCH_BACKEND": "gloo"}
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "100000", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=dict(env_clean, **env), timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["config"]["rccl"]["ranks_seen"] == 2 and rec["config"]["kb_rows_total"] == 200000
    assert rec["config"]["fixed_total_kb"]["kb_rows_total"] == 100000 and rec["config"]["fixed_total_kb"]["queries_per_s"] > 0
    assert rec["config"]["step_breakdown_ms"]["all_gather_ms"] > 0
