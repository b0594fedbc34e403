#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): exact inner-product top-100 of 4096 fp32 queries against a
synthetic 1.5M x 768 fp32 KB resident in HBM, on ONE MI355X.  A "step" is one search call over the
whole batch; inputs are already in HBM when the timed region starts, D/I are written inside it.
Two exact paths exist and return bit-identical results (checked in every run):
  --mode screened (default): mq_knn_search_screened_f32 = bf16 MFMA screening scan with a provable
      error margin + exact fp32 re-scoring of the survivors (csrc/knn_screen.inc);
  --mode exact_f32: mq_knn_search_f32 = fp32 MFMA scan with the top-k fused (csrc/knn.hip).
The headline `value` is the selected mode; the other path is timed next to it (`other_exact_path`).

N > 1 (BASELINE configs[4] shape): the KB is row-sharded, one 1.5M-row
shard per rank (weak scaling: per-GPU work fixed), queries replicated; each step = local scan +
one RCCL all-gather of the per-shard [nq,100] lists + merge on every rank.  `value` counts the
units all ranks processed: one unit = one query's exact top-100 over one 1.5M x 768 shard (the
BASELINE metric's unit), so value = N * nq * K / t; the end-to-end rate of finished queries over
the N x 1.5M KB is reported next to it as `global_queries_per_s`.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KB_ROWS = 1_500_000
DIM = 768
NQ = 4096
TOPK = 100
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak
PEAK_HBM_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=KB_ROWS, help="KB rows per GPU (default: BASELINE size)")
    ap.add_argument("--nq", type=int, default=NQ)
    ap.add_argument("--mode", choices=["screened", "exact_f32"], default="screened",
                    help="screened: bf16 screening scan + exact fp32 re-scoring (default, same results); exact_f32: fp32 MFMA scan")
    ap.add_argument("--no-other-path", action="store_true", help="do not time the other exact path next to the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encoders", action="store_true", help="skip the secondary encoder throughput figures")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample duration")
    return ap.parse_args()


def build_shard(idx, rows, seed, device):
    """Synthetic shard generated ON DEVICE (Philox, per-shard seed) and packed batch by batch."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    step = 1 << 16
    for s in range(0, rows, step):
        n = min(step, rows - s)
        idx.add(torch.randn((n, DIM), generator=g, device=device, dtype=torch.float32), total_hint=rows)


def cpu_baseline(idx, Q, seconds):
    """Times the CPU oracle (oracle/knn_oracle.c, OpenMP over 32-query blocks) on a bounded sample of
    the same workload: ALL queries of the step (so every host core has a block) against the first
    `rows_s` KB rows, sized by a calibration run for ~`seconds` of CPU work, then scaled by rows to
    the metric's unit (queries/s over the full 1.5M x 768 KB)."""
    from oracle import knn as ok
    threads = ok.num_threads()
    Qh = Q.cpu().numpy()
    nq = Qh.shape[0]
    cal_rows = min(idx.ntotal, 4096)
    X = idx.reconstruct_n(0, cal_rows)
    ok.knn(X[:256], Qh, TOPK, metric=0)  # warm the thread pool
    t0 = time.perf_counter()
    ok.knn(X, Qh, TOPK, metric=0)
    tc = time.perf_counter() - t0
    rows_s = int(min(idx.ntotal, max(cal_rows, cal_rows * seconds / max(tc, 1e-4))))
    X = idx.reconstruct_n(0, rows_s)
    t0 = time.perf_counter()
    ok.knn(X, Qh, TOPK, metric=0)
    t = time.perf_counter() - t0
    value = (nq / t) * rows_s / idx.ntotal  # the same queries against the full KB cost ntotal/rows_s more
    return {
        "value": round(value, 2), "unit": "queries/s", "cores": threads, "kind": "port",
        "sample": f"oracle/knn_oracle.c (fmaf-chain restatement of FAISS IndexFlatIP, OpenMP x{threads}) on {nq} queries x "
                  f"the first {rows_s} KB rows, top-{TOPK}: {t:.2f} s ({2.0 * nq * rows_s * DIM / t / 1e9:.0f} GFLOP/s); "
                  f"scaled by rows to {idx.ntotal} x {DIM}",
    }


def load_traffic(workload_key):
    """HBM bytes per scan launch from a committed rocprofv3 --pmc pass (profiles/knn_traffic.json)."""
    p = os.path.join(ROOT, "profiles", "knn_traffic.json")
    try:
        with open(p) as f:
            t = json.load(f)
        return t.get(workload_key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    args = parse()
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to the C-level stdout when the process
    # exits: keep a private handle on the real stdout for the JSON line and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from viquae_amd import _lib
    from viquae_amd.index import MI355XFlatIndex
    from viquae_amd.sharded import ShardedFlatIndex

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: viquae_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # MQ_BENCH_FORCE_DIST=1: go through the RCCL all-gather + shard merge even with one rank (exercises the
    # N > 1 code path on a 1-GPU box)
    force_dist = os.environ.get("MQ_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)
    lib = _lib.load()

    rows, nq, k = args.rows, args.nq, TOPK
    local = MI355XFlatIndex(device=local_rank, string_factory="Flat", metric_type=0, id_offset=rank * rows, screen=True)
    build_shard(local, rows, seed=rank, device=device)
    index = ShardedFlatIndex(string_factory="Flat", metric_type=0, local_index=local) if (world > 1 or force_dist) else None
    if index is not None:
        index.ntotal = rows * world
    g = torch.Generator(device=device)
    g.manual_seed(100)
    Q = torch.randn((nq, DIM), generator=g, device=device, dtype=torch.float32)  # same on every rank

    stream = torch.cuda.current_stream(device)
    ws_bytes = int(lib.mq_knn_workspace_bytes(rows, DIM, nq, k))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    D = torch.empty((nq, k), dtype=torch.float32, device=device)
    I = torch.empty((nq, k), dtype=torch.int64, device=device)

    mode = args.mode
    def local_step(ev0=None, ev1=None, which=None):
        e0 = ev0.cuda_event if ev0 is not None else None
        e1 = ev1.cuda_event if ev1 is not None else None
        if (which or mode) == "screened":
            _lib.check(lib.mq_knn_search_screened_f32(
                local._packed.data_ptr(), local._sqnorm.data_ptr(), local._rowmajor.data_ptr(), local._bf16.data_ptr(),
                local._xmax2.data_ptr(), rows, DIM, Q.data_ptr(), nq, k, 0, 0, local.id_offset, D.data_ptr(), I.data_ptr(),
                ws.data_ptr(), ws_bytes, stream.cuda_stream, e0, e1), "mq_knn_search_screened_f32")
        else:
            _lib.check(lib.mq_knn_search_f32_ev(local._packed.data_ptr(), local._sqnorm.data_ptr(), rows, DIM, Q.data_ptr(),
                                                nq, k, 0, 0, local.id_offset, D.data_ptr(), I.data_ptr(), ws.data_ptr(),
                                                ws_bytes, stream.cuda_stream, e0, e1), "mq_knn_search_f32_ev")
        return D, I

    def step(ev0=None, ev1=None):
        Dl, Il = local_step(ev0, ev1)
        if world == 1 and not force_dist:
            return Dl, Il
        Ds = torch.empty((world * nq, k), dtype=Dl.dtype, device=device)
        Is = torch.empty((world * nq, k), dtype=Il.dtype, device=device)
        dist.all_gather_into_tensor(Ds, Dl)
        dist.all_gather_into_tensor(Is, Il)
        return index.merge_fn(Ds.view(world, nq, k), Is.view(world, nq, k), 0)

    def make_events(n):
        evs_ = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in evs_:  # HIP events are created by a first record on the launch stream
            a.record(stream)
            b.record(stream)
        return evs_

    for _ in range(args.warmup):
        step()
    # HIP events that bracket the dominant kernel on the stream it is launched on
    evs = make_events(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(*evs[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    scan_ms = sum(a.elapsed_time(b) for a, b in evs) / args.steps

    # the other exact path, timed next to the headline (rank 0, N = 1): a few steps are enough
    other = None
    if world == 1 and not args.no_other_path:
        which = "exact_f32" if mode == "screened" else "screened"
        D_head, I_head = D.clone(), I.clone()
        n2 = max(2, min(5, args.steps))
        local_step(which=which)
        evs2 = make_events(n2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for a, b in evs2:
            local_step(a, b, which=which)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        other = {"path": which, "value": round(nq * n2 / el2, 1), "unit": "queries/s", "ms_per_step": round(el2 / n2 * 1e3, 3),
                 "kernel_ms": round(sum(a.elapsed_time(b) for a, b in evs2) / n2, 3),
                 "results_identical_to_headline_path": bool(torch.equal(D, D_head) and torch.equal(I, I_head))}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        units = world * nq * args.steps
        value = units / elapsed
        flops = 2.0 * nq * rows * DIM  # algorithmic FLOPs of one scan launch (2.304 GFLOP/query)
        achieved = flops / (scan_ms * 1e-3) / 1e12
        info = (ctypes_i64 * 8)()
        lib.mq_knn_launch_info(rows, DIM, nq, k, info)
        workload = f"{rows}x{DIM} fp32 KB per GPU, {nq} queries, exact IP top-{k}"
        if mode == "screened":
            peak, kernel = PEAK_BF16_MFMA_TFLOPS, "screen_scan_kernel (v_mfma_f32_32x32x16_bf16, relaxed top-k fused)"
            alg_bytes = rows * DIM * 2  # bf16 copy of the shard, one pass
            dtype = "bf16 screen + f32 exact re-score"
        else:
            peak, kernel = PEAK_F32_MFMA_TFLOPS, "knn_scan_kernel<IP> (v_mfma_f32_32x32x2_f32, top-k fused)"
            alg_bytes = rows * DIM * 4
            dtype = "f32"
        rec = {
            "metric": "queries/sec exact top-100 over 1.5M x 768 KB",
            "value": round(value, 1),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic (torch Philox randn generated on device, seed = shard rank; queries seed 100)",
            "config": {
                "workload": workload,
                "path": mode,
                "kb_rows_total": rows * world,
                "queries_per_step": nq,
                "k": k,
                "sharding": "single GPU" if world == 1 else f"row-sharded x{world}, RCCL all-gather of per-shard top-{k} + merge",
                "unit_definition": "one query's exact top-100 over one 1.5M x 768 shard",
                "global_queries_per_s": round(nq * args.steps / elapsed, 1),
                "scan_launch": {"workgroups": int(info[0]), "threads": int(info[6] if mode == "screened" else info[1]),
                                "lds_bytes": int(info[7] if mode == "screened" else info[2]),
                                "query_tiles": int(info[3]), "kb_slabs": int(info[4]), "kb_chunks": int(info[5])},
            },
            "roofline": {
                "bound": "mfma",
                "kernel": kernel,
                "achieved": round(achieved, 2),
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4),
                "kernel_ms": round(scan_ms, 3),
                "algorithmic_flops_per_launch": flops,
                "algorithmic_hbm_bytes_per_launch": alg_bytes,
                "hbm_frac_at_one_pass": round(alg_bytes / (scan_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                "traffic": load_traffic(f"{mode}_{rows}x{DIM}_nq{nq}_k{k}"),
            },
        }
        if mode == "screened":
            local._ws, keep = ws, local._ws
            st = local.screen_stats(nq, k)
            local._ws = keep
            rec["config"]["screen"] = {"query_tiles_recomputed_exactly": st[0], "candidates_rescored_per_query": round(st[1] / nq, 1),
                                       "max_candidates_of_a_query": st[2]}
        if other is not None:
            rec["other_exact_path"] = other
            if other["path"] == "exact_f32":
                a2 = flops / (other["kernel_ms"] * 1e-3) / 1e12
                other["roofline"] = {"bound": "mfma", "kernel": "knn_scan_kernel<IP> (v_mfma_f32_32x32x2_f32)", "achieved": round(a2, 2),
                                     "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(a2 / PEAK_F32_MFMA_TFLOPS, 4)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                rec["cpu_baseline"] = cpu_baseline(local, Q, args.cpu_seconds)
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                rec["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        if world == 1 and not args.no_encoders:
            # secondary BASELINE figures (configs[2], configs[3]); the headline `value` stays queries/s
            try:
                del local, ws
                torch.cuda.empty_cache()
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_encoders
                d = bench_encoders.dpr_throughput(B=2048, L=100, steps=2)
                try:
                    dp = bench_encoders.dpr_padded_throughput()
                    dq = bench_encoders.dpr_padded_throughput(mean_len=16, std_len=5, steps=5)
                except Exception as e:
                    dp = dq = {"error": repr(e)}
                c = bench_encoders.clip_throughput(B=3072, steps=2)
                tt = bench_encoders.clip_text_throughput(B=2048, L=77, steps=2)
                try:
                    import bench_image
                    ip = bench_image.main()
                except Exception as e:
                    ip = {"error": repr(e)}
                # BASELINE configs[3], search half: 512-d CLIP vectors, "L2norm,Flat" + inner product, 4096-query chunks
                g3 = torch.Generator(device=device)
                g3.manual_seed(3)
                clip_idx = MI355XFlatIndex(device=local_rank, string_factory="L2norm,Flat", metric_type=0, screen=True)
                for s0 in range(0, rows, 1 << 16):
                    clip_idx.add(torch.randn((min(1 << 16, rows - s0), 512), generator=g3, device=device), total_hint=rows)
                Q3 = torch.randn((nq, 512), generator=g3, device=device)
                clip_idx.search_device(Q3, k)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                for _ in range(5):
                    clip_idx.search_device(Q3, k)
                torch.cuda.synchronize()
                t3 = (time.perf_counter() - t3) / 5
                del clip_idx
                rec["secondary"] = {
                    "clip_kb_search": {"workload": f"{rows}x512 'L2norm,Flat' inner-product KB, {nq} queries, exact top-{k} (screened path)",
                                       "queries_per_s": round(nq / t3, 1), "ms_per_step": round(t3 * 1e3, 3)},
                    "kb_passages_encoded_per_s": round(d["passages_per_s"], 1),
                    "gemm_arithmetic": os.environ.get("MQ_ENC_GEMM", "split_bf16") + " (split_bf16 = 3 bf16 MFMA products per fp32 product, fp32-class accuracy; parity <= 1e-3 vs HF goldens)",
                    "dpr": {"workload": "DPR bert-base, 2048 x 100 synthetic tokens per batch", "ms_per_batch": round(d["ms_per_batch"], 2),
                            "algorithmic_tflops": round(d["tflops"], 2), "x_f32_mfma_peak": round(d["tflops"] / PEAK_F32_MFMA_TFLOPS, 3),
                            "executed_bf16_mfma_frac": round(3 * d["tflops"] / PEAK_BF16_MFMA_TFLOPS, 4)},
                    "dpr_reference_padding": dict(dp, workload="DPR bert-base, 2048 passages padded to max_length 256 as the reference's tokenization_kwargs do (synthetic lengths ~N(130,30)): padding-aware forward vs dense, identical outputs"),
                    "dpr_questions_reference_padding": dict(dq, workload="same with question-like lengths ~N(16,5), padded to 256 (experiments/ir/viquae/dpr/questions/config.json)"),
                    "images_encoded_per_s": round(c["images_per_s"], 1),
                    "clip": {"workload": "CLIP ViT-B/32, 3072 x 224x224 synthetic images per batch", "ms_per_batch": round(c["ms_per_batch"], 2),
                             "algorithmic_tflops": round(c["tflops"], 2), "x_f32_mfma_peak": round(c["tflops"] / PEAK_F32_MFMA_TFLOPS, 3),
                             "executed_bf16_mfma_frac": round(3 * c["tflops"] / PEAK_BF16_MFMA_TFLOPS, 4)},
                    "image_preprocess": dict(ip, workload="Pillow-exact bicubic resize + crop + normalise of 3072 decoded RGB images on the device (csrc/image.hip)"),
                    "titles_encoded_per_s": round(tt["titles_per_s"], 1),
                    "clip_text": {"workload": "CLIP ViT-B/32 text tower, 2048 x 77 synthetic tokens per batch (causal)",
                                  "ms_per_batch": round(tt["ms_per_batch"], 2), "algorithmic_tflops": round(tt["tflops"], 2)},
                }
            except Exception as e:
                rec["secondary"] = {"error": repr(e)}
        os.write(json_fd, (json.dumps(rec) + "\n").encode())
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


import ctypes  # noqa: E402

ctypes_i64 = ctypes.c_int64

if __name__ == "__main__":
    main()
