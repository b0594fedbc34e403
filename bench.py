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

N > 1 (BASELINE configs[4]: 12M x 768 over 8 GPUs, 16k queries): the KB is
row-sharded, one 1.5M-row shard per rank (weak scaling: per-GPU work fixed), 16,384 replicated
queries per step; a step = ShardedFlatIndex.search_device = per 4096-query chunk {local scan into the
rank's shard record, ONE RCCL all-gather of the records (async, overlapping the next chunk's scan),
merge on every rank}.  `value` counts the units all ranks processed: one unit = one query's exact
top-100 over one 1.5M x 768 shard (the BASELINE metric's unit), so value = N * nq * K / t; the
end-to-end rate of finished queries over the N x 1.5M KB is `global_queries_per_s`.  Rank 0 also
reports the per-chunk scan / all-gather / merge times of an un-overlapped pass, the RCCL rank count
it saw, and the same search with the TOTAL KB fixed at 1.5M rows (strong scaling).

`python bench.py --gpus N` without WORLD_SIZE in the environment starts the N ranks itself (a
`torch.distributed.run` child process, before this process touches any GPU) and relays their line.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

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
    ap.add_argument("--nq", type=int, default=None, help="queries per step (default: 4096 on one GPU = configs[1]; 16384 on several = configs[4])")
    ap.add_argument("--mode", choices=["screened", "exact_f32"], default="screened",
                    help="screened: bf16 screening scan + exact fp32 re-scoring (default, same results); exact_f32: fp32 MFMA scan")
    ap.add_argument("--no-other-path", action="store_true", help="do not time the other exact path next to the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encoders", action="store_true", help="skip the secondary encoder throughput figures")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed headline steps (no CPU baseline, encoders, other path, small-batch / big-k legs): the command "
                         "tools/profile_round.sh profiles, so that the rocprofv3 average of the scan kernel is the headline launch's")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample duration")
    args = ap.parse_args()
    if args.headline_only:
        args.no_cpu_baseline = args.no_encoders = args.no_other_path = True
    return args


def build_shard(idx, rows, seed, device):
    """Synthetic shard generated ON DEVICE (Philox, per-shard seed) and packed batch by batch."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    step = 1 << 16
    for s in range(0, rows, step):
        n = min(step, rows - s)
        idx.add(torch.randn((n, DIM), generator=g, device=device, dtype=torch.float32), total_hint=rows)


def mfma_ceiling(device, stream, achieved_tflops):
    """What the bf16 matrix pipe of THIS box sustains with no operand movement at all (csrc/diag.hip: 16 waves per CU looping
    over v_mfma_f32_32x32x16_bf16 on register operands), zero operands vs N(0,1)-like random ones: the power management lowers
    the clock on random data, so `peak` (the nominal 2.5 PFLOP/s) is not reachable by any kernel on the bench's operands.
    Measured outside the timed region; `frac` above stays relative to the nominal peak."""
    import torch
    from viquae_amd import _lib
    lib = _lib.load()
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    out = torch.empty(cus * 1024, dtype=torch.float32, device=device)
    iters = 8000
    flops = 2.0 * 32 * 32 * 16 * 16 * iters * 16 * cus
    res = {}
    for name, rnd in (("zero_operands", 0), ("random_operands", 1)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        times = []
        for rep in range(3):  # the first launch settles the clocks
            e0.record(stream)
            _lib.check(lib.mq_diag_mfma_bf16_loop(iters, rnd, cus, out.data_ptr(), stream.cuda_stream), "mq_diag_mfma_bf16_loop")
            e1.record(stream)
            e1.synchronize()
            times.append(e0.elapsed_time(e1))
        res[name + "_tflops"] = round(flops / (min(times[1:]) * 1e-3) / 1e12, 1)
    res["frac_of_random_operand_ceiling"] = round(achieved_tflops / res["random_operands_tflops"], 4)
    res["what"] = ("mq_diag_mfma_bf16_loop: register-only bf16 MFMA loop on all CUs (no LDS, no HBM); the random-operand figure is the "
                   "ceiling of any bf16 GEMM-shaped kernel on this data under the chip's power limit")
    return res


def host_cpus():
    """Physical cores, hardware threads and the threads this process may use -- stated ONCE for every CPU leg."""
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    physical = None
    try:
        cores = set()
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
        physical = len(cores) or None
    except OSError:
        pass
    # the container's CPU bandwidth (cgroup `cpu.max` = quota period): the GPU boxes of this pool show 256 hardware threads
    # and grant 16 CPUs' worth of time -- every CPU leg below is a number measured UNDER that quota, whatever its thread count
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
        if q != "max":
            quota = round(int(q) / int(period), 2)
    except (OSError, ValueError):
        pass
    return {"physical_cores": physical or logical, "hardware_threads": logical, "usable_threads": usable,
            "cgroup_cpu_quota": quota}


def cpu_baseline(idx, Q, seconds):
    """CPU legs on the GPU box's host cores, each on a BOUNDED sample of the same workload (ALL queries of the step, so
    every core has work, against the first `rows` KB rows sized by a calibration run), scaled by rows to the metric's
    unit (queries/s over the full 1.5M x 768 KB):
      * fmaf_chain_oracle:    oracle/knn_oracle.c, the bit-exact checker (OpenMP over 32-query blocks);
      * faiss_organisation:   oracle.knn.knn_blas = FAISS's own organisation for >= 20 queries: blocks of 4096 queries x
                              1024 database rows through the host BLAS's sgemm (torch.mm = MKL, all cores), then FAISS's
                              strict-'>' heap rule per query in C / OpenMP (oracle_heap_add_block) -- the faster leg, and
                              the one `value` reports.
    Neither is FAISS itself.  When `import faiss` works on the box (then "the reference path verbatim"), a
    third leg runs first and wins: IndexFlatIP.add + .search in the reference's own call shape -- batches of 256 queries
    (experiments/ir/viquae/dpr/search/config.json:25) -- kind = "faiss"; it never is in this image."""
    from oracle import knn as ok
    import torch
    cpus = host_cpus()
    threads = ok.num_threads()
    Qh = Q.cpu().numpy()
    nq = Qh.shape[0]

    def leg(fn, cal_rows, budget, name):
        X = idx.reconstruct_n(0, min(idx.ntotal, cal_rows))
        fn(X[:256], Qh)  # warm the thread pool
        t0 = time.perf_counter()
        fn(X, Qh)
        tc = time.perf_counter() - t0
        rows_s = int(min(idx.ntotal, max(X.shape[0], X.shape[0] * budget / max(tc, 1e-4))))
        if rows_s > X.shape[0]:
            X = idx.reconstruct_n(0, rows_s)
            t0 = time.perf_counter()
            fn(X, Qh)
            tc = time.perf_counter() - t0
        value = (nq / tc) * rows_s / idx.ntotal  # the same queries against the full KB cost ntotal/rows_s more
        return {"value": round(value, 2), "seconds": round(tc, 2), "kb_rows_sampled": rows_s,
                "gflops": round(2.0 * nq * rows_s * DIM / tc / 1e9, 1), "what": name}

    used_threads = threads
    legs = {"fmaf_chain_oracle": leg(lambda X, Qq: ok.knn(X, Qq, TOPK, metric=0), 4096, seconds * 0.5,
                                     f"oracle/knn_oracle.c (k-ordered fmaf chain = the bit-exact checker), OpenMP x{threads}")}
    try:
        # which host BLAS, how many threads, which database block: a 1-second calibration on the box (MKL takes a slow code
        # path on AMD hosts; numpy's OpenBLAS is capped at 64 threads; FAISS's own 1024-row blocks starve 128 threads)
        Xc = idx.reconstruct_n(0, min(idx.ntotal, 1 << 17))
        trials = []
        for backend, nthreads, block in (("torch", cpus["physical_cores"], 1024), ("torch", cpus["physical_cores"], 16384),
                                         ("torch", max(1, cpus["physical_cores"] // 2), 16384), ("numpy", None, 1024), ("numpy", None, 16384),
                                         ("c", cpus["physical_cores"], 16384), ("c", max(1, cpus["physical_cores"] // 2), 16384),
                                         ("c", max(1, cpus["physical_cores"] // 2), 65536)):
            try:
                ok.knn_blas(Xc[:2048], Qh, TOPK, metric=0, block=block, backend=backend, threads=nthreads)
                t0 = time.perf_counter()
                ok.knn_blas(Xc, Qh, TOPK, metric=0, block=block, backend=backend, threads=nthreads)
                trials.append((time.perf_counter() - t0, backend, nthreads, block))
            except Exception:
                pass
        _, backend, nthreads, block = min(trials)
        used_threads = nthreads or threads
        legs["faiss_organisation"] = leg(lambda X, Qq: ok.knn_blas(X, Qq, TOPK, metric=0, block=block, backend=backend, threads=nthreads), 1 << 16, seconds * 0.7,
                                         f"oracle.knn.knn_blas: 4096 x {block} sgemm blocks on the host BLAS ({backend}"
                                         f"{': oracle_sgemm_nt, OpenMP x%d' % nthreads if backend == 'c' else ', %d threads' % nthreads if nthreads else ' / OpenBLAS'}; the fastest of "
                                         f"{[(b, n, bl, round(2.0 * nq * Xc.shape[0] * DIM / t / 1e9)) for t, b, n, bl in trials]} (backend, threads, block, "
                                         f"GFLOP/s)) + FAISS's strict-'>' heap per query in C / OpenMP x{threads}")
    except Exception as e:
        legs["faiss_organisation"] = {"value": 0.0, "what": f"failed: {e!r}"}
    best = max(legs, key=lambda n: legs[n]["value"])
    try:
        import faiss  # the reference's own dependency (requirements.txt:14); absent from this image
        faiss.omp_set_num_threads(cpus["usable_threads"])

        def faiss_leg(X, Qq):
            index = faiss.IndexFlatIP(X.shape[1])
            index.add(np.ascontiguousarray(X, np.float32))
            for s0 in range(0, Qq.shape[0], 256):
                index.search(np.ascontiguousarray(Qq[s0:s0 + 256], np.float32), TOPK)
        legs["faiss"] = leg(faiss_leg, 1 << 16, seconds * 0.7, f"faiss {getattr(faiss, '__version__', '?')} IndexFlatIP.add + .search in batches "
                            f"of 256 queries, {cpus['usable_threads']} OpenMP threads (add included)")
        best, used_threads = "faiss", cpus["usable_threads"]
    except ImportError:
        pass
    ran_on = used_threads if best in ("faiss_organisation", "faiss") else threads
    quota = cpus.get("cgroup_cpu_quota")
    rec = {
        # `threads_of_reported_leg` = the threads the reported leg's matrix product ran on (its heap pass and the chain oracle use
        # every OpenMP thread); `cores` = the CPUs' worth of time those threads could actually get: the box shows 256 hardware
        # threads but its cgroup grants `cgroup_cpu_quota` CPUs (16 on this pool) -- VERDICT r4 weak 8: the headline record must say so
        "value": legs[best]["value"], "unit": "queries/s",
        "cores": int(min(ran_on, math.ceil(quota))) if quota else ran_on, "threads": threads, "threads_of_reported_leg": ran_on,
        "cgroup_cpu_quota": quota, "physical_cores": cpus["physical_cores"], "hardware_threads": cpus["hardware_threads"],
        "kind": "faiss" if best == "faiss" else "port (FAISS organisation)" if best == "faiss_organisation" else "port",
        "host": cpus,
        "sample": f"{best}: {legs[best]['what']}; {nq} queries x the first {legs[best]['kb_rows_sampled']} KB rows, top-{TOPK}: "
                  f"{legs[best]['seconds']} s ({legs[best]['gflops']} GFLOP/s); scaled by rows to {idx.ntotal} x {DIM}",
        "legs": legs,
    }
    return rec


def cpu_encoder_baseline(n=64):
    """the reference's OWN CPU encode path -- Hugging Face `DPRContextEncoder` /
    `CLIPModel.get_image_features` (what meerqat/ir/embedding.py:226 and meerqat/image/embedding.py:156-161 call) on torch-CPU
    fp32, all cores, `n` passages of 100 tokens / `n` 224 x 224 images, default-config random weights -- when `transformers`
    is importable on the box (it is in this image); otherwise the numpy restatement oracle/encoders.py (`kind: "port"`)."""
    import torch
    cpus = host_cpus()
    rng = np.random.default_rng(0)
    ids = rng.integers(1000, 30000, (n, 100)).astype(np.int64)
    px = rng.standard_normal((n, 3, 224, 224)).astype(np.float32)
    try:
        import transformers
        torch.set_num_threads(cpus["physical_cores"])
        out = {"cores": cpus["physical_cores"], "threads": torch.get_num_threads(), "kind": "hf-transformers torch-cpu",
               "sample": f"transformers {transformers.__version__} DPRContextEncoder / CLIPModel.get_image_features, default configs "
                         f"(bert-base / ViT-B/32), random weights, fp32, {n} passages x 100 tokens / {n} images, one batch"}
        with torch.no_grad():
            dpr = transformers.DPRContextEncoder(transformers.DPRConfig()).eval()
            t_ids = torch.from_numpy(ids)
            dpr(input_ids=t_ids[:4], attention_mask=torch.ones_like(t_ids[:4]))
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                dpr(input_ids=t_ids, attention_mask=torch.ones_like(t_ids))
                best = min(best, time.perf_counter() - t0)
            out["dpr_passages_per_s"] = round(n / best, 2)
            del dpr
            clip = transformers.CLIPModel(transformers.CLIPConfig()).eval()
            t_px = torch.from_numpy(px)
            clip.get_image_features(pixel_values=t_px[:4])
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                clip.get_image_features(pixel_values=t_px)
                best = min(best, time.perf_counter() - t0)
            out["clip_images_per_s"] = round(n / best, 2)
        return out
    except Exception as e:  # noqa: BLE001 - fall back to the port
        why = repr(e)
    from oracle import encoders as oe
    out = {"cores": cpus["physical_cores"], "threads": cpus["usable_threads"], "kind": "port",
           "sample": f"oracle/encoders.py (numpy fp32 on the host BLAS), {n} passages x 100 tokens / {n} images; transformers unusable: {why}"}
    state = oe.seeded_state(oe.bert_param_shapes(oe.BERT_BASE), 1)
    oe.bert_forward(state, oe.BERT_BASE, ids[:4], None, np.ones_like(ids[:4]))
    t0 = time.perf_counter()
    oe.bert_forward(state, oe.BERT_BASE, ids, None, np.ones_like(ids))
    out["dpr_passages_per_s"] = round(n / (time.perf_counter() - t0), 2)
    del state
    state = oe.seeded_state(oe.clip_vision_param_shapes(oe.CLIP_VITB32), 2)
    oe.clip_vision_forward(state, oe.CLIP_VITB32, px[:4])
    t0 = time.perf_counter()
    oe.clip_vision_forward(state, oe.CLIP_VITB32, px)
    out["clip_images_per_s"] = round(n / (time.perf_counter() - t0), 2)
    return out


def load_traffic(workload_key):
    """(bytes per scan launch, provenance).  PMC counters cannot be read from inside this process: the figure is the one
    committed with the rocprofv3 --pmc passes of THIS command (profiles/knn_traffic.json, written by
    tools/summarize_profiles.py; FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE), tagged with its profile run."""
    p = os.path.join(ROOT, "profiles", "knn_traffic.json")
    try:
        with open(p) as f:
            t = json.load(f)
        e = t.get(workload_key, {})
        return e.get("hbm_bytes_per_launch"), (f"profiles/knn_traffic.json[{workload_key}] from {e.get('profile', t.get('_profile', '?'))}"
                                                if e else None)
    except Exception:
        return None, None


def profile_kernel_ms(kernel_prefix, suffix="_kernel_stats.csv"):
    """Average launch duration (ms) of the kernel whose name starts with `kernel_prefix` in the newest TRACKED rocprofv3 summary
    profiles/rNN<suffix> -- the number DESIGN.md quotes, shown beside the HIP-event figure of this run (they come from different
    boxes and one of them ran under the profiler: a few per cent apart).  (None, None) when there is no such file / row."""
    import csv
    import glob
    import re
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]" + suffix))]
    if not files:
        return None, None
    f = max(files, key=lambda p: int(re.search(r"r(\d+)", os.path.basename(p)).group(1)))
    try:
        best = None
        for row in csv.DictReader(open(f)):
            name = row["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            if name.startswith(kernel_prefix) and float(row["AverageNs"]) > 1e5:  # (the fallback's no-op launches share a name)
                if best is None or float(row["TotalDurationNs"]) > float(best["TotalDurationNs"]):
                    best = row
        if best is None:
            return None, None
        return round(float(best["AverageNs"]) / 1e6, 4), os.path.relpath(f, ROOT)
    except Exception:
        return None, None


def spawn_ranks(args):
    """`bench.py --gpus N` started by hand: run the N ranks as children of a torch.distributed.run launcher.  Nothing in
    THIS process has touched a GPU (torch.cuda.device_count() does not initialise HIP on this image), and it never will:
    it only waits for the launcher and passes the ranks' JSON line through."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("MQ_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
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
    # Developer switches for exercising the N > 1 code path on a ONE-GPU box (never set by the driver): every rank on GPU 0
    # (RCCL refuses two ranks on one device, hence the gloo backend, which stages through host memory)
    if os.environ.get("MQ_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("MQ_BENCH_BACKEND", "nccl")
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
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    lib = _lib.load()

    multi = world > 1 or force_dist
    rows, k = args.rows, TOPK
    nq = args.nq or (4 * NQ if multi else NQ)
    # keep_panel: the exact fp32 scan is timed beside the screened search on the same shard (a screened index alone keeps no
    # panel copy; its fallback launch is the same no-op either way)
    local = MI355XFlatIndex(device=local_rank, string_factory="Flat", metric_type=0, id_offset=rank * rows, screen=True,
                            keep_panel=True)
    build_shard(local, rows, seed=rank, device=device)
    index = ShardedFlatIndex(string_factory="Flat", metric_type=0, local_index=local, always_gather=force_dist) if multi else None
    if index is not None:
        index.ntotal, index.d = rows * world, DIM
    g = torch.Generator(device=device)
    g.manual_seed(100)
    Q = torch.randn((nq, DIM), generator=g, device=device, dtype=torch.float32)  # same on every rank

    stream = torch.cuda.current_stream(device)
    nqc = min(nq, NQ)  # queries per C-ABI call (the screened scan's 4096-query chunk)
    ws_bytes = int(lib.mq_knn_workspace_bytes(rows, DIM, nqc, k))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    D = torch.empty((nqc, k), dtype=torch.float32, device=device)
    I = torch.empty((nqc, k), dtype=torch.int64, device=device)

    mode = args.mode
    def local_step(ev0=None, ev1=None, which=None, q=None, out=None):
        """ONE C-ABI search call over (at most) 4096 queries with HIP events around its dominant kernel."""
        e0 = ev0.cuda_event if ev0 is not None else None
        e1 = ev1.cuda_event if ev1 is not None else None
        q = Q[:nqc] if q is None else q
        Dq, Iq = (D, I) if out is None else out
        if (which or mode) == "screened":
            _lib.check(lib.mq_knn_search_screened_f32(
                local._packed.data_ptr() if local._packed is not None else None, local._sqnorm.data_ptr(), local._rowmajor.data_ptr(), local._bf16.data_ptr(),
                local._xmax2.data_ptr(), rows, DIM, q.data_ptr(), q.shape[0], k, local._screen_metric, 0, local.id_offset, Dq.data_ptr(), Iq.data_ptr(),
                ws.data_ptr(), ws_bytes, stream.cuda_stream, e0, e1), "mq_knn_search_screened_f32")
        else:
            _lib.check(lib.mq_knn_search_f32_ev(local._packed.data_ptr(), local._sqnorm.data_ptr(), rows, DIM, q.data_ptr(),
                                                q.shape[0], k, 0, 0, local.id_offset, Dq.data_ptr(), Iq.data_ptr(), ws.data_ptr(),
                                                ws_bytes, stream.cuda_stream, e0, e1), "mq_knn_search_f32_ev")
        return Dq, Iq

    if multi:
        local.screen = mode == "screened"
        local._ws = ws  # the index's own search_device reuses the bench workspace

    def step(ev0=None, ev1=None):
        if not multi:
            return local_step(ev0, ev1)
        return index.search_device(Q, k)  # scan chunks | async all-gather | merge, software-pipelined

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

    if multi:
        # un-overlapped pass: per 4096-query chunk, HIP events around the scan kernel (through the C ABI), the collective
        # and the merge -- what the pipelined step hides is the difference to `ms_per_step`
        from viquae_amd.sharded import record_layout, _record_views, _hip_merge_records
        nb = 3
        chunks = [(s0, min(s0 + nqc, nq)) for s0 in range(0, nq, nqc)]
        tev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
        marks = []
        for _ in range(nb):
            for s0, e0_ in chunks:
                n = e0_ - s0
                rec_bytes, _ = record_layout(n, k)
                record = torch.empty(rec_bytes, dtype=torch.uint8, device=device)
                gathered = torch.empty(world * rec_bytes, dtype=torch.uint8, device=device)
                Dv, Iv = _record_views(record, n, k)
                ka, kb_, a0, a1, a2, a3 = tev(), tev(), tev(), tev(), tev(), tev()
                ka.record(stream); kb_.record(stream)
                a0.record(stream)
                local_step(ka, kb_, q=Q[s0:e0_], out=(Dv, Iv))
                a1.record(stream)
                dist.all_gather_into_tensor(gathered, record)
                a2.record(stream)
                _hip_merge_records(gathered, world, n, k, 0)
                a3.record(stream)
                marks.append((ka, kb_, a0, a1, a2, a3))
        torch.cuda.synchronize()
        marks = marks[len(chunks):]  # first pass = warm-up
        per = lambda i, j: sum(m[i].elapsed_time(m[j]) for m in marks) / len(marks)  # noqa: E731
        scan_ms = per(0, 1)
        breakdown = {"per_chunk_of_queries": nqc, "chunks_per_step": len(chunks), "scan_call_ms": round(per(2, 3), 3),
                     "scan_kernel_ms": round(scan_ms, 3), "all_gather_ms": round(per(3, 4), 3), "merge_ms": round(per(4, 5), 3),
                     "all_gather_bytes_per_rank": record_layout(nqc, k)[0],
                     "unoverlapped_ms_per_step": round(len(chunks) * per(2, 5), 3)}
        t = torch.tensor([breakdown["all_gather_ms"], breakdown["merge_ms"], breakdown["scan_call_ms"]], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        breakdown["max_over_ranks"] = {"all_gather_ms": round(float(t[0]), 3), "merge_ms": round(float(t[1]), 3), "scan_call_ms": round(float(t[2]), 3)}
    else:
        scan_ms = sum(a.elapsed_time(b) for a, b in evs) / args.steps
        breakdown = None

    # strong scaling ("N total fixed"): the SAME 1.5M-row KB cut into `world` shards
    fixed_total = None
    if world > 1:
        from viquae_amd.sharded import shard_bounds
        lo, hi = shard_bounds(rows, world, rank)
        small = MI355XFlatIndex(device=local_rank, string_factory="Flat", metric_type=0, id_offset=lo, screen=mode == "screened")
        gs = torch.Generator(device=device)
        gs.manual_seed(1000 + rank)
        for s0 in range(0, hi - lo, 1 << 16):
            small.add(torch.randn((min(1 << 16, hi - lo - s0), DIM), generator=gs, device=device), total_hint=hi - lo)
        sidx = ShardedFlatIndex(string_factory="Flat", metric_type=0, local_index=small)
        sidx.ntotal, sidx.d = rows, DIM
        sidx.search_device(Q, k)
        torch.cuda.synchronize()
        dist.barrier()
        ts = time.perf_counter()
        n_s = max(2, min(5, args.steps))
        for _ in range(n_s):
            sidx.search_device(Q, k)
        torch.cuda.synchronize()
        dist.barrier()
        ts = torch.tensor([time.perf_counter() - ts], dtype=torch.float64, device=device)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        fixed_total = {"kb_rows_total": rows, "kb_rows_per_gpu": hi - lo, "queries_per_step": nq, "steps": n_s,
                       "ms_per_step": round(float(ts.item()) / n_s * 1e3, 3),
                       "queries_per_s": round(nq * n_s / float(ts.item()), 1)}
        del small, sidx

    # the other exact path, timed next to the headline (rank 0, N = 1): a few steps are enough
    # The reference's own call size: Dataset.map hands KnowledgeBase.search_batch 256 questions at a time
    # (experiments/ir/viquae/dpr/search/config.json:25 -> meerqat/ir/search.py:146).  One query tile = the KB is streamed once per
    # search: the regime north_star's "HBM-read roofline" is about.  Same shard, same C-ABI call, HIP events around the scan.
    small_batch = None
    if world == 1 and mode == "screened" and not args.headline_only:
        try:
            nqs, reps = 256, 30
            qs = Q[:nqs].contiguous()
            outs = (torch.empty((nqs, k), dtype=torch.float32, device=device), torch.empty((nqs, k), dtype=torch.int64, device=device))
            for _ in range(5):
                local_step(q=qs, out=outs)
            evs_s = make_events(reps)
            torch.cuda.synchronize()
            ts0 = time.perf_counter()
            for a_, b_ in evs_s:
                local_step(a_, b_, q=qs, out=outs)
            torch.cuda.synchronize()
            call_ms = (time.perf_counter() - ts0) / reps * 1e3
            k_ms = sum(a_.elapsed_time(b_) for a_, b_ in evs_s) / reps
            kb_bytes = rows * DIM * 2  # the bf16 screening copy, read once
            kind = int(lib.mq_knn_screen_scan_kind(rows, DIM, nqs, k, 0))
            # the same search through the 256 x 256 tile kernel (MQ_KNN_OPT_SMALL_SCAN = 0 through mq_knn_set_option), same process, for the A/B the notes quote
            with _lib.knn_option(_lib.KNN_OPT_SMALL_SCAN, 0):
                for _ in range(3):
                    local_step(q=qs, out=outs)
                evs_t = make_events(reps)
                torch.cuda.synchronize()
                for a_, b_ in evs_t:
                    local_step(a_, b_, q=qs, out=outs)
                torch.cuda.synchronize()
                tile_ms = sum(a_.elapsed_time(b_) for a_, b_ in evs_t) / reps
            # ... and through round 4's streaming kernel (one wave per SIMD, csrc/knn_small.inc), same process
            with _lib.knn_option(_lib.KNN_OPT_SMALL_WAVES, 4):
                for _ in range(3):
                    local_step(q=qs, out=outs)
                evs_4 = make_events(reps)
                torch.cuda.synchronize()
                for a_, b_ in evs_4:
                    local_step(a_, b_, q=qs, out=outs)
                torch.cuda.synchronize()
                one_wave_ms = sum(a_.elapsed_time(b_) for a_, b_ in evs_4) / reps
            waves = int(lib.mq_knn_get_option(_lib.KNN_OPT_SMALL_WAVES))
            small_name = "screen_small8_kernel" if waves == 8 else "screen_small_kernel"
            small_batch = {
                "workload": f"{nqs} queries (the reference's Dataset.map batch) x {rows}x{DIM} KB, exact IP top-{k}, one C-ABI call",
                "scan_kernel": {0: "none", 1: "screen_scan_kernel (256 x 256 tiles)",
                                2: f"{small_name}<12> (queries in registers, LDS ring of 32-row tiles, "
                                   f"{'two waves' if waves == 8 else 'one wave'} per SIMD)"}.get(kind, str(kind)),
                "ms": round(call_ms, 4), "queries_per_s": round(nqs / call_ms * 1e3, 1),
                "scan_kernel_ms": round(k_ms, 4),
                "algorithmic_hbm_bytes": kb_bytes, "achieved_gbps": round(kb_bytes / (k_ms * 1e-3) / 1e9, 1),
                "hbm_frac": round(kb_bytes / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                "mfma_frac": round(2.0 * nqs * rows * DIM / (k_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                "tile_kernel_scan_ms": round(tile_ms, 4),
                "one_wave_per_simd_kernel_scan_ms": round(one_wave_ms, 4),
                "scan_kernel_ms_profile": profile_kernel_ms(small_name, "_nq256_kernel_stats.csv")[0],
                "scan_kernel_ms_profile_from": profile_kernel_ms(small_name, "_nq256_kernel_stats.csv")[1],
                "traffic": load_traffic("screened_1500000x768_nq256_k100")[0],
                "traffic_from_profile": load_traffic("screened_1500000x768_nq256_k100")[1],
            }
        except Exception as e:  # never a reason to lose the line
            small_batch = {"error": repr(e)}
    # k beyond the screen's own range (`--k` is a user option of the reference, meerqat/ir/search.py:12): the same 4096 queries at
    # k = 256 and 512, served over row ranges (csrc/knn.hip partition_plan) where round 3 ran ceil(k / 128) exact scans
    big_k = None
    if world == 1 and mode == "screened" and not args.no_other_path and not args.headline_only:
        try:
            big_k = {"workload": f"{nqc} queries x {rows}x{DIM} KB, exact IP top-k through MI355XFlatIndex.search_device", "ms": {}}
            keep_ws = local._ws
            for kk in (256, 512):
                local.search_device(Q[:nqc], kk)
                torch.cuda.synchronize()
                tb0 = time.perf_counter()
                for _ in range(3):
                    local.search_device(Q[:nqc], kk)
                torch.cuda.synchronize()
                big_k["ms"][str(kk)] = round((time.perf_counter() - tb0) / 3 * 1e3, 3)
                big_k.setdefault("scan_kind", {})[str(kk)] = local.scan_kind(nqc, kk)
            big_k["x_k100_step"] = {kk: None for kk in big_k["ms"]}
            local._ws = keep_ws
            torch.cuda.empty_cache()
        except Exception as e:
            big_k = {"error": repr(e)}
    other = None
    if world == 1 and not args.no_other_path:
        which = "exact_f32" if mode == "screened" else "screened"
        D_head, I_head = D.clone(), I.clone()
        n2 = max(2, min(5, args.steps))
        local_step(which=which)
        torch.cuda.synchronize()
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
        flops = 2.0 * nqc * rows * DIM  # algorithmic FLOPs of one scan launch (2.304 GFLOP/query)
        achieved = flops / (scan_ms * 1e-3) / 1e12
        info = (ctypes_i64 * 8)()
        lib.mq_knn_launch_info(rows, DIM, nqc, k, info)
        workload = (f"{rows}x{DIM} fp32 KB per GPU, {nq} queries, exact IP top-{k}" if not multi else
                    f"{rows * world}x{DIM} fp32 KB row-sharded over {world} GPU(s) ({rows} rows each), {nq} replicated queries per step, "
                    f"exact IP top-{k} (BASELINE configs[4] shape)")
        if mode == "screened":
            peak, kernel = PEAK_BF16_MFMA_TFLOPS, "screen_scan_kernel (v_mfma_f32_32x32x16_bf16, relaxed top-k fused)"
            alg_bytes = rows * DIM * 2  # bf16 copy of the shard, one pass
            dtype = "bf16 screen + f32 exact re-score"
        else:
            peak, kernel = PEAK_F32_MFMA_TFLOPS, "knn_scan_kernel<IP> (v_mfma_f32_32x32x2_f32, top-k fused)"
            alg_bytes = rows * DIM * 4
            dtype = "f32"
        traffic = load_traffic(f"{mode}_{rows}x{DIM}_nq{nqc}_k{k}")
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
                "sharding": "single GPU" if not multi else f"row-sharded x{world}: per 4096-query chunk one RCCL all-gather of the shard records "
                            f"{{f32 score, i64 id}}[nq,{k}] (async, overlapping the next chunk's scan) + merge on every rank",
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
                "kernel_ms_profile": profile_kernel_ms(kernel.split("<")[0].split(" ")[0], "_kernel_stats.csv" if mode == "screened" else "_exact_kernel_stats.csv")[0],
                "kernel_ms_profile_from": profile_kernel_ms(kernel.split("<")[0].split(" ")[0], "_kernel_stats.csv" if mode == "screened" else "_exact_kernel_stats.csv")[1],
                "algorithmic_flops_per_launch": flops,
                "algorithmic_hbm_bytes_per_launch": alg_bytes,
                "hbm_frac_at_one_pass": round(alg_bytes / (scan_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                "traffic": traffic[0],
                "traffic_from_profile": traffic[1],
            },
        }
        if world == 1 and mode == "screened":
            try:
                rec["roofline"]["sustained_mfma_ceiling"] = mfma_ceiling(device, stream, achieved)
            except Exception as e:  # a diagnostic, never a reason to lose the line
                rec["roofline"]["sustained_mfma_ceiling"] = {"error": repr(e)}
        if multi:
            rec["config"]["rccl"] = {"backend": dist.get_backend(), "ranks_seen": dist.get_world_size(),
                                     "launched_by": "torch.distributed.run"}
            rec["config"]["step_breakdown_ms"] = breakdown
        if fixed_total is not None:
            rec["config"]["fixed_total_kb"] = fixed_total
        if mode == "screened":
            local._ws, keep = ws, local._ws
            st = local.screen_stats(nqc, k)
            local._ws = keep
            rec["config"]["screen"] = {"query_tiles_recomputed_exactly": st[0], "candidates_rescored_per_query": round(st[1] / nqc, 1),
                                       "max_candidates_of_a_query": st[2]}
        if small_batch is not None:
            rec.setdefault("secondary", {})["small_batch"] = small_batch
        if big_k is not None:
            if "ms" in big_k:
                big_k["x_k100_step"] = {kk: round(v / rec["ms_per_step"], 2) for kk, v in big_k["ms"].items()}
            rec.setdefault("secondary", {})["big_k"] = big_k
        if other is not None:
            rec["other_exact_path"] = other
            if other["path"] == "exact_f32":
                a2 = flops / (other["kernel_ms"] * 1e-3) / 1e12
                other["roofline"] = {"bound": "mfma", "kernel": "knn_scan_kernel<IP> (v_mfma_f32_32x32x2_f32)", "achieved": round(a2, 2),
                                     "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(a2 / PEAK_F32_MFMA_TFLOPS, 4)}
        if world == 1 and not args.no_encoders:
            # secondary BASELINE figures (configs[2], configs[3]); the headline `value` stays queries/s
            try:
                # the shard itself stays (the CPU legs at the end read its rows back); its search workspace and whatever the
                # caching allocator kept from the headline steps go back to the device first
                local._ws = None
                del ws
                torch.cuda.empty_cache()
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_encoders
                d = bench_encoders.dpr_throughput(B=2048, L=100, steps=5)   # (2 steps right after the headline: 104-122 ms; 5: steady)
                try:
                    dp = bench_encoders.dpr_padded_throughput()
                    dq = bench_encoders.dpr_padded_throughput(mean_len=16, std_len=5, steps=5)
                except Exception as e:
                    dp = dq = {"error": repr(e)}
                c = bench_encoders.clip_throughput(B=3072, steps=5)
                tt = bench_encoders.clip_text_throughput(B=2048, L=77, steps=5)
                try:
                    ec = bench_encoders.eca_throughput()
                    eca = {"workload": "ECAEncoder as shipped (experiments/mm/eca/config.yaml: bert-base, n_faces 0, one clip-RN50 feature of "
                                       "1024 dims; KB config: batch 2048, text padded to 256, ~130 real tokens) -- text + image token",
                           "passages_per_s": round(ec["passages_per_s"], 1), "ms_per_batch": round(ec["ms_per_batch"], 2)}
                except Exception as e:
                    eca = {"error": repr(e)}
                try:
                    af = bench_encoders.arcface_throughput()
                    arc = {"workload": f"ArcFace r50 (IResNet-50), {af['batch']} aligned 112x112 faces per batch, seeded weights; 3x3 convolutions = "
                                       "implicit split-bf16 GEMMs (mq_conv3x3_pair_f32, csrc/conv.hip), stem = direct fp32 convolution, downsamples / head = im2col + GEMM",
                           "faces_per_s": round(af["faces_per_s"], 1), "ms_per_batch": round(af["ms_per_batch"], 2),
                           "algorithmic_tflops": round(af["tflops"], 2),
                           "executed_bf16_mfma_frac": round(3 * af["tflops"] / PEAK_BF16_MFMA_TFLOPS, 4)}
                except Exception as e:
                    arc = {"error": repr(e)}
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
                # the headline's workload on DPR-LIKE data (zero-mean Gaussians flatter the screen): every
                # vector = a shared direction of norm 9 + isotropic noise of sd 0.25 (scores ~81 +- a few units, like real DPR
                # inner products); same index class, same search call, exact results
                try:
                    gd = torch.Generator(device=device)
                    gd.manual_seed(7)
                    mu = torch.randn((1, DIM), generator=gd, device=device)
                    mu = 9.0 * mu / mu.norm()
                    dpr_idx = MI355XFlatIndex(device=local_rank, string_factory="Flat", metric_type=0, screen=True)
                    for s0 in range(0, rows, 1 << 16):
                        dpr_idx.add(mu + 0.25 * torch.randn((min(1 << 16, rows - s0), DIM), generator=gd, device=device), total_hint=rows)
                    Qd = mu + 0.25 * torch.randn((nq, DIM), generator=gd, device=device)
                    dpr_idx.search_device(Qd, k)
                    torch.cuda.synchronize()
                    td = time.perf_counter()
                    for _ in range(5):
                        dpr_idx.search_device(Qd, k)
                    torch.cuda.synchronize()
                    td = (time.perf_counter() - td) / 5
                    std = dpr_idx.screen_stats(nq, k)
                    dpr_like = {"workload": f"{rows}x{DIM} KB of shared-direction (norm 9) + N(0, 0.25^2) noise vectors, {nq} such queries, exact IP "
                                            f"top-{k} (screened path, bf16 copy centred on the KB mean)",
                                "queries_per_s": round(nq / td, 1), "ms_per_step": round(td * 1e3, 3),
                                "candidates_rescored_per_query": round(std[1] / nq, 1), "query_tiles_recomputed_exactly": std[0]}
                    del dpr_idx, Qd
                except Exception as e:
                    dpr_like = {"error": repr(e)}
                try:
                    import bench_host_path
                    surface = bench_host_path.main(rows=rows)
                except Exception as e:
                    surface = {"error": repr(e)}
                try:
                    import bench_encode_surface
                    encode_surface = bench_encode_surface.main()
                except Exception as e:
                    encode_surface = {"error": repr(e)}
                try:
                    import fusion_config_search
                    fusion_search = fusion_config_search.main(rows=rows)
                except Exception as e:
                    fusion_search = {"error": repr(e)}
                try:
                    import small_batch_l2
                    l2_small = small_batch_l2.main(rows=rows)
                except Exception as e:
                    l2_small = {"error": repr(e)}
                rec.setdefault("secondary", {}).update({
                    "small_batch_l2": l2_small,
                    "fusion_config_search": fusion_search,
                    "reference_call_surface": surface,
                    "encode_call_surface": encode_surface,
                    "dpr_like_data": dpr_like,
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
                    "eca_multimodal_encoder": eca,
                    "arcface": arc,
                    "titles_encoded_per_s": round(tt["titles_per_s"], 1),
                    "clip_text": {"workload": "CLIP ViT-B/32 text tower, 2048 x 77 synthetic tokens per batch (causal)",
                                  "ms_per_batch": round(tt["ms_per_batch"], 2), "algorithmic_tflops": round(tt["tflops"], 2)},
                })
            except Exception as e:
                rec.setdefault("secondary", {})["error"] = repr(e)
        # The CPU legs come LAST: whatever they leave behind in the process (128-thread BLAS / OpenMP pools, Hugging Face
        # models) cost the host-bound call-surface legs a third of their rate when it ran before them (map_arrow 300 k -> 170-200 k
        # queries/s, measured); the GPU legs above do not care, the CPU legs neither.
        if world == 1 and not args.no_cpu_baseline:
            try:
                rec["cpu_baseline"] = cpu_baseline(local, Q, args.cpu_seconds)
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                rec["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port (FAISS organisation)",
                                       "sample": f"failed: {e!r}"}
            if not args.no_encoders:
                try:
                    rec["cpu_baseline"]["encoders"] = cpu_encoder_baseline()
                except Exception as e:
                    rec["cpu_baseline"]["encoders"] = {"error": repr(e)}
        os.write(json_fd, (json.dumps(rec) + "\n").encode())
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


import ctypes  # noqa: E402

ctypes_i64 = ctypes.c_int64

if __name__ == "__main__":
    main()
