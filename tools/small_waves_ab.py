#!/usr/bin/env python3
"""A/B of the two streaming kernels for one query tile -- one wave per SIMD (csrc/knn_small.inc, MQ_KNN_OPT_SMALL_WAVES = 4) against
two (csrc/knn_small8.inc, = 8) -- and the 256 x 256 tile kernel, same process, same shard: HIP events around the scan kernel of one
C-ABI search call, results compared bit for bit.   usage: python tools/small_waves_ab.py [d ...]   (ROWS, NQ from the environment)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viquae_amd import _lib
from viquae_amd.index import MI355XFlatIndex

rows, k, reps = int(os.environ.get("ROWS", 1_500_000)), 100, 30
dev = torch.device("cuda")
lib = _lib.load()
out = {}
for d in [int(x) for x in sys.argv[1:]] or [768, 512]:
    for nq in [int(x) for x in os.environ.get("NQ", "256").split(",")]:
        g = torch.Generator(device=dev).manual_seed(d)
        idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
        step = (1 << 26) // d // 64 * 64
        for s in range(0, rows, step):
            idx.add(torch.randn((min(step, rows - s), d), generator=g, device=dev), total_hint=rows)
        Q = torch.randn((nq, d), generator=g, device=dev)
        stream = torch.cuda.current_stream()
        wsb = int(lib.mq_knn_workspace_bytes_metric(rows, d, nq, k, 0))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        res = {}

        def run(D, I, e0=None, e1=None):
            _lib.check(lib.mq_knn_search_screened_f32(None, idx._sqnorm.data_ptr(), idx._rowmajor.data_ptr(), idx._bf16.data_ptr(),
                                                      idx._xmax2.data_ptr(), rows, d, Q.data_ptr(), nq, k, idx._screen_metric, 0, 0, D.data_ptr(), I.data_ptr(),
                                                      ws.data_ptr(), wsb, stream.cuda_stream, e0.cuda_event if e0 else None,
                                                      e1.cuda_event if e1 else None), "search")

        for name, small, waves in (("tile", 0, 8), ("stream4", 1, 4), ("stream8", 1, 8)):
            lib.mq_knn_set_option(_lib.KNN_OPT_SMALL_SCAN, small)
            lib.mq_knn_set_option(_lib.KNN_OPT_SMALL_WAVES, waves)
            D = torch.empty((nq, k), dtype=torch.float32, device=dev)
            I = torch.empty((nq, k), dtype=torch.int64, device=dev)
            for _ in range(5):
                run(D, I)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in evs:
                a.record(stream); b.record(stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for a, b in evs:
                run(D, I, a, b)
            torch.cuda.synchronize()
            call = (time.perf_counter() - t0) / reps * 1e3
            scan = sum(a.elapsed_time(b) for a, b in evs) / reps
            idx._ws = idx._last_ws = ws
            idx._last_call_nq = nq
            st = idx.screen_stats(nq, k)
            res[name] = {"scan_ms": round(scan, 4), "call_ms": round(call, 4), "hbm_frac": round(rows * ((d + 63) // 64 * 64) * 2 / (scan * 1e-3) / 8e12, 4),
                         "kind": int(lib.mq_knn_screen_scan_kind(rows, d, nq, k, 0)), "recomputed": st[0], "cand_per_q": round(st[1] / nq, 1),
                         "pool_keys_per_q": round(st[4] / nq, 1)}
            res[name + "_DI"] = (D, I)
        for name in ("stream4", "stream8"):
            res[name]["equal_to_tile"] = bool(torch.equal(res[name + "_DI"][0], res["tile_DI"][0]) and torch.equal(res[name + "_DI"][1], res["tile_DI"][1]))
        out[f"d{d}_nq{nq}"] = {n: v for n, v in res.items() if not n.endswith("_DI")}
        lib.mq_knn_set_option(_lib.KNN_OPT_SMALL_SCAN, 1)
        lib.mq_knn_set_option(_lib.KNN_OPT_SMALL_WAVES, 8)
        del idx, ws
        torch.cuda.empty_cache()
for cfg, r in out.items():
    print(cfg, json.dumps(r))
