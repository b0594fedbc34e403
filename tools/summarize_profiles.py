"""Condenses gpurun_out/<tag>/ (written by tools/profile_round.sh) into the tracked files under profiles/:
  <tag>_kernel_stats.csv, <tag>_encoders_kernel_stats.csv   rocprofv3 --kernel-trace --stats summaries
  <tag>_pmc.json                                            per kernel and counter: calls, max and mean per launch
  knn_traffic.json                                          what bench.py reports as roofline.traffic
  <tag>_bench_screened.json / <tag>_bench_exact_f32.json    the bench lines of that run
usage: python tools/summarize_profiles.py <tag>"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")


def first(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return hits[-1] if hits else None  # gpurun merges runs into the same directories: take the newest


for sub, name in (("kt", f"{tag}_kernel_stats.csv"), ("enc", f"{tag}_encoders_kernel_stats.csv"), ("arc", f"{tag}_arcface_kernel_stats.csv")):
    f = first(f"{sub}/**/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, name))
f = first("nq256_kt/**/*kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(dst, f"{tag}_nq256_kernel_stats.csv"))
f = first("exact_kt/**/*kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(dst, f"{tag}_exact_kernel_stats.csv"))


def collect(subs):
    pmc = {}
    for sub in subs:
        f = first(f"{sub}/**/*counter_collection.csv")
        if not f:
            continue
        for row in csv.DictReader(open(f)):
            kernel = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            if kernel.startswith("at::") or "rocclr" in kernel or "elementwise" in kernel:
                continue
            c = pmc.setdefault(kernel, {}).setdefault(row["Counter_Name"], {})
            c.setdefault("by_dispatch", {}).setdefault(row["Dispatch_Id"], 0.0)
            c["by_dispatch"][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for kernel, counters in pmc.items():
        for name, c in counters.items():
            vals = list(c.pop("by_dispatch").values())
            c.update(calls=len(vals), max_per_launch=max(vals), mean=sum(vals) / len(vals))
    return pmc


small = collect(("nq256_fetch", "nq256_write", "nq256_sq", "nq256_tcc"))
if small:
    json.dump(small, open(os.path.join(dst, f"{tag}_nq256_pmc.json"), "w"), indent=1)
pmc = {}
for sub in ("fetch", "write", "sq", "tcc", "exact_fetch", "exact_write"):
    f = first(f"{sub}/**/*counter_collection.csv")
    if not f:
        continue
    for row in csv.DictReader(open(f)):
        kernel = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
        if kernel.startswith("at::") or "rocclr" in kernel or "elementwise" in kernel:
            continue
        c = pmc.setdefault(kernel, {}).setdefault(row["Counter_Name"], {})
        c.setdefault("by_dispatch", {}).setdefault(row["Dispatch_Id"], 0.0)
        c["by_dispatch"][row["Dispatch_Id"]] += float(row["Counter_Value"])
for kernel, counters in pmc.items():
    for name, c in counters.items():
        vals = list(c.pop("by_dispatch").values())
        c.update(calls=len(vals), max_per_launch=max(vals), mean=sum(vals) / len(vals))
json.dump(pmc, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)


def traffic(kernel):
    c = pmc.get(kernel) or next((v for k, v in pmc.items() if k.startswith(kernel)), {})
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    fetch_kb, write_kb = c["FETCH_SIZE"]["max_per_launch"], c["WRITE_SIZE"]["max_per_launch"]
    # MI355X_MICROARCH.md (HBM section): FETCH_SIZE is in KB and reports half of a 16 B/lane coalesced stream on gfx950
    return {"kernel": kernel, "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024), "fetch_kb_raw": fetch_kb,
            "write_kb_raw": write_kb}


out = {"_profile": f"profiles/{tag}_pmc.json", "_note": "L2<->fabric bytes per full-size launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH "
                "doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streams; Infinity-Cache hits are included, so this is "
                f"an upper bound of HBM bytes.  Raw counters: profiles/{tag}_pmc.json."}
for key, kernel in (("screened_1500000x768_nq4096_k100", "screen_scan_kernel"), ("exact_f32_1500000x768_nq4096_k100", "knn_scan_kernel<0, false, false>")):
    t = traffic(kernel)
    if t:
        out[key] = t
# the 256-query search (its own rocprofv3 passes): the streaming scan reads every row once
small_name = next((k_ for k_ in (small or {}) if k_.startswith("screen_small")), None)  # screen_small8_kernel<12> since round 5
c = small[small_name] if small_name else None
if c and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    out["screened_1500000x768_nq256_k100"] = {"kernel": small_name, "profile": f"profiles/{tag}_nq256_pmc.json",
                                              "hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"]["max_per_launch"] + c["WRITE_SIZE"]["max_per_launch"]) * 1024),
                                              "fetch_kb_raw": c["FETCH_SIZE"]["max_per_launch"], "write_kb_raw": c["WRITE_SIZE"]["max_per_launch"]}
if len(out) > 1:
    json.dump(out, open(os.path.join(dst, "knn_traffic.json"), "w"), indent=1)
def stats_avg_ms(path, prefix):
    """average launch duration of the real (> 0.1 ms) launches of a kernel in a rocprofv3 kernel_stats.csv"""
    if not os.path.exists(path):
        return None
    best = None
    for row in csv.DictReader(open(path)):
        name = row["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if name.startswith(prefix) and float(row["AverageNs"]) > 1e5 and (best is None or float(row["TotalDurationNs"]) > float(best["TotalDurationNs"])):
            best = row
    return round(float(best["AverageNs"]) / 1e6, 4) if best else None


for a, b, stats, prefix in (("bench.json", f"{tag}_bench_screened.json", f"{tag}_kernel_stats.csv", "screen_scan_kernel"),
                            ("bench_exact.json", f"{tag}_bench_exact_f32.json", f"{tag}_exact_kernel_stats.csv", "knn_scan_kernel")):
    f = os.path.join(src, a)
    if os.path.exists(f) and os.path.getsize(f):
        rec = json.load(open(f))
        # the bench ran on the box BEFORE these summaries existed: it quoted the previous tracked file; point it at this round's
        ms = stats_avg_ms(os.path.join(dst, stats), prefix)
        if ms is not None and "roofline" in rec:
            rec["roofline"]["kernel_ms_profile"], rec["roofline"]["kernel_ms_profile_from"] = ms, f"profiles/{stats}"
            t = out.get(("screened" if prefix.startswith("screen") else "exact_f32") + "_1500000x768_nq4096_k100")
            if t:
                rec["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
        json.dump(rec, open(os.path.join(dst, b), "w"))
print(json.dumps({k: {n: round(c["max_per_launch"], 1) for n, c in v.items()} for k, v in pmc.items()
                  if k.startswith(("screen_scan_kernel", "knn_scan_kernel<0", "rescore_kernel", "cand_select_kernel"))}, indent=1))
