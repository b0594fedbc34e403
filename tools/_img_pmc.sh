cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/img_pmc
mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $O/a --output-format csv -- python3 /root/repo/tools/bench_image.py > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/b --output-format csv -- python3 /root/repo/tools/bench_image.py > $O/b.log 2>&1
cd /root/repo
python3 - <<'PY'
import csv, glob, collections
for tag in "ab":
    for f in glob.glob(f"gpurun_out/img_pmc/{tag}/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[-2].split("::")[-1] if "resample" in r["Kernel_Name"] else None
            if not k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, d in agg.items():
            print(tag, k, {c: f"{v:.3e}" for c, v in d.items()})
PY
tail -2 $O/a.log | cut -c1-300
