#!/bin/bash
# Profiles of one round on the GPU box (run through gpurun):  tools/profile_round.sh <tag>
#   kernel trace + stats of the default bench, then one PMC pass per counter group (never combined with tracing),
#   every rocprofv3 run bounded by `timeout`, python3 directly after `--`.
# Outputs: gpurun_out/<tag>/{kt,fetch,write,sq,tcc}/... ; summarise with tools/summarize_profiles.py <tag>.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 8 --warmup 2 --headline-only"   # only the headline launches: the stats' average IS the scan of the 4096-query step
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- $B > $O/kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- $B > $O/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- $B > $O/write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/sq --output-format csv -- $B > $O/sq.log 2>&1
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/tcc --output-format csv -- $B > $O/tcc.log 2>&1
# the exact fp32 scan, the same way (its own passes: in the screened runs above it only appears as the fallback's no-op launches)
BE="python3 $R/bench.py --mode exact_f32 --steps 3 --warmup 1 --headline-only"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/exact_kt --output-format csv -- $BE > $O/exact_kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/exact_fetch --output-format csv -- $BE > $O/exact_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/exact_write --output-format csv -- $BE > $O/exact_write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/enc --output-format csv -- python3 $R/tools/bench_encoders.py > $O/enc.log 2>&1
# the reference's own call size: 256 queries per search (one query tile -> the streaming scan, csrc/knn_small.inc)
S="python3 $R/tools/small_batch_search.py 256 50"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/nq256_kt --output-format csv -- $S > $O/nq256_kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/nq256_fetch --output-format csv -- $S > $O/nq256_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/nq256_write --output-format csv -- $S > $O/nq256_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/nq256_sq --output-format csv -- $S > $O/nq256_sq.log 2>&1
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/nq256_tcc --output-format csv -- $S > $O/nq256_tcc.log 2>&1
cd $R
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python3 bench.py --mode exact_f32 --no-encoders --no-other-path > $O/bench_exact.json 2>> $O/bench.err
ls $O
