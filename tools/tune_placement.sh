#!/bin/bash
# Sweeps MQ_PAD_NOPS (code placement of the MFMA loop in knn_scan_kernel) on a GPU box:
#   gpurun -- 'bash tools/tune_placement.sh'      (build variants first, here, with: bash tools/tune_placement.sh build)
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fno-fast-math -ffp-contract=off"
mkdir -p ab
if [ "$1" = build ]; then
  for n in 0 1 2 3 4 5 6 7; do hipcc $F -DMQ_PAD_NOPS=$n viquae_amd/csrc/*.hip -o ab/lib_pad$n.so & done; wait; exit 0
fi
for n in 0 1 2 3 4 5 6 7; do
  echo -n "pad$n "
  MEERQAT_HIP_LIB=$PWD/ab/lib_pad$n.so python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])"
done
