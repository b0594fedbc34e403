#!/bin/bash
# Builds a variant of libmeerqat_hip.so for same-box A/B runs:  tools/ab_build.sh <name> "<-D flags>"  -> ab/lib_<name>.so
# The -DMQ_ABL_* / MQ_PROBE* switches live in tools/ablate/lab_switches.patch (see tools/ablate/README.md): apply it first.
# (ab/ is git-ignored; it travels to the GPU box with gpurun.  Use with MEERQAT_HIP_LIB=ab/lib_<name>.so.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fno-fast-math -ffp-contract=off -Wno-unused-result $2 \
  $R/viquae_amd/csrc/knn.hip $R/viquae_amd/csrc/encoder.hip $R/viquae_amd/csrc/conv.hip $R/viquae_amd/csrc/fuse.hip $R/viquae_amd/csrc/image.hip $R/viquae_amd/csrc/diag.hip $R/viquae_amd/csrc/runfmt.cpp -lpthread -o $R/ab/lib_$1.so
echo built ab/lib_$1.so
