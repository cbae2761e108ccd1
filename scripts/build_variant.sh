#!/bin/bash
# build_variant.sh NAME "-DFLAG=..."  ->  gpurun_out/variants/libmtg_NAME.so  (kernel A/B experiments)
set -e
NAME=$1; shift
OUT=/root/repo/variants
mkdir -p $OUT
cd /root/repo/mind_the_gaps_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 "$@" -shared -o $OUT/libmtg_$NAME.so mtg_kernels.hip mtg_sampler.hip mtg_simulate.hip mtg_timeparallel.hip mtg_capi.hip -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib
echo built $OUT/libmtg_$NAME.so
