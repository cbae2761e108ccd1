#!/bin/bash
# Kernel traces of the single-light-curve chains (BASELINE configs[1], [2], [4]) through the device sampler:
#   bash scripts/profile_chains.sh r02     (on the GPU box, via gpurun)
# -> gpurun_out/<tag>_small_{1,2}_{stats.csv,trace.txt}, <tag>_tp_kernel_stats.csv, <tag>_tp_halfstep_trace.txt,
#    <tag>_c5_sweep.txt; copy the ones to keep into profiles/.
TAG=${1:-r03}
REPO=$PWD
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for k in 1 2; do
  rocprofv3 --kernel-trace -d $OUT/prof_${TAG}_s$k -o s$k -- python3 $REPO/scripts/small_trace.py $k > $OUT/${TAG}_small_$k.log 2>&1 || exit 1
  f=$(find $OUT/prof_${TAG}_s$k -name "*.db" | head -1)
  python3 $REPO/scripts/rocpd_kernels.py $f > $OUT/${TAG}_small_${k}_stats.csv
  python3 $REPO/scripts/rocpd_kernels.py $f --trace 12 > $OUT/${TAG}_small_${k}_trace.txt
  grep configs $OUT/${TAG}_small_$k.log
done
rocprofv3 --kernel-trace -d $OUT/prof_${TAG}_c5 -o c5 -- python3 $REPO/scripts/c5_trace.py > $OUT/${TAG}_c5.log 2>&1 || exit 1
f=$(find $OUT/prof_${TAG}_c5 -name "*.db" | head -1)
python3 $REPO/scripts/rocpd_kernels.py $f > $OUT/${TAG}_tp_kernel_stats.csv
python3 $REPO/scripts/rocpd_kernels.py $f --trace 24 > $OUT/${TAG}_tp_halfstep_trace.txt
tail -1 $OUT/${TAG}_c5.log
cd $REPO && python3 scripts/c5_sweep.py 8 16 32 64 128 256 512 1024 | grep config5 > $OUT/${TAG}_c5_sweep.txt
cat $OUT/${TAG}_c5_sweep.txt
