#!/bin/bash
# kernel stats of the single-light-curve chains (BASELINE configs[1], [2]) under rocprofv3
OUT=$PWD/gpurun_out/prof_small; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/scripts/small_probe.py > $OUT/trace.log 2>&1
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cut -c1-200 $f | head -20; done
tail -5 $OUT/trace.log
