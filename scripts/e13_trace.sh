#!/bin/bash
# where the E13 adjustment's time goes at configs[3] size: rocprofv3 kernel stats over scripts/e13_probe.py
REPO=$PWD; OUT=$REPO/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e13_trace -o t -- python3 $REPO/scripts/e13_probe.py 32 > $OUT/e13_trace.log 2>&1
grep "light curves" $OUT/e13_trace.log
f=$(find $OUT/e13_trace -name "*kernel_stats.csv" | head -1); head -22 $f | cut -c1-160
