#!/bin/bash
# per-kernel durations of the rank-10 half-step at B = 8, 32, 256 rows (scripts/c5_one.py): gpurun -- bash scripts/c5_breakdown.sh
OUT=$PWD/gpurun_out/prof_c5b; rm -rf $OUT; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for B in 8 32 256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b$B -- python3 $REPO/scripts/c5_one.py $B 6 > $OUT/b$B.log 2>&1
  echo "== B = $B: $(tail -1 $OUT/b$B.log)"
  for f in $(find $OUT/b$B -name "*kernel_stats.csv"); do cut -c1-150 $f | head -12; done
done
