#!/bin/bash
# kernel durations of small chains with the tables copied (default) and computed per workgroup (MTG_MEASURE build)
REPO=$PWD
OUT=$REPO/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MTG_HIP_LIB=$REPO/mind_the_gaps_amd/libmtg_var_measure.so
: > $OUT/tables_trace.txt
for mode in 1 0; do
  for c in null c1; do
    export MTG_GLOBAL_TABLES=$mode
    timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tt_${c}_$mode -o t -- python3 $REPO/scripts/chain_case.py $c 2000 > $OUT/tt_${c}_$mode.log 2>&1 || { echo "rocprofv3 failed ($c, $mode)" >> $OUT/tables_trace.txt; exit 1; }
    echo "== $c tables=$mode: $(grep 'iterations/s' $OUT/tt_${c}_$mode.log)" >> $OUT/tables_trace.txt
    f=$(find $OUT/tt_${c}_$mode -name "*kernel_stats.csv" | head -1); head -6 $f | cut -d, -f1-8 >> $OUT/tables_trace.txt
  done
done
cat $OUT/tables_trace.txt
