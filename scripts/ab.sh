#!/bin/bash
# run perf_probe for the default lib and every mind_the_gaps_amd/libmtg_var_*.so, interleaved 2 rounds
for round in 1 2; do
  echo "== default"; python scripts/perf_probe.py $@ 2>&1 | grep -v "^$" | awk "NR%3==0"
  for f in mind_the_gaps_amd/libmtg_var_*.so; do echo "== $f"; MTG_HIP_LIB=$PWD/$f python scripts/perf_probe.py $@ 2>&1 | grep -v "^$" | awk "NR%3==0"; done
done
