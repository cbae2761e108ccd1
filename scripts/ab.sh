#!/bin/bash
# run perf_probe for the default lib and every variants/*.so, interleaved 2 rounds
for round in 1 2; do
  echo "== default"; python scripts/perf_probe.py $@ 2>&1 | grep -v "^$" | awk "NR%3==0"
  for f in variants/*.so; do echo "== $f"; MTG_HIP_LIB=$PWD/$f python scripts/perf_probe.py $@ 2>&1 | grep -v "^$" | awk "NR%3==0"; done
done
