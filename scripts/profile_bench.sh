#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box via gpurun):
#   1. --kernel-trace --stats  -> per-kernel durations
#   2. --pmc FETCH_SIZE        -> HBM read traffic   (own pass)
#   3. --pmc WRITE_SIZE        -> HBM write traffic  (own pass)
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries into profiles/.
set -u
TAG=${1:-r01}
STEPS=${2:-5}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps $STEPS --warmup 1 --cpu-seconds 0 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $CMD > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $CMD > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -50
for f in $(find "$OUT/trace" -name "*kernel_stats.csv"); do echo "== $f"; cat "$f"; done
tail -2 "$OUT/trace.log"
