#!/bin/bash
# Round-6 profiles at the final HEAD (run on the GPU box via gpurun; gpurun_out/r06/ is scratch, keepers are copied to profiles/):
#   bash scripts/profile_r06.sh bench    rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes over the bench command
#   bash scripts/profile_r06.sh chains   kernel stats of the small chains (tutorial null / alt, configs[1], [2]) + sampler phase stamps
#   bash scripts/profile_r06.sh c5       per-kernel durations of the rank-10 half-step at 8 / 32 / 256 rows + its SQ counters at 256
WHAT=${1:-bench}
REPO=$PWD
OUT=$REPO/gpurun_out/r06
HEAD=$(cat $REPO/.git_head 2>/dev/null || echo unknown)
mkdir -p $OUT
case $WHAT in
bench)
  bash scripts/profile_bench.sh r06 5 > $OUT/profile_bench.log 2>&1
  python3 scripts/summarize_profile.py r06 > $OUT/summarize.log 2>&1
  tail -30 $OUT/summarize.log ;;
chains)
  cd /tmp && export TMPDIR=/tmp
  F=$OUT/r06_chain_kernel_stats.txt
  echo "# HEAD $HEAD -- rocprofv3 --kernel-trace --stats over scripts/chain_case.py CASE 2000 (derive_posteriors on the device sampler, one MI355X)" > $F
  for c in null alt c1 c2; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain_$c -o t -- python3 $REPO/scripts/chain_case.py $c 2000 > $OUT/chain_$c.log 2>&1 || { echo "rocprofv3 failed ($c)" >> $F; exit 1; }
    echo "== $c: $(grep 'iterations/s' $OUT/chain_$c.log) (under the profiler)" >> $F
    f=$(find $OUT/chain_$c -name "*kernel_stats.csv" | head -1); head -5 $f | cut -c1-170 >> $F
  done
  cd $REPO
  (echo "# HEAD $HEAD -- scripts/sampler_stamps.sh (a -DMTG_SAMPLER_STAMPS build in a scratch copy: phases inside the speculative sampler kernel, 10 ns ticks)"
   timeout -k 10 600 bash scripts/sampler_stamps.sh 2>&1 | grep -v amdgpu.ids) > $OUT/r06_sampler_stamps.txt
  cat $F $OUT/r06_sampler_stamps.txt ;;
c5)
  (echo "# HEAD $HEAD -- scripts/c5_breakdown.sh: rank-10 half-step (5 x SHO, N = 2e5) per kernel, rocprofv3 --kernel-trace --stats over scripts/c5_one.py B 6"
   timeout -k 10 500 bash scripts/c5_breakdown.sh 2>&1 | grep -v amdgpu.ids) > $OUT/r06_c5_breakdown.txt
  (echo "# HEAD $HEAD -- scripts/c5_pmc.sh 256: SQ counters of the composition and scan kernels per launch (two counter passes p1, p2)"
   timeout -k 10 500 bash scripts/c5_pmc.sh 256 r06 2>&1 | grep -v amdgpu.ids) > $OUT/r06_c5_counters.txt
  cat $OUT/r06_c5_breakdown.txt $OUT/r06_c5_counters.txt ;;
esac
