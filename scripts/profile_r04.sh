#!/bin/bash
# Round-4 profiles (run on the GPU box via gpurun; everything lands in gpurun_out/r04/, the keepers are copied to profiles/):
#   bash scripts/profile_r04.sh bench     rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes over the bench command
#   bash scripts/profile_r04.sh midbatch  kernel stats and SQ counters of the pipelined sweep (250 x 128 rows, both models)
#   bash scripts/profile_r04.sh probes    pipe_probe, halfwave_probe, slice_probe, c3_share_phases
WHAT=${1:-bench}
REPO=$PWD
OUT=$REPO/gpurun_out/r04
HEAD=$(cat $REPO/.git_head 2>/dev/null || echo unknown)
mkdir -p $OUT
case $WHAT in
bench)
  bash scripts/profile_bench.sh r04 5 > $OUT/profile_bench.log 2>&1
  python3 scripts/summarize_profile.py r04 > $OUT/summarize.log 2>&1
  tail -30 $OUT/summarize.log ;;
midbatch)
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/midbatch_trace -- python3 $REPO/scripts/pipe_ab.py > $OUT/midbatch_trace.log 2>&1
  f=$(find $OUT/midbatch_trace -name "*kernel_stats.csv" | head -1)
  (echo "# HEAD $HEAD -- rocprofv3 --kernel-trace --stats over scripts/pipe_ab.py: 250 light curves x 128 rows, N = 1e4, both models,"
   echo "# 7 launches each of the one-lane sweep and of its two-wave pipeline (interleaved)"; grep -v "rocprim" $f | head -14) > $OUT/r04_midbatch_kernel_stats.csv
  cd $REPO && bash scripts/pipe_pmc.sh r04mid > $OUT/pipe_pmc.log 2>&1
  (echo "# HEAD $HEAD -- scripts/pipe_pmc.sh: SQ counters per launch (means of 7), pipelined sweep and one-lane sweep at 250 x 128 rows, N = 1e4"
   echo "# units: SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* in quad-cycles summed over the waves; p1, p2 = the two counter passes"
   grep "pipe_kernel\|solve_kernel_multi" $OUT/pipe_pmc.log) > $OUT/r04_midbatch_sq_counters.txt
  cat $OUT/r04_midbatch_kernel_stats.csv; head -50 $OUT/r04_midbatch_sq_counters.txt ;;
probes)
  (echo "# HEAD $HEAD"; python3 scripts/pipe_probe.py 2>&1 | grep -v amdgpu.ids) > $OUT/r04_pipe_probe.txt
  (echo "# HEAD $HEAD"; python3 scripts/halfwave_probe.py 2>&1 | grep -v amdgpu.ids) > $OUT/r04_halfwave_probe.txt
  (echo "# HEAD $HEAD"; python3 scripts/slice_probe.py 2>&1 | grep -v amdgpu.ids) > $OUT/r04_slice_probe.txt
  (echo "# HEAD $HEAD"; python3 scripts/c3_share_phases.py 2>&1 | grep -v amdgpu.ids) > $OUT/r04_c3_share_phases.txt
  cat $OUT/r04_pipe_probe.txt $OUT/r04_halfwave_probe.txt $OUT/r04_slice_probe.txt $OUT/r04_c3_share_phases.txt ;;
esac
