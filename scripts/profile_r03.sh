#!/bin/bash
# Round-3 profiles (run on the GPU box via gpurun; summaries land in gpurun_out/, copy the keepers into profiles/):
#   bash scripts/profile_r03.sh c5      per-kernel breakdown of the rank-10 half-step at B = 8, 32, 256
#   bash scripts/profile_r03.sh chains  kernel traces of the single-light-curve chains (configs[1], [2], [4])
#   bash scripts/profile_r03.sh pmc     FETCH_SIZE / WRITE_SIZE passes: time-parallel kernels (configs[1], [2], [4])
#   bash scripts/profile_r03.sh order   scripts/order_sweep.py
WHAT=${1:-c5}
TAG=${2:-r03}
REPO=$PWD
OUT=$REPO/gpurun_out
HEAD=$(cat $REPO/.git_head 2>/dev/null || echo unknown)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
case $WHAT in
c5)
  : > $OUT/${TAG}_c5_breakdown.txt
  echo "# rank-10 half-step (5 x SHO, N = 2e5) per kernel: rocprofv3 --kernel-trace over scripts/c5_one.py B 6; HEAD $HEAD" >> $OUT/${TAG}_c5_breakdown.txt
  for B in 8 32 256; do
    rocprofv3 --kernel-trace -d $OUT/prof_${TAG}_c5_$B -o c5 -- python3 $REPO/scripts/c5_one.py $B 6 > $OUT/${TAG}_c5_$B.log 2>&1 || exit 1
    f=$(find $OUT/prof_${TAG}_c5_$B -name "*.db" | head -1)
    echo "== B = $B: $(tail -1 $OUT/${TAG}_c5_$B.log)" >> $OUT/${TAG}_c5_breakdown.txt
    python3 $REPO/scripts/rocpd_kernels.py $f | grep -v "lc_setup" >> $OUT/${TAG}_c5_breakdown.txt
    echo "-- last half-step, dispatch by dispatch" >> $OUT/${TAG}_c5_breakdown.txt
    python3 $REPO/scripts/rocpd_kernels.py $f --trace 14 >> $OUT/${TAG}_c5_breakdown.txt
  done
  cat $OUT/${TAG}_c5_breakdown.txt ;;
chains)
  cd $REPO && bash scripts/profile_chains.sh $TAG ;;
pmc)
  for k in 1 2; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_s${k}_$c -- python3 $REPO/scripts/small_trace.py $k > $OUT/pmc_${TAG}_s${k}_$c.log 2>&1 || exit 1
    done
  done
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_c5_$c -- python3 $REPO/scripts/c5_one.py 256 4 > $OUT/pmc_${TAG}_c5_$c.log 2>&1 || exit 1
  done
  python3 $REPO/scripts/summarize_tp_pmc.py $TAG $HEAD > $OUT/${TAG}_tp_pmc_hbm.txt
  cat $OUT/${TAG}_tp_pmc_hbm.txt ;;
order)
  cd $REPO && python3 scripts/order_sweep.py > $OUT/${TAG}_order_sweep.txt 2>&1; cat $OUT/${TAG}_order_sweep.txt ;;
esac
