REPO=$PWD; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; HEAD=$(cat $REPO/.git_head 2>/dev/null || echo unknown)
cd /tmp && export TMPDIR=/tmp
: > $OUT/r04_small_chain_gaps.txt
echo "# HEAD $HEAD -- rocprofv3 --kernel-trace over scripts/small_trace.py (device sampler, speculative iterations, N = 1e4), scripts/chain_gaps.py" >> $OUT/r04_small_chain_gaps.txt
for k in 1 2; do
  rocprofv3 --kernel-trace -d $OUT/gaps_s$k -o s$k -- python3 $REPO/scripts/small_trace.py $k > $OUT/gaps_$k.log 2>&1
  f=$(find $OUT/gaps_s$k -name "*.db" | head -1)
  echo "== configs[$k]: $(grep configs $OUT/gaps_$k.log)" >> $OUT/r04_small_chain_gaps.txt
  python3 $REPO/scripts/chain_gaps.py $f >> $OUT/r04_small_chain_gaps.txt
done
cat $OUT/r04_small_chain_gaps.txt
