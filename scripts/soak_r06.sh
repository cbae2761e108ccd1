#!/bin/bash
# Soaks at the final HEAD of round 6 (gpurun): fresh offsets / seeds, none used in an earlier round.
REPO=$PWD; OUT=$REPO/gpurun_out/r06; mkdir -p $OUT; HEAD=$(cat $REPO/.git_head 2>/dev/null || echo unknown)
F=$OUT/r06_soak.txt
echo "# soaks at HEAD $HEAD (round 6), one MI355X" > $F
echo "# (1) fuzz: MTG_FUZZ_OFFSET=k python -m pytest tests/test_fuzz_gpu.py -m gpu, k = 201000..260000 step 1000 (121 random models against the oracle + 60 pipelined sweeps against the one-lane sweep per offset)" >> $F
for k in $(seq 201000 1000 260000); do
  r=$(MTG_FUZZ_OFFSET=$k timeout -k 10 300 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -1)
  echo "offset $k	$r" >> $F
  case "$r" in *failed*|*error*) echo "FUZZ FAILURE at offset $k" >> $F;; esac
done
echo "# (2) sampler: scripts/sampler_soak.py 400 SEED, SEED = 11..16 (device sampler against its host replay with the oracle likelihood)" >> $F
for s in 11 12 13 14 15 16; do timeout -k 10 600 python scripts/sampler_soak.py 400 $s 2>&1 | grep -v amdgpu.ids | tail -3 >> $F; done
echo "# (3) pairs + chirp-z: scripts/r05_soak.py 400 60 SEED, SEED = 11..13" >> $F
for s in 11 12 13; do timeout -k 10 600 python scripts/r05_soak.py 400 60 $s 2>&1 | grep -v amdgpu.ids | tail -2 >> $F; done
echo "# (4) blocks: scripts/block_soak.py 60 SEED, SEED = 11, 12" >> $F
for s in 11 12; do timeout -k 10 600 python scripts/block_soak.py 60 $s 2>&1 | grep -v amdgpu.ids | tail -1 >> $F; done
echo "# (5) rank-10 path: scripts/tp_big_soak.py" >> $F
timeout -k 10 900 python scripts/tp_big_soak.py 200 11 2>&1 | grep -v amdgpu.ids | tail -4 >> $F
echo "# (6) numpy-stream simulator: scripts/numpy_stream_soak.py" >> $F
timeout -k 10 900 python scripts/numpy_stream_soak.py 200 11 2>&1 | grep -v amdgpu.ids | tail -3 >> $F
tail -40 $F
