#!/bin/bash
# does placing kernel arguments in device memory (HIP_FORCE_DEV_KERNARG) change the small chains' iteration rate?
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v MTG_SAMPLER_LDS=1 python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
src = open("scripts/small_chain_ab.py").read()
child = src.split("CHILD = r'''")[1].split("''' % ROOT")[0] % os.getcwd()
exec(child)
PY
done
