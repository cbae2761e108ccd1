#!/bin/bash
# PMC passes over the configs[4] half-step (scripts/c5_one.py B): where the wave cycles of the compose / scan kernels go.
# Run on the GPU box: gpurun -- bash scripts/c5_pmc.sh [B] [tag]
B=${1:-256}
TAG=${2:-c5}
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -- python3 $REPO/scripts/c5_one.py $B 4 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -- python3 $REPO/scripts/c5_one.py $B 4 > $OUT/p2.log 2>&1
python3 - <<PY
import csv,glob,re
for sub in ("p1","p2"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv"%sub):
        acc={}
        for r in csv.DictReader(open(f)):
            k=re.sub(r"\(anonymous namespace\)::|void ","",r["Kernel_Name"]).split("(")[0]
            if "tpb" not in k or "finish" in k or "down" in k or "filter" in k: continue
            acc.setdefault((k,r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
        for (k,c),v in sorted(acc.items()): print(sub,"%-34s %-22s max %.4g mean %.4g n %d"%(k,c,max(v),sum(v)/len(v),len(v)))
PY
grep config5 $OUT/p1.log
