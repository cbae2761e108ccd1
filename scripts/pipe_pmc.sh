#!/bin/bash
# PMC passes over scripts/pipe_ab.py: where the wave cycles of the pipeline kernel go.  gpurun -- bash scripts/pipe_pmc.sh TAG [lib]
TAG=${1:-pipe}
LIB=${2:-}
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT; REPO=$PWD
if [ -n "$LIB" ]; then export MTG_HIP_LIB=$REPO/mind_the_gaps_amd/$LIB; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -- python3 $REPO/scripts/pipe_ab.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -- python3 $REPO/scripts/pipe_ab.py > $OUT/p2.log 2>&1
python3 - <<PY
import csv,glob,re
for sub in ("p1","p2"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv"%sub):
        acc={};dur={}
        for r in csv.DictReader(open(f)):
            k=re.sub(r"\(anonymous namespace\)::|void ","",r["Kernel_Name"]).split("(")[0]
            if "pipe" not in k and "solve_kernel" not in k: continue
            acc.setdefault((k,r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
            dur.setdefault(k,[]).append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
        for (k,c),v in sorted(acc.items()): print("$TAG",sub,"%-36s %-22s mean %.5g n %d"%(k,c,sum(v)/len(v),len(v)))
        for k,v in dur.items(): print("$TAG",sub,k,"min duration ns",min(v))
PY
