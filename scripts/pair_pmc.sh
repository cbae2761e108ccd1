#!/bin/bash
# PMC passes over scripts/pair_probe.py: where the wave cycles of the paired pipeline kernel go, next to the unpaired
# kernels of the same run.   gpurun -- bash scripts/pair_pmc.sh
OUT=$PWD/gpurun_out/pmc_pair; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -- python3 $REPO/scripts/pair_probe.py 250 128 6 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -- python3 $REPO/scripts/pair_probe.py 250 128 6 > $OUT/p2.log 2>&1
python3 - <<PY
import csv,glob,re
for sub in ("p1","p2"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv"%sub):
        acc={};dur={}
        for r in csv.DictReader(open(f)):
            k=re.sub(r"\(anonymous namespace\)::|void ","",r["Kernel_Name"]).split("(")[0]
            if "pipe" not in k: continue
            k=k[:60]
            acc.setdefault((k,r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
            dur.setdefault(k,[]).append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
        for (k,c),v in sorted(acc.items()): print(sub,"%-60s %-22s mean %.5g n %d"%(k,c,sum(v)/len(v),len(v)))
        for k,v in dur.items(): print(sub,k,"min / median duration us %.1f %.1f"%(min(v)/1e3,sorted(v)[len(v)//2]/1e3))
PY
