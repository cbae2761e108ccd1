#!/bin/bash
# PMC passes over perf_probe (alt model only): clock + issue utilisation
OUT=$PWD/gpurun_out/pmc_probe; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -- python3 $REPO/scripts/perf_probe.py 10000 2000 256 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -- python3 $REPO/scripts/perf_probe.py 10000 2000 256 > $OUT/p2.log 2>&1
python3 - <<PY
import csv,glob
for sub in ("p1","p2"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv"%sub):
        acc={}
        for r in csv.DictReader(open(f)):
            if "mtg_solve_kernel_multi<1, 2, 2, 1>" in r["Kernel_Name"] or "mtg_solve_kernel<1, 2" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
                dur=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
        for k,v in acc.items(): print(sub,k,sum(v)/len(v))
        print("last duration ns",dur)
PY
tail -3 $OUT/p1.log
