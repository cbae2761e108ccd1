#!/bin/bash
OUT=$PWD/gpurun_out/prof_c5; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/scripts/c5_trace.py > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cut -c1-160 $f | head -14; done
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/trace/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-60:]
prev=None
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(s-prev)/1e3 if prev else 0
    print("%-60s dur %9.1f us  gap %9.1f us"%(r["Kernel_Name"][:60],(e-s)/1e3,gap))
    prev=e
PY
