#!/bin/bash
# scripts/hbm_regime_probe.py plain, then under rocprofv3 --pmc FETCH_SIZE (own pass) and TCC hit / miss counters:
# per-launch HBM read traffic of the sweep where every row streams its own light curve.  gpurun -- bash scripts/hbm_regime.sh [L]
L=${1:-131072}
OUT=$PWD/gpurun_out/hbm_regime; mkdir -p $OUT; REPO=$PWD
python3 $REPO/scripts/hbm_regime_probe.py $L 262144 524288 > $OUT/plain.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $REPO/scripts/hbm_regime_probe.py $L > $OUT/fetch.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcc -- python3 $REPO/scripts/hbm_regime_probe.py $L > $OUT/tcc.log 2>&1
python3 - <<PY
import csv, glob
print(open("$OUT/plain.txt").read())
for sub, names in (("fetch", ("FETCH_SIZE",)), ("tcc", ("TCC_HIT_sum", "TCC_MISS_sum"))):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % sub):
        rows = [r for r in csv.DictReader(open(f)) if "mtg_solve_kernel" in r["Kernel_Name"] or "mtg_lc_setup" in r["Kernel_Name"]]
        by = {}
        for r in rows:
            by.setdefault((r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        for (d, k, g), c in sorted(by.items(), key=lambda x: int(x[0][0])):
            print(sub, "dispatch", d, k, "grid", g, {n: c.get(n) for n in names},
                  ("FETCH_SIZE x 1024 x 2 = %.3f GB" % (c["FETCH_SIZE"] * 2048 / 1e9)) if "FETCH_SIZE" in c else
                  ("L2 hit rate %.3f" % (c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1))))
PY
