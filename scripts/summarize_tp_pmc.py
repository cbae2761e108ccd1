#!/usr/bin/env python3
"""HBM traffic of the time-parallel kernels from the --pmc passes of scripts/profile_r03.sh pmc:
FETCH_SIZE / WRITE_SIZE per launch (KiB, separate passes), read side doubled as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE tallies 128-B requests at 64 B).
    summarize_tp_pmc.py TAG HEAD  -> text table on stdout"""
import csv, glob, os, re, sys
tag, head = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "unknown")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
print("# HBM traffic per launch of the time-parallel kernels (rocprofv3 --pmc, one counter per pass); HEAD %s" % head)
print("# read bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction), write bytes = WRITE_SIZE x 1024")
algo = {"s1": ("configs[1] half-step: 64 evaluations, N = 1e4", 64 * (24 * 10000 + 8 * 5 + 12)),
        "s2": ("configs[2] half-step: 128 evaluations, N = 1e4", 128 * (24 * 10000 + 8 * 8 + 12)),
        "c5": ("configs[4] half-step: 256 evaluations, N = 2e5", 256 * (24 * 200000 + 8 * 15 + 12))}
for run in ("s1", "s2", "c5"):
    acc = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_%s_%s" % (tag, run, counter), "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != counter:
                    continue
                k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
                if not ("tp" in k or "sampler" in k):
                    continue
                acc.setdefault(k, {}).setdefault(counter, []).append(float(r["Counter_Value"]))
    print("== %s; algorithmic bytes of the half-step %.4g" % algo[run])
    for k, v in sorted(acc.items()):
        fs, ws = v.get("FETCH_SIZE", [0.0]), v.get("WRITE_SIZE", [0.0])
        # steady state: the median launch (the first ones of a process include cold caches)
        fs, ws = sorted(fs)[len(fs) // 2], sorted(ws)[len(ws) // 2]
        rd, wr = 2.0 * fs * 1024.0, ws * 1024.0
        print("%-44s launches %4d  read %.4g B  write %.4g B  total/algorithmic %.3f" % (
            k, len(v.get("FETCH_SIZE", [])), rd, wr, (rd + wr) / algo[run][1]))
