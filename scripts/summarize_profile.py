#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (scripts/profile_bench.sh) -> profiles/<tag>_* summaries.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE
and WRITE_SIZE come from separate --pmc passes, are in KiB, and on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, so the read side is doubled.  The
doubling is calibrated in the same run on mtg_lc_setup_kernel, whose byte count
is known (it reads y[L*N] + yerr[L*N] + t[N] + y_offset[L] and writes the
interleaved (y, var)[L*N] and (dx, t)[N] pairs).
"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
W = int(sys.argv[4]) if len(sys.argv) > 4 else 256
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)



def head_commit():
    """The commit the profiled tree was at: the .git_head file written before the tree travelled to the GPU box (which has
    no .git) -- the profiles are summarised HERE, afterwards, possibly a few commits later -- else `git rev-parse HEAD`."""
    import subprocess
    try:
        return open(os.path.join(root, ".git_head")).read().strip()
    except OSError:
        pass
    try:
        return subprocess.check_output(["git", "-C", root, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:
        return "unknown"


HEAD = head_commit()
stats = max(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)   # the latest run
with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as fh:   # (every summary says which commit it measured)
    fh.write("# HEAD %s\n" % HEAD)
    for line in open(stats):   # rocPRIM's kernel names run to two thousand characters: keep what tells them apart
        if line.startswith('"void rocprim::'):
            name, rest = line[1:].split('",', 1)
            what = name.split("detail::wrapped_")[1].split("<")[0] if "detail::wrapped_" in name else "kernel"
            step = "iteration" if "onesweep_iteration" in name else "global_offsets" if "global_offsets" in name else ""
            line = '"rocprim::%s %s (the sweep\'s sort by structure, light curve)",%s' % (what, step, rest)
        fh.write(line)


def mean_counter(sub, counter):
    f = max(glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")), key=os.path.getmtime)
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch = mean_counter("pmc_fetch", "FETCH_SIZE")
write = mean_counter("pmc_write", "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    rows.append({"kernel": k, "launches": max(nf, nw), "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
                 "hbm_read_bytes_corrected": 2.0 * f * 1024.0, "hbm_write_bytes": w * 1024.0})
with open(os.path.join(dst, tag + "_pmc_hbm.csv"), "w", newline="") as fh:
    fh.write("# HEAD %s\n" % HEAD)
    wr = csv.DictWriter(fh, fieldnames=list(rows[0]))
    wr.writeheader()
    wr.writerows(rows)

setup = [r for r in rows if r["kernel"].startswith("mtg_lc_setup_kernel")][0]
known_read = (2 * L * N + N + L) * 8.0     # y, yerr, t, y_offset
known_write = (L * N + N) * 16.0            # interleaved (y, var) and (dx, t) pairs
solve = max((r for r in rows if "mtg_solve_kernel" in r["kernel"]),
            key=lambda r: r["hbm_read_bytes_corrected"])
rec = {
    "tag": tag, "head": HEAD, "round": "round " + tag[1:3].lstrip("0") if tag[:1] == "r" and tag[1:3].isdigit() else tag,
    "N": N, "B": L * W, "kernel": solve["kernel"],
    "hbm_bytes_per_launch": solve["hbm_read_bytes_corrected"] + solve["hbm_write_bytes"],
    "hbm_read_bytes_per_launch": solve["hbm_read_bytes_corrected"],
    "hbm_write_bytes_per_launch": solve["hbm_write_bytes"],
    "algorithmic_bytes_per_launch": L * W * (24 * N + 8 * 8 + 12),
    "calibration": {
        "kernel": "mtg_lc_setup_kernel", "known_read_bytes": known_read,
        "FETCH_SIZE_x1024": setup["FETCH_SIZE_KiB_raw"] * 1024.0,
        "read_ratio_raw": setup["FETCH_SIZE_KiB_raw"] * 1024.0 / known_read,
        "known_write_bytes": known_write, "WRITE_SIZE_x1024": setup["WRITE_SIZE_KiB_raw"] * 1024.0,
        "write_ratio_raw": setup["WRITE_SIZE_KiB_raw"] * 1024.0 / known_write,
    },
    "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); "
            "separate --pmc passes; per-launch means",
}
json.dump(rec, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1)
json.dump(rec, open(os.path.join(dst, "bench_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
print(open(os.path.join(dst, tag + "_kernel_stats.csv")).read())
