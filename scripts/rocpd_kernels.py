#!/usr/bin/env python3
"""Per-dispatch and per-kernel durations from a rocprofv3 rocpd database (the default output format
of ROCm 7.2): rocpd_kernels.py results.db [--trace N]  -> kernel stats (csv on stdout); with --trace the
last N dispatches in order."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [c[1] for c in cur.execute("pragma table_info(kernels)")]
rows = list(cur.execute("select * from kernels"))
ix = {c: i for i, c in enumerate(cols)}
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0]
name_col = "name" if "name" in ix else [c for c in cols if "name" in c][0]
rows.sort(key=lambda r: r[ix["start"]])
if "--trace" in sys.argv:
    n = int(sys.argv[sys.argv.index("--trace") + 1])
    t0 = rows[-n][ix["start"]]
    prev_end = None
    for r in rows[-n:]:
        gap = (r[ix["start"]] - prev_end) / 1e3 if prev_end else 0.0
        print("%10.1f us  +%6.1f us gap  %9.1f us  grid %s  %s" % ((r[ix["start"]] - t0) / 1e3, gap, (r[ix["end"]] - r[ix["start"]]) / 1e3,
              "x".join(str(r[ix[c]]) for c in ("grid_x", "grid_y", "grid_z") if c in ix) if "grid_x" in ix else "", short(r[ix[name_col]])))
        prev_end = r[ix["end"]]
else:
    agg = {}
    for r in rows:
        a = agg.setdefault(short(r[ix[name_col]]), [0, 0.0, 1e30, 0.0])
        d = (r[ix["end"]] - r[ix["start"]]) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print("Name,Calls,TotalDurationUs,AverageUs,MinUs,MaxUs,Percentage")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f' % (k, a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
