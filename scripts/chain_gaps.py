#!/usr/bin/env python3
"""Where an iteration of a single-light-curve chain goes: from a rocprofv3 rocpd database of scripts/small_trace.py, the
steady alternation solve kernel / sampler kernel -- durations and the idle gaps between them.
    python scripts/chain_gaps.py results.db"""
import re, sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [c[1] for c in cur.execute("pragma table_info(kernels)")]
ix = {c: i for i, c in enumerate(cols)}
name_col = "name" if "name" in ix else [c for c in cols if "name" in c][0]
rows = sorted(cur.execute("select * from kernels"), key=lambda r: r[ix["start"]])
short = lambda n: re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n)).split("(")[0]
seq = [(short(r[ix[name_col]]), r[ix["start"]], r[ix["end"]]) for r in rows]
# the longest run of strictly alternating (solve, sampler_spec) dispatches
best, cur_run = [], []
for k, (n, s, e) in enumerate(seq):
    want_solve = len(cur_run) % 2 == 0
    is_solve = n.startswith("mtg_tp_")
    is_samp = n.startswith("mtg_sampler_spec")
    if (want_solve and is_solve) or (not want_solve and is_samp):
        cur_run.append((n, s, e))
    else:
        if len(cur_run) > len(best):
            best = cur_run
        cur_run = [(n, s, e)] if is_solve else []
if len(cur_run) > len(best):
    best = cur_run
best = best[: len(best) // 2 * 2]
solve = np.array([(e - s) for n, s, e in best[0::2]]) / 1e3
samp = np.array([(e - s) for n, s, e in best[1::2]]) / 1e3
gap_a = np.array([best[i + 1][1] - best[i][2] for i in range(0, len(best) - 1, 2)]) / 1e3     # solve end -> sampler start
gap_b = np.array([best[i + 1][1] - best[i][2] for i in range(1, len(best) - 1, 2)]) / 1e3     # sampler end -> next solve start
period = np.array([best[i + 2][1] - best[i][1] for i in range(0, len(best) - 2, 2)]) / 1e3
print("%d iterations in steady alternation: %s | %s" % (len(solve), best[0][0], best[1][0]))
for label, v in (("solve kernel", solve), ("gap solve -> sampler", gap_a), ("sampler kernel", samp), ("gap sampler -> next solve", gap_b),
                 ("iteration period", period)):
    print("  %-28s median %7.2f us   mean %7.2f   min %7.2f   max %7.2f" % (label, np.median(v), v.mean(), v.min(), v.max()))
print("  gaps are %.1f %% of the period; a persistent kernel could save at most the gaps plus the per-launch table fill"
      % (100 * (np.median(gap_a) + np.median(gap_b)) / np.median(period)))
