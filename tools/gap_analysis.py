#!/usr/bin/env python
"""Where does a hipGraph replay spend its time?  Reads a rocprofv3 --kernel-trace .db, cuts the dispatch stream into steps at
a marker kernel (default: the stem kernel), and reports for the last steps: busy time, idle gaps between consecutive
dispatches, and the gap preceding each kernel name.  usage: gap_analysis.py results.db [marker substring] [steps to average]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "stem_conv7x7"
nlast = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = db.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if marker in r[0]]
if len(idx) < nlast + 1:
    sys.exit("marker %r found %d times" % (marker, len(idx)))
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").strip()[:70]
tot = collections.Counter(); gaps = collections.Counter(); cnt = collections.Counter()
span = busy = 0.0
for a, b in zip(idx[-nlast - 1:-1], idx[-nlast:]):
    step = rows[a:b]
    span += rows[b][1] - step[0][1]
    prev_end = None
    for name, s, e in step:
        busy += e - s
        k = short(name)
        tot[k] += e - s; cnt[k] += 1
        if prev_end is not None:
            gaps[k] += max(0, s - prev_end)
        prev_end = max(prev_end or e, e)
    gaps["(step boundary)"] += max(0, rows[b][1] - prev_end); cnt["(step boundary)"] += 1
n = float(nlast)
print("steps averaged: %d   dispatches/step: %.1f" % (nlast, sum(v for k, v in cnt.items() if k != "(step boundary)") / n))
print("step span %.1f us = busy %.1f us + idle %.1f us (%.1f %%)" % (span / n / 1e3, busy / n / 1e3, (span - busy) / n / 1e3, 100 * (span - busy) / span))
print("| kernel | calls/step | busy us/step | gap-before us/step | avg gap us |")
print("|---|---|---|---|---|")
for k, _ in sorted(cnt.items(), key=lambda kv: -(tot[kv[0]] + gaps[kv[0]])):
    print("| %s | %.1f | %.1f | %.1f | %.2f |" % (k, cnt[k] / n, tot[k] / n / 1e3, gaps[k] / n / 1e3, gaps[k] / max(cnt[k], 1) / 1e3))
if len(sys.argv) > 4:                                            # dump the dispatch sequence of the last step: start offset, duration, gap, name
    a, b = idx[-2], idx[-1]
    t0 = rows[a][1]; prev_end = None
    with open(sys.argv[4], "w") as f:
        for name, s, e in rows[a:b]:
            f.write("%9.1f %8.1f %6.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev_end is None else max(0, s - prev_end) / 1e3, short(name)[:110]))
            prev_end = max(prev_end or e, e)
