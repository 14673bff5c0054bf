#!/usr/bin/env python
"""Turn a rocprofv3 results .db (--kernel-trace --stats) into the per-kernel summary kept under profiles/.
usage: python tools/prof_summary.py gpurun_out/prof_r1/r1_results.db [steps] > profiles/rNN_<what>_kernel_stats.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print("| kernel | calls | total us | avg us | % |" + (" us/step |" if steps else ""))
print("|---|---|---|---|---|" + ("---|" if steps else ""))
for name, calls, tot, avg, pct in rows:
    short = re.sub(r"\(.*", "", name).replace("void ", "").strip()[:120] or name[:60]
    line = "| %s | %d | %.1f | %.2f | %.2f |" % (short, calls, tot, avg, pct)
    if steps:
        line += " %.1f |" % (tot / steps)
    print(line)
