# how much do the two sub-batch streams of the default inference step overlap?  sum of kernel durations vs the union of their intervals
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && rm -rf /tmp/prof_ov && timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_ov -o r04 -- python $R/bench.py --legs none --no-cpu-baseline --steps 20 --repeats 1 > $R/gpurun_out/prof_ov.log 2>&1
cd $R; DB=$(find /tmp/prof_ov -name "*.db" | head -1)
python - $DB <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "decode_boxes" in r[0]]
a, b = idx[-6], idx[-1]          # five steps (two decode_boxes per step: one per sub-batch? use pairs)
seg = rows[a + 1:b + 1]
t0, t1 = seg[0][1], max(r[2] for r in seg)
tot = sum(e - s for _, s, e in seg)
ev = sorted([(s, 1) for _, s, e in seg] + [(e, -1) for _, s, e in seg])
cur = 0; last = None; union = 0; two = 0
for t, d in ev:
    if cur > 0: union += t - last
    if cur > 1: two += t - last
    cur += d; last = t
print("kernels %d  span %.1f us  sum of durations %.1f us  union %.1f us  >=2 kernels resident %.1f us (%.0f %% of the union)" % (
    len(seg), (t1 - t0) / 1e3, tot / 1e3, union / 1e3, two / 1e3, 100.0 * two / union))
PY
