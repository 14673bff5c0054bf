# per-kernel durations of the DCNv2 backward group (64 -> 64 @ 96x320, B = 8, bf16) for both forms (option dcn_bt_fly = 1 / 0)
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$PWD}
cat > /tmp/run_bt.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from monoflex_amd import lib as L
L.check(L.load().mfx_set_option(b"dcn_bt_fly", int(sys.argv[1])), "opt")
from tools.train_layer_bench import dominant_kernel_roofline
r = dominant_kernel_roofline("bf16", 8, torch.device("cuda", 0))
print(r["avg_launch_ms"])
PY
for f in ${1:-1 0}; do cd /tmp; rm -rf /tmp/p$f; rocprofv3 --kernel-trace --stats -d /tmp/p$f -o p -- python /tmp/run_bt.py $f > /dev/null 2>&1
python - <<PY
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob("/tmp/p$f/**/*.db", recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
tot = collections.Counter(); cnt = collections.Counter()
for n, s, e in rows[len(rows)//2:]:
    k = n.split("(")[0][-60:]
    tot[k] += e - s; cnt[k] += 1
print("dcn_bt_fly=$f   group total %.1f us" % (sum(v / cnt[k] for k, v in tot.items() if "dcn_patch" not in k and "pack_conv" not in k and "elementwise" not in k) / 1e3))
for k, v in tot.most_common(9): print("  %-62s n=%3d avg %.1f us" % (k, cnt[k], v / cnt[k] / 1e3))
PY
done
