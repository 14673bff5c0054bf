export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for o in 1 0 1 0; do
cd /tmp && rm -rf /tmp/p_h && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/p_h -o r -- python $R/bench.py --no-cpu-baseline --steps 20 --opts heads_planes=$o > /tmp/b.log 2>&1
cd $R; DB=$(find /tmp/p_h -name "*.db" | head -1); echo "planes=$o $(tail -1 /tmp/b.log | cut -c160-215)"; python tools/prof_summary.py $DB | grep -E "heads_fused" | cut -c1-110
done
