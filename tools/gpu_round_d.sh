mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o x -- python $R/tools/one_op.py dcnbwd 8 96 320 64 64 --reps 5 > /dev/null 2>&1
python $R/tools/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) > $R/gpurun_out/dcnbwd_prof_64.md
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o x -- python $R/tools/one_op.py dcnbwd 8 24 80 256 128 --reps 5 > /dev/null 2>&1
python $R/tools/prof_summary.py $(find /tmp/p2 -name "*.db" | head -1) > $R/gpurun_out/dcnbwd_prof_256.md
head -14 $R/gpurun_out/dcnbwd_prof_64.md | cut -c1-160; head -12 $R/gpurun_out/dcnbwd_prof_256.md | cut -c1-160
