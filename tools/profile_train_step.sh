# rocprofv3 kernel trace of `bench.py --mode train`: per-kernel totals, the per-kernel timeline of one graph replay and its dispatch sequence
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && rm -rf /tmp/prof_tr && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o prof -- python $R/bench.py --mode train --no-cpu-baseline --steps 10 > $R/gpurun_out/prof_tr.log 2>&1
cd $R; DB=$(find /tmp/prof_tr -name "*.db" | head -1)
python tools/prof_summary.py $DB > gpurun_out/${RTAG:-r05}_a_train_step_kernel_stats.md
python tools/gap_analysis.py $DB "stem_conv7x7" 5 gpurun_out/${RTAG:-r05}_a_train_step_sequence.txt > gpurun_out/${RTAG:-r05}_a_train_replay_kernel_timeline.md 2>&1; head -12 gpurun_out/${RTAG:-r05}_a_train_replay_kernel_timeline.md | cut -c1-130
