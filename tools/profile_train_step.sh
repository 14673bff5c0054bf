mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o r02 -- python $R/bench.py --mode train --no-cpu-baseline --steps 10 > $R/gpurun_out/prof_tr.log 2>&1
cd $R; tail -1 gpurun_out/prof_tr.log | cut -c1-300; DB=$(find /tmp/prof_tr -name "*.db" | head -1)
python tools/gap_analysis.py $DB "mfx::bf16_t, mfx::bf16_t, 256, 16" 5 gpurun_out/train_step_sequence.txt > gpurun_out/r02_g_train_replay_kernel_timeline.md 2>&1; head -70 gpurun_out/r02_g_train_replay_kernel_timeline.md | cut -c1-150
