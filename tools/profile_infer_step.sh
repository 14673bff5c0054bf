# rocprofv3 kernel trace of the inference step (single-stream form, so that the dispatch sequence is the network's order): per-kernel
# totals, idle gaps and the dispatch sequence of one graph replay
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && rm -rf /tmp/prof_inf1 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_inf1 -o r04 -- python $R/bench.py --legs none --no-cpu-baseline --streams 1 --steps 20 --repeats 1 > $R/gpurun_out/prof_inf1.log 2>&1
cd $R; DB=$(find /tmp/prof_inf1 -name "*.db" | head -1)
python tools/gap_analysis.py $DB "f1_fused" 5 gpurun_out/r04_b_inference_step_sequence.txt > gpurun_out/r04_b_inference_replay_kernel_timeline.md 2>&1; head -12 gpurun_out/r04_b_inference_replay_kernel_timeline.md | cut -c1-130
