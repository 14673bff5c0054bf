export RTAG=r05; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
cd /tmp && rm -rf /tmp/prof_inf && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_inf -o prof -- python $R/bench.py --legs none --no-cpu-baseline --no-families --steps 20 --repeats 1 > $R/gpurun_out/prof_inf.log 2>&1
cd $R; python tools/prof_summary.py $(find /tmp/prof_inf -name "*.db" | head -1) > gpurun_out/r05_b_inference_kernel_stats.md; head -12 gpurun_out/r05_b_inference_kernel_stats.md | cut -c1-140
bash tools/profile_infer_step.sh > /dev/null 2>&1; head -50 gpurun_out/r05_b_inference_replay_kernel_timeline.md | cut -c1-140
