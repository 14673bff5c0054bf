mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
bash tools/pmc_dcn.sh r02 > gpurun_out/pmc_dcn.log 2>&1
head -60 gpurun_out/r02_dcn_pmc.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_inf -o r02 -- python $R/bench.py --no-cpu-baseline --steps 20 > $R/gpurun_out/prof_inf.log 2>&1
cd $R; DB=$(find /tmp/prof_inf -name "*.db" | head -1); python tools/prof_summary.py $DB > gpurun_out/r02_i_inference_kernel_stats.md; python tools/gap_analysis.py $DB > gpurun_out/r02_i_graph_replay_kernel_timeline.md 2>&1; head -16 gpurun_out/r02_i_graph_replay_kernel_timeline.md | cut -c1-120
