# Round artefacts of the current HEAD -> gpurun_out/${RTAG:-r05}_*: the exact default bench command (all legs), the B=32 / training / split-precision
# lines, rocprofv3 kernel summaries of the inference and training steps, the PMC passes of the heads kernel and of the DCN backward group.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python bench.py > gpurun_out/${RTAG:-r05}_bench_default.json 2> gpurun_out/${RTAG:-r05}_bench_default.err; cut -c1-200 gpurun_out/${RTAG:-r05}_bench_default.json; echo
timeout 600 python bench.py --batch 32 --no-cpu-baseline --legs none > gpurun_out/${RTAG:-r05}_bench_infer_b32.json 2>/dev/null; cut -c1-200 gpurun_out/${RTAG:-r05}_bench_infer_b32.json; echo
timeout 600 python bench.py --dtype fp16x2 --no-cpu-baseline --legs none > gpurun_out/${RTAG:-r05}_bench_infer_fp16x2.json 2>/dev/null; cut -c1-200 gpurun_out/${RTAG:-r05}_bench_infer_fp16x2.json; echo
timeout 900 python bench.py --mode train > gpurun_out/${RTAG:-r05}_bench_train.json 2> gpurun_out/${RTAG:-r05}_bench_train.err; cut -c1-200 gpurun_out/${RTAG:-r05}_bench_train.json; echo
cd /tmp && rm -rf /tmp/prof_inf && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_inf -o prof -- python $R/bench.py --legs none --no-cpu-baseline --no-families --steps 20 --repeats 1 > $R/gpurun_out/prof_inf.log 2>&1
cd $R; python tools/prof_summary.py $(find /tmp/prof_inf -name "*.db" | head -1) > gpurun_out/${RTAG:-r05}_b_inference_kernel_stats.md; head -8 gpurun_out/${RTAG:-r05}_b_inference_kernel_stats.md | cut -c1-140
bash tools/profile_infer_step.sh > /dev/null 2>&1; head -6 gpurun_out/${RTAG:-r05}_b_inference_replay_kernel_timeline.md | cut -c1-140
bash tools/profile_train_step.sh > /dev/null 2>&1; head -6 gpurun_out/${RTAG:-r05}_a_train_step_kernel_stats.md | cut -c1-140
bash tools/pmc_heads.sh ${RTAG:-r05} > /dev/null 2>&1; cat gpurun_out/${RTAG:-r05}_heads_traffic.json
bash tools/pmc_dcnbwd.sh ${RTAG:-r05} > /dev/null 2>&1; cat gpurun_out/${RTAG:-r05}_dcnbwd_traffic.json
