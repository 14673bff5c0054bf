python -m pytest tests/test_gpu_train.py -q -m gpu -p no:cacheprovider -k "dcn" 2>&1 | tail -5 | cut -c1-300
for st in 1.5 4.0 7.0; do python tools/one_op.py dcnbwd 8 96 320 64 64 --reps 3 --std $st 2>&1 | tail -1; done
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o r02b -- python $R/bench.py --mode train --no-cpu-baseline --steps 5 > $R/gpurun_out/prof_train_b.log 2>&1
cd $R; python tools/prof_summary.py $(find /tmp/prof_train -name "*.db" | head -1) 5 > gpurun_out/r02_b_train_step_kernel_stats.md; tail -1 gpurun_out/prof_train_b.log | cut -c1-300; head -30 gpurun_out/r02_b_train_step_kernel_stats.md | cut -c1-130
