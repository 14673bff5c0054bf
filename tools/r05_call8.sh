mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
bash tools/profile_train_step.sh > /dev/null 2>&1
ls gpurun_out | grep r04_a
head -100 gpurun_out/r04_a_train_replay_kernel_timeline.md | cut -c1-130
