#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export RTAG=r06c
bash tools/profile_train_step.sh > gpurun_out/r06c_prof.log 2>&1
