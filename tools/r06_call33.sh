#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export RTAG=r06c33
bash tools/profile_train_step.sh > gpurun_out/r06c33_prof.log 2>&1
