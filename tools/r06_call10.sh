#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c10; mkdir -p $O
timeout 2700 python -m pytest tests -q -m gpu -p no:cacheprovider -x -s -k "e2e or training_step_vs_oracle or dcn_lds or project_then_sample or test_conv2d or gram_heads" > $O/gpu_tests.log 2>&1
grep -E "passed|failed|FAILED|Error|full-size|vs reference" $O/gpu_tests.log | tail -40 > $O/gpu_tests_tail.txt
