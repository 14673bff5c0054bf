#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c9; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/full_gpu_tests.log 2>&1
tail -15 $O/full_gpu_tests.log > $O/full_gpu_tests_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
