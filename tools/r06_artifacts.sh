#!/bin/bash
# round-6 artefacts of the current HEAD: full GPU suite, smoke, default bench line, rocprof timelines / kernel stats, PMC passes (heads, DCN forward, DCN backward)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/full_gpu_tests.log 2>&1; (grep -E "passed|failed" gpurun_out/full_gpu_tests.log | tail -1) > gpurun_out/r06_full_gpu_tests_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r06_full_gpu_tests_tail.txt
RTAG=r06 bash tools/round_artifacts.sh > gpurun_out/r06_artifacts.log 2>&1
bash tools/pmc_dcn.sh r06 > /dev/null 2>&1
