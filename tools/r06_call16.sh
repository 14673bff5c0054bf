#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c16; mkdir -p $O
B="python bench.py --legs none --no-cpu-baseline --no-families --steps 30 --warmup 10 --repeats 5"
run() { echo "## $1" >> $O/ab.txt; env $2 $B $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['timing']['ms_per_step_each'])" >> $O/ab.txt; }
run "default" "X=1" ""
run "MFX_PARALLEL_PROJ=1" "MFX_PARALLEL_PROJ=1" ""
run "--streams 2" "X=1" "--streams 2"
run "--streams 2 + PARALLEL_PROJ" "MFX_PARALLEL_PROJ=1" "--streams 2"
run "default again" "X=1" ""
run "batch 16" "X=1" "--batch 16"
