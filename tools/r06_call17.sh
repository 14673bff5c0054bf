#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c17; mkdir -p $O
B="python bench.py --legs none --no-cpu-baseline --no-families --steps 30 --warmup 10 --repeats 5"
run() { echo "## $1" >> $O/ab.txt; env $2 $B $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['timing']['ms_per_step_each'])" >> $O/ab.txt; }
run "streams 1" "X=1" "--streams 1"
run "streams 2" "X=1" "--streams 2"
run "streams 4" "X=1" "--streams 4"
run "streams 2" "X=1" "--streams 2"
run "streams 1" "X=1" "--streams 1"
run "streams 2 fp16" "X=1" "--streams 2 --dtype fp16"
run "streams 1 fp16" "X=1" "--streams 1 --dtype fp16"
run "b32 streams 1" "X=1" "--streams 1 --batch 32"
run "b32 streams 2" "X=1" "--streams 2 --batch 32"
run "b32 streams 4" "X=1" "--streams 4 --batch 32"
