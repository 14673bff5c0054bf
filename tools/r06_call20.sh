#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c20; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "conv_cws or (e2e and fp16x2)" > $O/t.log 2>&1; tail -15 $O/t.log > $O/t_tail.txt
B="python bench.py --legs none --no-cpu-baseline --no-families --steps 20 --warmup 5 --repeats 3 --dtype fp16x2"
run() { echo "## $1" >> $O/ab.txt; $B --opts "$2" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['timing']['ms_per_step_each'])" >> $O/ab.txt; }
run "fp16x2 halo_cws=0" "halo_cws=0"
run "fp16x2 halo_cws=1" ""
run "fp16x2 halo_cws=0" "halo_cws=0"
run "fp16x2 halo_cws=1" ""
