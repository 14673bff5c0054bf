#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c19; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "dcn_lds_split or (e2e and fp16x2)" > $O/t.log 2>&1; tail -15 $O/t.log > $O/t_tail.txt
B="python bench.py --legs none --no-cpu-baseline --no-families --steps 20 --warmup 5 --repeats 3 --dtype fp16x2"
run() { echo "## $1" >> $O/ab.txt; $B --opts "$2" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['timing']['ms_per_step_each'], d['config'].get('vs_reference'))" >> $O/ab.txt; }
run "fp16x2 dcn_lds=0" "dcn_lds=0"
run "fp16x2 dcn_lds=1 (split LDS kernel on the 64->64 layers)" ""
run "fp16x2 dcn_lds=2 (also 128->64, 256->64)" "dcn_lds=2"
run "fp16x2 dcn_lds=0" "dcn_lds=0"
