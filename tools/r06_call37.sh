#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c37; mkdir -p $O
timeout 900 python -m pytest tools/probes/syncbn_flaky_dbg.py -q -p no:cacheprovider -s -k "bitwise or zz_syncbn" 2>&1 | grep -E "state equal|eager step|replay loss|modules whose|passed|failed" > $O/dbg3.txt
