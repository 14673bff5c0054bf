#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c14; mkdir -p $O
for i in 1 2; do
echo "## new" >> $O/up.txt; python tools/upsample_bench.py 2>/dev/null >> $O/up.txt
echo "## old" >> $O/up.txt; MFX_LIB_PATH=build_variants/lib_misc_old.so python tools/upsample_bench.py 2>/dev/null >> $O/up.txt
done
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -k "upsample or dlaup or e2e_small" > $O/t.log 2>&1; tail -2 $O/t.log >> $O/up.txt
