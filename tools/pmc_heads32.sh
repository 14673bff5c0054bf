#!/bin/bash
# PMC A/B of the two heads kernels (16x16x32 vs 32x32x16 MFMA form) at the bench shape: MFMA pipe, VALU, LDS and wait-state counters per launch.
#   usage (GPU box): bash tools/pmc_heads32.sh   -> gpurun_out/r06_heads32_pmc.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  OUT=$R/gpurun_out/pmc_heads32_$v; mkdir -p $OUT; i=0
  for c in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_WAVES"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- python $R/tools/one_op.py heads --reps 4 --eager --opts heads_mfma32=$v > $OUT/p$i.log 2>&1
  done
  echo "## heads_mfma32=$v" >> $R/gpurun_out/r06_heads32_pmc.txt
  python $R/tools/pmc_summary.py $OUT >> $R/gpurun_out/r06_heads32_pmc.txt
  rm -rf $OUT
done
