# PMC passes for the LDS-patch DCN kernel (64->64 @ B=8, offsets std 2.5 px).  Every rocprofv3 call is wrapped in its own
# `timeout`: an earlier version of this script (derived TCC_* counters in one pass) hung a GPU box until the outer limit.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_dp/p$i -- python $R/tools/one_dcn.py 8 96 320 64 64 "" 4 2.5 > /dev/null 2>&1
done
python $R/tools/pmc_summary.py /tmp/pmc_dp > $R/gpurun_out/pmc_dcn_patch_final.txt
