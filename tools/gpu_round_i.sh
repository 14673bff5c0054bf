export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcb/p$i -- python $R/tools/one_op.py dcnbwd 8 96 320 64 64 --reps 3 > /dev/null 2>&1
done
python $R/tools/pmc_summary.py /tmp/pmcb | grep -A24 "dcn_bwd_tile"
