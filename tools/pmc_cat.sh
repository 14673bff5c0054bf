#!/bin/bash
# PMC passes of one Root-node launch (cat_igemm_kernel): where does a one-round LDS-tiled launch spend its time?   usage (GPU box): bash tools/pmc_cat.sh "8 24 80 256,256,128,256 256" [tag]
SHP=${1:-"8 24 80 256,256,128,256 256"}
TAG=${2:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_cat; rm -rf $OUT; mkdir -p $OUT
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- python $R/tools/probes/one_cat.py $SHP > $OUT/p$i.log 2>&1 || echo "pass $i failed: $c" >> $OUT/failed.txt
done
echo "== Root node, B H W segments Cout = $SHP" > $R/gpurun_out/${TAG}_cat_pmc.txt
python $R/tools/pmc_summary.py $OUT >> $R/gpurun_out/${TAG}_cat_pmc.txt 2>&1
cat $OUT/failed.txt >> $R/gpurun_out/${TAG}_cat_pmc.txt 2>/dev/null
cat $R/gpurun_out/${TAG}_cat_pmc.txt
