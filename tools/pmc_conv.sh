#!/bin/bash
# PMC passes of the trunk's 3x3 kernel (conv3x3_cw_kernel) at the four DLA level shapes (B = 8, bf16): MFMA busy, LDS conflicts, instruction mix,
# HBM bytes.  --pmc only (no tracing), one counter group per pass, each under `timeout`.   usage (GPU box): bash tools/pmc_conv.sh [tag]
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/${TAG}_conv_pmc.txt
for shp in "8 96 320 64 64" "8 48 160 128 128" "8 24 80 256 256" "8 12 40 512 512"; do
  OUT=$R/gpurun_out/pmc_conv_$(echo $shp | tr ' ' '_')
  rm -rf $OUT; mkdir -p $OUT
  i=0
  for c in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- python $R/tools/one_op.py conv $shp --reps 4 --eager > $OUT/p$i.log 2>&1
  done
  echo "== conv 3x3 + BN + residual + ReLU, B H W Cin Cout = $shp" >> $R/gpurun_out/${TAG}_conv_pmc.txt
  python $R/tools/pmc_summary.py $OUT >> $R/gpurun_out/${TAG}_conv_pmc.txt 2>&1
done
