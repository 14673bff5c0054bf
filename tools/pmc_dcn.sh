cd /tmp && export TMPDIR=/tmp
R=/root/repo
for std in 0.0 1.5; do for o in dcn_patch=2 dcn_patch=3 dcn_patch=0; do python $R/tools/one_dcn.py 8 96 320 64 64 $o 20 $std; done; done
python $R/tools/one_dcn.py 8 48 160 128 64 dcn_patch=3 20 1.5; python $R/tools/one_dcn.py 8 48 160 128 64 dcn_patch=2 20 1.5; python $R/tools/one_dcn.py 8 48 160 128 64 dcn_patch=0 20 1.5
python $R/tools/one_dcn.py 8 48 160 128 128 dcn_patch=3 20 1.5; python $R/tools/one_dcn.py 8 48 160 128 128 dcn_patch=0 20 1.5
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_dp/p$i -- python $R/tools/one_dcn.py 8 96 320 64 64 dcn_patch=2 4 1.5 > /dev/null 2>&1
done
python $R/tools/pmc_summary.py /tmp/pmc_dp > $R/gpurun_out/pmc_dcn_patch.txt
