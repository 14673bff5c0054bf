#!/bin/bash
# Where does a DCN gather kernel wait?  PMC passes (one counter group per pass, --pmc only) of the DCN module at one layer shape with the stall /
# texture-path / L2 counters gfx950 offers.   usage (GPU box): bash tools/pmc_dcn_stalls.sh "8 48 160 128 64" [tag]
SHP=${1:-"8 48 160 128 64"}
TAG=${2:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_dcn_stalls_$(echo $SHP | tr ' ' '_')
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -oE "\b(SQ_[A-Z_0-9]+|TA_[A-Za-z_0-9]+|TCP_[A-Za-z_0-9]+|TD_[A-Za-z_0-9]+|TCC_[A-Za-z_0-9]+)\b" $OUT/avail.txt | sort -u > $OUT/names.txt
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA" "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" \
         "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_TOTAL_ACCESSES_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- python $R/tools/one_op.py dcnmod $SHP --reps 4 --eager > $OUT/p$i.log 2>&1 || echo "pass $i failed: $c" >> $OUT/failed.txt
done
echo "== DCN module, B H W Cin Cout = $SHP" > $R/gpurun_out/${TAG}_dcn_stalls_pmc.txt
python $R/tools/pmc_summary.py $OUT >> $R/gpurun_out/${TAG}_dcn_stalls_pmc.txt 2>&1
cat $OUT/failed.txt >> $R/gpurun_out/${TAG}_dcn_stalls_pmc.txt 2>/dev/null
grep -c . $OUT/names.txt >> $R/gpurun_out/${TAG}_dcn_stalls_pmc.txt
cp $OUT/names.txt $R/gpurun_out/${TAG}_pmc_counter_names.txt
cat $R/gpurun_out/${TAG}_dcn_stalls_pmc.txt
