#!/bin/bash
# PMC passes over the WHOLE eager inference step (B = 8, bf16): per kernel, how busy are the vector, matrix, LDS and texture paths and how long do waves wait?
# --pmc only, one counter group per pass.   usage (GPU box): bash tools/pmc_step.sh [tag]   -> gpurun_out/<tag>_step_pmc.txt (+ a per-kernel table .md)
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_step; rm -rf $OUT; mkdir -p $OUT
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- python $R/bench.py --legs none --no-families --no-cpu-baseline --no-graph --steps 4 --warmup 2 --repeats 1 > $OUT/p$i.log 2>&1 || echo "pass $i failed: $c" >> $OUT/failed.txt
done
python $R/tools/pmc_summary.py $OUT > $R/gpurun_out/${TAG}_step_pmc.txt 2>&1
cat $OUT/failed.txt >> $R/gpurun_out/${TAG}_step_pmc.txt 2>/dev/null
python - <<'PY' > $R/gpurun_out/${TAG}_step_pmc.md
import re, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); TAG = os.environ.get("TAG_", "r05")
cur, data = None, {}
for l in open(R + "/gpurun_out/%s_step_pmc.txt" % TAG):
    if not l.startswith(" "):
        cur = l.strip(); data[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+([\d.]+)", l)
        if m and cur: data[cur][m.group(1)] = float(m.group(2))
print("| kernel | GUI-active cycles / XCD | waves | wave-cycles waiting % | issuing % | VALU insts / MFMA insts | MFMA pipe busy % | LDS conflict % of LDS-active | L1 hit % | avg L2 latency cyc |")
print("|---|---|---|---|---|---|---|---|---|---|")
rows = []
for k, c in data.items():
    if "SQ_WAVE_CYCLES" not in c or c.get("SQ_WAVE_CYCLES", 0) == 0: continue
    wc = c["SQ_WAVE_CYCLES"]; g = c.get("GRBM_GUI_ACTIVE", 0) / 8
    mf = c.get("SQ_INSTS_MFMA", 0)
    rows.append((g, "| %s | %.0f | %.0f | %.0f | %.0f | %s | %.0f | %.0f | %s | %s |" % (
        k[:70], g, c.get("SQ_WAVES", 0), 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        ("%.1f" % (c.get("SQ_INSTS_VALU", 0) / mf)) if mf else "-", 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / max(g, 1) if g else 0,
        100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1),
        ("%.0f" % (100 * (1 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]))) if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") else "-",
        ("%.0f" % (c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"])) if c.get("TCP_TCC_READ_REQ_sum") else "-")))
for _, r in sorted(rows, reverse=True): print(r)
PY
head -40 $R/gpurun_out/${TAG}_step_pmc.md
