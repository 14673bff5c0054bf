#!/bin/bash
# PMC passes for the fused heads kernel at the bench shape (B=8 bf16): HBM traffic per launch as
# MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, --pmc only, no tracing;
# FETCH_SIZE x2 on gfx950 for wide coalesced reads) plus MFMA-busy / LDS counters.  Every rocprofv3 call has its
# own `timeout`.  Writes raw CSVs to gpurun_out/pmc_heads/ and the summary + bench input to gpurun_out/.
#   usage (GPU box): bash tools/pmc_heads.sh [tag]        then copy gpurun_out/<tag>_heads_* into profiles/
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_heads
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- python $R/tools/one_op.py heads --reps 4 --eager > $OUT/p$i.log 2>&1
done
python $R/tools/pmc_summary.py $OUT > $R/gpurun_out/${TAG}_heads_pmc.txt
# raw per-dispatch rows of the heads kernel only (small): the CSV the bench's `traffic` comes from
python - <<PY
import csv, glob, json
rows = []
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "heads_fused" in r.get("Kernel_Name", ""):
            rows.append({k: r[k] for k in ("Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value") if k in r})
with open("$R/gpurun_out/${TAG}_heads_pmc.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
    w.writeheader()
    for r in rows:
        r["Kernel_Name"] = r["Kernel_Name"][:60]
        w.writerow(r)
def mean(name):
    v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == name]
    return sum(v) / len(v) if v else None
fs, ws = mean("FETCH_SIZE"), mean("WRITE_SIZE")
if fs is not None and ws is not None:
    json.dump({"kernel": "heads_fused_kernel<bf16>", "batch": 8, "dtype": "bf16", "fetch_size_kb": fs, "write_size_kb": ws,
               "traffic_bytes": int(2 * fs * 1024 + ws * 1024),
               "source": "profiles/${TAG}_heads_pmc.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                         "2 x FETCH_SIZE + WRITE_SIZE: gfx950 tallies 128-B read requests at 64 B)"},
              open("$R/gpurun_out/${TAG}_heads_traffic.json", "w"))
PY
