#!/bin/bash
# HBM traffic of the DCN backward group (64->64 @ 96x320, B=8, bf16): FETCH_SIZE and WRITE_SIZE in separate --pmc passes (no tracing),
# summed over the group's kernels per backward call.  Writes gpurun_out/<tag>_dcnbwd_pmc.{txt,csv} and <tag>_dcnbwd_traffic.json.
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_dcnbwd
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- python $R/tools/one_op.py dcnbwd 8 96 320 64 64 --reps 4 --eager > $OUT/p$i.log 2>&1
done
python $R/tools/pmc_summary.py $OUT > $R/gpurun_out/${TAG}_dcnbwd_pmc.txt
python - <<PY
import csv, glob, json, collections
group = ("conv_igemm_kernel", "dcn_bwd_sample_kernel", "dcn_bwd_sample_wgrad_kernel", "dcn_bwd_tile_kernel", "dcn_bwd_far_kernel", "conv_wgrad_mfma_kernel",
         "dcn_bwd_sample_wgrad_fly_kernel", "dcn_bwd_tile_fly_kernel", "dcn_bwd_far_fly_kernel",
         "wgrad_reduce_kernel", "colsum_chunk_kernel", "bt_pack_weight_t", "zero_fill_kernel")
rows = []
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if any(g in r.get("Kernel_Name", "") for g in group):
            rows.append({k: r[k] for k in ("Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value") if k in r})
with open("$R/gpurun_out/${TAG}_dcnbwd_pmc.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
    w.writeheader()
    for r in rows:
        r["Kernel_Name"] = r["Kernel_Name"][:70]
        w.writerow(r)
tot = collections.defaultdict(float); calls = collections.Counter()
for r in rows:
    tot[r["Counter_Name"]] += float(r["Counter_Value"])
    if "dcn_bwd_tile" in r["Kernel_Name"]:
        calls[r["Counter_Name"]] += 1
n = max(1, min(calls.values()) if calls else 1)                     # backward calls seen by every pass (one tile kernel per call)
fs, ws = tot["FETCH_SIZE"] / n, tot["WRITE_SIZE"] / n
json.dump({"kernel": "DCNv2 backward group 64->64 @ 96x320", "batch": 8, "dtype": "bf16", "calls": n, "fetch_size_kb": fs, "write_size_kb": ws,
           "traffic_bytes": int(2 * fs * 1024 + ws * 1024),
           "source": "profiles/${TAG}_dcnbwd_pmc.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, summed over the group's kernels per "
                     "backward call; 2 x FETCH_SIZE + WRITE_SIZE: gfx950 tallies 128-B read requests at 64 B)"},
          open("$R/gpurun_out/${TAG}_dcnbwd_traffic.json", "w"))
print(open("$R/gpurun_out/${TAG}_dcnbwd_traffic.json").read())
PY
