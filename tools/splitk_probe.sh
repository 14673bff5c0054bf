cd /root/repo
for cfg in "halo=0" "halo=8" "halo=12" "halo=13" "halo=14" "halo=15"; do
 for shp in "8 12 40 512 512 3 1" "8 24 80 256 256 3 1" "8 48 160 128 128 3 1" "8 96 320 64 64 3 1"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sk -- python tools/one_kernel.py $shp $cfg 20 > /dev/null 2>&1
  f=$(find /tmp/sk -name "*kernel_stats.csv" | head -1)
  echo "$cfg shape=[$shp] $(python - <<PY
import csv
tot=0
for r in csv.DictReader(open("$f")):
    if 'conv_igemm' in r['Name'] or 'splitk' in r['Name'] or 'conv3x3_wave' in r['Name']:
        tot+=float(r['AverageNs'])
print('%.1f us'%(tot/1e3))
PY
)"
  rm -rf /tmp/sk
 done
done
