cd /root/repo
for cfg in "ksplit=1" "conv_tile=5,ksplit=1" "conv_tile=5,ksplit=4" "conv_tile=5,ksplit=8" "conv_tile=6,ksplit=4" "conv_tile=9,ksplit=8" "conv_tile=8,ksplit=8"; do
 for shp in "8 12 40 512 512 3 1" "8 24 80 256 256 3 1" "8 24 80 128 256 3 2" "8 12 40 256 512 3 2"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sk -- python tools/one_kernel.py $shp halo=0,$cfg 20 > /dev/null 2>&1
  f=$(find /tmp/sk -name "*kernel_stats.csv" | head -1)
  echo "$cfg shape=[$shp] $(python - <<PY
import csv
tot=0
for r in csv.DictReader(open("$f")):
    if 'conv_igemm' in r['Name'] or 'splitk' in r['Name']:
        tot+=float(r['AverageNs'])
print('%.1f us'%(tot/1e3))
PY
)"
  rm -rf /tmp/sk
 done
done
