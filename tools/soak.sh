O=gpurun_out/r06_soak; mkdir -p $O; rm -f $O/*.txt
echo "## inference B=8 bf16, 4000 timed steps per repeat (about 10 s of back-to-back replays each)" >> $O/soak.txt
timeout 600 python bench.py --steps 4000 --warmup 50 --no-cpu-baseline --legs none --no-families 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('timing'))" >> $O/soak.txt
echo "## default 30 steps, same box" >> $O/soak.txt
timeout 600 python bench.py --no-cpu-baseline --legs none --no-families 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('timing'))" >> $O/soak.txt
echo "## training B=8 bf16, 600 timed steps per repeat" >> $O/soak.txt
timeout 900 python bench.py --mode train --steps 600 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('timing'), d['config'].get('loss_last_step'))" >> $O/soak.txt
cat $O/soak.txt
