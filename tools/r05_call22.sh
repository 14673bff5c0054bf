#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
MFX_TRACE_CW=1 timeout 900 python bench.py --mode train --train-steps 2 --train-warmup 1 --train-repeats 1 --legs none 2> gpurun_out/cw_trace.err | tail -1 | cut -c1-300
grep cw-fallback gpurun_out/cw_trace.err | sort | uniq -c | sort -rn > gpurun_out/cw_fallbacks.txt
cat gpurun_out/cw_fallbacks.txt
