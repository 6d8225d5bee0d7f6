#!/bin/bash
# round 5: sub-batch / stream shape of the sampler after this round's kernels (real bench.py lines, no extra legs)
set -u
OUT=gpurun_out/r5_streams; rm -rf $OUT; mkdir -p $OUT
run() { echo -n "$1: " | tee -a $OUT/streams.txt; env $2 python bench.py --batch $3 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-extra-legs --no-roofline --no-smi 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'slices/s', d['ms_per_unet_forward_per_slice'], 'ms/slice-forward')" | tee -a $OUT/streams.txt; }
for i in 1 2; do
  run "2 x 8 (default)" "FOUNDDIFF_STREAMS=2" 16
  run "3 x 8" "FOUNDDIFF_STREAMS=3" 24
  run "4 x 8" "FOUNDDIFF_STREAMS=4" 32
  run "2 x 12" "FOUNDDIFF_STREAMS=2" 24
  run "3 x 6" "FOUNDDIFF_STREAMS=3" 18
done
