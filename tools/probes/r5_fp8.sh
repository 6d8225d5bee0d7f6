#!/bin/bash
# round 5: the fp8 mode's up-sampling convolutions -- e4m3 9-tap (FOUNDDIFF_FP8_UPCONV=1, rounds 2-4) against the bf16 four-2x2 form
set -u
OUT=gpurun_out/r5_fp8; rm -rf $OUT; mkdir -p $OUT
run() { echo -n "$1: " | tee -a $OUT/fp8.txt; env $2 python bench.py --precision fp8 --ddim-steps 25 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-extra-legs --no-roofline --no-smi 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'slices/s')" | tee -a $OUT/fp8.txt; }
for i in 1 2 3; do
  run "e4m3 9-tap up-sampling convs" "FOUNDDIFF_FP8_UPCONV=1"
  run "bf16 four 2x2 (default)" "FOUNDDIFF_FP8_UPCONV=0"
done
python tools/fp8_check.py 2>/dev/null | tail -3 | tee -a $OUT/fp8.txt
