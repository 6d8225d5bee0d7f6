#!/bin/bash
# round 5: what would eliminating the small reduction launches buy (timing only; buffers hold the previous forward's values)?
set -u
OUT=gpurun_out/r5_skip; rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do
  AB_SKIP=fd_gn_finalize,fd_chan_attn_weff python tools/ab_forward.py "skip gn_finalize+weff" --sample 2>/dev/null | tail -1 | tee -a $OUT/skip_small.txt
  python tools/ab_forward.py "default" --sample 2>/dev/null | tail -1 | tee -a $OUT/skip_small.txt
done
AB_SKIP=fd_gn_finalize,fd_chan_attn_weff,fd_ln_modulate,fd_ln_gate python tools/ab_forward.py "skip + ln_modulate + ln_gate" --sample 2>/dev/null | tail -1 | tee -a $OUT/skip_small.txt
python tools/ab_forward.py "default" --sample 2>/dev/null | tail -1 | tee -a $OUT/skip_small.txt
