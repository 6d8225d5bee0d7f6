#!/bin/bash
# round 5: conflict-free swizzle of the depthwise tile (ts_off) -- tests, kernel-level and forward-level A/B against the previous commit's library
set -u
OUT=gpurun_out/r5_ts; rm -rf $OUT; mkdir -p $OUT
A=founddiff_amd/lib/ab/e90b04f.so; B=founddiff_amd/lib/libfounddiff_hip.so
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "pw_dw or dwconv or gram or mamba or fused" > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
for i in 1 2; do
  FOUNDDIFF_LIB=$A python tools/kbench.py pwdw 2>/dev/null | sed 's/^/A /' | tee -a $OUT/kbench.txt
  FOUNDDIFF_LIB=$B python tools/kbench.py pwdw 2>/dev/null | sed 's/^/B /' | tee -a $OUT/kbench.txt
done
bash tools/probes/ab.sh $A $B 3 --sample | tee $OUT/ab.txt
AB_SKIP=fd_gn_finalize,fd_chan_attn_weff python tools/ab_forward.py "skip gn_finalize+weff (timing only)" --sample | tail -1 | tee $OUT/skip_small.txt
python tools/ab_forward.py "default" --sample | tail -1 | tee -a $OUT/skip_small.txt
