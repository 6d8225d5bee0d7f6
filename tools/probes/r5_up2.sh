#!/bin/bash
# round 5: small variants of the up-sampling form (no GroupNorm sums in its epilogue; 8-row tiles at Cout <= 64)
set -u
OUT=gpurun_out/r5_up2; rm -rf $OUT; mkdir -p $OUT
A=founddiff_amd/lib/ab/head.so; B=founddiff_amd/lib/libfounddiff_hip.so
timeout 600 python -m pytest tests/test_gpu_round5.py -x -q -k "upsample" > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
for i in 1 2 3; do
  FOUNDDIFF_LIB=$A python tools/kbench.py conv3 2>/dev/null | grep "up=1" | sed 's/^/A /' | tee -a $OUT/kbench.txt
  FOUNDDIFF_LIB=$B python tools/kbench.py conv3 2>/dev/null | grep "up=1" | sed 's/^/B /' | tee -a $OUT/kbench.txt
  FD_CONV3_UP_TH8=1 FOUNDDIFF_LIB=$B python tools/kbench.py conv3 2>/dev/null | grep "up=1" | head -1 | sed 's/^/B(th8) /' | tee -a $OUT/kbench.txt
done
