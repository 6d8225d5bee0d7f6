#!/bin/bash
# round 5: batch-1 latency under the round's changes (up-sampling convs as four 2x2, the low-latency flag on the fused-x_proj scan)
set -u
OUT=gpurun_out/r5_b1; rm -rf $OUT; mkdir -p $OUT
bash tools/probes/knob_sweep_b1.sh "FD_NO_CONV3_UP2X=1" "FD_DBG_NO_LL_XPROJ=1" "FD_NO_CONV3_UP2X=1 FD_DBG_NO_LL_XPROJ=1" | tee $OUT/latency_b1_sweep.txt
