#!/bin/bash
# round 5: up-sampling 3x3 convs as four 2x2 convs on the source grid -- tests, kernel-level and forward-level A/B (FD_NO_CONV3_UP2X=1 = the 9-tap form)
set -u
OUT=gpurun_out/r5_up2x; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -s -k "upsample" > $OUT/pytest_up.txt 2>&1; tail -3 $OUT/pytest_up.txt; grep "up-sampling conv" $OUT/pytest_up.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" > $OUT/pytest_conv.txt 2>&1; tail -2 $OUT/pytest_conv.txt
for i in 1 2; do
  FD_NO_CONV3_UP2X=1 python tools/kbench.py conv3 2>/dev/null | grep "up=1" | sed 's/^/A /' | tee -a $OUT/kbench.txt
  python tools/kbench.py conv3 2>/dev/null | grep "up=1" | sed 's/^/B /' | tee -a $OUT/kbench.txt
done
bash tools/probes/ab_env.sh "FD_NO_CONV3_UP2X=1" 3 --sample | tee $OUT/ab.txt
timeout 1500 python -m pytest tests/test_gpu_e2e.py -x -q -k "512 or 256 or full_arch or drift" > $OUT/pytest_e2e.txt 2>&1; tail -3 $OUT/pytest_e2e.txt
