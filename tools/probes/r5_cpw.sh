#!/bin/bash
# round 5: classes per workgroup of the up-sampling form (4: one workgroup per tile; 1: one per (tile, class)) at batch 8 and at batch 1
set -u
OUT=gpurun_out/r5_cpw; rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_round5.py -x -q -k "upsample" > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
for i in 1 2; do
  for c in 4 2 1; do
    FD_CONV3_UP_CPW=$c python tools/kbench.py conv3 2>/dev/null | grep "up=1" | sed "s/^/cpw=$c /" | tee -a $OUT/kbench_b8.txt
  done
done
for i in 1 2; do
  echo "low-latency default (cpw=1) $(python tools/latency_b1.py 2>/dev/null | tail -1 | cut -c1-20)" | tee -a $OUT/latency.txt
  echo "cpw=4 $(FD_CONV3_UP_CPW=4 python tools/latency_b1.py 2>/dev/null | tail -1 | cut -c1-20)" | tee -a $OUT/latency.txt
  echo "9-tap $(FD_NO_CONV3_UP2X=1 python tools/latency_b1.py 2>/dev/null | tail -1 | cut -c1-20)" | tee -a $OUT/latency.txt
done
bash tools/probes/ab_env.sh "FD_CONV3_UP_CPW=1" 2 --sample | tee $OUT/ab_cpw1.txt
