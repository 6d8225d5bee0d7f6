#!/bin/bash
# round 5: chunk length of the chunked scans after the block-wise phase A (kernel level; FD_SCAN_CL forces every level)
set -u
OUT=gpurun_out/r5_cl; rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do
  for cl in 0 64 128 256; do
    if [ $cl = 0 ]; then python tools/kbench.py scanx 2>/dev/null | sed "s/^/default /" | tee -a $OUT/kbench.txt
    else FD_SCAN_CL=$cl python tools/kbench.py scanx 2>/dev/null | sed "s/^/CL=$cl /" | tee -a $OUT/kbench.txt; fi
  done
done
python tools/latency_b1.py 2>/dev/null | tail -1 | cut -c1-60 | tee $OUT/latency_b1.txt
