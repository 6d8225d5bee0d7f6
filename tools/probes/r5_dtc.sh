#!/bin/bash
# round 5: dt cache of the level-0 scan (phase A stores dt, phase C loads it) -- tests, kernel-level and forward-level A/B against HEAD's library
set -u
OUT=gpurun_out/r5_dtc; rm -rf $OUT; mkdir -p $OUT
A=founddiff_amd/lib/ab/head.so; B=founddiff_amd/lib/libfounddiff_hip.so
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "scan" > $OUT/pytest_scan.txt 2>&1; tail -2 $OUT/pytest_scan.txt
timeout 1200 python -m pytest tests/test_gpu_e2e.py -x -q -k "mamba or odd or properties_512 or concurrent" > $OUT/pytest_e2e.txt 2>&1; tail -2 $OUT/pytest_e2e.txt
for i in 1 2; do
  FOUNDDIFF_LIB=$A python tools/kbench.py scanx 2>/dev/null | head -1 | sed 's/^/A /' | tee -a $OUT/kbench.txt
  FOUNDDIFF_LIB=$B python tools/kbench.py scanx 2>/dev/null | head -1 | sed 's/^/B /' | tee -a $OUT/kbench.txt
  FD_SCAN_NO_DTC=1 FOUNDDIFF_LIB=$B python tools/kbench.py scanx 2>/dev/null | head -1 | sed 's/^/B(no dt cache) /' | tee -a $OUT/kbench.txt
done
bash tools/probes/ab.sh $A $B 3 --sample | tee $OUT/ab.txt
