#!/bin/bash
# round 5: phase A with the u block staged in LDS (one read of u) against the round-4 library, alternated in one call
set -u
OUT=gpurun_out/r5_scan_ab; rm -rf $OUT; mkdir -p $OUT
R4=founddiff_amd/lib/ab/r4.so; NEW=founddiff_amd/lib/libfounddiff_hip.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "scan" > $OUT/pytest_scan.txt 2>&1; tail -3 $OUT/pytest_scan.txt
timeout 1200 python -m pytest tests/test_gpu_round5.py -x -q -s > $OUT/pytest_round5.txt 2>&1; tail -3 $OUT/pytest_round5.txt; grep "vs oracle\|ln rows" $OUT/pytest_round5.txt
for i in 1 2; do
  FOUNDDIFF_LIB=$R4 python tools/kbench.py scanx 2>/dev/null | sed 's/^/A /' | tee -a $OUT/kbench.txt
  FOUNDDIFF_LIB=$NEW python tools/kbench.py scanx 2>/dev/null | sed 's/^/B /' | tee -a $OUT/kbench.txt
done
bash tools/probes/ab.sh $R4 $NEW 3 --sample | tee $OUT/ab.txt
