#!/bin/bash
# round 5 mid-way: the full GPU suite, the forward launch by launch, and an A/B of the fused in_proj kernel at 128x128 (FD_PWDW_MINPIX)
set -u
OUT=gpurun_out/r5_mid; rm -rf $OUT; mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
python tools/forward_table.py > $OUT/forward_launches.md 2> $OUT/forward_table.err; head -3 $OUT/forward_launches.md
bash tools/probes/ab_env.sh "FD_PWDW_MINPIX=16384" 3 | tee $OUT/ab_minpix.txt
