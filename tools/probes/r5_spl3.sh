#!/bin/bash
mkdir -p gpurun_out
python tools/tail_table.py > gpurun_out/tail_occ2.md 2>&1
FOUNDDIFF_LIB=$PWD/founddiff_amd/lib/ab/occ3.so python tools/tail_table.py > gpurun_out/tail_occ3.md 2>&1
for f in occ2 occ3; do echo "== $f"; grep -E "kid 15|total|sum" gpurun_out/tail_$f.md | head -30; done
python -m pytest tests/test_gpu_round5.py -q -x -k "split or fp32s" 2>&1 | tail -3
