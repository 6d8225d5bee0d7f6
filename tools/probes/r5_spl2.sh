#!/bin/bash
# SPL halo conv: committed lib vs reload-skip vs reload-skip + 8-row tiles (tail table, conv lines only) + the kernel tests
mkdir -p gpurun_out
FOUNDDIFF_LIB=$PWD/founddiff_amd/lib/ab/head.so python tools/tail_table.py > gpurun_out/tail_head.md 2>&1
python tools/tail_table.py > gpurun_out/tail_new.md 2>&1
FD_CONV3_SPLIT_TH8=1 python tools/tail_table.py > gpurun_out/tail_th8.md 2>&1
for f in head new th8; do echo "== $f"; grep -E "kid 15|total|sum" gpurun_out/tail_$f.md | head -30; done
python -m pytest tests/test_gpu_round5.py -q -x -k "split or fp32s" 2>&1 | tail -3
FD_CONV3_SPLIT_TH8=1 python -m pytest tests/test_gpu_round5.py -q -x -k "split" 2>&1 | tail -3
