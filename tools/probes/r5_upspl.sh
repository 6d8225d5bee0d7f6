#!/bin/bash
python -m pytest tests/test_gpu_round5.py -q -x -k "split or fp32s or upsample" 2>&1 | tail -4
python tools/forward_table.py --precision fp32s > gpurun_out/fwd_fp32s_c.md 2>/dev/null; head -1 gpurun_out/fwd_fp32s_c.md; grep "kid\|split-bf16" gpurun_out/fwd_fp32s_c.md | grep -E "four 2x2|up" 
python tools/tail_table.py > gpurun_out/tail_c.md 2>/dev/null; head -1 gpurun_out/tail_c.md; grep "kid 16" gpurun_out/tail_c.md
