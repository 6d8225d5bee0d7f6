#!/bin/bash
# forward time (batch 8, one stream) under single environment knobs, baseline interleaved: bash tools/probes/knob_sweep.sh "A=1" "B=2" ...
for SW in "$@"; do
  python tools/ab_forward.py "default" 2>/dev/null | tail -1
  env $SW python tools/ab_forward.py "$SW" 2>/dev/null | tail -1
done
python tools/ab_forward.py "default" 2>/dev/null | tail -1
