#!/bin/bash
# A/B of two builds inside one gpurun call: ab_bench.sh libA.so libB.so [rounds]   (headline workload, short)
A=$1; B=$2; N=${3:-2}
for i in $(seq 1 $N); do
  for L in $A $B; do
    FOUNDDIFF_LIB=$L python bench.py --no-cpu-baseline --no-fp32-leg --no-extra-legs --no-roofline --no-smi --steps 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $L)', d['value'], d['ms_per_unet_forward_per_slice'])"
  done
done
