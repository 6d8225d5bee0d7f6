#!/bin/bash
# alternate two library builds inside one gpurun call:  bash tools/probes/ab.sh <libA.so> <libB.so> [rounds] [--sample]
A=$1; B=$2; N=${3:-3}; EXTRA=$4
for i in $(seq 1 $N); do
  FOUNDDIFF_LIB=$A python tools/ab_forward.py A $EXTRA 2>/dev/null | tail -1
  FOUNDDIFF_LIB=$B python tools/ab_forward.py B $EXTRA 2>/dev/null | tail -1
done
