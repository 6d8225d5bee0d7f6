#!/bin/bash
# alternate several environment settings inside one gpurun call:
#   bash tools/probes/ab_env3.sh [rounds] [--sample] -- "FOO=1" "FOO=2 BAR=3" ...      (a bare "-" = default environment)
N=${1:-2}; shift; EXTRA=""
if [ "$1" != "--" ]; then EXTRA=$1; shift; fi
shift
for i in $(seq 1 $N); do
  for SW in "$@"; do
    if [ "$SW" = "-" ]; then python tools/ab_forward.py "default" $EXTRA 2>/dev/null | tail -1
    else env $SW python tools/ab_forward.py "$SW" $EXTRA 2>/dev/null | tail -1; fi
  done
done
