#!/bin/bash
# alternate an environment switch inside one gpurun call:  bash tools/probes/ab_env.sh "FD_NO_PWGEMM=1" [rounds] [--sample]
SW=$1; N=${2:-3}; EXTRA=${3:-}
for i in $(seq 1 $N); do
  env $SW python tools/ab_forward.py "A($SW)" $EXTRA 2>/dev/null | tail -1
  python tools/ab_forward.py "B(default)" $EXTRA 2>/dev/null | tail -1
done
