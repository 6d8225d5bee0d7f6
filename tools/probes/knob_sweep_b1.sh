#!/bin/bash
# batch-1 latency (ms per 50-step slice) under single environment knobs, baseline interleaved
for SW in "$@"; do
  echo "default $(python tools/latency_b1.py 2>/dev/null | tail -1 | cut -c1-20)"
  echo "$SW $(env $SW python tools/latency_b1.py 2>/dev/null | tail -1 | cut -c1-20)"
done
echo "default $(python tools/latency_b1.py 2>/dev/null | tail -1 | cut -c1-20)"
