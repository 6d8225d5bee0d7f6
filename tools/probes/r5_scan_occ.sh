#!/bin/bash
# scan_chunk_kernel occupancy steps: N = 16 at 4 waves/SIMD (no scratch), N = 32 at 2 (no scratch; only the one-slice kernel set runs it)
A=$PWD/founddiff_amd/lib/libfounddiff_hip.so; B=$PWD/founddiff_amd/lib/ab/scan_n16occ4.so; C=$PWD/founddiff_amd/lib/ab/scan_n32occ2.so
for r in 1 2; do
  for L in $A $B; do echo "== $(basename $L) kbench scan"; FOUNDDIFF_LIB=$L python tools/kbench.py scan 2>/dev/null | grep -E "512, 16|256, 16"; FOUNDDIFF_LIB=$L python tools/kbench.py scanx 2>/dev/null | grep -E "16"; done
done
for r in 1 2; do
  for L in $A $C; do echo "== $(basename $L) latency_b1"; FOUNDDIFF_LIB=$L python tools/latency_b1.py 2>/dev/null | tail -1 | cut -c1-120; done
done
