#!/bin/bash
# pw_gemm_kernel<256> (one 8-wave workgroup per CU) against <128> (two 4-wave workgroups per CU): kbench pw + the kernel test
for r in 1 2; do
  echo "== TNC 256"; python tools/kbench.py pw 2>/dev/null | grep "^pw"
  echo "== TNC 128"; FD_PWGEMM_TNC=128 python tools/kbench.py pw 2>/dev/null | grep "^pw"
done
FD_PWGEMM_TNC=128 python -m pytest tests -q -m gpu -k "pointwise_gemm" 2>&1 | tail -3
