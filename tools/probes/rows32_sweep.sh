export FOUNDDIFF_LIB=$PWD/founddiff_amd/lib/ab/dev.so
for v in "FD_ROWS32_PER_CU=3" "FD_ROWS32_PER_CU=4" "FD_ROWS32_PER_CU=2" "FD_ROWS32_PER_CU=3"; do
  echo "== $v"
  env $v python tools/forward_table.py --precision fp32s 2>/dev/null | grep -E "row-GEMM fp32|^# One" | awk -F'|' '{s+=$5} /One/ {print} END {print "rows32 sum us", s}'
done
