export FOUNDDIFF_LIB=$PWD/founddiff_amd/lib/ab/dev.so
for v in "" "FD_SCAN_NO_CPL2=1" "FD_SCAN_CL=64" "FD_SCAN_CL=256" "FD_SCAN_NO_SEQ=1"; do
  echo "== $v"
  env $v python tools/forward_table.py 2>/dev/null | grep -E "selective_scan|^# One" | cut -c1-100
done
