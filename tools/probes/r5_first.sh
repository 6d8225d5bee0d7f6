#!/bin/bash
# round 5, first GPU call: VALU issue-rate probe, the new round-5 tests, a short bench line (scan roofline entries, fp32s leg)
set -u
OUT=gpurun_out/r5_first; rm -rf $OUT; mkdir -p $OUT
(tools/probes/bin/valu_rate || (hipcc --offload-arch=gfx950 -O3 -o /tmp/vr tools/probes/valu_rate.hip && /tmp/vr)) > $OUT/valu_rate.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_round5.py -x -q -s > $OUT/pytest_round5.txt 2>&1
tail -5 $OUT/pytest_round5.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_first/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "fp32s", d.get("fp32s_parity_mode", {}).get("value"), "fp32", d.get("fp32_parity_mode", {}).get("value"))
print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "scan ms", d["roofline"].get("scan_family_ms_per_forward"))
for o in d["roofline"]["others"][:4]:
    print(o["kernel"][:60], o["frac"], o["avg_launch_us"])
PY
