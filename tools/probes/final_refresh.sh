set -u
O=gpurun_out/final_refresh; rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 tools/forward_table.py --precision fp16 > $O/forward_launches_fp16.md 2> /dev/null
python3 tools/forward_table.py > $O/forward_launches.md 2> /dev/null
timeout 2400 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -4 $O/smoke.txt
head -1 $O/forward_launches_fp16.md; head -1 $O/forward_launches.md
python3 -c "
import json; b=json.load(open('$O/bench.json')); print(b['value'], b['fp16_mode']['value'], b['fp32s_parity_mode']['value'], b['roofline']['traffic'])"
