#!/bin/bash
# round 5: split-bf16 form of the halo kernel (fp32s engine) -- kernel test, e2e fp32s tests, the fp32s leg and the production sample() A/B (FD_NO_CONV3_SPLIT=1 = generic)
set -u
OUT=gpurun_out/r5_spl; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -s -k "split_bf16 or fp32s" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt; grep "split-bf16 3x3" $OUT/pytest.txt
python - <<'PY' 2>/dev/null | tee $OUT/fp32s_leg.txt
import os, subprocess, sys, json
for tag, env in (("generic split igemm", {"FD_NO_CONV3_SPLIT": "1"}), ("halo-tiled split", {})) * 2:
    r = subprocess.run([sys.executable, "-c", "import torch, bench; from founddiff_amd import synth; dev=torch.device('cuda'); _, ld = synth.ct_phantom(16, 512, seed=10); x=torch.from_numpy(ld).to(dev); n=torch.randn(16,1,512,512,device=dev); print(bench.fp32_parity_leg(dev, x, n, precision='fp32s')['value'])"],
                       env={**os.environ, **env}, capture_output=True, text=True)
    print(tag, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
PY
bash tools/probes/ab_env.sh "FD_NO_CONV3_SPLIT=1" 3 --sample | tee $OUT/ab.txt
