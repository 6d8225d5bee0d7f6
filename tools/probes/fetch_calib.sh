#!/bin/bash
# FETCH_SIZE per access shape (tools/probes/fetch_calib.hip); run from the repo root on the GPU box
set -u
OUT=gpurun_out/fetch_calib; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/fc tools/probes/fetch_calib.hip 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f -o fc --output-format csv -- /tmp/fc > $OUT/run.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    kb = sum(v) / len(v)
    print(f"{k:10s} FETCH_SIZE {kb:12.0f} KiB per launch = {kb / (512 * 1024):.3f} x the 512 MiB read  (n = {len(v)})")
PY
