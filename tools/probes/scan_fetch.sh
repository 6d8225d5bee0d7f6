#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the level-0 scan launches alone (kbench scanx); run from the repo root on the GPU box
set -u
OUT=gpurun_out/scan_fetch_${1:-x}; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f -o k --output-format csv -- python3 tools/kbench.py scanx > $OUT/run.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and ("scan" in r["Kernel_Name"]):
            acc[r["Kernel_Name"][:75]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:75s} n={len(v):3d} FETCH x2 = {2 * 1024 * sum(v) / len(v) / 1e6:9.1f} MB per launch")
PY
grep "scanx" $OUT/run.txt
find $OUT -name "*.csv" -delete
