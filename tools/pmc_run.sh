#!/bin/bash
# SQ / LDS counter table (tools/pmc_table.py) of any python3 command:  bash tools/pmc_run.sh <tag> tools/kbench.py attn
# Two counter passes, each with --kernel-trace only (gpurun refuses --pmc next to the other trace domains).
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $OUT/sq1 -o k --output-format csv -- python3 "$@" > $OUT/sq1.out 2> $OUT/sq1.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d $OUT/sq2 -o k --output-format csv -- python3 "$@" > $OUT/sq2.out 2> $OUT/sq2.err
PMC_ROWS=${PMC_ROWS:-16} python3 tools/pmc_table.py $OUT/sq1 $OUT/sq2 > $OUT/pmc_table.md 2> $OUT/pmc_table.err
python3 - $OUT <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
c = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        c[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(d + "/insts.txt", "w") as o:
    for k, m in c.items():
        if "SQ_INSTS_VALU" in m:
            n = len(m["SQ_INSTS_VALU"])
            o.write(f"{k}: launches {n} VALU {sum(m['SQ_INSTS_VALU'])/n:.3e} LDS {sum(m['SQ_INSTS_LDS'])/n:.3e} SALU {sum(m['SQ_INSTS_SALU'])/n:.3e} WAIT_INST {sum(m.get('SQ_WAIT_INST_ANY',[0]))/max(len(m.get('SQ_WAIT_INST_ANY',[0])),1):.3e}\n")
PY
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/pmc_table.md; cat $OUT/insts.txt
