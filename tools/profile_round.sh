#!/bin/bash
# Evidence run for profiles/: rocprofv3 kernel stats of the default bench, HBM traffic (two --pmc passes) and the SQ /
# LDS counter table of one batch-8 forward.  Run on the GPU box from the repo root:  bash tools/profile_round.sh r02
# (counters in their own passes with --kernel-trace only: gpurun refuses --pmc next to the trace domains)
set -u
TAG=${1:-rXX}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench --output-format csv -- python3 bench.py --no-cpu-baseline --no-fp32-leg --no-extra-legs > $OUT/bench_prof.json 2> $OUT/bench_prof.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# the same bench on ONE stream at the sub-batch size (8): per-kernel durations are exclusive there (with the default two concurrent sub-batches a
# kernel's duration includes the time it shares the chip with the other stream's kernels) -- the cross-check of roofline.avg_launch_us
export FOUNDDIFF_STREAMS=1
rocprofv3 --kernel-trace --stats -d $OUT/stats_s1 -o bench --output-format csv -- python3 bench.py --batch 8 --no-cpu-baseline --no-fp32-leg --no-extra-legs > $OUT/bench_prof_s1.json 2> $OUT/bench_prof_s1.err
unset FOUNDDIFF_STREAMS
find $OUT/stats_s1 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_one_stream.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o fwd --output-format csv -- python3 tools/run_forward.py --n 2 --batch 8 > /dev/null 2> $OUT/$c.err
done
python3 tools/traffic.py $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/traffic.json > $OUT/traffic_summary.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace -d $OUT/sq1 -o fwd --output-format csv -- python3 tools/run_forward.py --n 2 --batch 8 > /dev/null 2> $OUT/sq1.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/sq2 -o fwd --output-format csv -- python3 tools/run_forward.py --n 2 --batch 8 > /dev/null 2> $OUT/sq2.err
PMC_ROWS=30 python3 tools/pmc_table.py $OUT/sq1 $OUT/sq2 > $OUT/pmc_table.md 2> $OUT/pmc_table.err
# keep only the summaries (the raw traces are large)
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
ls -la $OUT
