#!/bin/bash
# the two rocprofv3 --stats passes of tools/profile_round4.sh alone
set -u
OUT=gpurun_out/prof_r04; mkdir -p $OUT; export TMPDIR=/tmp
rm -rf $OUT/stats $OUT/stats_s1
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench --output-format csv -- python3 bench.py --no-cpu-baseline --no-fp32-leg --no-extra-legs --no-clock-replay > $OUT/bench_under_rocprof.json 2> $OUT/bench_prof.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_default_bench.csv \;
export FOUNDDIFF_STREAMS=1
rocprofv3 --kernel-trace --stats -d $OUT/stats_s1 -o bench --output-format csv -- python3 bench.py --batch 8 --no-cpu-baseline --no-fp32-leg --no-extra-legs --no-clock-replay > $OUT/bench_one_stream_b8_under_rocprof.json 2> $OUT/bench_prof_s1.err
unset FOUNDDIFF_STREAMS
find $OUT/stats_s1 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_one_stream_b8.csv \;
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
head -8 $OUT/kernel_stats_one_stream_b8.csv | cut -c1-140
