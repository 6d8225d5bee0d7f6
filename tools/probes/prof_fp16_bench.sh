# rocprofv3 kernel stats of the fp16 bench (one stream, batch 8: exclusive durations) and the plain fp16 bench line
set -u
O=gpurun_out/prof_fp16; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
python3 bench.py --precision fp16 --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err
export FOUNDDIFF_STREAMS=1
rocprofv3 --kernel-trace --stats -d $O/stats_s1 -o bench --output-format csv -- python3 bench.py --precision fp16 --batch 8 --no-cpu-baseline --no-fp32-leg --no-extra-legs --no-roofline --no-clock-replay > $O/bench_fp16_one_stream_b8_under_rocprof.json 2> $O/prof.err
unset FOUNDDIFF_STREAMS
find $O/stats_s1 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_fp16_one_stream_b8.csv \;
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
head -8 $O/kernel_stats_fp16_one_stream_b8.csv | cut -c1-170
python3 -c "
import json; b=json.load(open('$O/bench_fp16.json')); print(b['value'], b['dtype'][:90], b['config']['workload'][:80])"
