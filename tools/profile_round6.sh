#!/bin/bash
# Evidence run of round 6 (run on the GPU box from the repo root): bash tools/profile_round6.sh r06
# Counters in their own passes with --kernel-trace only (gpurun refuses --pmc next to the other trace domains).
set -u
TAG=${1:-r06}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
# 1. HBM traffic first (bench.py's roofline.forward reads profiles/traffic.json and checks its csrc hash): FETCH_SIZE / WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o fwd --output-format csv -- python3 tools/run_forward.py --n 2 --batch 8 > /dev/null 2> $OUT/$c.err
done
python3 tools/traffic.py $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/traffic.json > $OUT/traffic_summary.txt 2>&1
cp $OUT/traffic.json profiles/traffic.json
# 2. the benches WITHOUT a profiler (the numbers, with clock / power of the timed region)
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
FOUNDDIFF_STREAMS=1 python3 bench.py --batch 8 --no-cpu-baseline --no-fp32-leg --no-extra-legs --no-roofline > $OUT/bench_one_stream_b8.json 2> $OUT/bench_s1.err
# 3. rocprofv3 kernel stats of the default bench and of the one-stream batch-8 bench (exclusive durations)
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench --output-format csv -- python3 bench.py --no-cpu-baseline --no-fp32-leg --no-extra-legs --no-clock-replay > $OUT/bench_under_rocprof.json 2> $OUT/bench_prof.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_default_bench.csv \;
export FOUNDDIFF_STREAMS=1
rocprofv3 --kernel-trace --stats -d $OUT/stats_s1 -o bench --output-format csv -- python3 bench.py --batch 8 --no-cpu-baseline --no-fp32-leg --no-extra-legs --no-clock-replay > $OUT/bench_one_stream_b8_under_rocprof.json 2> $OUT/bench_prof_s1.err
unset FOUNDDIFF_STREAMS
find $OUT/stats_s1 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_one_stream_b8.csv \;
# 4. SQ / LDS counters (6 forwards: the one-off DA-CLIP encode is < 4 % of the sums)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace -d $OUT/sq1 -o fwd --output-format csv -- python3 tools/run_forward.py --n 6 --batch 8 > /dev/null 2> $OUT/sq1.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/sq2 -o fwd --output-format csv -- python3 tools/run_forward.py --n 6 --batch 8 > /dev/null 2> $OUT/sq2.err
PMC_ROWS=32 python3 tools/pmc_table.py $OUT/sq1 $OUT/sq2 > $OUT/pmc_table.md 2> $OUT/pmc_table.err
python3 tools/pmc_two_stream.py $OUT/sq1 $OUT/sq2 $OUT/traffic.json $OUT/bench.json $OUT/bench_one_stream_b8.json 6 > $OUT/pmc_two_stream.md 2> $OUT/pmc_two_stream.err
# 4b. the forward launch by launch (each launch replayed alone between events)
python3 tools/forward_table.py > $OUT/forward_launches.md 2> /dev/null
python3 tools/forward_table.py --precision fp32s > $OUT/forward_launches_fp32s.md 2> /dev/null
python3 tools/tail_table.py > $OUT/tail_launches.md 2> /dev/null
# 5. per-stage times of one forward
python3 tools/stage_times.py --batches 8 --detail > $OUT/stage_detail_b8.md 2> /dev/null
# keep only the summaries (the raw traces are large)
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -2 $OUT/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
FOUNDDIFF_LOW_LATENCY=1 python3 tools/forward_table.py --batch 1 > $OUT/forward_launches_b1_low_latency.md 2> /dev/null
# 6. precision='fp16' (the binary16 build): loop drift against the CPU oracle at both sizes beside bf16, stage by stage against the fp32
# engine, repeatability / batch invariance, the two GEMM kernels bit for bit, and the forward launch by launch
python3 tools/probes/fp16_drift.py --size 256 --modes bf16,fp16 --tails 1:0,1:2,0:2 > $OUT/fp16_drift_256.md 2> /dev/null
python3 tools/probes/fp16_drift.py --size 512 --modes bf16,fp16 --tails 1:0,1:2,0:2 > $OUT/fp16_drift_512.md 2> /dev/null
python3 tools/probes/fp16_stages.py --size 256 > $OUT/fp16_stage_errors_256.md 2> /dev/null
python3 tools/probes/fp16_repeat.py > $OUT/fp16_repeat.txt 2> /dev/null
python3 tools/probes/pwg_vs_igemm.py > $OUT/fp16_pwg_vs_igemm.txt 2> /dev/null
python3 tools/forward_table.py --precision fp16 > $OUT/forward_launches_fp16.md 2> /dev/null
ls -la $OUT; cat $OUT/pmc_two_stream.md; cat $OUT/traffic_summary.txt; cat $OUT/fp16_drift_512.md
