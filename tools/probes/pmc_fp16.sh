set -u
OUT=gpurun_out/pmc_f16; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace -d $OUT/sq1 -o fwd --output-format csv -- python3 tools/run_forward.py --n 6 --batch 8 --precision fp16 > /dev/null 2> $OUT/sq1.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/sq2 -o fwd --output-format csv -- python3 tools/run_forward.py --n 6 --batch 8 --precision fp16 > /dev/null 2> $OUT/sq2.err
PMC_ROWS=32 python3 tools/pmc_table.py $OUT/sq1 $OUT/sq2 > $OUT/pmc_table_fp16.md 2> $OUT/pmc_table.err
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
head -40 $OUT/pmc_table_fp16.md | cut -c1-220
python3 -m pytest tests/test_gpu_fp16.py tests/test_gpu_round5.py tests/test_gpu_round6.py -m gpu -q -k "oracle" -s 2>&1 | grep -v "^$" | tail -12
