#!/usr/bin/env bash
# Run on the GPU box (through gpurun): the rocprofv3 evidence behind bench.py's numbers, as text under gpurun_out/.
#   kernel-trace + stats of the default bench run, separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the headline,
#   the per-query timeline at 1 M rows, and the SQ counters of the matrix-core batch pass.
set -uo pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_r06
rm -rf "$OUT"; mkdir -p "$OUT"
R="rocprofv3 --output-format csv"
# (the headline alone: the default run also measures configs[1] and configs[4] in the same process, whose launches of the same
#  kernel would be averaged in)
$R --kernel-trace --stats -d $OUT/stats -- python bench.py --no-cpu-baseline --no-configs > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
python scripts/rocprof_csv_summary.py stats $OUT/stats > $OUT/kernel_trace_stats.txt 2>&1
$R --kernel-trace --stats -d $OUT/stats_all -- python bench.py --no-cpu-baseline > $OUT/bench_all_under_rocprof.json 2> $OUT/stats_all.err
python scripts/rocprof_csv_summary.py stats $OUT/stats_all > $OUT/kernel_trace_stats_all_configs.txt 2>&1
SMALL="--no-cpu-baseline --no-configs --steps 2 --warmup 1 --queries-per-step 2"
$R --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python bench.py $SMALL > /dev/null 2> $OUT/fetch.err
$R --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python bench.py $SMALL > /dev/null 2> $OUT/write.err
( python scripts/rocprof_csv_summary.py pmc $OUT/fetch; python scripts/rocprof_csv_summary.py pmc $OUT/write ) > $OUT/pmc_hbm_traffic_raw.txt 2>&1
$R --kernel-trace -d $OUT/tl1m -- python scripts/time_single.py 1000000 > $OUT/tl1m.log 2>&1
python scripts/rocprof_csv_summary.py timeline $OUT/tl1m fused_kernel 4 > $OUT/query_timeline_1M.txt 2>&1
BATCH="--no-cpu-baseline --fp-bits 2048 --batch-queries 256 --rows-per-gpu 125000000 --steps 3 --warmup 1"
$R --kernel-trace --stats -d $OUT/bstats -- python bench.py $BATCH > /dev/null 2> $OUT/bstats.err
python scripts/rocprof_csv_summary.py stats $OUT/bstats > $OUT/batch_kernel_trace_stats.txt 2>&1
$R --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/bpmc1 -- python bench.py $BATCH > /dev/null 2> $OUT/bpmc1.err
$R --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT -d $OUT/bpmc2 -- python bench.py $BATCH > /dev/null 2> $OUT/bpmc2.err
( python scripts/rocprof_csv_summary.py pmc $OUT/bpmc1; python scripts/rocprof_csv_summary.py pmc $OUT/bpmc2 ) > $OUT/batch_mfma_pmc_raw.txt 2>&1
rm -rf $OUT/stats $OUT/stats_all $OUT/fetch $OUT/write $OUT/tl1m $OUT/bstats $OUT/bpmc1 $OUT/bpmc2   # keep the text only
ls -la $OUT
