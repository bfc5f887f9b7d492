#!/usr/bin/env bash
# Round-5 late evidence, part B (GPU box): the default bench line and the rocprofv3 summaries behind it, on the new single launch.
set -uo pipefail
export TMPDIR=/tmp
OUT=gpurun_out/final_b
rm -rf "$OUT"; mkdir -p "$OUT"
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
R="rocprofv3 --output-format csv"
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
rm -rf $OUT/stats $OUT/stats_all $OUT/fetch $OUT/write $OUT/tl1m
python scripts/server_latency.py > $OUT/server_latency.json 2> $OUT/server_latency.err || true
ls -la $OUT; tail -c 2500 $OUT/bench_n1.json; head -12 $OUT/kernel_trace_stats.txt
