#!/usr/bin/env bash
# Run on the GPU box after collect_profiles.sh / sweeps.sh: phase profiles, soak, the two-rank test mode, the smaller reports.
set -uo pipefail
OUT=gpurun_out/rest_r05
rm -rf "$OUT"; mkdir -p "$OUT"
GSIM_FUSED_DEBUG=1 python scripts/time_single.py 1000000 2>&1 | tail -30 > $OUT/fused_phases_1M.txt
GSIM_FUSED_DEBUG=1 TS_REPS=20 python scripts/time_single.py 100000000 2>&1 | tail -30 > $OUT/fused_phases_100M.txt
TS_K=8192 GSIM_FUSED_DEBUG=1 python scripts/time_single.py 1000000 2>&1 | tail -30 > $OUT/fused_phases_1M_k8192.txt
( python scripts/soak_fused.py 70000 300000; python scripts/soak_fused.py 300000 300000; python scripts/soak_fused.py 1000000 1000000
  python scripts/soak_fused.py 4000000 300000; python scripts/soak_fused.py 33000000 100000
  SOAK_KIND=morgan python scripts/soak_fused.py 1000000 800000; SOAK_KIND=morgan python scripts/soak_fused.py 10000000 150000
  SOAK_BITS=128 python scripts/soak_fused.py 8000000 150000; SOAK_BITS=256 python scripts/soak_fused.py 8000000 150000; SOAK_BITS=512 python scripts/soak_fused.py 4000000 150000
  SOAK_BITS=128 python scripts/soak_fused.py 500000 300000
  SOAK_LARGE_K=1 python scripts/soak_fused.py 3000000 40000; SOAK_LARGE_K=1 SOAK_BITS=512 python scripts/soak_fused.py 6000000 20000
  SOAK_LARGE_K=1 SOAK_KIND=morgan python scripts/soak_fused.py 30000000 6000 ) 2>&1 | grep -E "soak|handed back by|MISMATCH" > $OUT/soak.txt
GSIM_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --no-cpu-baseline --steps 4 --warmup 2 > $OUT/bench_n2_two_ranks_sharing_one_gpu.json 2> $OUT/bench_n2.err
python scripts/merge_cost.py > $OUT/merge_cost.json 2> $OUT/merge_cost.err
python scripts/time_folded.py > $OUT/folded_search.txt 2>&1
python scripts/time_device_block.py 1000000 1000 8192 > $OUT/device_block.txt 2>&1
python scripts/soak_batch.py > $OUT/soak_batch.txt 2>&1
ls -la $OUT
