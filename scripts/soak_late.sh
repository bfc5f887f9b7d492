#!/usr/bin/env bash
# A longer soak of the rebuilt single launch (tags, eight selector waves, sampled election): table sizes either side of every
# geometry switch, Morgan-shaped rows, widths, large k, the two forced paths.  ~10 min on the GPU box.
set -uo pipefail
OUT=gpurun_out/soak_late
rm -rf "$OUT"; mkdir -p "$OUT"
( python scripts/soak_fused.py 5000 150000; python scripts/soak_fused.py 40000 300000; python scripts/soak_fused.py 130000 300000
  python scripts/soak_fused.py 600000 600000; python scripts/soak_fused.py 1000000 1500000; python scripts/soak_fused.py 2200000 400000
  python scripts/soak_fused.py 9000000 200000; python scripts/soak_fused.py 33000000 60000
  SOAK_KIND=morgan python scripts/soak_fused.py 300000 400000; SOAK_KIND=morgan python scripts/soak_fused.py 1000000 1000000; SOAK_KIND=morgan python scripts/soak_fused.py 10000000 100000
  SOAK_BITS=128 python scripts/soak_fused.py 8000000 100000; SOAK_BITS=256 python scripts/soak_fused.py 2000000 200000; SOAK_BITS=2048 python scripts/soak_fused.py 500000 300000
  SOAK_BITS=896 python scripts/soak_fused.py 1000000 200000; SOAK_BITS=160 python scripts/soak_fused.py 3000000 150000
  SOAK_LARGE_K=1 python scripts/soak_fused.py 3000000 40000; SOAK_LARGE_K=1 SOAK_KIND=morgan python scripts/soak_fused.py 1000000 40000
  echo "GSIM_FUSED_FLAGS=4096 (late tags):"; GSIM_FUSED_FLAGS=4096 python scripts/soak_fused.py 1000000 60000; GSIM_FUSED_FLAGS=4096 SOAK_KIND=morgan python scripts/soak_fused.py 300000 40000
  echo "GSIM_FUSED_FLAGS=8192 (every report ranked):"; GSIM_FUSED_FLAGS=8192 SOAK_KIND=morgan python scripts/soak_fused.py 1000000 200000 ) 2>&1 | grep -E "soak|handed back by|MISMATCH|GSIM_FUSED_FLAGS" > $OUT/soak.txt
cat $OUT/soak.txt
