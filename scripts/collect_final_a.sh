#!/usr/bin/env bash
# Round-5 late evidence, part A (GPU box): the whole GPU suite, a soak of the single launch's new protocol, phase profiles.
set -uo pipefail
OUT=gpurun_out/final_a
rm -rf "$OUT"; mkdir -p "$OUT"
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $OUT/pytest_gpu.txt
( python scripts/soak_fused.py 70000 200000; python scripts/soak_fused.py 300000 200000; python scripts/soak_fused.py 1000000 600000
  python scripts/soak_fused.py 4000000 150000; SOAK_KIND=morgan python scripts/soak_fused.py 1000000 400000
  SOAK_BITS=128 python scripts/soak_fused.py 500000 150000; SOAK_BITS=512 python scripts/soak_fused.py 4000000 100000
  SOAK_LARGE_K=1 python scripts/soak_fused.py 3000000 20000
  echo "GSIM_FUSED_FLAGS=4096 (late tags):"; GSIM_FUSED_FLAGS=4096 python scripts/soak_fused.py 300000 30000
  echo "GSIM_FUSED_FLAGS=8192 (every report ranked):"; GSIM_FUSED_FLAGS=8192 python scripts/soak_fused.py 1000000 100000 ) 2>&1 | grep -E "soak|handed back by|MISMATCH|GSIM_FUSED_FLAGS" > $OUT/soak.txt
GSIM_FUSED_DEBUG=1 python scripts/time_single.py 1000000 2>&1 | tail -30 > $OUT/fused_phases_1M.txt
GSIM_FUSED_DEBUG=1 TS_REPS=20 python scripts/time_single.py 100000000 2>&1 | tail -30 > $OUT/fused_phases_100M.txt
( echo "== single queries, 1024-bit rows, Tanimoto top-1000 (scripts/time_single.py) =="
  python scripts/time_single.py 100000 1000000 10000000 30000000 100000000 2>&1 | grep "^rows"
  echo "== Morgan-shaped rows =="; TS_KIND=morgan python scripts/time_single.py 1000000 10000000 2>&1 | grep "^rows"
  echo "== k sweep, 1 M x 1024-bit =="
  for k in 1 100 1000 2048 4096 8192; do TS_K=$k python scripts/time_single.py 1000000 2>&1 | grep "^rows" | sed "s/^/k $k  /"; done
  echo "== other widths, 10 M rows =="
  for b in 128 256 512 2048; do TS_BITS=$b python scripts/time_single.py 10000000 2>&1 | grep "^rows" | sed "s/^/bits $b  /"; done ) > $OUT/sweeps.txt 2>&1
tail -6 $OUT/pytest_gpu.txt; cat $OUT/soak.txt; cat $OUT/sweeps.txt
