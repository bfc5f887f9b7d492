#!/usr/bin/env bash
# GSIM_FUSED_FLAGS=8192 forces the single launch's rank-every-report election (the fallback behind the sampled one).  On the GPU box:
# the parity test with both forced paths, soaks of the forced path against the four-kernel pipeline, the hand-backs of a short
# tie-heavy table under the exact and the sampled election, then the whole GPU suite.
set -uo pipefail
OUT=gpurun_out/s9
rm -rf "$OUT"; mkdir -p "$OUT"
(timeout 400 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "entries_that_arrive" 2>&1 | tail -4) > $OUT/pytest_forced_paths.txt
( echo "GSIM_FUSED_FLAGS=8192 (every report ranked), against the four-kernel pipeline:"
  GSIM_FUSED_FLAGS=8192 python scripts/soak_fused.py 1000000 100000
  GSIM_FUSED_FLAGS=8192 SOAK_KIND=morgan python scripts/soak_fused.py 1000000 100000
  GSIM_FUSED_FLAGS=8192 python scripts/soak_fused.py 130000 60000
  echo "Morgan-shaped 300 k rows (k = 8192 is ranked inside the launch there): every report ranked ..."
  GSIM_FUSED_FLAGS=8192 SOAK_KIND=morgan python scripts/soak_fused.py 300000 40000
  echo "... and the sampled election (default):"
  SOAK_KIND=morgan python scripts/soak_fused.py 300000 40000 ) 2>&1 | grep -E "soak|handed back by|MISMATCH|GSIM_FUSED_FLAGS|Morgan-shaped|sampled" > $OUT/soak_ranked_election.txt
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $OUT/pytest_gpu.txt
cat $OUT/pytest_forced_paths.txt; cat $OUT/soak_ranked_election.txt; grep -E "passed|failed|error" $OUT/pytest_gpu.txt
