#!/usr/bin/env bash
# Quick A/B on one box: ab/libgsim_hip_base.so (GSIM_LIB) against the tree's build, 1 M-row single queries, three rounds.
set -uo pipefail
OUT=gpurun_out/s4
rm -rf $OUT; mkdir -p $OUT
for i in 1 2 3; do
  (GSIM_LIB=/root/repo/ab/libgsim_hip_base.so timeout 100 python scripts/time_single.py 100000 1000000 2>&1 | grep "^rows") >> $OUT/ab_base.txt
  (timeout 100 python scripts/time_single.py 100000 1000000 2>&1 | grep "^rows") >> $OUT/ab_new.txt
done
for i in 1 2; do (TS_KIND=morgan timeout 100 python scripts/time_single.py 1000000 10000000 2>&1 | grep "^rows") >> $OUT/ab_new.txt; done
for i in 1 2; do (TS_KIND=morgan GSIM_LIB=/root/repo/ab/libgsim_hip_base.so timeout 100 python scripts/time_single.py 1000000 10000000 2>&1 | grep "^rows") >> $OUT/ab_base.txt; done
GSIM_FUSED_DEBUG=1 TS_KIND=morgan timeout 100 python scripts/time_single.py 1000000 2>&1 | tail -30 > $OUT/fused_phases_1M.txt
(timeout 300 python -m pytest tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -3) > $OUT/pytest_fused.txt
echo BASE; cat $OUT/ab_base.txt; echo NEW; cat $OUT/ab_new.txt; head -22 $OUT/fused_phases_1M.txt; cat $OUT/pytest_fused.txt
