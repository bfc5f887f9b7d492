#!/usr/bin/env bash
# A/B of two builds of the library on the same box: ab/libgsim_hip_base.so (GSIM_LIB) against the tree's.
set -uo pipefail
OUT=gpurun_out/s3
rm -rf $OUT; mkdir -p $OUT
(timeout 100 python scripts/time_single.py 1000000 2>&1 | tail -3) > $OUT/new_1M_first.txt; cat $OUT/new_1M_first.txt
(timeout 500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8) > $OUT/pytest_subset.txt; cat $OUT/pytest_subset.txt
for i in 1 2; do
  (GSIM_LIB=/root/repo/ab/libgsim_hip_base.so timeout 100 python scripts/time_single.py 100000 1000000 10000000 2>&1 | tail -3) >> $OUT/ab_base.txt
  (timeout 100 python scripts/time_single.py 100000 1000000 10000000 2>&1 | tail -3) >> $OUT/ab_new.txt
done
(TS_KIND=morgan timeout 100 python scripts/time_single.py 1000000 2>&1 | tail -1) >> $OUT/ab_new.txt
(TS_KIND=morgan GSIM_LIB=/root/repo/ab/libgsim_hip_base.so timeout 100 python scripts/time_single.py 1000000 2>&1 | tail -1) >> $OUT/ab_base.txt
GSIM_FUSED_DEBUG=1 timeout 100 python scripts/time_single.py 1000000 2>&1 | tail -30 > $OUT/fused_phases_1M.txt
echo BASE; cat $OUT/ab_base.txt; echo NEW; cat $OUT/ab_new.txt; head -22 $OUT/fused_phases_1M.txt
