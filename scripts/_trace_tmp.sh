O=gpurun_out/r06_suite; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest_gpu.txt
python scripts/fuzz_large_k.py 0 200 > $O/fuzz_large_k.txt 2>&1
( SOAK_KIND=morgan python scripts/soak_fused.py 300000 100000
  python scripts/soak_fused.py 100000 100000
  SOAK_LARGE_K=1 python scripts/soak_fused.py 300000 40000
  SOAK_LARGE_K=1 SOAK_KIND=morgan python scripts/soak_fused.py 150000 40000
  SOAK_LARGE_K=1 python scripts/soak_fused.py 3000000 40000 ) 2>&1 | grep -E "soak|handed back by|MISMATCH" > $O/soak.txt
tail -30 $O/pytest_gpu.txt; tail -5 $O/fuzz_large_k.txt; cat $O/soak.txt
