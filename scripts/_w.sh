for n in 150000 300000 500000; do SOAK_BITS=128 timeout 600 python scripts/soak_fused.py $n 60000 2>&1 | tail -2; done
SOAK_BITS=256 timeout 600 python scripts/soak_fused.py 400000 60000 2>&1 | tail -2
SOAK_BITS=160 timeout 600 python scripts/soak_fused.py 400000 60000 2>&1 | tail -2
for n in 300000 500000; do TS_BITS=128 TS_REPS=100 timeout 300 python scripts/time_single.py $n 2>&1 | grep rows; done
