cd /root/repo
mkdir -p gpurun_out/r05a
O=gpurun_out/r05a/sweep20.txt
rm -f $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "matrix_core" 2>&1 | tail -3 >> $O
for i in 1 2; do
echo "# ref bits=1024 Q=256" >> $O
GSIM_LIB=scripts/build/libgsim_hip_ref.so BB_BITS=1024 BB_ROWS=200000000 BB_Q=256 timeout 300 python scripts/bench_batch.py 2>&1 | tail -1 | cut -c1-200 >> $O
echo "# new bits=1024 Q=256" >> $O
BB_BITS=1024 BB_ROWS=200000000 BB_Q=256 timeout 300 python scripts/bench_batch.py 2>&1 | tail -1 | cut -c1-200 >> $O
echo "# ref Q=256 cutoff 0.1" >> $O
GSIM_LIB=scripts/build/libgsim_hip_ref.so BB_CUTOFF=0.1 BB_Q=256 timeout 300 python scripts/bench_batch.py 2>&1 | tail -1 | cut -c1-200 >> $O
echo "# new Q=256 cutoff 0.1" >> $O
BB_CUTOFF=0.1 BB_Q=256 timeout 300 python scripts/bench_batch.py 2>&1 | tail -1 | cut -c1-200 >> $O
done
