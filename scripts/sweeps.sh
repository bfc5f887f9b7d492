#!/usr/bin/env bash
# Run on the GPU box: latency / throughput sweeps behind DESIGN.md section 5 (text to stdout).
set -uo pipefail
echo "== single queries, 1024-bit rows, Tanimoto top-1000 (scripts/time_single.py) =="
python scripts/time_single.py 100000 1000000 10000000 30000000 100000000 2>&1 | grep rows
echo "== Morgan-shaped rows (GSIM_SYNTH_MORGAN), 1024-bit =="
TS_KIND=morgan python scripts/time_single.py 1000000 10000000 100000000 2>&1 | grep rows
echo "== other widths, 100 M rows (2048-bit: 60 M) =="
for b in 128 256 512; do TS_BITS=$b TS_REPS=50 python scripts/time_single.py 100000000 2>&1 | grep rows | sed "s/^/bits $b  /"; done
TS_BITS=2048 TS_REPS=50 python scripts/time_single.py 60000000 2>&1 | grep rows | sed "s/^/bits 2048  /"
TS_BITS=4096 TS_REPS=50 python scripts/time_single.py 30000000 2>&1 | grep rows | sed "s/^/bits 4096  /"
echo "== widths off the power-of-two template (word-streamed 160 / 192-bit, register-streamed 896 / 1536-bit: inside the single launch) =="
for b in 160 192 896 1536; do TS_BITS=$b TS_REPS=30 python scripts/time_single.py 50000000 2>&1 | grep rows | sed "s/^/bits $b  /"; done
echo "== four-kernel pipeline for comparison (GSIM_FUSED=0), 1024-bit =="
GSIM_FUSED=0 TS_REPS=50 python scripts/time_single.py 1000000 10000000 100000000 2>&1 | grep rows
echo "== k sweep, 100 M x 1024-bit =="
for k in 1 10 100 1000 2048 4096 8192 20000; do TS_K=$k TS_REPS=30 python scripts/time_single.py 100000000 2>&1 | grep rows | sed "s/^/k $k  /"; done
echo "== large k, 100 M x 1024-bit: the single launch scans and publishes (default) / the four-kernel pipeline's scan (GSIM_FUSED_PUBLISH=0) =="
for k in 8193 20000 32768 50000; do for p in 1 0; do GSIM_FUSED_PUBLISH=$p TS_K=$k TS_REPS=30 python scripts/time_single.py 100000000 2>&1 | grep rows | sed "s/^/k $k publish $p  /"; done; done
for n in 1000000 10000000; do for p in 1 0; do GSIM_FUSED_PUBLISH=$p TS_K=10000 TS_REPS=50 python scripts/time_single.py $n 2>&1 | grep rows | sed "s/^/k 10000 publish $p  /"; done; done
echo "== k sweep, 1 M x 1024-bit (Morgan-shaped: second block) =="
for k in 1 100 1000 2048 4096 8192; do TS_K=$k python scripts/time_single.py 1000000 2>&1 | grep rows | sed "s/^/k $k  /"; done
for k in 1000 4096 8192; do TS_KIND=morgan TS_K=$k python scripts/time_single.py 1000000 2>&1 | grep rows | sed "s/^/morgan k $k  /"; done
echo "== 256-query Tversky(0.3,0.7) batches on the matrix cores (scripts/time_batch.py) =="
python scripts/time_batch.py 125000000
TB_BITS=1024 python scripts/time_batch.py 125000000
TB_BITS=512 python scripts/time_batch.py 200000000
TB_BITS=256 python scripts/time_batch.py 200000000
TB_Q=64 python scripts/time_batch.py 125000000
TB_Q=32 python scripts/time_batch.py 125000000
echo "== the same batches with a cutoff: selective (exact path counts), dense (counted from the accumulators), and the VALU route for comparison =="
python scripts/time_batch_cutoff.py 125000000 0.3 0.15 0.1 0.05
TB_BITS=1024 python scripts/time_batch_cutoff.py 125000000 0.1
GSIM_BATCH_MFMA_DENSE=0 python scripts/time_batch_cutoff.py 125000000 0.1
