#!/bin/bash
# Side builds of libgsim_hip.so for A/B runs of the matrix-core batch pass on one box (select with GSIM_LIB=...), into scripts/build/:
#   libgsim_hip_timing.so      the current kernel with -DGSIM_MF_TIMING=1 (per-phase cycle counters, GSIM_DEBUG_BATCH=1 prints them)
#   libgsim_hip_ref.so         the kernel file of a git revision (default HEAD) in place of the working tree's
#   libgsim_hip_whatifN.so     timing-only builds with -DGSIM_MF_WHATIF=N (WRONG results; N from the arguments after the revision)
set -e
cd "$(dirname "$0")/../gpusimilarity_amd/csrc"
REV=${1:-HEAD}
shift || true
mkdir -p build_t ../../scripts/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I. "
OBJS=$(ls build/*.o | grep -v "gsim_batch_mfma.o\|_hooks.o")
LINK="-L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib"
/opt/rocm/bin/hipcc $FLAGS -DGSIM_MF_TIMING=1 -c -o build_t/gsim_batch_mfma_timing.o gsim_batch_mfma.hip &
git show $REV:gpusimilarity_amd/csrc/gsim_batch_mfma.hip > build_t/gsim_batch_mfma_ref.hip
/opt/rocm/bin/hipcc $FLAGS -c -o build_t/gsim_batch_mfma_ref.o build_t/gsim_batch_mfma_ref.hip &
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DGSIM_MF_WHATIF=$n -c -o build_t/gsim_batch_mfma_whatif$n.o gsim_batch_mfma.hip &
done
wait
for v in timing ref $(for n in "$@"; do echo whatif$n; done); do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../scripts/build/libgsim_hip_$v.so $OBJS build_t/gsim_batch_mfma_$v.o $LINK
done
ls -la ../../scripts/build/*.so
