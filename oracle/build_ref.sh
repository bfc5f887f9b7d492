#!/usr/bin/env bash
# Compile the REFERENCE's own CPU functor sources, in place, into oracle/_ref/.
# Outputs only under oracle/_ref/ (git-ignored; travels to the GPU box with the
# repo snapshot).  No reference source is copied; no stand-in headers are
# written: the only external need is the Qt5 *headers* that
# calculation_functors.h includes (<QVector>), which this image ships under
# /opt/conda/include/qt.  Nothing from Qt is linked into libgsim_ref.so.
#
#   libgsim_ref.so       calculation_functors.cpp (TanimotoFunctorCPU, Fold...)
#                        + oracle/ref_shim.cpp (our extern "C" driver)
#   libgsim_ref_sort.so  fingerprintdb_cuda.cpp (top_results_bubble_sort), needs
#                        libQt5Core/libQt5Concurrent at load time -> used only by
#                        the in-container tests, optional.
#
# What is NOT built: fingerprintdb_cuda.cu / gpusim.cpp (need CUDA + Thrust +
# a Qt event loop) -- unbuildable here without stand-ins, see DESIGN.md.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${GSIM_REFERENCE_DIR:-/root/reference}"
QI="${GSIM_QT_INCLUDE:-/opt/conda/include/qt}"
QL="${GSIM_QT_LIB:-/opt/conda/lib}"
OUT="$HERE/_ref"
if [ ! -f "$REF/calculation_functors.cpp" ]; then
    echo "build_ref: $REF not present; keeping any prebuilt $OUT" >&2
    exit 0
fi
if [ ! -d "$QI/QtCore" ]; then
    echo "build_ref: Qt headers not found under $QI; skipping" >&2
    exit 0
fi
mkdir -p "$OUT"
FL="-O2 -std=c++14 -fPIC -mpopcnt -DQT_NO_VERSION_TAGGING -I$REF -I$QI -I$QI/QtCore"
g++ $FL -c "$REF/calculation_functors.cpp" -o "$OUT/calculation_functors.o"
g++ $FL -c "$HERE/ref_shim.cpp" -o "$OUT/ref_shim.o"
g++ -shared -o "$OUT/libgsim_ref.so" "$OUT/calculation_functors.o" "$OUT/ref_shim.o" -lpthread
echo "built $OUT/libgsim_ref.so"
# optional: the reference's bubble sort (links real Qt)
if g++ $FL -I$QI/QtConcurrent -c "$REF/fingerprintdb_cuda.cpp" -o "$OUT/fingerprintdb_cpu.o" 2>"$OUT/sort_build.log" \
   && g++ -O2 -std=c++14 -fPIC -c "$HERE/ref_shim_sort.cpp" -o "$OUT/ref_shim_sort.o" 2>>"$OUT/sort_build.log" \
   && g++ -shared -o "$OUT/libgsim_ref_sort.so" "$OUT/fingerprintdb_cpu.o" "$OUT/calculation_functors.o" "$OUT/ref_shim_sort.o" \
        "$QL/libQt5Core.so.5" "$QL/libQt5Concurrent.so.5" -Wl,-rpath,/usr/lib/x86_64-linux-gnu:"$QL" \
        -Wl,-rpath-link,"$QL" 2>>"$OUT/sort_build.log"; then
    echo "built $OUT/libgsim_ref_sort.so"
else
    echo "build_ref: libgsim_ref_sort.so not built (see $OUT/sort_build.log)" >&2
fi
