#!/usr/bin/env bash
# Syntax-check docs/fingerprintdb_hip.cpp (the Seam-B adapter of INTEGRATION.md) against the REFERENCE's own
# fingerprintdb_cuda.h / types.h where they lie, the image's Qt headers and include/gpusim_hip.h -- the flags
# of oracle/build_ref.sh.  Dev container only (the GPU box has no /root/reference); nothing is copied.
#
#   check_seam_b.sh                  compile + symbol check only
#   check_seam_b.sh --link <dir>     ... and LINK the reference's own server around the adapter: the reference's main.cpp, gpusim.cpp,
#                                    fingerprintdb_cuda.cpp and calculation_functors.cpp compiled where they lie (plain g++, the image's
#                                    real Qt 5.9 headers and libraries; `-include QElapsedTimer` for gpusim.cpp: Qt 5.9 lacks the
#                                    transitive include its newer Qt had) + docs/fingerprintdb_hip.cpp + -lgsim_hip instead of
#                                    fingerprintdb_cuda.cu (reference CMakeLists.txt:60-67) -> <dir>/refserver_hip, a SCRATCH binary:
#                                    never committed, never shipped, never sent to the GPU box (tests/test_host_cpp.py replays the
#                                    golden protocol frames against it with --cpu_only and removes it)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
REF="${GSIM_REFERENCE_DIR:-/root/reference}"
QI="${GSIM_QT_INCLUDE:-/opt/conda/include/qt}"
if [ ! -f "$REF/fingerprintdb_cuda.h" ] || [ ! -d "$QI/QtCore" ]; then
    echo "check_seam_b: reference or Qt headers not present; nothing checked" >&2
    exit 77
fi
g++ -std=c++14 -fPIC -fsyntax-only -Wall -Wextra -Werror -DQT_NO_VERSION_TAGGING \
    -I"$REF" -isystem "$QI" -isystem "$QI/QtCore" -I"$HERE/include" "$HERE/docs/fingerprintdb_hip.cpp"
# ... and every symbol fingerprintdb_cuda.cu defines is defined by the adapter (compile to an object, compare)
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
g++ -std=c++14 -fPIC -DQT_NO_VERSION_TAGGING -I"$REF" -isystem "$QI" -isystem "$QI/QtCore" -I"$HERE/include" -c "$HERE/docs/fingerprintdb_hip.cpp" -o "$TMP/a.o"
nm -C --defined-only "$TMP/a.o" > "$TMP/syms.txt"
for sym in get_gpu_free_memory get_gpu_count get_next_gpu get_available_gpu_memory \
           'FingerprintDBStorage::FingerprintDBStorage' 'FingerprintDBStorage::getOffsetIndex' \
           'FingerprintDB::FingerprintDB' 'FingerprintDB::copyToGPU' 'FingerprintDB::getStorageAndLocalIndex' \
           'FingerprintDB::getFingerprint' 'FingerprintDB::search_storage' 'FingerprintDB::search(' \
           'FingerprintDB::tanimoto_similarity_cpu'; do
    grep -qF "gpusim::$sym" "$TMP/syms.txt" || { echo "check_seam_b: $sym not defined" >&2; exit 1; }
done
echo "check_seam_b: docs/fingerprintdb_hip.cpp compiles against $REF/fingerprintdb_cuda.h and defines every symbol of fingerprintdb_cuda.cu"
if [ "${1:-}" = "--link" ]; then
    OUT="${2:?check_seam_b.sh --link <dir>}"
    QL="${GSIM_QT_LIB:-/opt/conda/lib}"
    LIB="$HERE/gpusimilarity_amd"
    [ -f "$LIB/libgsim_hip.so" ] || { echo "check_seam_b: build libgsim_hip.so first (make -C gpusimilarity_amd/csrc)" >&2; exit 1; }
    FL="-std=c++14 -fPIC -O1 -DQT_NO_VERSION_TAGGING -I$REF -isystem $QI -isystem $QI/QtCore -isystem $QI/QtNetwork -isystem $QI/QtConcurrent -I$HERE/include"
    for f in calculation_functors fingerprintdb_cuda main; do g++ $FL -c "$REF/$f.cpp" -o "$TMP/$f.o"; done
    g++ $FL -include QElapsedTimer -c "$REF/gpusim.cpp" -o "$TMP/gpusim.o"
    mkdir -p "$OUT"
    # (not -L$QL: its old libstdc++ would shadow the system's)
    g++ -o "$OUT/refserver_hip" "$TMP/main.o" "$TMP/gpusim.o" "$TMP/fingerprintdb_cuda.o" "$TMP/calculation_functors.o" "$TMP/a.o" \
        -L"$LIB" -lgsim_hip "$QL/libQt5Core.so.5" "$QL/libQt5Network.so.5" "$QL/libQt5Concurrent.so.5" \
        -Wl,-rpath,"$LIB":/usr/lib/x86_64-linux-gnu:"$QL":/opt/rocm/lib -Wl,-rpath-link,"$QL":/opt/rocm/lib
    echo "check_seam_b: linked $OUT/refserver_hip (the reference's server on libgsim_hip.so)"
fi
