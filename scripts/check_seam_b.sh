#!/usr/bin/env bash
# Syntax-check docs/fingerprintdb_hip.cpp (the Seam-B adapter of INTEGRATION.md) against the REFERENCE's own
# fingerprintdb_cuda.h / types.h where they lie, the image's Qt headers and include/gpusim_hip.h -- the flags
# of oracle/build_ref.sh.  Dev container only (the GPU box has no /root/reference); nothing is built or copied.
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
