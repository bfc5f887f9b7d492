// fingerprintdb_hip.cpp -- "Seam B" of INTEGRATION.md as a real translation unit: what a maintainer of
// schrodinger/gpusimilarity would drop in INSTEAD of fingerprintdb_cuda.cu to run the search on
// libgsim_hip.so.  It defines exactly the symbols that file defines (fingerprintdb_cuda.cu:33-413,
// declared in fingerprintdb_cuda.h) on top of the C ABI in include/gpusim_hip.h; fingerprintdb_cuda.cpp
// (search_cpu, fold_data, top_results_bubble_sort), gpusim.cpp and main.cpp stay as they are.
//
// This file is DOCUMENTATION THAT COMPILES: scripts/check_seam_b.sh syntax-checks it in the dev
// container against the reference's own header (/root/reference/fingerprintdb_cuda.h, types.h) and the
// image's Qt headers.  It is not built into this repository's products (they use the Qt-free twin,
// gpusimilarity_amd/csrc/host/fingerprintdb.{h,cpp}) and it never travels to the GPU box.
//
// Build in the reference tree, replacing the CUDA target of its CMakeLists.txt:
//     g++ -std=c++14 -fPIC -I<Qt>/include ... -c fingerprintdb_hip.cpp
//     ... link with -lgsim_hip instead of the CUDA runtime
#include "fingerprintdb_cuda.h" // the reference's header, unchanged

#include <climits>
#include <cstdint>
#include <stdexcept>
#include <string>

#include <QDebug>
#include <QString>

#include "gpusim_hip.h"

namespace gpusim
{

// The reference hides `std::shared_ptr<thrust::device_vector<int>>` behind this name
// (fingerprintdb_cuda.cu:106-109, "Used to conceal cuda types"); here it conceals the engine handle.
// One handle serves the whole FingerprintDB: it lives in the first storage's m_priv, the other storages
// keep their host copy (m_data) and their offsets only.
class FingerprintDBPriv
{
  public:
    ~FingerprintDBPriv()
    {
        if (db) gsim_db_destroy(db);
    }
    gsim_db* db = nullptr;
};

// (declared in the header for search_storage; the per-storage result object of :106-109 is not needed:
// the engine merges the shards itself)
struct StorageResultObject {
};

namespace
{
[[noreturn]] void throw_last(const char* what)
{
    throw std::runtime_error(std::string(what) + ": " + gsim_last_error());
}
} // namespace

size_t get_gpu_free_memory(unsigned int device_index) // fingerprintdb_cuda.cu:33-38
{
    size_t b = 0;
    gsim_device_free_bytes(static_cast<int>(device_index), &b);
    return b;
}

unsigned int get_gpu_count() // :40-52
{
    int n = 0;
    gsim_device_count(&n);
    return static_cast<unsigned int>(n);
}

unsigned int get_next_gpu(size_t required_memory) // :54-68
{
    int dev = 0;
    if (gsim_next_device(required_memory, &dev) != GSIM_OK)
        throw std::runtime_error("Can't find a GPU with enough memory to copy data.");
    return static_cast<unsigned int>(dev);
}

size_t get_available_gpu_memory() // :401-413
{
    size_t b = 0;
    gsim_available_device_bytes(&b);
    return b;
}

FingerprintDBStorage::FingerprintDBStorage(FingerprintDB* parent, std::vector<char>& fp_data, int index_offset,
                                           int fp_bitcount) // :111-126
    : m_parent(parent), m_index_offset(index_offset), m_count(static_cast<int>(fp_data.size() / (fp_bitcount / CHAR_BIT))),
      m_gpu_device(0)
{
    const int* int_data = reinterpret_cast<const int*>(fp_data.data());
    m_data.assign(int_data, int_data + fp_data.size() / sizeof(int)); // host copy: search_cpu / getFingerprint read it
    m_priv = std::make_shared<FingerprintDBPriv>();
}

unsigned int FingerprintDBStorage::getOffsetIndex(unsigned int without_offset) // :128-131
{
    return without_offset + m_index_offset;
}

FingerprintDB::FingerprintDB(int fp_bitcount, int fp_count, const QString& dbkey, std::vector<std::vector<char>>& data,
                             std::vector<char*>& smiles_vector, std::vector<char*>& ids_vector) // :133-166
{
    m_dbkey = dbkey;
    m_fp_intsize = fp_bitcount / (sizeof(int) * 8); // ASSUMES INT-DIVISIBLE SIZE (:140)
    m_total_count = fp_count;
    m_fold_factor = 1;
    int current_fp_count = 0;
    for (auto& dataset : data) {
        auto storage = std::make_shared<FingerprintDBStorage>(this, dataset, current_fp_count, fp_bitcount);
        m_storage.push_back(storage);
        current_fp_count += storage->m_count;
    }
    if (current_fp_count != m_total_count)
        throw std::runtime_error("Mismatch between FP count and data, potential database corruption."); // :153-156
    m_total_data_size = static_cast<size_t>(m_total_count) * static_cast<size_t>(m_fp_intsize) * sizeof(int);
    qDebug() << "Database loaded with" << m_total_count << "molecules";

    // the engine's table: one add_rows slice per storage, in order (global row = storage offset + local row,
    // the same numbering as getOffsetIndex)
    if (!m_storage.empty()) {
        gsim_db* db = nullptr;
        if (gsim_db_create(static_cast<uint32_t>(fp_bitcount), &db) != GSIM_OK) throw_last("FingerprintDB");
        m_storage[0]->m_priv->db = db;
        for (auto& s : m_storage) {
            if (gsim_db_add_rows(db, reinterpret_cast<const uint32_t*>(s->m_data.data()), static_cast<uint64_t>(s->m_count)) != GSIM_OK)
                throw_last("FingerprintDB");
        }
    }
    m_smiles.swap(smiles_vector); // :164-165
    m_ids.swap(ids_vector);
}

void FingerprintDB::copyToGPU(unsigned int fold_factor) // :168-195
{
    if (m_storage.empty()) return;
    gsim_db* db = m_storage[0]->m_priv->db;
    // the engine makes the factor divide the word count exactly as :170-173 does, and folds on upload
    if (fold_factor > 1 && gsim_db_set_fold_factor(db, fold_factor) != GSIM_OK) throw_last("copyToGPU");
    // unfolded: contiguous row shards over all GPUs; folded: one storage per GPU, round-robin (:176-188)
    if (gsim_db_finalize(db, /*device*/ -1, /*ndevices: 0 = all GPUs*/ 0) != GSIM_OK) throw_last("copyToGPU");
    m_fold_factor = static_cast<int>(gsim_db_fold_factor(db));
}

void FingerprintDB::getStorageAndLocalIndex(unsigned int offset_index, FingerprintDBStorage** storage,
                                            unsigned int* local_index) const // :197-210
{
    unsigned int slice_index_offset = 0;
    *storage = m_storage[0].get();
    for (unsigned int i = 1; i < m_storage.size(); i++) {
        if (m_storage[i]->m_index_offset >= offset_index) break;
        *storage = m_storage[i].get();
        slice_index_offset = m_storage[i]->m_index_offset;
    }
    *local_index = offset_index - slice_index_offset;
}

Fingerprint FingerprintDB::getFingerprint(unsigned int index) const // :212-226
{
    Fingerprint output(static_cast<size_t>(m_fp_intsize));
    if (m_storage.empty() || gsim_db_row(m_storage[0]->m_priv->db, index, reinterpret_cast<uint32_t*>(output.data())) != GSIM_OK)
        throw_last("getFingerprint");
    return output;
}

// :228-339 -- the Thrust pipeline of one storage.  Nothing calls it any more: FingerprintDB::search hands the
// whole query to the engine, which scans every shard, selects the exact top-k on the GPUs and merges.  Kept
// because the header declares it.
void FingerprintDB::search_storage(const Fingerprint&, const std::shared_ptr<FingerprintDBStorage>&, StorageResultObject*,
                                   unsigned int, float) const
{
    throw std::logic_error("search_storage: the per-storage pipeline lives inside libgsim_hip");
}

void FingerprintDB::search(const Fingerprint& query, const QString& dbkey, unsigned int max_return_count, float similarity_cutoff,
                           std::vector<char*>& results_smiles, std::vector<char*>& results_ids, std::vector<float>& results_scores,
                           unsigned long& approximate_result_count) const // :341-381
{
    if (dbkey != m_dbkey) {
        qDebug() << "Key check failed, returning empty results";
        return;
    }
    if (m_storage.empty()) return;
    if (max_return_count > count()) max_return_count = count();
    std::vector<gsim_hit> hits(max_return_count ? max_return_count : 1);
    uint32_t n = 0;
    uint64_t approx = 0;
    if (gsim_db_search(m_storage[0]->m_priv->db, reinterpret_cast<const uint32_t*>(query.data()), 1, max_return_count, similarity_cutoff,
                       GSIM_METRIC_TANIMOTO, 0.f, 0.f, hits.data(), &n, &approx) != GSIM_OK)
        throw_last("search");
    approximate_result_count = static_cast<unsigned long>(approx);
    for (uint32_t i = 0; i < n; i++) { // the ABI speaks row indices; the strings live up here (:297-304)
        results_scores.push_back(hits[i].score);
        results_smiles.push_back(m_smiles[hits[i].row]);
        results_ids.push_back(m_ids[hits[i].row]);
    }
}

float FingerprintDB::tanimoto_similarity_cpu(const Fingerprint& fp1, const Fingerprint& fp2) const // :387-399
{
    int total = 0, common = 0;
    for (int i = 0; i < m_fp_intsize; i++) {
        total += __builtin_popcount(fp1[i]) + __builtin_popcount(fp2[i]);
        common += __builtin_popcount(fp1[i] & fp2[i]);
    }
    return static_cast<float>(common) / static_cast<float>(total - common);
}

} // namespace gpusim
