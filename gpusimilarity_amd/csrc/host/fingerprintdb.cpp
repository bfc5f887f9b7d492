// fingerprintdb.cpp -- gpusim::FingerprintDB over the C ABI.  See fingerprintdb.h.
#include "fingerprintdb.h"

#include <climits>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <map>
#include <mutex>
#include <thread>

#include "../../../include/gpusim_hip.h"

namespace gpusim
{

namespace
{
[[noreturn]] void throw_last(const char* what)
{
    throw std::runtime_error(std::string(what) + ": " + gsim_last_error());
}
} // namespace

unsigned int get_gpu_count()
{
    int n = 0;
    gsim_device_count(&n);
    return static_cast<unsigned int>(n);
}

unsigned int get_next_gpu(size_t required_memory)
{
    int dev = 0;
    if (gsim_next_device(required_memory, &dev) != GSIM_OK) {
        // reference message, fingerprintdb_cuda.cu:65-66
        throw std::runtime_error("Can't find a GPU with enough memory to copy data.");
    }
    return static_cast<unsigned int>(dev);
}

size_t get_available_gpu_memory()
{
    size_t b = 0;
    gsim_available_device_bytes(&b);
    return b;
}

size_t get_gpu_free_memory(unsigned int device)
{
    size_t b = 0;
    gsim_device_free_bytes(static_cast<int>(device), &b);
    return b;
}

FingerprintDB::FingerprintDB(int fp_bitcount, int fp_count, const std::string& dbkey,
                             std::vector<std::vector<char>>& data, std::vector<char*>& smiles_vector,
                             std::vector<char*>& ids_vector)
    : m_dbkey(dbkey)
{
    m_fp_intsize = fp_bitcount / static_cast<int>(sizeof(int) * 8);
    m_total_count = fp_count;
    if (gsim_db_create(static_cast<uint32_t>(fp_bitcount), &m_db) != GSIM_OK) throw_last("FingerprintDB");
    const size_t row_bytes = static_cast<size_t>(fp_bitcount / CHAR_BIT);
    long current_fp_count = 0;
    for (auto& dataset : data) {
        const uint64_t rows = row_bytes ? dataset.size() / row_bytes : 0;
        if (gsim_db_add_rows(m_db, reinterpret_cast<const uint32_t*>(dataset.data()), rows) != GSIM_OK) {
            gsim_db_destroy(m_db);
            m_db = nullptr;
            throw_last("FingerprintDB");
        }
        current_fp_count += static_cast<long>(rows);
    }
    if (current_fp_count != m_total_count) {
        gsim_db_destroy(m_db);
        m_db = nullptr;
        throw std::runtime_error("Mismatch between FP count and data, potential database corruption.");
    }
    m_total_data_size = static_cast<size_t>(m_total_count) * static_cast<size_t>(m_fp_intsize) * sizeof(int);
    std::fprintf(stderr, "Database loaded with %d molecules\n", m_total_count);
    m_smiles.swap(smiles_vector);
    m_ids.swap(ids_vector);
}

FingerprintDB::FingerprintDB(int fp_bitcount, unsigned long long fp_count, const std::string& dbkey, unsigned long long seed, int kind)
    : m_dbkey(dbkey), m_synthetic(true), m_seed(seed), m_kind(kind)
{
    if (fp_count > 0x7FFFFFFFull) throw std::runtime_error("synthetic table: too many rows for one device");
    m_fp_intsize = fp_bitcount / static_cast<int>(sizeof(int) * 8);
    m_total_count = static_cast<int>(fp_count);
    if (gsim_db_create(static_cast<uint32_t>(fp_bitcount), &m_db) != GSIM_OK) throw_last("FingerprintDB");
    m_total_data_size = static_cast<size_t>(m_total_count) * static_cast<size_t>(m_fp_intsize) * sizeof(int);
    // "S%010d\0" and "ZINC%010d\0": 12 + 15 bytes per row, one arena
    const size_t per = 12 + 15;
    m_arena.resize(static_cast<size_t>(m_total_count) * per);
    m_smiles.resize(static_cast<size_t>(m_total_count));
    m_ids.resize(static_cast<size_t>(m_total_count));
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 32) nt = 32;
    std::vector<std::thread> pool;
    const size_t n = static_cast<size_t>(m_total_count), chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        const size_t lo = chunk * t, hi = lo + chunk < n ? lo + chunk : n;
        if (lo >= hi) break;
        pool.emplace_back([this, lo, hi, per] {
            for (size_t r = lo; r < hi; r++) {
                char* sm = m_arena.data() + r * per;
                char* id = sm + 12;
                std::snprintf(sm, 12, "S%010zu", r);
                std::snprintf(id, 15, "ZINC%010zu", r);
                m_smiles[r] = sm;
                m_ids[r] = id;
            }
        });
    }
    for (auto& th : pool) th.join();
    std::fprintf(stderr, "Synthetic database of %d molecules (kind %d, %d bits)\n", m_total_count, kind, fp_bitcount);
}

namespace
{
// One communicator per distinct device list, shared by every database that spans those devices (RCCL keeps device and host
// buffers per communicator, and ncclCommInitAll takes seconds on a full node; the server answers one request at a time, so
// the databases never use it concurrently) -- ADVICE r04.
struct SharedComm {
    gsim_comm* comm = nullptr;
    int users = 0;
};
std::mutex g_comm_mutex;
std::map<std::vector<int>, SharedComm> g_comms;

gsim_comm* acquire_comm(const std::vector<int>& devs)
{
    std::lock_guard<std::mutex> guard(g_comm_mutex);
    SharedComm& sc = g_comms[devs];
    if (!sc.comm && gsim_comm_create(devs.data(), static_cast<int>(devs.size()), &sc.comm) != GSIM_OK) {
        g_comms.erase(devs);
        return nullptr;
    }
    sc.users++;
    return sc.comm;
}

void release_comm(gsim_comm* comm)
{
    std::lock_guard<std::mutex> guard(g_comm_mutex);
    for (auto it = g_comms.begin(); it != g_comms.end(); ++it) {
        if (it->second.comm != comm) continue;
        if (--it->second.users == 0) {
            gsim_comm_destroy(comm);
            g_comms.erase(it);
        }
        return;
    }
}
} // namespace

FingerprintDB::~FingerprintDB()
{
    if (m_db && m_comm) gsim_db_set_comm(m_db, nullptr);
    if (m_db) gsim_db_destroy(m_db);
    if (m_comm) release_comm(m_comm);
}

void FingerprintDB::copyToGPU(unsigned int fold_factor, int ndevices, bool full_on_device, bool rccl_merge)
{
    if (m_synthetic) { // generated in HBM: on the device get_next_gpu picks, or split over the first `ndevices` like an uploaded table
        (void) fold_factor;
        if (ndevices > 1) {
            if (gsim_db_generate_sharded(m_db, m_seed, m_kind, 0, static_cast<uint64_t>(m_total_count), 0, ndevices) != GSIM_OK)
                throw_last("copyToGPU (synthetic, sharded)");
        } else {
            const unsigned int dev = get_next_gpu(m_total_data_size);
            if (gsim_db_generate(m_db, m_seed, m_kind, 0, static_cast<uint64_t>(m_total_count), static_cast<int>(dev)) != GSIM_OK)
                throw_last("copyToGPU (synthetic)");
        }
        m_fold_factor = 1;
    } else {
        if (fold_factor > 1 && gsim_db_set_fold_factor(m_db, fold_factor) != GSIM_OK) throw_last("copyToGPU");
        if (fold_factor > 1 && gsim_db_set_fold_full_on_device(m_db, full_on_device ? 1 : 0) != GSIM_OK) throw_last("copyToGPU");
        if (gsim_db_finalize(m_db, ndevices == 1 ? -1 : 0, ndevices) != GSIM_OK) throw_last("copyToGPU");
        m_fold_factor = static_cast<int>(gsim_db_fold_factor(m_db));
    }
    // the collective only where there is something to gather: a single-shard table answers from its own block
    if (rccl_merge && m_fold_factor <= 1 && gsim_db_shard_count(m_db) > 1) {
        std::vector<int> devs;
        for (int i = 0; i < gsim_db_shard_count(m_db); i++) devs.push_back(gsim_db_shard_device(m_db, i));
        m_comm = acquire_comm(devs);
        if (!m_comm) throw_last("copyToGPU (gsim_comm_create)");
        if (gsim_db_set_comm(m_db, m_comm) != GSIM_OK) throw_last("copyToGPU (gsim_db_set_comm)");
    }
    m_on_gpu = true;
}

Fingerprint FingerprintDB::getFingerprint(unsigned int index) const
{
    Fingerprint output(static_cast<size_t>(m_fp_intsize));
    if (gsim_db_row(m_db, index, reinterpret_cast<uint32_t*>(output.data())) != GSIM_OK) throw_last("getFingerprint");
    return output;
}

void FingerprintDB::search(const Fingerprint& query, const std::string& dbkey, unsigned int max_return_count,
                           float similarity_cutoff, std::vector<char*>& results_smiles,
                           std::vector<char*>& results_ids, std::vector<float>& results_scores,
                           unsigned long& approximate_result_count) const
{
    if (dbkey != m_dbkey) {
        std::fprintf(stderr, "Key check failed, returning empty results\n");
        return;
    }
    if (static_cast<int>(query.size()) != m_fp_intsize) throw std::invalid_argument("query has the wrong width");
    // no more hits than rows (the reference bounds its work by the row count, :282-287): a huge count from the
    // socket must not size the buffers
    if (max_return_count > count()) max_return_count = count();
    std::vector<gsim_hit> hits(max_return_count ? max_return_count : 1);
    uint32_t count = 0;
    uint64_t approx = 0;
    if (gsim_db_search(m_db, reinterpret_cast<const uint32_t*>(query.data()), 1, max_return_count, similarity_cutoff,
                       GSIM_METRIC_TANIMOTO, 0.f, 0.f, hits.data(), &count, &approx) != GSIM_OK)
        throw_last("search");
    approximate_result_count = static_cast<unsigned long>(approx);
    for (uint32_t i = 0; i < count; i++) {
        results_scores.push_back(hits[i].score);
        results_smiles.push_back(m_smiles[hits[i].row]);
        results_ids.push_back(m_ids[hits[i].row]);
    }
}

void FingerprintDB::search_cpu(const Fingerprint& query, const std::string& dbkey, unsigned int max_return_count,
                               float similarity_cutoff, std::vector<char*>& results_smiles,
                               std::vector<char*>& results_ids, std::vector<float>& results_scores,
                               unsigned long& approximate_result_count) const
{
    (void) approximate_result_count; // fingerprintdb_cuda.cpp:39 "not giving approximate total count back"
    if (dbkey != m_dbkey) {
        std::fprintf(stderr, "Key check failed, returning empty results\n");
        return;
    }
    if (static_cast<int>(query.size()) != m_fp_intsize) throw std::invalid_argument("query has the wrong width");
    // the reference indexes indices[i] for i < max_return_count even past the table (UB); clamp instead
    const unsigned int k = max_return_count < count() ? max_return_count : count();
    std::vector<gsim_hit> hits(k ? k : 1);
    uint32_t n = 0;
    if (gsim_db_search_cpu(m_db, reinterpret_cast<const uint32_t*>(query.data()), 1, k, similarity_cutoff, hits.data(),
                           &n) != GSIM_OK)
        throw_last("search_cpu");
    for (uint32_t i = 0; i < n; i++) {
        results_smiles.push_back(m_smiles[hits[i].row]);
        results_ids.push_back(m_ids[hits[i].row]);
        results_scores.push_back(hits[i].score);
    }
}

void top_results_bubble_sort(std::vector<int>& indices, std::vector<float>& scores, int number_required)
{
    const int count = static_cast<int>(indices.size());
    for (int i = 0; i < number_required; i++) {
        for (int j = count - 1; j > i; j--) {
            if (scores[j] > scores[j - 1]) {
                std::swap(indices[j], indices[j - 1]);
                std::swap(scores[j], scores[j - 1]);
            }
        }
    }
}

} // namespace gpusim
