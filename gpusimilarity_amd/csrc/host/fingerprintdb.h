// fingerprintdb.h -- Qt-free C++ twin of the reference's gpusim::FingerprintDB
// (fingerprintdb_cuda.h:53-140), implemented over the C ABI (include/gpusim_hip.h).
//
// Same method names, argument meaning and error behaviour as the reference class;
// QString becomes std::string.  This is the class the reference's GPUSimServer
// (gpusim.cpp) talks to; host C++ reaches the GPU only through the C ABI.
#pragma once

#include <cstddef>
#include <string>
#include <utility>
#include <vector>

struct gsim_db;
struct gsim_comm;

namespace gpusim
{

typedef std::vector<int> Fingerprint; // types.h:11
typedef std::pair<char*, char*> ResultData;          // fingerprintdb_cuda.h:29
typedef std::pair<float, ResultData> SortableResult; // fingerprintdb_cuda.h:30

unsigned int get_gpu_count();                    // fingerprintdb_cuda.cu:40-52
unsigned int get_next_gpu(size_t required_memory); // fingerprintdb_cuda.cu:54-68 (throws std::runtime_error)
size_t get_available_gpu_memory();               // fingerprintdb_cuda.cu:401-413
size_t get_gpu_free_memory(unsigned int device); // fingerprintdb_cuda.cu:33-38

class FingerprintDB
{
  public:
    // fingerprintdb_cuda.cu:133-166.  One element of `data` per storage block; throws
    // std::runtime_error when fp_count does not match the data (:153-156).  Steals
    // the smiles / ids vectors (:164-165).
    FingerprintDB(int fp_bitcount, int fp_count, const std::string& dbkey, std::vector<std::vector<char>>& data,
                  std::vector<char*>& smiles_vector, std::vector<char*>& ids_vector);
    // A synthetic table for benchmarks (no reference counterpart): fp_count rows of the counter-based generator
    // (gsim_db_generate: GSIM_SYNTH_SPARSE / _DENSE / _MORGAN) made directly in HBM by copyToGPU; SMILES / ID strings are
    // "S<row>" / "ZINC<row>".  No host copy of the fingerprints: search_cpu is not available.
    FingerprintDB(int fp_bitcount, unsigned long long fp_count, const std::string& dbkey, unsigned long long seed, int kind);
    ~FingerprintDB();
    FingerprintDB(const FingerprintDB&) = delete;
    FingerprintDB& operator=(const FingerprintDB&) = delete;

    // fingerprintdb_cuda.cu:168-195.  fold_factor > 1: the GPU holds an OR-folded copy and
    // search() is the reference's approximate folded search (candidates re-scored with the
    // full fingerprints).  ndevices: 1 = one GPU (round-robin placement like
    // get_next_gpu), 0 = shard over all GPUs.
    // full_on_device: a folded table may also keep its full fingerprints in HBM for the re-score (gsim_db_set_fold_full_on_device)
    // rccl_merge: the shards' results meet through the C ABI's collective (gsim_comm: RCCL all-gather over xGMI + a merge
    // kernel) instead of on the host -- unfolded tables only
    void copyToGPU(unsigned int fold_factor, int ndevices = 1, bool full_on_device = true, bool rccl_merge = false);

    unsigned int count() const { return static_cast<unsigned int>(m_total_count); }
    Fingerprint getFingerprint(unsigned int index) const; // :212-226

    // :341-381.  Appends to the result vectors (borrowed smiles / id pointers) and sets
    // approximate_result_count; a wrong dbkey logs and leaves the outputs untouched (:349-352).
    void search(const Fingerprint& query, const std::string& dbkey, unsigned int max_return_count,
                float similarity_cutoff, std::vector<char*>& results_smiles, std::vector<char*>& results_ids,
                std::vector<float>& results_scores, unsigned long& approximate_result_count) const;

    // fingerprintdb_cuda.cpp:20-54 (cutoff ignored, approximate_result_count untouched)
    void search_cpu(const Fingerprint& query, const std::string& dbkey, unsigned int max_return_count,
                    float similarity_cutoff, std::vector<char*>& results_smiles, std::vector<char*>& results_ids,
                    std::vector<float>& results_scores, unsigned long& approximate_result_count) const;

    char* getSmiles(int index) const { return m_smiles[index]; }
    char* getID(int index) const { return m_ids[index]; }
    size_t getFingerprintDataSize() const { return m_total_data_size; }
    int getFingerprintBitcount() const { return m_fp_intsize * static_cast<int>(sizeof(int)) * 8; }
    bool onGPU() const { return m_on_gpu; }

  private:
    gsim_db* m_db = nullptr;
    gsim_comm* m_comm = nullptr;
    int m_total_count = 0, m_fp_intsize = 0, m_fold_factor = 1;
    size_t m_total_data_size = 0;
    std::vector<char*> m_smiles;
    std::vector<char*> m_ids;
    std::string m_dbkey;
    bool m_on_gpu = false;
    bool m_synthetic = false;
    unsigned long long m_seed = 0;
    int m_kind = 0;
    std::vector<char> m_arena; // synthetic tables: the strings' storage
};

// fingerprintdb_cuda.cpp:92-103
void top_results_bubble_sort(std::vector<int>& indices, std::vector<float>& scores, int number_required);

} // namespace gpusim
