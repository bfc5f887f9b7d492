// capi_cpu.cpp -- the reference's explicit host path (FingerprintDB::search_cpu, fingerprintdb_cuda.cpp:20-103).
// Only gsim_db_search_cpu runs it; it is never a fallback of the GPU path.
#include "capi_internal.h"

using namespace gsim_host;

extern "C" {

// The reference's explicit host path, fingerprintdb_cuda.cpp:20-54: score every
// row on all host threads (QtConcurrent::blockingMap -> std::thread here), then
// top_results_bubble_sort (:92-103) and the first k.  Not a fallback: only this
// entry point runs it.
int gsim_db_search_cpu(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k, float cutoff, gsim_hit* hits,
                       uint32_t* counts)
{
    (void) cutoff; // ignored by the reference's CPU path
    if (!db || !queries || (!hits && k) || !counts) return fail(GSIM_ERR_INVALID, "NULL argument");
    if (!db->has_host_copy) return fail(GSIM_ERR_STATE, "search_cpu needs the host copy of the rows");
    if (k > db->nrows) return fail(GSIM_ERR_INVALID, "search_cpu: k exceeds the row count");
    const uint64_t n = db->nrows;
    const uint32_t W = db->W;
    std::vector<int> indices(n);
    std::vector<float> scores(n);
    std::vector<uint16_t> cm(n), pc(n);
    unsigned nthreads = std::thread::hardware_concurrency();
    if (nthreads == 0) nthreads = 1;
    if (nthreads > n) nthreads = n ? static_cast<unsigned>(n) : 1;
    for (uint32_t q = 0; q < nq; q++) {
        const uint32_t* query = queries + static_cast<size_t>(q) * W;
        auto work = [&](uint64_t lo, uint64_t hi) {
            for (uint64_t r = lo; r < hi; r++) {
                const uint32_t* d = db->host_rows.data() + r * W;
                int total = 0, common = 0;
                int pd = 0;
                for (uint32_t i = 0; i < W; i++) {
                    const int p2 = __builtin_popcount(d[i]);
                    total += __builtin_popcount(query[i]) + p2;
                    pd += p2;
                    common += __builtin_popcount(query[i] & d[i]);
                }
                scores[r] = static_cast<float>(common) / static_cast<float>(total - common);
                cm[r] = static_cast<uint16_t>(common);
                pc[r] = static_cast<uint16_t>(pd);
                indices[r] = static_cast<int>(r);
            }
        };
        std::vector<std::thread> pool;
        const uint64_t per = (n + nthreads - 1) / nthreads;
        for (unsigned t = 0; t < nthreads; t++) {
            const uint64_t lo = per * t, hi = std::min<uint64_t>(lo + per, n);
            if (lo < hi) pool.emplace_back(work, lo, hi);
        }
        for (auto& th : pool) th.join();
        // partial bubble sort, strict '>' (stable)
        for (uint32_t i = 0; i < k; i++) {
            for (uint64_t j = n - 1; j > i; j--) {
                if (scores[j] > scores[j - 1]) {
                    std::swap(indices[j], indices[j - 1]);
                    std::swap(scores[j], scores[j - 1]);
                }
            }
        }
        for (uint32_t i = 0; i < k; i++) {
            gsim_hit& h = hits[static_cast<size_t>(q) * k + i];
            h.row = static_cast<uint32_t>(indices[i]) + db->row_base;
            h.score = scores[i];
            h.common = cm[indices[i]];
            h.popc_db = pc[indices[i]];
        }
        counts[q] = k;
    }
    return GSIM_OK;
}

} // extern "C"
