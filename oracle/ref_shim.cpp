// ref_shim.cpp -- extern "C" driver around the REFERENCE's own CPU functors.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes
// the reference's header where it lies (/root/reference/calculation_functors.h)
// and is linked with the reference's calculation_functors.cpp compiled in place
// by oracle/build_ref.sh into oracle/_ref/libgsim_ref.so (git-ignored).  It lets
// the tests check the restatement in gsim_oracle.c against the real
// TanimotoFunctorCPU / FoldFingerprintFunctorCPU, and lets bench.py time the
// reference's host functor path as cpu_baseline kind "reference".
//
// Row parallelism mirrors QtConcurrent::blockingMap over the index vector
// (fingerprintdb_cuda.cpp:42-44): every index is handed to the functor once,
// split across `nthreads` std::threads.
#include "calculation_functors.h"

#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

namespace
{
struct RefTable {
    int fp_intsize;
    std::vector<int> data;
};
} // namespace

extern "C" {

void* gsref_table_create(const int* rows, uint64_t nrows, int fp_intsize)
{
    auto* t = new RefTable;
    t->fp_intsize = fp_intsize;
    t->data.assign(rows, rows + nrows * static_cast<uint64_t>(fp_intsize));
    return t;
}

void gsref_table_destroy(void* table)
{
    delete static_cast<RefTable*>(table);
}

// Scores every row with gpusim::TanimotoFunctorCPU (calculation_functors.cpp:6-20).
void gsref_tanimoto_scan(void* table, const int* query, float* out_scores,
                         uint64_t nrows, int nthreads)
{
    auto* t = static_cast<RefTable*>(table);
    gpusim::Fingerprint ref(query, query + t->fp_intsize);
    std::vector<float> scores(nrows);
    gpusim::TanimotoFunctorCPU functor(ref, t->fp_intsize, t->data, scores);
    if (nthreads < 1)
        nthreads = 1;
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            functor(static_cast<int>(i));
        }
    };
    if (nthreads == 1) {
        work(0, nrows);
    } else {
        std::vector<std::thread> pool;
        const uint64_t per = (nrows + nthreads - 1) / nthreads;
        for (int k = 0; k < nthreads; k++) {
            uint64_t lo = per * k, hi = lo + per < nrows ? lo + per : nrows;
            if (lo >= hi)
                break;
            pool.emplace_back(work, lo, hi);
        }
        for (auto& th : pool)
            th.join();
    }
    for (uint64_t i = 0; i < nrows; i++)
        out_scores[i] = scores[i];
}

// The reference's whole CPU search with a selection instead of its O(k N) bubble sort: every row
// scored by gpusim::TanimotoFunctorCPU on `nthreads` threads (as above), then the canonical top-k
// (score desc, row asc): every thread selects the k best of its slice (std::nth_element +
// sort), one thread merges the per-thread lists.  What bench.py times as "the reference's host
// functor path + top-k" (BASELINE.md section 3).  Returns the number of hits written.
int gsref_search_topk(void* table, const int* query, uint64_t nrows, int nthreads, int k, int* out_rows,
                      float* out_scores)
{
    auto* t = static_cast<RefTable*>(table);
    gpusim::Fingerprint ref(query, query + t->fp_intsize);
    std::vector<float> scores(nrows);
    gpusim::TanimotoFunctorCPU functor(ref, t->fp_intsize, t->data, scores);
    if (nthreads < 1)
        nthreads = 1;
    if (static_cast<uint64_t>(nthreads) > nrows)
        nthreads = nrows ? static_cast<int>(nrows) : 1;
    const uint64_t per = (nrows + nthreads - 1) / nthreads;
    auto before = [&](int a, int b) { return scores[a] > scores[b] || (scores[a] == scores[b] && a < b); };
    std::vector<std::vector<int>> best(nthreads);
    auto work = [&](int tix) {
        const uint64_t lo = per * tix, hi = lo + per < nrows ? lo + per : nrows;
        for (uint64_t i = lo; i < hi; i++)
            functor(static_cast<int>(i));
        std::vector<int>& idx = best[tix];
        idx.resize(hi > lo ? hi - lo : 0);
        for (uint64_t i = lo; i < hi; i++)
            idx[i - lo] = static_cast<int>(i);
        if (idx.size() > static_cast<size_t>(k)) {
            std::nth_element(idx.begin(), idx.begin() + k, idx.end(), before);
            idx.resize(k);
        }
    };
    std::vector<std::thread> pool;
    for (int tix = 1; tix < nthreads; tix++)
        pool.emplace_back(work, tix);
    work(0);
    for (auto& th : pool)
        th.join();
    std::vector<int> all;
    for (auto& b : best)
        all.insert(all.end(), b.begin(), b.end());
    const size_t n = all.size() < static_cast<size_t>(k) ? all.size() : static_cast<size_t>(k);
    std::partial_sort(all.begin(), all.begin() + n, all.end(), before);
    for (size_t i = 0; i < n; i++) {
        out_rows[i] = all[i];
        out_scores[i] = scores[all[i]];
    }
    return static_cast<int>(n);
}

// gpusim::FoldFingerprintFunctorCPU (calculation_functors.cpp:22-41) on one FP.
void gsref_fold(const int* unfolded, int unfolded_intsize, int factor, int* folded)
{
    std::vector<int> in(unfolded, unfolded + unfolded_intsize);
    std::vector<int> out(unfolded_intsize / factor, 0);
    gpusim::FoldFingerprintFunctorCPU(factor, unfolded_intsize, in, out)(0);
    for (size_t i = 0; i < out.size(); i++)
        folded[i] = out[i];
}

} // extern "C"
