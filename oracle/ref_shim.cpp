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
