/* gsim_oracle.h -- CPU restatement of the reference's fingerprint scan + top-k.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (gpusimilarity_amd/,
 * include/) may include, link or call this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Every function cites the reference (schrodinger/gpusimilarity) file:line whose
 * behaviour it restates.  Parity pinning: tests/test_oracle_golden.py checks this
 * oracle against (1) the reference's own known-answer test
 * test/test_gpusim.cpp:101-128 (TestSimilarityCutoff), (2) :134-146 (CPUSort),
 * (3) :148-166 (FoldFingerprint), (4) the top-15 (row, common, popc, score bits)
 * vectors captured from the running reference for test/small.fsim (SURVEY.md
 * Appendix C, committed as tests/golden/small_fsim_reference_topk.json) and
 * (5) oracle/_ref -- the reference's own calculation_functors.cpp compiled in
 * place -- when it has been built.
 *
 * Tversky and multi-query batching do not exist in the reference: for those the
 * oracle is the definition ("parity unpinned" by the reference; cross-checked
 * only through Tversky(1,1) == Tanimoto).
 */
#ifndef GSIM_ORACLE_H
#define GSIM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSO_METRIC_TANIMOTO 0
#define GSO_METRIC_TVERSKY 1

#define GSO_KIND_SPARSE 0 /* bit density 1/16, Morgan-like */
#define GSO_KIND_DENSE 1  /* bit density 1/2 */
#define GSO_KIND_MORGAN 2 /* Morgan-shaped: popcount 20..53, scaffold clusters, duplicates */

typedef struct {
    uint32_t row;     /* row index in the scanned table               */
    float score;      /* f32 score, reference arithmetic              */
    uint16_t common;  /* popc(query & row)                            */
    uint16_t popc_db; /* popc(row)                                    */
} gso_hit;

/* ---- synthetic data: counter-based, regenerable row by row -- */
uint64_t gso_splitmix64(uint64_t x);
/* word j of row i of the table with this seed/kind (W words per row). */
uint32_t gso_synth_word(uint64_t seed, int kind, uint64_t row, uint32_t W, uint32_t j);
void gso_synth_rows(uint32_t* out, uint64_t seed, int kind, uint64_t first_row,
                    uint64_t nrows, uint32_t W);
/* one row of the GSO_KIND_MORGAN table (gso_synth_word serves the other kinds word by word) */
void gso_synth_row_morgan(uint32_t* out, uint64_t seed, uint64_t row, uint32_t W);
/* row index used for query number q of an N-row table. */
uint64_t gso_query_row(uint64_t q, uint64_t nrows);

/* ---- scoring ------------------------------------------------------------- */
/* calculation_functors.cpp:6-20 (TanimotoFunctorCPU): raw score, no cutoff, NaN
 * kept.  Also emits the integer popcounts.  common/popc may be NULL. */
void gso_tanimoto_raw(const uint32_t* query, const uint32_t* db, uint64_t nrows,
                      uint32_t W, float* scores, uint16_t* common, uint16_t* popc);
/* Build-defined Tversky: c / (alpha*(a-c) + beta*(b-c) + c), f32, this order. */
float gso_score_one(int metric, float alpha, float beta, uint32_t a, uint32_t b,
                    uint32_t c);
/* fingerprintdb_cuda.cu:99-102: score >= cutoff ? score : 0 (NaN -> 0). */
float gso_score_den(int metric, float alpha, float beta, uint32_t a, uint32_t b, uint32_t c);
float gso_apply_cutoff(float score, float cutoff);

/* ---- search: fingerprintdb_cuda.cu:228-381 in canonical form --------------
 * rows kept: all rows when cutoff <= 0 (:263-273), else rows whose cut score
 * is non-zero (:265-271); approx = #kept; hits = first min(k, #kept) of the kept
 * rows ordered by (score desc, row asc) -- the set Thrust's stable sort_by_key
 * yields, in canonical order.  row_base is added to every returned row index
 * (shards / getOffsetIndex :128-131).  nthreads <= 1 -> single thread.        */
int gso_search(const uint32_t* query, const uint32_t* db, uint64_t nrows, uint32_t W,
               uint32_t k, float cutoff, int metric, float alpha, float beta,
               uint32_t row_base, int nthreads, gso_hit* hits, uint32_t* nhits,
               uint64_t* approx);

/* Merge G sorted hit lists (each canonical order) into the first k of the union,
 * canonical order: fingerprintdb_cuda.cu:363-380 (std::sort + truncate). */
void gso_merge_hits(const gso_hit* lists, const uint32_t* counts, uint32_t nlists,
                    uint32_t stride, uint32_t k, gso_hit* out, uint32_t* nout);

/* ---- CPU path pieces ------------------------------------------------------ */
/* fingerprintdb_cuda.cpp:73-103 top_results_bubble_sort (strict '>' => stable) */
void gso_bubble_sort(int* indices, float* scores, int count, int number_required);
/* fingerprintdb_cuda.cpp:20-54 search_cpu: raw scores (no cutoff, NaN kept),
 * bubble sort, first k.  Requires k <= nrows (the reference reads out of
 * bounds otherwise).  */
int gso_search_cpu(const uint32_t* query, const uint32_t* db, uint64_t nrows,
                   uint32_t W, uint32_t k, int* out_rows, float* out_scores);
/* calculation_functors.cpp:22-41 FoldFingerprintFunctorCPU for one fingerprint;
 * folded must be zero-initialised by the caller (as fold_data does, :60-61). */
void gso_fold(const int* unfolded, int unfolded_intsize, int factor, int* folded);

/* fold_data (fingerprintdb_cuda.cpp:56-69): every row folded by `factor`. */
void gso_fold_rows(const uint32_t* rows, uint64_t nrows, uint32_t W, int factor, uint32_t* folded);
/* copyToGPU's adjustment (fingerprintdb_cuda.cu:170-173): smallest factor >= requested
 * that divides the word count. */
int gso_effective_fold_factor(uint32_t W, int requested);
/* Folded search of ONE storage, fingerprintdb_cuda.cu:228-339 with m_fold_factor > 1:
 * folded query vs folded rows (cutoff applied to the FOLDED score, :258-273), canonical
 * top-R with R = k * F * (int)log2(2F) (:284-287), re-score those R rows with the full
 * fingerprints (:307-314, tanimoto_similarity_cpu :387-399), stable partial bubble sort
 * (:315), keep the first min(k, R) and stop at the first re-scored value < cutoff
 * (:317-331).  approx = number of folded survivors (:272-277).  Rows get row_base added. */
int gso_search_folded(const uint32_t* query, const uint32_t* db, uint64_t nrows, uint32_t W,
                      int fold_factor, uint32_t k, float cutoff, uint32_t row_base,
                      gso_hit* hits, uint32_t* nhits, uint64_t* approx);

#ifdef __cplusplus
}
#endif
#endif
