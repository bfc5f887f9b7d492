/* gpusim_hip.h -- C ABI of the MI355X-native fingerprint scan engine.
 *
 * This is the drop-in boundary for the ONE hot path of schrodinger/gpusimilarity:
 * "score every packed fingerprint of a table against a query (popcount
 * Tanimoto / Tversky), apply a cutoff, return the exact top-k".  In the
 * reference that path lives behind the C++ class gpusim::FingerprintDB
 * (fingerprintdb_cuda.h:53-140), whose stated purpose is to keep CUDA types
 * away from its callers (fingerprintdb_cuda.h:47, fingerprintdb_cuda.cpp:2-3).
 * The reference has no FFI of its own; the entry points below are what a
 * binding for that class would bind, one for each thing the class does with the
 * device.  Host C++ (gpusimilarity_amd/csrc/host/fingerprintdb.h, the Qt-free
 * twin of the reference class) reaches HIP only through these functions.
 *
 * Rules of the ABI: plain C, opaque handle, plain pointers and sizes, no HIP /
 * STL / torch types, never throws, every call returns a status (0 = ok, < 0 =
 * error, text via gsim_last_error()).  One search at a time per handle (the
 * reference serialises searches behind a function-static mutex,
 * fingerprintdb_cuda.cu:236, and its server is single threaded).
 *
 * Fingerprint layout: row-major uint32_t[nrows][fp_bits/32]; word i of a row is
 * bytes 4i..4i+3 of the RDKit BitVectToBinaryText string read little-endian --
 * the same pass-through reinterpretation as fingerprintdb_cuda.cu:117-126.
 */
#ifndef GPUSIM_HIP_H
#define GPUSIM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSIM_OK 0
#define GSIM_ERR_INVALID (-1)   /* bad argument                                  */
#define GSIM_ERR_NO_DEVICE (-2) /* no usable GPU / device index out of range     */
#define GSIM_ERR_HIP (-3)       /* a HIP runtime call failed                     */
#define GSIM_ERR_NOMEM (-4)     /* host or device allocation failed              */
#define GSIM_ERR_STATE (-5)     /* call not valid in the handle's current state  */

#define GSIM_METRIC_TANIMOTO 0 /* c / (a + b - c)          fingerprintdb_cuda.cu:89-103 */
#define GSIM_METRIC_TVERSKY 1  /* c / (al*(a-c)+be*(b-c)+c) build extension (BASELINE config 5) */

#define GSIM_SYNTH_SPARSE 0 /* bit density 1/16 (Morgan-like) */
#define GSIM_SYNTH_DENSE 1  /* bit density 1/2                */
#define GSIM_SYNTH_MORGAN 2 /* Morgan-shaped rows: popcount 20..53 of 1024, a dozen very frequent bits, series of
                               analogs, table-wide scaffolds, ~3 % exact duplicates (csrc/gsim_synth.h) -- coarse,
                               tie-heavy scores like the reference's ChEMBL / Zinc / Enamine tables */

typedef struct gsim_db gsim_db;

/* One result row.  `row` is the index in the table (plus the handle's row base,
 * see gsim_db_set_row_base) -- the ABI speaks row indices; SMILES / ID strings
 * live above it (reference: m_smiles/m_ids lookups, fingerprintdb_cuda.cu:297-304).
 * `score` is the float the reference would return; `common` = popc(q & row),
 * `popc_db` = popc(row) are the integers it was computed from. */
typedef struct {
    uint32_t row;
    float score;
    uint16_t common;
    uint16_t popc_db;
} gsim_hit;

/* Header of a device-resident result block written by gsim_db_search_device and
 * gsim_merge_device: {gsim_result_header; gsim_hit hits[k];}. */
typedef struct {
    uint32_t count;  /* number of valid hits (<= k)                         */
    uint32_t flags;  /* bit 0: block produced by the general (large) path; bit 1: the single launch handed the query back
                        (internal: never set in a block a call returns); bits 8..31: zero in every returned block
                        (the library's own pinned blocks carry the query's epoch there while it polls them) */
    uint64_t approx; /* "approximate matching results", see gsim_db_search  */
} gsim_result_header;

/* Per-handle timing of the last searches, measured with HIP events on the
 * stream the kernels were launched on (gsim_db_enable_timing). */
typedef struct {
    uint64_t queries;        /* searches accumulated since enable/reset           */
    double scan_ms_sum;      /* sum of scan-kernel (dominant kernel) durations    */
    double select_ms_sum;    /* sum of compaction + final select kernel durations */
    uint64_t candidates_sum; /* rows that survived the in-scan threshold filter   */
    uint64_t finalists_sum;  /* rows handed to the final select                   */
    uint64_t handed_back;    /* queries the single-launch path handed back to the four-kernel pipeline */
    uint64_t batches;        /* multi-query passes timed (gsim_db_search with nq >= 4)                  */
    double batch_kernel_ms_sum; /* sum of their dominant kernel's durations (the matrix-core contraction,
                                   or all table passes of the VALU route)                               */
    uint64_t handed_back_why;   /* union of the reasons, since the handle was created: 1 a wave's candidate store or a
                                   workgroup's list overflowed (heavy ties, ascending scores), 2 / 4 a wait ran out
                                   (GPU shared with another process), 8 more rows at the final threshold than a
                                   selector holds, 16 more of them owned by one selector than its list holds, 64 (k in (GSIM_FUSED_SELECT_MAX_K = 2048, 32768], rows of at least 512 bits: the publishing route)
                                   more rows in one of the top score bins than the bin-ranked emission takes (ties)     */
    uint64_t batches_dense_cutoff; /* multi-query passes, since the handle was created, whose cutoff kept so many rows that the
                                      matrix-core pass counted them from its accumulators (gsim_prefilter.h cutoff_band)      */
    uint64_t collectives;          /* searches (single queries or <= 256-query batches) merged through gsim_db_set_comm's route */
    double gather_ms_sum;          /* ... their all-gather: last shard kernel enqueued on the first shard's stream -> blocks gathered */
    double merge_ms_sum;           /* ... their merge_kernel                                                                  */
    uint64_t blocks_rechecked;     /* since the handle was created: result blocks of the single launch whose checksum did not match
                                      the hits when the header arrived in host memory (the hits were still on their way) ...   */
    uint64_t blocks_torn;          /* ... and those that never matched: the query was re-run on the four-kernel pipeline      */
    uint64_t batches_regrown;      /* since the handle was created: multi-query passes run again because a wave's candidate segment
                                      overflowed (the segments start at 4 Ki slots and grow to what was asked for, up to 64 Ki)   */
    uint64_t large_k_single_scan;  /* since the handle was created: shard queries with k above 2048 whose scan was the single launch's
                                      (it publishes, other kernels rank: DESIGN.md section 3); the rest of them were ranked inside it (k up to 8192)
                                      or took the four-kernel pipeline's scan */
    /* Why synchronous queries (gsim_db_search, _each, _timed) ran a second time on the four-kernel pipeline, counted on the host
     * since the handle was created -- `handed_back` above is the DEVICE's count of single launches that ended with a hand-back and
     * equals rerun_own + rerun_publish when nothing else went wrong: */
    uint64_t rerun_own;     /* the query's own single launch handed it back (reasons: handed_back_why)                          */
    uint64_t rerun_publish; /* large k: its publishing launch, or the bin-ranked emission behind it, handed it back             */
    uint64_t rerun_behind;  /* it ran behind a launch that ended without closing its query (GPU shared: a wait ran out and the
                               kernel was gone before its header): every query in flight on that state is run again            */
    uint64_t rerun_torn;    /* = blocks_torn                                                                                    */
    uint64_t lane_queries;  /* since the handle was created: shard queries of gsim_db_search_each that ran on one of the shard's two
                               half-grid lanes (small tables: consecutive queries overlap)                  */
    uint64_t backoff_skips; /* queries routed AROUND the single launch (or its publishing mode / the bin-ranked emission) because
                               earlier ones were handed back: they scan once, on the four-kernel pipeline, and are not hand-backs */
} gsim_timing;

/* ---- device enumeration / placement ------------------------------------- */
/* get_gpu_count()            fingerprintdb_cuda.cu:40-52  */
int gsim_device_count(int* count);
/* get_gpu_free_memory(i)     fingerprintdb_cuda.cu:33-38  */
int gsim_device_free_bytes(int device, size_t* free_bytes);
/* get_available_gpu_memory() fingerprintdb_cuda.cu:401-413: sum of free bytes */
int gsim_available_device_bytes(size_t* total_free_bytes);
/* get_next_gpu(required)     fingerprintdb_cuda.cu:54-68: round-robin over the
 * devices, skipping those without `required_bytes` free.  (The reference tests
 * device i but returns the round-robin device -- a latent bug, :57-62; here the
 * device that is returned is the one that was checked.)  GSIM_ERR_NO_DEVICE with
 * no GPU, GSIM_ERR_NOMEM when none has room (reference: throws, :65-66). */
int gsim_next_device(size_t required_bytes, int* device);

/* ---- table lifecycle ------------------------------------------------------ */
/* FingerprintDB::FingerprintDB   fingerprintdb_cuda.cu:133-166.  fp_bits must be
 * a positive multiple of 32 (":140 ASSUMES INT-DIVISIBLE SIZE"), <= 32768. */
int gsim_db_create(uint32_t fp_bits, gsim_db** out);
/* FingerprintDBStorage ctor      fingerprintdb_cuda.cu:111-126: appends one slice
 * of host rows (copied; the caller keeps its buffer).  Slices are concatenated in
 * call order; global row = slice offset + local (getOffsetIndex, :128-131). */
int gsim_db_add_rows(gsim_db* db, const uint32_t* rows, uint64_t nrows);
/* FingerprintDB::copyToGPU(1)    fingerprintdb_cuda.cu:168-183: uploads the rows
 * and allocates the search scratch.  ndevices == 1: the whole table goes to
 * `device` (device < 0: gsim_next_device picks).  ndevices > 1: rows are split
 * into ndevices contiguous shards on devices device .. device+ndevices-1
 * (ndevices == 0: all devices).  The host copy is kept (reference keeps m_data,
 * fingerprintdb_cuda.h:46) for gsim_db_row and gsim_db_search_cpu. */
int gsim_db_finalize(gsim_db* db, int device, int ndevices);
/* FingerprintDB::copyToGPU(fold_factor > 1)  fingerprintdb_cuda.cu:168-195 + fold_data
 * fingerprintdb_cuda.cpp:56-69: call before gsim_db_finalize.  The table is then kept
 * on the GPU OR-folded to fp_bits / F bits (F = the smallest factor >= fold_factor that
 * divides the word count, :170-173), one shard per add_rows slice (one reference
 * "storage" each).  gsim_db_search on a folded table is the reference's approximate
 * search: the k*F*(int)log2(2F) best folded scores per storage (:284-287) are
 * re-scored with the full fingerprints from the host copy (:307-331).  Tanimoto only.
 * Not needed on MI355X for capacity (288 GB hold 2.25 G unfolded 1024-bit rows). */
int gsim_db_set_fold_factor(gsim_db* db, uint32_t fold_factor);
uint32_t gsim_db_fold_factor(const gsim_db* db);
/* Folded tables: may gsim_db_finalize also place the storages' FULL fingerprints in HBM, so that the candidates are
 * re-scored on the GPU instead of on the host?  1 (default): yes, when all of them fit next to the folded rows and the
 * search scratch with 2 GB to spare, on every device -- all storages or none; 0: never (callers that fold because the
 * tables do NOT fit and still have tables to place, e.g. a server loading several databases).  Before gsim_db_finalize. */
int gsim_db_set_fold_full_on_device(gsim_db* db, int allow);
/* FoldFingerprintFunctorCPU (calculation_functors.cpp:22-41) for one fingerprint of
 * `words` 32-bit words: out receives words / fold_factor words (host function). */
int gsim_fold_fingerprint(const uint32_t* fingerprint, uint32_t words, uint32_t fold_factor,
                          uint32_t* out);
/* Synthetic table generated directly in HBM (no host copy): row r, word j =
 * the counter-based generator of SURVEY.md 8d (oracle/gsim_oracle.c
 * gso_synth_word is the CPU twin), rows first_row .. first_row+nrows-1.
 * Replaces add_rows+finalize for benchmark tables that exceed host memory. */
int gsim_db_generate(gsim_db* db, uint64_t seed, int kind, uint64_t first_row,
                     uint64_t nrows, int device);
/* The same table split over `ndevices` GPUs (device .. device + ndevices - 1) exactly as
 * gsim_db_finalize(db, device, ndevices) splits uploaded rows -- contiguous, equal shares -- each
 * shard generated on its own device: the one-process multi-GPU handle of fingerprintdb_cuda.cu:176-182
 * for tables that exceed host memory (bench.py --in-process, gpusimserver synthetic:... --gpus N). */
int gsim_db_generate_sharded(gsim_db* db, uint64_t seed, int kind, uint64_t first_row,
                             uint64_t nrows, int device, int ndevices);
/* Row `row` of that synthetic table, computed on the host by the generator's own code (benchmark
 * queries are rows of the table; no device, no handle needed). */
int gsim_synth_row(uint64_t seed, int kind, uint64_t row, uint32_t fp_bits, uint32_t* out_words);
/* Borrow rows that already live in device memory (e.g. a torch tensor); the
 * caller keeps ownership and must keep them alive.  16-byte aligned. */
int gsim_db_attach_device_rows(gsim_db* db, const void* d_rows, uint64_t nrows,
                               int device);
int gsim_db_destroy(gsim_db* db);

/* FingerprintDB::count()                  fingerprintdb_cuda.h:74  */
uint64_t gsim_db_count(const gsim_db* db);
/* FingerprintDB::getFingerprintBitcount() fingerprintdb_cuda.h:123-126 */
uint32_t gsim_db_fp_bits(const gsim_db* db);
/* FingerprintDB::getFingerprintDataSize() fingerprintdb_cuda.h:122 */
size_t gsim_db_data_bytes(const gsim_db* db);
/* FingerprintDB::getFingerprint(index)    fingerprintdb_cuda.cu:212-226 */
int gsim_db_row(const gsim_db* db, uint64_t row, uint32_t* out_words);
/* number of device shards the table was placed on (1 unless ndevices > 1) */
int gsim_db_shard_count(const gsim_db* db);
/* the device shard `shard` lives on (-1: no such shard) -- the order gsim_comm_create wants */
int gsim_db_shard_device(const gsim_db* db, int shard);

/* ---- search --------------------------------------------------------------- */
/* FingerprintDB::search          fingerprintdb_cuda.cu:341-381 (+ search_storage
 * :228-339).  For each of the nq queries (nq * fp_bits/32 words, host memory):
 *   score(row) = metric(query, row); score = score >= cutoff ? score : 0  (:101;
 *   NaN -> 0); rows kept = all rows when cutoff <= 0, else rows with score != 0
 *   (:263-273); approx[q] = #kept (:272-277, :367-369); hits = the first
 *   min(k, #kept) kept rows in the order (score desc, row asc) -- the set the
 *   reference's sort_by_key returns, in canonical order (within a tie group the
 *   reference's own order depends on heap addresses, :366).
 * hits: caller-allocated nq*k entries, query q's hits at hits[q*k ..];
 * counts[q] = number of valid hits; approx may be NULL.  alpha/beta are used by
 * GSIM_METRIC_TVERSKY only.  Four or more queries in one call share table passes
 * (up to 256 per pass; on the matrix cores for 256..2048-bit rows) -- purely an
 * execution detail: every query gets exactly the result a call with nq = 1 would
 * return. */
int gsim_db_search(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k,
                   float cutoff, int metric, float alpha, float beta, gsim_hit* hits,
                   uint32_t* counts, uint64_t* approx);
/* The same call with the queries answered strictly one after the other through the single-query path
 * (FingerprintDB::search called nq times, as the reference's server does, gpusim.cpp:306-374): no table
 * pass is shared.  On a single-shard handle up to eight queries are enqueued ahead of the one being waited for
 * (each with its own pinned result block): they still run one at a time on the GPU, but the next one starts when
 * the previous kernel retires, not after a host round trip. */
int gsim_db_search_each(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k,
                        float cutoff, int metric, float alpha, float beta, gsim_hit* hits,
                        uint32_t* counts, uint64_t* approx);
/* gsim_db_search for nq queries strictly ONE AT A TIME -- nothing enqueued ahead, unlike gsim_db_search_each -- with the
 * wall time of each, measured inside the library around the single-query path: seconds[q] = query q in host memory ->
 * its hits in host memory (the reference's server logs the same interval per request, gpusim.cpp:420-429).  What a
 * caller of gsim_db_search with nq = 1 waits for, without the caller's own binding overhead. */
int gsim_db_search_timed(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k, float cutoff, int metric,
                         float alpha, float beta, gsim_hit* hits, uint32_t* counts, uint64_t* approx, double* seconds);
/* FingerprintDB::search_cpu      fingerprintdb_cuda.cpp:20-54: the reference's
 * explicit host path (TanimotoFunctorCPU on all host threads + the partial
 * bubble sort of :92-103).  Same outputs as gsim_db_search, but reference
 * semantics: cutoff is ignored, NaN scores are kept, approx is not written,
 * and k > count is an error (the reference reads out of bounds).  Needs the host
 * copy.  Never used by gsim_db_search -- there is no CPU fallback. */
int gsim_db_search_cpu(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k,
                       float cutoff, gsim_hit* hits, uint32_t* counts);

/* ---- device-resident results (one process per GPU + RCCL gather) ---------- */
/* All kernels of this handle are enqueued on `hip_stream` (a hipStream_t passed
 * as void*; NULL = the handle's own stream). */
int gsim_db_set_stream(gsim_db* db, void* hip_stream);
/* Added to every returned row index: the shard's first global row. */
int gsim_db_set_row_base(gsim_db* db, uint32_t row_base);
/* bytes of one result block for top-k: sizeof(header) + k*sizeof(gsim_hit),
 * rounded up to 16. */
size_t gsim_result_block_bytes(uint32_t k);
/* One query, single-shard handle: leaves {header; hits[k]} in device memory at
 * d_result (gsim_result_block_bytes(k) bytes), enqueued on the handle's stream,
 * no host synchronisation for any k (k > 8192: a radix select on the device finds
 * the k-th best key, nothing is sized by a count the host would have to read) --
 * except that the FIRST call with a k above the largest one seen so far allocates
 * its buffers (hipMalloc synchronises the device once).
 * query is host memory (fp_bits/32 words). */
int gsim_db_search_device(gsim_db* db, const uint32_t* query, uint32_t k, float cutoff,
                          int metric, float alpha, float beta, void* d_result);
/* Merge nblocks result blocks (contiguous in device memory, block_bytes apart,
 * e.g. the output of an RCCL all-gather) into one block holding the first k of
 * their union in canonical order; approx = sum.  FingerprintDB::search's host
 * merge, fingerprintdb_cuda.cu:363-380.  Enqueued on hip_stream of `device`. */
int gsim_merge_device(int device, void* hip_stream, const void* d_blocks,
                      uint32_t nblocks, size_t block_bytes, uint32_t k, void* d_result);

/* nq queries (host memory, nq x fp_bits/32 words), single-shard handle: result
 * block q = {header; hits[k]} at d_results + q * gsim_result_block_bytes(k) in
 * device memory, enqueued on the handle's stream.  Batches share table passes
 * (the multi-query / matrix-core pass) exactly as in gsim_db_search; the call
 * waits for the handle's stream once per 256 queries (queries that need the
 * single-query pipeline -- heavy ties, candidate overflow -- are re-enqueued
 * before it returns).  Reference: the per-storage half of FingerprintDB::search,
 * fingerprintdb_cuda.cu:228-339, for a batch (build-defined). */
int gsim_db_search_batch_device(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k,
                                float cutoff, int metric, float alpha, float beta,
                                void* d_results);
/* gsim_merge_device for nq queries at once: d_blocks holds nranks arrays of nq
 * result blocks (rank-major, the layout an all-gather of each rank's
 * gsim_db_search_batch_device output produces); merged block q goes to
 * d_results + q * block_bytes. */
int gsim_merge_device_batch(int device, void* hip_stream, const void* d_blocks,
                            uint32_t nranks, uint32_t nq, size_t block_bytes, uint32_t k,
                            void* d_results);

/* Host twin of gsim_merge_device for result blocks that are in host memory (the
 * in-process multi-device path, and the gloo/CPU tests of the gather+merge
 * logic): std::sort + truncate exactly as fingerprintdb_cuda.cu:363-380. */
int gsim_merge_host(const void* blocks, uint32_t nblocks, size_t block_bytes, uint32_t k,
                    void* result);

/* ---- in-process collective: per-GPU top-k merged via an RCCL all-gather ---- */
/* FingerprintDB::search runs one host thread per storage and merges their results on the host
 * (fingerprintdb_cuda.cu:356-380); gsim_db_search on a multi-device handle does the same by default.  With a
 * communicator attached, every shard's kernels leave their result block in their own device's HBM, ONE grouped
 * ncclAllGather over the shards' streams (RCCL over xGMI; 16 + 12 k bytes per device) brings the blocks to every
 * device and a merge kernel on the first shard's device writes the merged block into pinned host memory -- no
 * PyTorch, no second process.  Results are identical to the host merge's, bit for bit. */
typedef struct gsim_comm gsim_comm;
/* ncclCommInitAll over `devices` (distinct device indices, in the order of the handle's shards:
 * gsim_db_finalize(db, device, n) places shard i on device + i). */
int gsim_comm_create(const int* devices, int ndevices, gsim_comm** out);
int gsim_comm_destroy(gsim_comm* comm); /* detach it from every handle first (gsim_db_set_comm(db, NULL)) */
/* The RCCL this process is bound to: NCCL_VERSION_CODE of the header the library was built with, ncclGetVersion() of the
 * library the loader resolved, and that library's file (a process that imported PyTorch first runs on PyTorch's own
 * librccl.so, not /opt/rocm/lib's).  gsim_comm_create refuses a different MAJOR version; a different minor one is this
 * call's to report (the entry points used are unchanged across 2.x).  Any pointer may be NULL. */
int gsim_rccl_info(int* header_version, int* runtime_version, char* path, size_t path_bytes);
int gsim_comm_size(const gsim_comm* comm);
/* Route gsim_db_search / gsim_db_search_each of this handle through the communicator (NULL: back to the host
 * merge).  The communicator's devices must be the shards' devices, in order.  Not for folded tables. */
int gsim_db_set_comm(gsim_db* db, gsim_comm* comm);
/* Which shard's device merges the gathered blocks and answers the host (default 0; after the all-gather every
 * device holds all blocks, so any of them can). */
int gsim_db_set_comm_root(gsim_db* db, int shard);

/* ---- instrumentation ------------------------------------------------------ */
int gsim_db_enable_timing(gsim_db* db, int enable); /* resets the accumulators */
int gsim_db_get_timing(gsim_db* db, gsim_timing* out); /* synchronises the stream */
/* With timing enabled: one byte per query of the handle's LAST gsim_db_search / _each / _timed call (first min(n, queries) bytes of
 * `flags`; *written = how many) -- 1 its own single launch handed it back, 2 run again behind a launch that did not close its query,
 * 4 its block's checksum never matched, 8 routed around the single launch by the back-off, 16 large k: publishing launch or
 * bin-ranked emission handed it back, 32 large k: routed around those by a back-off.  For the tests that compare the pipelined
 * entry point with one query at a time. */
int gsim_debug_query_flags(gsim_db* db, uint8_t* flags, uint32_t n, uint32_t* written);
/* Litmus tests of the hardware behaviours the single launch rests on, with its own instructions (gsim_litmus.hip; nothing of the product
 * calls them).  test 1: aligned 16-byte write-through (sc1) stores are seen whole by 16-byte sc1 loads of another workgroup; test 2: a
 * 16-byte system-scope store into pinned host memory is whole for a host that polls one of its words and then reads the others (how
 * gsim_db_search reads a result header); test 3: entry then header from one lane -- how often a reader sees the header first, and that
 * a re-read always finds the entry; test 4: the same with the product's shape (entries stored by another wave, a workgroup barrier, then
 * the header).  `workgroups` even (pairs of writer + reader; test 2: one writer each), `iterations` stores per slot
 * (64 slots per pair; test 2: one per workgroup).  stats[8] = {loads (test 2: host observations), torn values, headers seen, entries
 * behind their header at the first read, entries that never caught up, re-reads, workgroups that ran out of time, stores}. */
int gsim_debug_litmus(int device, int test, uint32_t workgroups, uint32_t iterations, unsigned long long* stats);
/* score of every (common, popc_db) pair for a query of popcount a, computed ON
 * THE DEVICE with the scan kernel's arithmetic: out[c * (max_b+1) + b].  Used
 * by the parity tests to pin the f32 divide bit for bit. */
int gsim_debug_score_table(int device, int metric, float alpha, float beta, uint32_t a,
                           uint32_t max_b, uint32_t max_c, float* out);

/* The constants of the matrix-core pass's division-free pre-filter (gsim_prefilter.h) for query
 * popcounts 0..max_qa: without a cutoff for every threshold bin, out[(qa * 512 + bin) * 4 + i],
 * with one for the cutoff's level, out[qa * 4 + i]; i = {ka, kb, u, v}.  device >= 0: computed by
 * a kernel on that device; device < 0: computed on the host by the same code.  The parity tests
 * compare the two (the exhaustive filter-never-rejects-an-accepted-pair test runs on the host). */
int gsim_debug_prefilter_constants(int device, int metric, float alpha, float beta, uint32_t max_qa,
                                   int has_cutoff, float cutoff, float* out);

/* The device sort behind k > 8192 and the folded tables' re-score (launch_sort_desc: tiles sorted in LDS, positions by
 * counting): keys[0 .. n), n a power of two, sorted in place, descending; equal keys (the callers pad with zeros) keep a
 * deterministic order.  For the parity tests. */
int gsim_debug_sort_desc(int device, unsigned long long* keys, uint32_t n);

const char* gsim_last_error(void);
const char* gsim_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GPUSIM_HIP_H */
