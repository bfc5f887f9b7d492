// gsim_device.h -- launch interface between the C-ABI host code (capi_*.cpp)
// and the gfx950 kernels (gsim_*.hip).  Internal; not part of the ABI.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <cstddef>
#include <stdint.h>

namespace gsim
{

// In-scan coarse histogram: linear bins over the score range [0, 1].
constexpr int kScanBins = 1024;
// Largest finalist count the single-workgroup LDS select handles.
constexpr int kSelectCap = 8192;
constexpr int kScanBlock = 256; // 4 wavefronts
constexpr int kFusedCheckpoints = 16; // single-launch path: threshold checkpoints (trip counts 1, 4, 16, ... and 3/4 of the trips)

// Device-resident per-query state, zeroed before every scan.
// Device-resident per-query state.  Zero when a query starts: allocated zeroed,
// and the last workgroup of the select kernel re-zeroes it for the next query
// (no per-query memset on the stream).
struct QueryState {
    uint32_t ghist[kScanBins];   // candidates per coarse bin (bins >= each workgroup's final threshold)
    unsigned long long kept;     // rows with score >= cutoff (cutoff > 0 only)
    unsigned long long ncand;    // total candidates emitted by the scan
    uint32_t nfinal;             // finalists appended by the compaction
    uint32_t done;               // select-kernel workgroups that have finished (ticket)
    uint32_t gtau;               // table-wide threshold bin shared by all workgroups of the scan (monotone)
    uint32_t elected;            // single-launch path: highest in-loop checkpoint (1-based) whose threshold election has been held
                                 // ({gtau, elected} are polled with ONE 64-bit load: static_assert below)
    uint32_t redo;               // single-launch path gave up (candidate overflow, heavy ties): the gated
                                 // classic kernels behind it run the query; their select kernel clears it
    uint32_t pad0;
    // --- single-launch path (fused_kernel) ---
    unsigned long long sel_done; // top-level closing ticket: groups of selectors that have written their hits (bits 0..15), those
                                 // that saw the query fail (bits 16..31), the sum of the words of all hits written (bits 32..63)
    // --- not reset per query (the reset boundary is offsetof(redo_why)) ---
    uint32_t redo_why;           // the union of the reasons (kRedo*) of every query handed back so far
    uint32_t pad1;
    unsigned long long ncand_sum; // running totals for gsim_db_get_timing
    unsigned long long nfinal_sum;
    unsigned long long queries;
    unsigned long long redo_sum; // queries the single-launch path handed back to the four-kernel pipeline
};
static_assert(offsetof(QueryState, gtau) % 8 == 0 && offsetof(QueryState, elected) == offsetof(QueryState, gtau) + 4,
              "the single launch's poller reads {gtau, elected} with one 64-bit load");
static_assert(offsetof(QueryState, sel_done) % 8 == 0, "64-bit atomic");

// The single launch's closing word carries a checksum of the result block for synchronous callers: the sum (mod 2^32) of
// the 32-bit words of every hit written, plus epoch * kBlockCheckMul, travels in the upper half of the header's approx
// field (a shard holds < 2^31 rows, so the true upper half is zero) -- the host verifies it against the hits it reads
// before it trusts the block, and clears it (capi_query.cpp finish_query_sync).
constexpr uint32_t kBlockCheckMul = 0x9E3779B1u;

// Why the single-launch path handed a query back (bits of QueryState::redo; the gated kernels only test for non-zero).
constexpr uint32_t kRedoStore = 1;        // a wave met more rows at its threshold than its LDS store holds / a region overflowed
constexpr uint32_t kRedoElectionWait = 2; // small table: a threshold election did not arrive within wait_ticks (shared GPU)
constexpr uint32_t kRedoArrivalWait = 4;  // the grid-wide arrival did not complete within wait_ticks (shared GPU)
constexpr uint32_t kRedoFinalists = 8;    // more rows at or above the final threshold than a selector's LDS holds
constexpr uint32_t kRedoOwned = 16;       // ... or more of them owned by one selector than its list holds
constexpr uint32_t kRedoSeen = 32;        // a selector found the query already failed
constexpr uint32_t kRedoBinTies = 64;     // large k, ranked by coarse bin: one of the top bins holds more rows than kBinRankCap (ties)

struct ScanGeometry {
    uint32_t lanes_per_row; // 16-byte lanes per fingerprint (fp words / 4), 0 = generic path
    uint32_t unroll;        // 16-byte loads in flight per lane
    uint32_t chunk_rows;    // rows per wave iteration
    uint32_t nwaves;        // wavefronts in the grid
    uint32_t seg_cap;       // candidate slots per wavefront
    uint64_t nchunks;
    uint32_t ragged_loads;  // generic widths that are whole 16-byte units per row, not a power of two (896, 1536 ... bits): loads per
                            // chunk of scan_ragged_kernel (the odd part of W / 4: 3, 5 or 7), 0 = the LDS-staged generic scan
    uint32_t ragged_words;  // 1: rows are NOT whole 16-byte units (W = 3, 5, 7, 9, 11 or twice that: 96 ... 704 bits) -- the single launch's
                            // word-granular variant (scan_rows_wragged), ragged_loads = the odd part of W; the four-kernel pipeline
                            // keeps the LDS-staged generic scan for them
};

struct ScanArgs {
    const void* rows;       // device, row-major uint32[nrows][W]
    uint64_t nrows;
    uint32_t W;             // words per fingerprint
    const uint32_t* query;  // W words; device memory or device-visible pinned host memory (read once per wave)
    uint32_t* query_dev;    // device copy for the kernels after the scan (the scan writes it when != query)
    uint32_t qpop;          // popc(query)
    uint32_t k;
    float cutoff;
    int metric;
    float alpha, beta;
    unsigned long long* cand; // device, nwaves*seg_cap keys
    uint32_t* cand_cb;        // device, nwaves*seg_cap (common << 16 | popc_db), parallel to cand
    uint32_t* seg_count;      // device, nwaves
    QueryState* state;        // device
    const uint32_t* gate;     // NULL, or device word: the classic kernels return at once unless *gate != 0
                              // (they are enqueued behind the single-launch path as its fallback)
};

// ---- single-launch path: scan + publish + select in ONE kernel --------------------------------
constexpr uint32_t kFusedMaxK = 8192;       // largest k the single-launch path serves
constexpr uint32_t kFusedPublishMaxK = 262144; // ... and the largest it scans and publishes for (k > kSelectCap: the large-k kernels rank what it published)
// M of the publishing launch's in-loop reports: up to 64 as long as 64 rows per wave cover k (65 536 hits on 1024 waves), up to 256 beyond
// (mth_best keeps four keys per lane -- eight from M = 65 on -- i.e. 256 / 512 per wave: a report is the wave's M-th best of those,
// valid at any M, tight while M is well below that)
constexpr uint32_t fused_publish_max_m(uint32_t k) { return k > 65536u ? 256u : 64u; }
constexpr uint32_t kFusedPublishOnly = 2048u; // FusedArgs::xflags: scan + publish, no selection (launch_fused_handoff follows)
constexpr int kFusedWaveCap = 2048;         // candidate slots per wavefront, in LDS
constexpr int kFusedSelectors = 256;        // workgroups of the grid: every one of them ranks its share of the finalists
constexpr uint32_t kFusedRegion = 4 * kFusedWaveCap; // published entries (16 B each) of one workgroup: its fixed region of the list
constexpr uint32_t kFusedHeaderBytes = 16;  // per workgroup: {entries | sorted << 31, bucket shift | launch tag << 5 | failed << 31, its end-of-scan report (64-bit key)}
constexpr uint32_t kFusedArriveWords = 25;  // arrival (publishing launches of large-k queries only: the last workgroup tidies up): 8 group counters (b % 8), 1 top counter, [8 unused]; closing: 8 group tickets -- 128 B apart

struct FusedArgs {
    void* pub;            // device, nwg regions of kFusedRegion x 16 B {key, cb, launch tag}: what each workgroup publishes
    void* hdr;            // device, nwg headers of kFusedHeaderBytes
    uint32_t* arrive;     // device, kFusedArriveWords words 128 B apart (zero between queries)
    uint32_t* summ;       // device, nwaves score keys: the waves' in-loop checkpoint summaries (zero between queries)
    uint32_t summ_keys;   // M: at the in-loop checkpoints every wave reports its M-th best key (fused_summary_keys), 0 = no checkpoints
    uint32_t final_keys;  // Mw: at the end of the scan every workgroup reports its Mw-th best key (fused_final_keys), 0 = no final threshold
    uint32_t* tickets;    // device, kFusedCheckpoints x 9 counters 128 B apart (8 per-group + 1 top), zero between queries
    void* result;         // result block: device memory or device-visible pinned host memory
    uint32_t row_base;
    uint32_t* done_flag;  // non-NULL: a synchronous caller polls the result header (in pinned host memory): its flags word carries `epoch` << 8
    uint32_t epoch;
    uint32_t pub_tag;     // non-zero, new for every launch on this handle: the fourth word of every entry this launch publishes, its low 26
                          // bits in every header -- readers tell this query's lists from what a region held before (no arrival counter)
    uint32_t wait_ticks;     // bound of the grid-wide wait (100 MHz ticks): a few scan times, see fused_kernel
    uint32_t xflags;         // 4 = QueryState::gtau was seeded by the sample kernel (a coarse bin); experiments (GSIM_FUSED_FLAGS): 2 = no in-loop checkpoints, 1024 = release fence before the closing ticket, 4096 = the published entries get their tags a few microseconds after the header (drives the selectors' read-again path), 8192 = the final threshold always by ranking every report (the path behind a sampled election that found no sample)
    unsigned long long* dbg; // NULL, or 24 timestamps (100 MHz wall clock) per workgroup: phase profile (GSIM_FUSED_DEBUG)
};

// bytes of the three device buffers of the single-launch path for a grid of nwg workgroups
inline size_t fused_pub_bytes(uint32_t nwg) { return static_cast<size_t>(nwg) * kFusedRegion * 16; }
inline size_t fused_hdr_bytes(uint32_t nwg) { return static_cast<size_t>(nwg) * kFusedHeaderBytes; }

hipError_t launch_fused(const ScanArgs& a, const ScanGeometry& g, const FusedArgs& f, hipStream_t s);
// Behind a kFusedPublishOnly launch: the published rows of the coarse bins at or above the k-th best's become the finalists of
// the large-k kernels (their keys into `finalists`, QueryState::nfinal set; ::ghist was filled by the launch) -- unless the
// launch handed the query back (QueryState::redo != 0).
// ... or, for callers that look at the result block (a hand-back can be run again by the host): the same rows placed BY COARSE BIN,
// highest bin first (the histogram gives every bin's first position; `cursors`: kScanBins counters, zero between queries, + kScanBins words that receive the layout), so
// that launch_binrank_emit only has to order each bin's rows among themselves.  A top bin of more than kBinRankCap rows: handed
// back (kRedoBinTies), nothing is placed.
constexpr uint32_t kBinRankCap = 16384; // (a hand-back costs a second scan; a bin of 16 Ki rows ~40 us of compares)
hipError_t launch_fused_binsort(const ScanArgs& a, const FusedArgs& f, uint32_t nwg, unsigned long long* finalists, uint32_t cap, uint32_t* cursors, hipStream_t s);
hipError_t launch_fused_handoff(const ScanArgs& a, const FusedArgs& f, uint32_t nwg, unsigned long long* finalists, uint32_t cap, hipStream_t s);
bool fused_supported(const ScanGeometry& g);
uint32_t fused_summary_keys(uint32_t nwaves, uint32_t k, uint32_t max_m = 16);
uint32_t fused_final_keys(uint32_t nwg, uint32_t k);

// Geometry of the scan grid for a table (host side, no device work).
// Every tuning knob of the library, read from the environment ONCE per handle -- by gsim_db_create -- and nowhere else
// (INTEGRATION.md lists them with their meaning; nothing under a search entry point calls getenv).
struct Knobs {
    int scan_waves_per_cu = 4;       // GSIM_SCAN_WAVES_PER_CU
    int scan_ragged = 1;             // GSIM_SCAN_RAGGED         0: odd widths take the LDS-staged generic scan
    int sample_chunks = 4;           // GSIM_SAMPLE_CHUNKS
    int sample_shift = 15;           // GSIM_SAMPLE_SHIFT
    int fused = 1;                   // GSIM_FUSED               0: every query through the four-kernel pipeline
    long long fused_max_rows = -1;   // GSIM_FUSED_MAX_ROWS
    int fused_debug = 0;             // GSIM_FUSED_DEBUG         in-kernel phase stamps
    int fused_flags = 0;             // GSIM_FUSED_FLAGS
    int fused_seed_narrow = 1;       // GSIM_FUSED_SEED_NARROW
    int fused_publish = 1;           // GSIM_FUSED_PUBLISH       0: k in (2048, 8192] is ranked inside the single launch, k above 8192 scans with the four-kernel pipeline
    int fused_select_max_k = 2048;   // GSIM_FUSED_SELECT_MAX_K  largest k the single launch ranks itself (2048 ... 8192) where it can also publish for the large-k kernels
    int largek_one_block_max = 32768; // GSIM_LARGEK_ONE_BLOCK_MAX
    int fused_publish_max_k = 100000; // GSIM_FUSED_PUBLISH_MAX_K (up to kFusedPublishMaxK; round 5: 32768).  Measured at 100 M rows (profiles/r06_large_k.txt):
                                      // k = 50 000 2.18 -> 1.93 ms (0.83 of the roofline), 100 000 2.31 -> 2.17; at 131 072 and beyond the waves' stores
                                      // overflow before the first useful election and the queries are handed back: slower than the four-kernel pipeline
    int publish_narrow = 1;           // GSIM_PUBLISH_NARROW      0: 128 / 256-bit rows never take the publishing launch for large k (round 5)
    int largek_binrank_max_k = 65536; // GSIM_LARGEK_BINRANK_MAX_K  above it the published rows go through the radix select + sort (measured: 100 M rows, k = 100 000
                                      // 2.31 ms bin-ranked with hand-backs for crowded bins, 2.11 ms by the radix tail; at k = 50 000 1.94 against 1.97)
    int fused_backoff = 1;           // GSIM_FUSED_BACKOFF       0: a query handed back never routes later ones around the single launch
    int publish_min_rows_per_k = 0;  // GSIM_PUBLISH_MIN_ROWS_PER_K  tables shorter than this many rows per hit rank large k inside the launch (up
                                     // to round 5: 64 -- short tables publish most of their rows; measured in round 6 the publishing route is
                                     // still 1.5 ... 6 x faster there and hands nothing back: profiles/r06_short_tables_large_k.txt)
    int largek_binrank = 1;          // GSIM_LARGEK_BINRANK      0: the published rows of a large-k query always go through the radix select + sort
    int each_pipeline = 1;           // GSIM_EACH_PIPELINE       0: gsim_db_search_each waits for every query before the next
    int each_lanes = 2;              // GSIM_EACH_LANES          2 (or 4): pipelined queries of small tables alternate between that many part-grid lanes; 0: never
    int each_lanes_share = 2;        // GSIM_EACH_LANES_SHARE    a lane's grid is sized for the CUs divided by this
    int each_lanes_publish = 1;      // GSIM_EACH_LANES_PUBLISH  0: large k (the publishing route) stays on one stream
    int each_lanes_max_mb = 4096;    // GSIM_EACH_LANES_MAX_MB   ... only shards of at most this many MB do (larger ones are HBM-bound: nothing to overlap)
    int batch = 1;                   // GSIM_BATCH               0: no shared table passes
    int batch_waves_per_cu = 12;     // GSIM_BATCH_WAVES_PER_CU
    int batch_seg_cap = 65536;       // GSIM_BATCH_SEG_CAP
    int batch_seg_cap_init = 4096;   // GSIM_BATCH_SEG_CAP_INIT
    int batch_sample_chunks = 8;     // GSIM_BATCH_SAMPLE_CHUNKS
    int batch_rpl = 0;               // GSIM_BATCH_RPL           rows per lane of the VALU pass (0: by width)
    int batch_mfma_min_q = 4;        // GSIM_BATCH_MFMA_MIN_Q    0: batches never take the matrix cores
    int batch_mfma_sample = 1;       // GSIM_BATCH_MFMA_SAMPLE
    int batch_mfma_dense = 1;        // GSIM_BATCH_MFMA_DENSE    0: dense cutoffs go to the VALU pass
    int debug_batch = 0;             // GSIM_DEBUG_BATCH         print the batch flags (instrumented builds: phase counters)
    int fold_full_on_device = 1;     // GSIM_FOLD_FULL_ON_DEVICE
    int fold_rescore_host = 0;       // GSIM_FOLD_RESCORE=host
};

ScanGeometry scan_geometry(uint64_t nrows, uint32_t W, int num_cus, int waves_per_cu, int unroll, bool ragged = true);
// ... of the single launch where it differs from the four-kernel pipeline's: rows of 3, 5, 7, 9, 11 or twice that many WORDS
// (false: it does not -- use scan_geometry's)
bool fused_word_geometry(uint64_t nrows, uint32_t W, int num_cus, ScanGeometry* out, bool ragged = true);

// Optional K0: starting threshold from a strided sample of chunks_per_wave chunks per scan wave.
hipError_t launch_sample(const ScanArgs& a, const ScanGeometry& g, uint32_t chunks_per_wave, hipStream_t s, bool* launched = nullptr, int sample_shift = 15);
hipError_t launch_scan(const ScanArgs& a, const ScanGeometry& g, hipStream_t s);

// Compaction of candidates at or above the k-th best coarse bin into `finalists`.
hipError_t launch_compact(const ScanArgs& a, const ScanGeometry& g, unsigned long long* finalists,
                          uint32_t* finalists_cb, uint32_t finalists_cap, hipStream_t s);

// Final exact select + sort of the finalists (k <= kSelectCap, any finalist count);
// writes {gsim_result_header; gsim_hit[k]}.
hipError_t launch_select(const ScanArgs& a, const unsigned long long* finalists, const uint32_t* finalists_cb,
                         uint32_t finalists_cap, uint32_t row_base, void* d_result, hipStream_t s);
// Large-k path (k > kSelectCap): device state of the radix select (zero between queries).
struct LargeKState {
    unsigned long long prefix; // leading bytes of the k-th largest finalist key found so far
    uint32_t remaining;        // its rank among the finalists that share the prefix
    uint32_t ticket;
    uint32_t count;            // keys gathered
    uint32_t all;              // fewer finalists than k: every finalist is taken
    uint32_t hist[256];
};
hipError_t launch_largek_select(const ScanArgs& a, const unsigned long long* finalists, uint32_t finalists_cap, LargeKState* lk,
                                unsigned long long* out, uint32_t out_cap, uint32_t* hint, bool one_block, hipStream_t s);

// Large-k path (k > kSelectCap): the gathered top-k keys sorted in two launches (tiles in LDS, positions by counting).
// (n_pow2 keys, unique apart from zero padding; `tmp` holds n_pow2 more; *sorted = where the result is: keys or tmp)
hipError_t launch_sort_desc(unsigned long long* keys, unsigned long long* tmp, uint32_t n_pow2, hipStream_t s, unsigned long long** sorted);
// ... the large-k path's last two launches: tiles sorted (slots past LargeKState::count read as padding -- nobody zero-fills
// the buffer), then every key's position by counting AND its hit written there, the header, and -- the last workgroup -- the
// per-query state re-zeroed (until round 4: a fill, the sort, an emission kernel and a reset kernel)
hipError_t launch_largek_sort_emit(const ScanArgs& a, unsigned long long* keys, uint32_t n_pow2, LargeKState* lk, uint32_t row_base,
                                   uint64_t approx_if_no_cutoff, uint32_t flags, void* d_result, hipStream_t s);

// Behind launch_fused_binsort: every finalist's position = its bin's first position + the number of larger keys in its bin; the
// hits of the first k, the header (flag 2 and no hits if QueryState::redo is set: the host runs the query again), the per-query
// state and the cursors re-zeroed by the last workgroup.  Two launches instead of the radix select + gather + two-launch sort.
hipError_t launch_binrank_emit(const ScanArgs& a, const unsigned long long* finalists, uint32_t cap, uint32_t* cursors, LargeKState* lk,
                               uint32_t row_base, uint64_t approx_if_no_cutoff, uint32_t flags, void* d_result, hipStream_t s);

// Folded tables: re-score the candidates of a folded search (result block `folded_block`, device memory) with the full
// fingerprints, stable sort by the new score, first min(k, .) at or above the cutoff -> out_block (fingerprintdb_cuda.cu:
// 307-331).  npad = candidates rounded up to a power of two (<= 65536): size of keys / cbs.  *nan_flag is set when a
// score is NaN (the caller then takes the host path).
hipError_t launch_fold_rescore(const void* folded_block, const uint32_t* full_rows, const uint32_t* full_query, uint32_t W, uint32_t qpop,
                               unsigned long long* keys, uint32_t* cbs, uint32_t npad, uint32_t* nan_flag, uint32_t k, float cutoff,
                               uint32_t row_base, void* out_block, hipStream_t s);

// Merge of result blocks (multi-GPU gather) for nq queries: the lists of query q are the blocks
// q, q + nq, ... (nblocks of them); merged block q goes to d_results + q * block_bytes.
hipError_t launch_merge_batch(const void* d_blocks, uint32_t nblocks, uint32_t nq, size_t block_bytes, uint32_t k,
                              void* d_results, hipStream_t s);

// ---- multi-query batches (gsim_batch.hip) ------------------------------------------
constexpr int kBQ = 32;     // queries per pass over the table
constexpr int kBBins = 512; // linear coarse bins of the batch filter

struct BatchQueryState { // one per query of a batch, device memory, zeroed before the batch
    uint32_t ghist[kBBins];
    uint32_t gtau;
    uint32_t nfinal;
    uint32_t bstar;
    uint32_t pad;             // matrix-core pass: candidates emitted so far (threshold update trigger)
    unsigned long long kept;
};

// Rarely used arguments (emission / compaction paths) live in device memory so that the
// scan's inner loop has scalar registers left for double-buffered query loads.
struct BatchRare {
    BatchQueryState* qstate; // device, Q
    unsigned long long* cand; // per-wave segments, seg_cap entries each
    uint32_t* cand_cb;
    uint32_t* cand_q;
    uint32_t* seg_count;      // nwaves
    unsigned long long* fin_key; // Q x kSelectCap
    uint32_t* fin_cb;
    uint32_t* flags;          // bit 0: a candidate segment overflowed
    uint32_t* ticket;
    uint32_t seg_cap;
    uint32_t pad;
};

struct BatchArgs {
    const void* rows;
    uint64_t nrows;
    const uint32_t* queries; // device, Q x W words
    const uint32_t* qpop;    // device, Q
    const BatchRare* rare;   // device copy of the rare arguments
    const uint16_t* rowpop;  // matrix-core pass: popc(row) per row (launch_row_popcounts), padded to whole 16-byte chunks
    uint32_t W;
    uint32_t q0, nq;         // this pass: queries q0 .. q0+nq-1 (nq <= kBQ, matrix-core pass: kMfmaQueries)
    uint32_t k;
    float cutoff;
    int metric;
    float alpha, beta;
    uint32_t opts;           // host side only -- bit 0: matrix-core sample pass allowed, bit 1: dense-cutoff variant allowed,
                             // bits 8-15: rows per lane of the VALU pass (0: by width) -- the handle's Knobs
};

bool batch_supported(uint32_t W);
hipError_t launch_batch_pass(const BatchArgs& a, const BatchRare& rare_host, const ScanGeometry& g,
                             uint32_t sample_chunks, uint32_t row_base, void* results, size_t block_bytes,
                             hipStream_t s);

// Matrix-core variant of the scan (gsim_batch_mfma.hip): up to kMfmaQueries queries per table pass.
constexpr int kMfmaQueries = 256;
bool batch_mfma_supported(uint32_t W);
// the side array of row popcounts the matrix-core pass reads: nrows 16-bit counts (allocate row_popcount_bytes(nrows))
hipError_t launch_row_popcounts(const void* rows, uint64_t nrows, uint32_t W, uint16_t* d_out, hipStream_t s);
inline size_t row_popcount_bytes(uint64_t nrows) { return static_cast<size_t>((nrows + 7) / 8 + 1) * 16; }
uint32_t batch_mfma_waves(int num_cus);
hipError_t launch_batch_mfma_scan(const BatchArgs& a, int num_cus, hipStream_t s);
bool batch_mfma_dense_applies(int metric, float alpha, float beta, float cutoff, bool enabled);
bool launch_batch_mfma_sample(const BatchArgs& a, int num_cus, hipStream_t s, hipError_t* err);
bool batch_mfma_sample_applies(uint32_t W, uint64_t nrows, uint32_t nq, uint32_t k, int num_cus, bool enabled);
hipError_t launch_batch_mfma_pass(const BatchArgs& a, const ScanGeometry& g, int num_cus, uint32_t sample_chunks,
                                  uint32_t row_base, void* results, size_t block_bytes, hipStream_t s,
                                  hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);

// debug hooks for the pre-filter of the matrix-core pass (gsim_prefilter.h): device and host twins
hipError_t launch_prefilter_table(int tversky, float alpha, float beta, uint32_t max_qa, int has_cutoff, float cutoff,
                                  float* d_out, hipStream_t s);
void prefilter_table_host(int tversky, float alpha, float beta, uint32_t max_qa, int has_cutoff, float cutoff, float* out);

hipError_t launch_generate(void* rows, uint64_t seed, int kind, uint64_t first_row, uint64_t nrows,
                           uint32_t W, hipStream_t s);

// gsim_litmus.hip: the hardware behaviours the single launch rests on, as litmus kernels (gsim_debug_litmus)
hipError_t launch_litmus_pair(void* entries, void* headers, unsigned long long* stats, uint32_t nblocks, uint32_t iters, int with_header,
                              unsigned long long budget_ticks, hipStream_t s);
hipError_t launch_litmus_host(void* slots, const uint32_t* ack, unsigned long long* stats, uint32_t nblocks, uint32_t iters, unsigned long long budget_ticks,
                              hipStream_t s);
hipError_t launch_score_table(int metric, float alpha, float beta, uint32_t a, uint32_t max_b,
                              uint32_t max_c, float* d_out, hipStream_t s);

} // namespace gsim
