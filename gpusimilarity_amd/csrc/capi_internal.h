// capi_internal.h -- what the translation units of the C-ABI implementation (capi_*.cpp) share: the per-shard and
// per-handle state, error plumbing, and the host-side steps one TU implements and another calls.  Internal; the ABI is
// include/gpusim_hip.h.  No exception crosses the ABI.
#pragma once

#include "../../include/gpusim_hip.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "gsim_device.h"
#include "gsim_synth.h"

namespace gsim_host
{

extern thread_local std::string g_last_error;
extern thread_local bool g_force_each; // gsim_db_search_each: queries one by one, never a shared table pass

int fail(int code, const std::string& msg);
int fail_hip(hipError_t e, const char* what);

#define GSIM_HIP(call)                                   \
    do {                                                 \
        hipError_t e__ = (call);                         \
        if (e__ != hipSuccess) return fail_hip(e__, #call); \
    } while (0)


gsim::Knobs read_knobs(); // (the one place that calls getenv for tuning knobs; gsim_db_create)
#ifdef GSIM_TEST_HOOKS
int env_int(const char* name, int dflt);
#endif
// Test hook (compiled only with -DGSIM_TEST_HOOKS: testhooks/libgsim_hip.so, never the shipped library):
// GSIM_TEST_ALIAS_DEVICES=N makes the library present N logical devices that all
// live on physical device 0, so that the in-process multi-device code -- gsim_db_finalize(db, dev, n > 1), the shard
// fan-out and host merge of search_one / the batch path / folded tables, gsim_next_device's round robin,
// gpusimserver --gpus N -- runs on a one-GPU box exactly as it would on N GPUs (own stream, state and scratch per
// shard; only the physical device index differs).
int alias_devices();
int phys_device(int logical);
hipError_t set_device(int logical);

// Host memory the kernels write and the host reads WHILE the kernel still runs (completion words, result blocks): it has
// to be coherent (fine-grained) whatever the runtime's default for pinned memory is (HIP_HOST_COHERENT).
// Portable: pinned for, and mapped into, EVERY device of the process -- a multi-device handle's kernels on GPU 1..7 read
// queries from and write blocks into buffers that were allocated while another device was current.
constexpr unsigned kHostPolled = hipHostMallocPortable | hipHostMallocCoherent;
constexpr unsigned kHostPinned = hipHostMallocPortable; // staging the kernels read (queries) or copies go through
constexpr int kTimingRing = 1024;
// single-launch path: 4096 summary keys + the checkpoint tickets (kFusedCheckpoints x 9 counters, 128 B apart) + the
// arrival counters (128 B apart)
constexpr size_t kTicketWords = static_cast<size_t>(gsim::kFusedCheckpoints) * 9 * 32;
constexpr size_t kSummBytes = 4096 * 4 + kTicketWords * 4 + static_cast<size_t>(gsim::kFusedArriveWords) * 128;
constexpr int kQueryRing = 16;
constexpr int kPipe = 8; // single queries of one gsim_db_search_each call enqueued ahead of the one being waited for (< kQueryRing)

struct Shard {
    int device = 0;
    int num_cus = 256;
    int cu_share = 1;       // > 1: a LANE of another shard -- its grids are sized for num_cus / cu_share compute units (see `lanes`)
    // Small tables, gsim_db_search_each: the query's launch streams for half its time and selects for the other half, and
    // launches on one stream serialise.  Two lanes = two half-grid copies of this shard's search state (own stream, per-query
    // state, regions, exchange buffer; the SAME rows): consecutive queries alternate between them, so one query's scan
    // overlaps the other's selection (capi_query.cpp search_each_pipelined; made on first use).
    std::vector<Shard> lanes;
    uint64_t first_row = 0; // offset inside the handle's table
    uint64_t nrows = 0;
    uint32_t W = 0;         // words per row ON THE DEVICE (table width / fold factor)
    void* d_rows = nullptr;
    uint16_t* d_rowpop = nullptr; // popc(row) side array of the matrix-core batch pass (2 B per row, made on first use)
    bool rowpop_valid = false;
    bool owns_rows = false;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr; // the stream in use (own or caller's)
    gsim::ScanGeometry geo{};
    gsim::ScanGeometry fgeo{}; // single-launch path: fewer waves on small tables (every wave gets >= 4 chunks)
    bool state_dirty = false; // set when an enqueue failed: the device state is re-zeroed before the next one
    int sample_chunks = 4; // chunks per scan wave scored by the sample kernel (0 = off)
    uint32_t* d_query = nullptr;
    gsim::QueryState* d_state = nullptr;
    unsigned long long* d_cand = nullptr;
    uint32_t* d_cand_cb = nullptr;
    uint32_t* d_seg_count = nullptr;
    unsigned long long* d_final = nullptr;
    uint32_t* d_final_cb = nullptr;
    uint32_t final_cap = 0;
    // folded tables: the storage's FULL fingerprints in HBM too (when they fit), and the re-score's buffers
    uint32_t* d_full = nullptr;
    uint32_t* d_fq = nullptr;              // the full query (+ one word: the NaN flag)
    uint32_t* h_fq = nullptr;              // ... its pinned staging
    unsigned long long* d_key2 = nullptr;  // re-scored keys, 2 x 64 Ki (the sort's second buffer)
    uint32_t* d_cb2 = nullptr;
    unsigned long long* d_large = nullptr; // k > kSelectCap: the gathered top-k keys and the sort's second buffer, 2 x next_pow2(k) entries
    uint32_t large_cap = 0;
    gsim::LargeKState* d_lk = nullptr;
    uint32_t* d_bincur = nullptr;    // kScanBins cursors of launch_fused_binsort (zero between queries) + kScanBins words: the bins' first positions
    uint32_t binrank_skip = 0;       // large-k queries left that take the radix tail: the bin-ranked one handed a query back (ties)
    uint32_t binrank_streak = 0;     // ... consecutive times: the skip doubles (16, 32, ... 1024)
    bool classic_ready = false; // candidate / finalist scratch of the four-kernel pipeline (allocated on first use)
    void* d_pub = nullptr;      // single-launch path: the workgroups' published-candidate regions (128 KB each)
    void* d_hdr = nullptr;      // ... and their headers (64 B each)
    uint32_t* d_summ = nullptr; // single-launch path: per-wave checkpoint summaries (16 KB, zero between queries)
    uint32_t* h_done = nullptr; // single-launch path: pinned words, one per pipeline slot (since round 3 only their address is used: "the caller polls the header")
    uint32_t epoch = 0;
    uint32_t pub_tag = 0;           // the last single launch's FusedArgs::pub_tag
    // One synchronous query in flight on this shard (gsim_db_search: slot 0; gsim_db_search_each: up to kPipe, slot = query mod kPipe):
    // how it was enqueued, what its caller waits for, and why it had to be run again.
    struct PipeSlot {
        bool inflight = false;  // enqueued by a synchronous caller and not finished yet (whatever route it took)
        bool fused = false;     // it went through the single launch, which announces itself in the block's header ...
        uint32_t epoch = 0;     // ... with this epoch
        bool publish = false;   // a large-k query scanned by the publishing launch (header flag 2: run it again)
        bool binrank = false;   // ... and ranked by coarse bin (a hand-back sends the next ones to the radix tail)
        hipEvent_t ev = nullptr; // recorded behind the last kernel of an enqueue that is not the single launch's own: the caller waits
        bool ev_set = false;    // for THIS query, not for the queries enqueued behind it on the stream
        bool rerun = false;     // it ran behind a launch that left the per-query state dirty: not to be trusted, run again
        uint8_t why = 0;        // kQ* bits: how it was routed and why it was run again (gsim_debug_query_flags)
    } slot[kPipe];
    char* h_pipe = nullptr;          // kPipe pinned result blocks (gsim_db_search_each)
    size_t h_pipe_block = 0;
    // Tables whose scores tie heavily (narrow or very sparse fingerprints) make the single-launch path hand every
    // query back, i.e. scan twice: after consecutive hand-backs the synchronous path skips it for 2, 4, ... 64 queries.
    uint32_t redo_streak = 0, fused_skip = 0;
    uint32_t publish_streak = 0, publish_skip = 0; // the same back-off for the publishing launch of k above fused_select_max_k (synchronous callers)
    unsigned long long* d_dbg = nullptr; // GSIM_FUSED_DEBUG: per-workgroup phase timestamps
    void* d_result = nullptr;
    size_t result_bytes = 0;
    // pinned host staging; queries go through a ring so that back-to-back
    // asynchronous searches never overwrite a query whose upload is still queued
    uint32_t* h_query = nullptr; // kQueryRing slots of W words
    std::vector<hipEvent_t> q_ev; // scan-done event per slot (asynchronous searches)
    bool q_pending[kQueryRing] = {};
    uint32_t q_next = 0;
    unsigned char* h_result = nullptr;
    size_t h_result_bytes = 0;
    gsim::QueryState* h_state = nullptr; // staging for the running totals
    // timing
    std::vector<hipEvent_t> ev; // 3 per slot
    uint32_t ev_used = 0;
    std::vector<hipEvent_t> bev; // multi-query passes: 2 per slot
    uint32_t bev_used = 0;
    unsigned long long base_ncand = 0, base_nfinal = 0, base_nredo = 0; // device totals when timing was enabled
    // multi-query batches (allocated on first use)
    gsim::ScanGeometry bgeo{};
    uint32_t bq_cap = 0;          // queries the batch buffers hold
    uint32_t bseg_cap = 0;        // candidate slots per wave segment, now / at most (grown on overflow) / number of segments
    uint32_t bseg_max = 0, bseg_waves = 0;
    uint32_t* d_bqueries = nullptr;
    uint32_t* d_bqpop = nullptr;
    gsim::BatchQueryState* d_bstate = nullptr;
    unsigned long long* d_bcand = nullptr;
    uint32_t* d_bcand_cb = nullptr;
    uint32_t* d_bcand_q = nullptr;
    uint32_t* d_bseg_count = nullptr;
    unsigned long long* d_bfin_key = nullptr;
    uint32_t* d_bfin_cb = nullptr;
    uint32_t* d_bflags = nullptr; // [0] overflow flags, [1] ticket
    gsim::BatchRare* d_brare = nullptr;
    gsim::BatchRare* h_brare = nullptr; // pinned
    uint32_t* h_bflags = nullptr;
    uint32_t* h_bqueries = nullptr; // pinned staging: queries + popcounts
    unsigned char* h_bresult = nullptr;
    unsigned char* d_bresult = nullptr; // the select kernel writes here; one bulk copy to h_bresult
    size_t h_bresult_bytes = 0;
    // in-process collective route (gsim_db_set_comm, capi_comm.cpp): every shard's gather buffer holds one slot per
    // shard (an in-place all-gather: the shard's own kernels write slot `index of the shard`)
    unsigned char* d_gather = nullptr;
    size_t gather_bytes = 0;
    unsigned char* d_merged = nullptr; // first shard only: the merged blocks of a batch (single queries merge straight into h_result)
    size_t merged_bytes = 0;
    hipEvent_t gather_ev = nullptr;    // loop-back comm (aliased devices): "this shard's block is in its slot"
    hipEvent_t cev[3] = {};            // first shard, timing: scan done / gather done / merge done
};

} // namespace gsim_host

// resize() without zero-filling: the rows are copied in right away, by several threads, which then
// also take the first-touch page faults in parallel
template <class T> struct NoInitAllocator : std::allocator<T> {
    template <class U> struct rebind {
        using other = NoInitAllocator<U>;
    };
    template <class U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
    template <class U, class... Args> void construct(U* p, Args&&... args) { ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...); }
};

struct gsim_db {
    uint32_t fp_bits = 0;
    uint32_t W = 0;
    uint64_t nrows = 0;
    std::vector<uint32_t, NoInitAllocator<uint32_t>> host_rows; // host copy (reference: m_data)
    bool has_host_copy = false;
    bool finalized = false;
    std::vector<gsim_host::Shard> shards;
    std::vector<uint64_t> slice_first; // first row of every add_rows slice (reference: one storage each)
    uint32_t fold_requested = 1;       // gsim_db_set_fold_factor
    uint32_t fold = 1;                 // effective factor (divides W), fixed at finalize
    bool fold_full_on_device = true;   // gsim_db_set_fold_full_on_device
    uint32_t row_base = 0;
    gsim::Knobs knobs;                 // the environment's tuning knobs as gsim_db_create found them
    bool timing = false;
    gsim_timing acc{};
    unsigned long long dense_batches = 0; // multi-query passes whose dense cutoff the matrix-core pass counted itself
    unsigned long long batch_regrown = 0; // batches run again with larger candidate segments
    std::atomic<unsigned long long> large_k_published{0}; // shard queries with k > kSelectCap scanned by the single launch
    // why synchronous queries were run a second time / routed around the single launch (host-side counts, gsim_timing.rerun_*)
    unsigned long long lane_queries = 0; // shard queries of gsim_db_search_each answered by a half-grid lane
    unsigned long long rerun_own = 0, rerun_behind = 0, rerun_torn = 0, rerun_publish = 0, backoff_skips = 0;
    size_t query_flags_at = 0;        // ... the query search_one is answering
    std::vector<uint8_t> query_flags; // kQ* bits of every query of the last gsim_db_search / _each call (timing enabled)
    unsigned long long blocks_checked = 0, blocks_rechecked = 0, blocks_torn = 0; // single launch, synchronous callers: result blocks whose checksum
                                                               // did not match at first sight / never did (re-run)
    gsim_comm* comm = nullptr; // gsim_db_set_comm: shard results meet through an RCCL all-gather + merge_kernel instead of on the host
    size_t comm_root = 0;      // ... on this shard's device (every device's buffer holds all blocks after the gather)
    // One search at a time per handle (the reference serialises searches behind a function-static
    // mutex, fingerprintdb_cuda.cu:236): concurrent callers queue here.
    std::mutex search_mutex;
};

namespace gsim_host
{

inline uint32_t next_pow2_u32(uint64_t x)
{
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p > 0x80000000ull ? 0x80000000u : static_cast<uint32_t>(p);
}

inline uint32_t popcount_words(const uint32_t* q, uint32_t W)
{
    uint32_t a = 0;
    for (uint32_t i = 0; i < W; i++) a += static_cast<uint32_t>(__builtin_popcount(q[i]));
    return a;
}

enum QueryMode { kAuto = 0, kClassic = 1 };
// Per-query record of the synchronous routes (Shard::PipeSlot::why -> gsim_db::query_flags -> gsim_debug_query_flags)
constexpr uint8_t kQHandedBack = 1;   // the single launch ran the query and handed it back itself (gsim_timing.handed_back counts these)
constexpr uint8_t kQRerunBehind = 2;  // run again because a launch ahead of it on the stream ended without closing its query
constexpr uint8_t kQTorn = 4;         // run again because its block's checksum never matched
constexpr uint8_t kQSkipFused = 8;    // routed around the single launch by the back-off after earlier hand-backs
constexpr uint8_t kQPublishBack = 16; // large k: the publishing launch or the bin-ranked emission handed it back
constexpr uint8_t kQSkipPublish = 32; // large k: routed around the publishing launch / the bin-ranked emission by a back-off
constexpr uint32_t kBatchMaxQ = 256; // queries per batch call on a shard (larger requests are split)

// capi_lifecycle.cpp
int free_shard(Shard& s);
int setup_shard(gsim_db* db, Shard& s);
// capi_query.cpp: the single-query path
int ensure_classic_scratch(Shard& s);
int ensure_result_capacity(Shard& s, uint32_t k);
int enqueue_query(gsim_db* db, Shard& s, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha, float beta,
                  uint32_t row_base, void* out, bool caller_syncs, QueryMode mode = kAuto, uint32_t pipe_slot = 0);
int wait_stream(hipStream_t st);
int wait_event(hipEvent_t ev);
int finish_query_sync(gsim_db* db, Shard& s, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha, float beta,
                      uint32_t row_base, void* out, uint32_t pipe_slot = 0);
int search_one(gsim_db* db, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha, float beta, gsim_hit* hits,
               uint32_t* count, uint64_t* approx, std::vector<gsim_hit>& merged);
int check_search_args(gsim_db* db, const uint32_t* queries, int metric);
// capi_batch.cpp: multi-query passes
int enqueue_batch(gsim_db* db, Shard& s, const uint32_t* queries, uint32_t nq, uint32_t k, float cutoff, int metric, float alpha,
                  float beta, uint32_t row_base, void* results, bool allow_mfma = true);
int run_batch(gsim_db* db, const uint32_t* qb, uint32_t nb, uint32_t k, float cutoff, int metric, float alpha, float beta,
              const std::vector<void*>& outs);
int search_batched(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k, uint32_t kout, float cutoff, int metric, float alpha,
                   float beta, gsim_hit* hits, uint32_t* counts, uint64_t* approx);
// capi_folded.cpp
void fold_row(const uint32_t* row, uint32_t W, uint32_t F, uint32_t* out);
void fold_rows_mt(const uint32_t* rows, uint64_t nrows, uint32_t W, uint32_t F, uint32_t* out);
int search_folded(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k, float cutoff, gsim_hit* hits, uint32_t* counts,
                  uint64_t* approx);
// capi_merge.cpp
bool hit_before(const gsim_hit& x, const gsim_hit& y);
uint32_t merge_canonical_lists(const std::vector<gsim_hit>& lists, const std::vector<size_t>& ends, uint32_t k, gsim_hit* out);
// capi_comm.cpp: the collective route of a multi-device handle
int search_one_comm(gsim_db* db, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha, float beta, gsim_hit* hits,
                    uint32_t* count, uint64_t* approx);
int search_batch_comm(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k, uint32_t kout, float cutoff, int metric, float alpha,
                      float beta, gsim_hit* hits, uint32_t* counts, uint64_t* approx);
void free_comm_buffers(Shard& s);
// capi_batch.cpp: nb <= kBatchMaxQ queries on one shard, result blocks in device memory at `out` (the per-shard half of
// gsim_db_search_batch_device)
int batch_to_device(gsim_db* db, Shard& s, const uint32_t* qb, uint32_t nb, uint32_t k, float cutoff, int metric, float alpha,
                    float beta, uint32_t row_base, unsigned char* out);
// capi_debug.cpp: timing, the phase profile of instrumented runs
int drain_timing(gsim_db* db, Shard& s);
int read_totals(Shard& s, unsigned long long* ncand, unsigned long long* nfinal, unsigned long long* nredo = nullptr);
void dump_fused_phases(Shard& s);

} // namespace gsim_host
