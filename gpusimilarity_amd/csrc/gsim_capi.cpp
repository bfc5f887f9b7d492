// gsim_capi.cpp -- implementation of the C ABI in include/gpusim_hip.h.
//
// Host side of the scan engine: table placement (shards), per-query launch
// sequence (scan -> compact -> select), result transfer, host merge across
// in-process shards, the reference's explicit CPU path, timing.  All device work
// goes through gsim_device.h.  No exception crosses the ABI.
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

namespace
{

thread_local std::string g_last_error;
thread_local bool g_force_each = false; // gsim_db_search_each: queries one by one, never a shared table pass

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

int fail_hip(hipError_t e, const char* what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return GSIM_ERR_HIP;
}

#define GSIM_HIP(call)                                   \
    do {                                                 \
        hipError_t e__ = (call);                         \
        if (e__ != hipSuccess) return fail_hip(e__, #call); \
    } while (0)

int env_int(const char* name, int dflt)
{
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    return std::atoi(v);
}

// Test hook (no production use): GSIM_TEST_ALIAS_DEVICES=N makes the library present N logical devices that all
// live on physical device 0, so that the in-process multi-device code -- gsim_db_finalize(db, dev, n > 1), the shard
// fan-out and host merge of search_one / the batch path / folded tables, gsim_next_device's round robin,
// gpusimserver --gpus N -- runs on a one-GPU box exactly as it would on N GPUs (own stream, state and scratch per
// shard; only the physical device index differs).
int alias_devices()
{
    static const int n = env_int("GSIM_TEST_ALIAS_DEVICES", 0);
    return n > 0 ? n : 0;
}

int phys_device(int logical)
{
    return alias_devices() ? 0 : logical;
}

hipError_t set_device(int logical)
{
    return hipSetDevice(phys_device(logical));
}

// Host memory the kernels write and the host reads WHILE the kernel still runs (completion words, result blocks): it has
// to be coherent (fine-grained) whatever the runtime's default for pinned memory is (HIP_HOST_COHERENT).
constexpr unsigned kHostPolled = hipHostMallocCoherent;
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
    unsigned long long* d_key2 = nullptr;  // re-scored keys, 64 Ki
    uint32_t* d_cb2 = nullptr;
    unsigned long long* d_large = nullptr; // k > kSelectCap: the gathered top-k keys (sorted in place), next_pow2(k) entries
    uint32_t large_cap = 0;
    gsim::LargeKState* d_lk = nullptr;
    bool classic_ready = false; // candidate / finalist scratch of the four-kernel pipeline (allocated on first use)
    void* d_pub = nullptr;      // single-launch path: the workgroups' published-candidate regions (128 KB each)
    void* d_hdr = nullptr;      // ... and their headers (64 B each)
    uint32_t* d_summ = nullptr; // single-launch path: per-wave checkpoint summaries (16 KB, zero between queries)
    uint32_t* h_done = nullptr; // single-launch path: pinned words, one per pipeline slot (since round 3 only their address is used: "the caller polls the header")
    uint32_t epoch = 0;
    bool slot_fused[kPipe] = {};    // the synchronous enqueue of the slot went through the single-launch path ...
    uint32_t slot_epoch[kPipe] = {}; // ... with this epoch
    char* h_pipe = nullptr;          // kPipe pinned result blocks (gsim_db_search_each)
    size_t h_pipe_block = 0;
    // Tables whose scores tie heavily (narrow or very sparse fingerprints) make the single-launch path hand every
    // query back, i.e. scan twice: after consecutive hand-backs the synchronous path skips it for 2, 4, ... 64 queries.
    uint32_t redo_streak = 0, fused_skip = 0;
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
    uint32_t bseg_cap = 0;
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
};

} // namespace

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
    std::vector<Shard> shards;
    std::vector<uint64_t> slice_first; // first row of every add_rows slice (reference: one storage each)
    uint32_t fold_requested = 1;       // gsim_db_set_fold_factor
    uint32_t fold = 1;                 // effective factor (divides W), fixed at finalize
    uint32_t row_base = 0;
    bool timing = false;
    gsim_timing acc{};
    unsigned long long dense_batches = 0; // multi-query passes whose dense cutoff the matrix-core pass counted itself
    // One search at a time per handle (the reference serialises searches behind a function-static
    // mutex, fingerprintdb_cuda.cu:236): concurrent callers queue here.
    std::mutex search_mutex;
};

namespace
{

int free_shard(Shard& s)
{
    (void) set_device(s.device);
    if (s.stream) (void) hipStreamSynchronize(s.stream);
    if (s.owns_rows && s.d_rows) (void) hipFree(s.d_rows);
    if (s.d_rowpop) (void) hipFree(s.d_rowpop);
    if (s.d_query) (void) hipFree(s.d_query);
    if (s.d_state) (void) hipFree(s.d_state);
    if (s.d_cand) (void) hipFree(s.d_cand);
    if (s.d_cand_cb) (void) hipFree(s.d_cand_cb);
    if (s.d_final_cb) (void) hipFree(s.d_final_cb);
    if (s.d_seg_count) (void) hipFree(s.d_seg_count);
    if (s.d_final) (void) hipFree(s.d_final);
    if (s.d_result) (void) hipFree(s.d_result);
    if (s.d_full) (void) hipFree(s.d_full);
    if (s.d_fq) (void) hipFree(s.d_fq);
    if (s.h_fq) (void) hipHostFree(s.h_fq);
    if (s.d_key2) (void) hipFree(s.d_key2);
    if (s.d_cb2) (void) hipFree(s.d_cb2);
    if (s.d_large) (void) hipFree(s.d_large);
    if (s.d_lk) (void) hipFree(s.d_lk);
    if (s.d_pub) (void) hipFree(s.d_pub);
    if (s.d_hdr) (void) hipFree(s.d_hdr);
    if (s.d_summ) (void) hipFree(s.d_summ);
    if (s.d_dbg) (void) hipFree(s.d_dbg);
    if (s.h_done) (void) hipHostFree(s.h_done);
    if (s.h_pipe) (void) hipHostFree(s.h_pipe);
    if (s.h_query) (void) hipHostFree(s.h_query);
    if (s.h_result) (void) hipHostFree(s.h_result);
    if (s.h_state) (void) hipHostFree(s.h_state);
    if (s.d_bqueries) (void) hipFree(s.d_bqueries);
    if (s.d_bqpop) (void) hipFree(s.d_bqpop);
    if (s.d_bstate) (void) hipFree(s.d_bstate);
    if (s.d_bcand) (void) hipFree(s.d_bcand);
    if (s.d_bcand_cb) (void) hipFree(s.d_bcand_cb);
    if (s.d_bcand_q) (void) hipFree(s.d_bcand_q);
    if (s.d_bseg_count) (void) hipFree(s.d_bseg_count);
    if (s.d_bfin_key) (void) hipFree(s.d_bfin_key);
    if (s.d_bfin_cb) (void) hipFree(s.d_bfin_cb);
    if (s.d_bflags) (void) hipFree(s.d_bflags);
    if (s.d_brare) (void) hipFree(s.d_brare);
    if (s.h_brare) (void) hipHostFree(s.h_brare);
    if (s.h_bflags) (void) hipHostFree(s.h_bflags);
    if (s.h_bqueries) (void) hipHostFree(s.h_bqueries);
    if (s.h_bresult) (void) hipHostFree(s.h_bresult);
    if (s.d_bresult) (void) hipFree(s.d_bresult);
    for (auto e : s.ev) (void) hipEventDestroy(e);
    for (auto e : s.bev) (void) hipEventDestroy(e);
    for (auto e : s.q_ev) (void) hipEventDestroy(e);
    if (s.own_stream) (void) hipStreamDestroy(s.own_stream);
    s = Shard{};
    return 0;
}

uint32_t next_pow2_u32(uint64_t x)
{
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p > 0x80000000ull ? 0x80000000u : static_cast<uint32_t>(p);
}

// Allocate the per-shard search scratch once the rows are in place.
int setup_shard(gsim_db* db, Shard& s)
{
    if (s.nrows > 0x7FFFFFFFull) // candidate / finalist slots are 32-bit indexed, capacity a power of two <= 2^31
        return fail(GSIM_ERR_INVALID, "more than 2^31-1 rows on one device: shard the table over more devices");
    GSIM_HIP(set_device(s.device));
    hipDeviceProp_t prop;
    GSIM_HIP(hipGetDeviceProperties(&prop, phys_device(s.device)));
    s.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    GSIM_HIP(hipStreamCreateWithFlags(&s.own_stream, hipStreamNonBlocking));
    s.stream = s.own_stream;
    const int wpc = env_int("GSIM_SCAN_WAVES_PER_CU", 4);
    const int unroll = env_int("GSIM_SCAN_UNROLL", 8);
    if (s.W == 0) s.W = db->W;
    s.geo = gsim::scan_geometry(s.nrows, s.W, s.num_cus, wpc, unroll);
    s.sample_chunks = env_int("GSIM_SAMPLE_CHUNKS", 4);
    s.fgeo = s.geo;
    if (s.geo.nchunks < 4ull * s.geo.nwaves) { // small table: threshold checkpoints need a few trips per wave
        uint64_t nw = s.geo.nchunks / 4 / 4 * 4;
        s.fgeo.nwaves = static_cast<uint32_t>(nw < 4 ? 4 : nw);
    }
    GSIM_HIP(hipMalloc(&s.d_query, static_cast<size_t>(s.W) * 4));
    GSIM_HIP(hipMalloc(&s.d_state, sizeof(gsim::QueryState)));
    GSIM_HIP(hipMemset(s.d_state, 0, sizeof(gsim::QueryState))); // the kernels keep it zero between queries
    GSIM_HIP(hipMalloc(&s.d_pub, gsim::fused_pub_bytes(s.fgeo.nwaves / 4)));
    GSIM_HIP(hipMalloc(&s.d_hdr, gsim::fused_hdr_bytes(s.fgeo.nwaves / 4)));
    GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_summ), kSummBytes));
    GSIM_HIP(hipMemset(s.d_summ, 0, kSummBytes));
    GSIM_HIP(hipHostMalloc(reinterpret_cast<void**>(&s.h_done), 64, kHostPolled));
    std::memset(s.h_done, 0, 64);
    GSIM_HIP(hipHostMalloc(&s.h_query, static_cast<size_t>(s.W) * 4 * kQueryRing, hipHostMallocDefault));
    for (int i = 0; i < kQueryRing; i++) {
        hipEvent_t e;
        GSIM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s.q_ev.push_back(e);
    }
    GSIM_HIP(hipHostMalloc(&s.h_state, sizeof(gsim::QueryState), hipHostMallocDefault));
    return GSIM_OK;
}

// Scratch of the four-kernel pipeline (per-wave candidate segments + finalists: the worst case is
// every row a candidate, 24 B per row).  The single-launch path keeps its candidates in LDS and
// needs none of it, so it is only allocated when a query takes the classic pipeline: k above
// kFusedMaxK, widths without a specialised scan, or a query the single-launch path handed back
// (heavy ties, adversarial row orders).
int ensure_classic_scratch(Shard& s)
{
    if (s.classic_ready) return GSIM_OK;
    GSIM_HIP(set_device(s.device));
    const uint64_t slots = static_cast<uint64_t>(s.geo.nwaves) * s.geo.seg_cap;
    GSIM_HIP(hipMalloc(&s.d_cand, static_cast<size_t>(slots) * 8));
    GSIM_HIP(hipMalloc(&s.d_cand_cb, static_cast<size_t>(slots) * 4));
    GSIM_HIP(hipMalloc(&s.d_seg_count, static_cast<size_t>(s.geo.nwaves) * 4));
    s.final_cap = next_pow2_u32(slots);
    GSIM_HIP(hipMalloc(&s.d_final, static_cast<size_t>(s.final_cap) * 8));
    GSIM_HIP(hipMalloc(&s.d_final_cb, static_cast<size_t>(s.final_cap) * 4));
    s.classic_ready = true;
    return GSIM_OK;
}

int ensure_result_capacity(Shard& s, uint32_t k)
{
    const size_t need = gsim_result_block_bytes(k);
    if (need > s.result_bytes) {
        GSIM_HIP(set_device(s.device));
        if (s.d_result) GSIM_HIP(hipFree(s.d_result));
        s.d_result = nullptr;
        GSIM_HIP(hipMalloc(&s.d_result, need));
        s.result_bytes = need;
    }
    if (need > s.h_result_bytes) {
        if (s.h_result) GSIM_HIP(hipHostFree(s.h_result));
        s.h_result = nullptr;
        GSIM_HIP(hipHostMalloc(&s.h_result, need, kHostPolled));
        s.h_result_bytes = need;
    }
    return GSIM_OK;
}

uint32_t popcount_words(const uint32_t* q, uint32_t W)
{
    uint32_t a = 0;
    for (uint32_t i = 0; i < W; i++) a += static_cast<uint32_t>(__builtin_popcount(q[i]));
    return a;
}

enum QueryMode { kAuto = 0, kClassic = 1 };

bool fused_applies(const Shard& s, uint32_t k)
{
    static const int enabled = env_int("GSIM_FUSED", 1);
    static const long long max_rows = std::getenv("GSIM_FUSED_MAX_ROWS") ? std::atoll(std::getenv("GSIM_FUSED_MAX_ROWS")) : -1;
    if (!enabled || k == 0 || k > gsim::kFusedMaxK || s.nrows == 0 || !gsim::fused_supported(s.fgeo)) return false;
    // thresholds need >= k summary keys; without them every row is published (tiny tables only)
    if ((gsim::fused_summary_keys(s.fgeo.nwaves, k) == 0 || gsim::fused_final_keys(s.fgeo.nwaves / 4, k) == 0) && s.nrows > 8192) return false;
    return max_rows < 0 || s.nrows <= static_cast<uint64_t>(max_rows);
}

// Enqueue one query on one shard; the result block ends up at `out`, which is
// device memory or device-visible pinned host memory (zero-copy).
//
// Single-launch path (fused_applies): ONE kernel does scan + publish + select.  Synchronous
// callers (caller_syncs) get the query's epoch stored into s.h_done when the block is complete
// and check header flag 2 ("handed back": re-run with mode kClassic).  Enqueue-only callers
// (the RCCL path) get the four classic kernels enqueued behind it, gated on QueryState::redo:
// they return at once unless the single launch handed the query back.
//
// Classic path: sample -> scan -> compact -> select.  The query is read by the kernels straight
// from a pinned ring slot (no upload op) and the last kernel re-zeroes the per-query state (no
// memset op).  Nothing here synchronises with the host, whatever k.
int enqueue_query_impl(gsim_db* db, Shard& s, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha,
                       float beta, uint32_t row_base, void* out, bool caller_syncs, QueryMode mode, uint32_t pipe_slot)
{
    GSIM_HIP(set_device(s.device));
    if (s.state_dirty) { // a previous enqueue failed half way: the per-query state may not be zero
        GSIM_HIP(hipMemsetAsync(s.d_state, 0, offsetof(gsim::QueryState, ncand_sum), s.stream));
        GSIM_HIP(hipMemsetAsync(s.d_summ, 0, kSummBytes, s.stream));
        s.state_dirty = false;
    }
    bool fused = mode == kAuto && fused_applies(s, k);
    if (fused && caller_syncs && s.fused_skip) {
        s.fused_skip--;
        fused = false;
    }
    const bool classic = !fused || !caller_syncs;
    if (classic) {
        const int rc = ensure_classic_scratch(s);
        if (rc != GSIM_OK) return rc;
    }
    const uint32_t slot = s.q_next++ % kQueryRing;
    uint32_t* hq = s.h_query + static_cast<size_t>(slot) * s.W;
    if (s.q_pending[slot]) { // only set by asynchronous searches
        GSIM_HIP(hipEventSynchronize(s.q_ev[slot]));
        s.q_pending[slot] = false;
    }
    std::memcpy(hq, query, static_cast<size_t>(s.W) * 4); // `query` is already folded for a folded table

    gsim::ScanArgs a{};
    a.rows = s.d_rows;
    a.nrows = s.nrows;
    a.W = s.W;
    a.query = hq; // hipHostMalloc memory: device-visible at the same address
    a.query_dev = s.d_query;
    a.qpop = popcount_words(query, s.W);
    a.k = k;
    a.cutoff = cutoff;
    a.metric = metric;
    a.alpha = alpha;
    a.beta = beta;
    a.cand = s.d_cand;
    a.cand_cb = s.d_cand_cb;
    a.seg_count = s.d_seg_count;
    a.state = s.d_state;
    a.gate = nullptr;
    if (s.geo.lanes_per_row == 0 || s.nrows == 0) {
        // generic-width scan reads the query per word: give it a device copy
        GSIM_HIP(hipMemcpyAsync(s.d_query, hq, static_cast<size_t>(s.W) * 4, hipMemcpyHostToDevice, s.stream));
        a.query = s.d_query;
    }

    hipEvent_t* ev = nullptr;
    if (db->timing && s.ev_used < kTimingRing) {
        if (s.ev.size() < static_cast<size_t>(3 * (s.ev_used + 1))) {
            for (int i = 0; i < 3; i++) {
                hipEvent_t e;
                GSIM_HIP(hipEventCreate(&e));
                s.ev.push_back(e);
            }
        }
        ev = &s.ev[3 * s.ev_used];
    }
    if (caller_syncs) s.slot_fused[pipe_slot] = false;
    if (fused) {
        gsim::FusedArgs f{};
        f.pub = s.d_pub;
        f.hdr = s.d_hdr;
        f.arrive = s.d_summ + 4096 + kTicketWords;
        f.summ = s.d_summ;
        f.summ_keys = gsim::fused_summary_keys(s.fgeo.nwaves, k);
        f.final_keys = gsim::fused_final_keys(s.fgeo.nwaves / 4, k);
        f.tickets = s.d_summ + 4096;
        f.result = out;
        f.row_base = row_base;
        // Synchronous callers poll the result block's own header: the closing workgroup stores {count, flags | epoch << 8,
        // approx} in ONE 16-byte write when the hits are out (a separate completion word meant waiting for the header's
        // acknowledgement over PCIe first: ~1.3 us per query); finish_query_sync clears the epoch bits again.
        f.done_flag = caller_syncs ? s.h_done + pipe_slot : nullptr; // (non-null = "the caller polls the header")
        f.epoch = ++s.epoch & 0xFFFFFFu;
        if (f.epoch == 0) f.epoch = ++s.epoch & 0xFFFFFFu; // 0: what a clean header holds
        if (caller_syncs) static_cast<gsim_result_header*>(out)->flags = 0;
        static const int dbg_on = env_int("GSIM_FUSED_DEBUG", 0);
        if (dbg_on && !s.d_dbg) GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_dbg), (static_cast<size_t>(s.fgeo.nwaves / 4) * 24 + 8) * 8));
        f.dbg = s.d_dbg;
        // grid-wide waits give up after 2 ms + four scan times at 4 TB/s (only reached when the GPU is shared)
        f.wait_ticks = static_cast<uint32_t>(std::min<uint64_t>(200000ull + static_cast<uint64_t>(s.nrows) * s.W * 4 / 10000ull, 0xFFFFFFFFull));
        static const int xflags = env_int("GSIM_FUSED_FLAGS", 0);
        f.xflags = static_cast<uint32_t>(xflags);
        if (ev) GSIM_HIP(hipEventRecord(ev[0], s.stream));
        GSIM_HIP(gsim::launch_fused(a, s.fgeo, f, s.stream));
        if (ev) GSIM_HIP(hipEventRecord(ev[1], s.stream));
        if (caller_syncs) {
            s.slot_fused[pipe_slot] = true;
            s.slot_epoch[pipe_slot] = f.epoch;
            if (ev) {
                GSIM_HIP(hipEventRecord(ev[2], s.stream));
                s.ev_used++;
            }
            return GSIM_OK;
        }
        a.gate = &s.d_state->redo; // the classic kernels behind it run only if it handed the query back
    }
    if (s.nrows > 0 && s.sample_chunks > 0)
        GSIM_HIP(gsim::launch_sample(a, s.geo, static_cast<uint32_t>(s.sample_chunks), s.stream));
    if (ev && !fused) GSIM_HIP(hipEventRecord(ev[0], s.stream));
    if (s.nrows > 0) GSIM_HIP(gsim::launch_scan(a, s.geo, s.stream));
    if (!caller_syncs) { // the ring slot is free once the scan has run
        GSIM_HIP(hipEventRecord(s.q_ev[slot], s.stream));
        s.q_pending[slot] = true;
    }
    if (ev && !fused) GSIM_HIP(hipEventRecord(ev[1], s.stream));
    if (s.nrows > 0) GSIM_HIP(gsim::launch_compact(a, s.geo, s.d_final, s.d_final_cb, s.final_cap, s.stream));
    if (k <= static_cast<uint32_t>(gsim::kSelectCap)) {
        GSIM_HIP(gsim::launch_select(a, s.d_final, s.d_final_cb, s.final_cap, row_base, out, s.stream));
    } else {
        // large k: the k-th largest finalist key by a radix select on the device (the finalist count never reaches the
        // host: nothing here waits), the keys at or above it gathered and sorted in global memory (sized by k)
        const uint32_t np2 = next_pow2_u32(k);
        if (np2 > s.large_cap) {
            if (s.d_large) GSIM_HIP(hipFree(s.d_large));
            s.d_large = nullptr;
            s.large_cap = 0;
            GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_large), static_cast<size_t>(np2) * 8));
            s.large_cap = np2;
        }
        if (!s.d_lk) {
            GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_lk), sizeof(gsim::LargeKState)));
            GSIM_HIP(hipMemsetAsync(s.d_lk, 0, sizeof(gsim::LargeKState), s.stream));
        }
        GSIM_HIP(hipMemsetAsync(s.d_large, 0, static_cast<size_t>(np2) * 8, s.stream));
        GSIM_HIP(gsim::launch_largek_select(a, s.d_final, s.final_cap, s.d_lk, s.d_large, np2, s.stream));
        GSIM_HIP(gsim::launch_bitonic_global(s.d_large, np2, s.stream));
        GSIM_HIP(gsim::launch_emit_hits(a, s.d_large, s.d_lk, row_base, s.nrows, 1u, out, s.stream));
        GSIM_HIP(gsim::launch_reset_state(s.d_state, s.d_lk, s.stream));
    }
    if (ev) {
        GSIM_HIP(hipEventRecord(ev[2], s.stream));
        s.ev_used++;
    }
    return GSIM_OK;
}

int enqueue_query(gsim_db* db, Shard& s, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha,
                  float beta, uint32_t row_base, void* out, bool caller_syncs, QueryMode mode = kAuto, uint32_t pipe_slot = 0)
{
    const int rc = enqueue_query_impl(db, s, query, k, cutoff, metric, alpha, beta, row_base, out, caller_syncs, mode, pipe_slot);
    if (rc != GSIM_OK) s.state_dirty = true;
    return rc;
}

// Fold the recorded events of a shard into the handle's accumulators.
int drain_timing(gsim_db* db, Shard& s)
{
    if (s.ev_used == 0 && s.bev_used == 0) return GSIM_OK;
    GSIM_HIP(set_device(s.device));
    GSIM_HIP(hipStreamSynchronize(s.stream));
    for (uint32_t i = 0; i < s.bev_used; i++) {
        float ms = 0.f;
        GSIM_HIP(hipEventElapsedTime(&ms, s.bev[2 * i], s.bev[2 * i + 1]));
        db->acc.batch_kernel_ms_sum += ms;
        db->acc.batches++;
    }
    s.bev_used = 0;
    for (uint32_t i = 0; i < s.ev_used; i++) {
        float scan = 0.f, sel = 0.f;
        GSIM_HIP(hipEventElapsedTime(&scan, s.ev[3 * i], s.ev[3 * i + 1]));
        GSIM_HIP(hipEventElapsedTime(&sel, s.ev[3 * i + 1], s.ev[3 * i + 2]));
        db->acc.scan_ms_sum += scan;
        db->acc.select_ms_sum += sel;
        db->acc.queries++;
    }
    s.ev_used = 0;
    return GSIM_OK;
}

// Running candidate / finalist totals kept on the device by the select kernel.
int read_totals(Shard& s, unsigned long long* ncand, unsigned long long* nfinal, unsigned long long* nredo = nullptr)
{
    GSIM_HIP(set_device(s.device));
    GSIM_HIP(hipMemcpyAsync(s.h_state, s.d_state, sizeof(gsim::QueryState), hipMemcpyDeviceToHost, s.stream));
    GSIM_HIP(hipStreamSynchronize(s.stream));
    *ncand = s.h_state->ncand_sum;
    *nfinal = s.h_state->nfinal_sum;
    if (nredo) *nredo = s.h_state->redo_sum;
    return GSIM_OK;
}

// Wait for a stream: poll for a short while (a query takes ~2 ms and the blocking
// wait's interrupt wake-up costs 10-20 us), then block.
int wait_stream(hipStream_t st)
{
    for (int i = 0; i < 200000; i++) {
        hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return GSIM_OK;
        if (e != hipErrorNotReady) return fail_hip(e, "hipStreamQuery");
    }
    GSIM_HIP(hipStreamSynchronize(st));
    return GSIM_OK;
}

// Wait for the result block of the last synchronous enqueue on `s` (it was given s.h_result or any
// pinned block `out`).  The single-launch path signals through the pinned epoch word -- the block
// is complete when it changes, a few microseconds before the stream reports the kernel retired;
// a query it handed back (header flag 2) is re-run by the classic kernels here.
int finish_query_sync(gsim_db* db, Shard& s, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha,
                      float beta, uint32_t row_base, void* out, uint32_t pipe_slot = 0)
{
    if (!s.slot_fused[pipe_slot]) return wait_stream(s.stream);
    s.slot_fused[pipe_slot] = false;
    volatile uint32_t* flag = &static_cast<gsim_result_header*>(out)->flags; // (flags | epoch << 8: one 16-byte store with the rest of the header)
    const uint32_t want = s.slot_epoch[pipe_slot];
    bool done = false;
    for (uint64_t spins = 0;; spins++) {
        if ((*flag >> 8) == want) {
            done = true;
            break;
        }
        if ((spins & 0x3FFu) == 0x3FFu) { // now and then: did the launch fail or end without the header?
            const hipError_t e = hipStreamQuery(s.stream);
            if (e == hipSuccess) {
                done = (*flag >> 8) == want;
                break;
            }
            if (e != hipErrorNotReady) return fail_hip(e, "hipStreamQuery");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (done) *flag &= 0xFFu; // the header as every other route leaves it
    const gsim_result_header* h = static_cast<const gsim_result_header*>(out);
    if (s.d_dbg && done) { // phase profile of this query (instrumented runs only)
        const size_t nwg = s.fgeo.nwaves / 4;
        std::vector<unsigned long long> t(nwg * 24 + 8);
        (void) hipStreamSynchronize(s.stream);
        (void) hipMemcpy(t.data(), s.d_dbg, t.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (size_t g = 0; g < nwg; g++) t0 = std::min(t0, t[g * 24]);
        auto stat = [&](int slot, int nslots, double* mn, double* av, double* mx) {
            double lo = 1e30, hi = 0, sum = 0;
            size_t n = 0;
            for (size_t g = 0; g < nwg; g++)
                for (int q = 0; q < nslots; q++) {
                    const unsigned long long v = t[g * 24 + slot + q];
                    if (v < t0 || v - t0 > 100000000ull) continue;
                    const double us = (v - t0) / 100.0;
                    lo = std::min(lo, us), hi = std::max(hi, us), sum += us, n++;
                }
            *mn = n ? lo : 0, *mx = hi, *av = n ? sum / n : 0;
        };
        double a, b, c;
        { // what the workgroups published: the headers stay as the query left them
            std::vector<uint32_t> hd(nwg * 4);
            (void) hipMemcpy(hd.data(), s.d_hdr, hd.size() * 4, hipMemcpyDeviceToHost);
            unsigned long long rows = 0;
            uint32_t unsorted = 0, most = 0;
            for (size_t g = 0; g < nwg; g++) {
                const uint32_t n = hd[4 * g] & 0x7FFFFFFFu;
                rows += n, most = std::max(most, n), unsorted += (hd[4 * g] >> 31) ? 0u : 1u;
            }
            std::fprintf(stderr, "published: %llu rows by %zu workgroups (most: %u; %u lists not in order)\n", rows, nwg, most, unsorted);
        }
        std::fprintf(stderr, "fused phases, us after the first workgroup started (min/avg/max over workgroups):\n");
        const char* names[] = {"start", "scan-end(w0)", "compacted", "published", "sel:all-arrived", "sel:filtered", "sel:ranked", "sel:fenced",
                               "tau-first-seen", "ckpt0-done", "elect-start", "elect-end"};
        for (int i = 0; i < 12; i++) {
            stat(i, 1, &a, &b, &c);
            std::fprintf(stderr, "  %-16s %8.2f %8.2f %8.2f\n", names[i], a, b, c);
        }
        stat(16, 1, &a, &b, &c);
        std::fprintf(stderr, "  %-16s %8.2f %8.2f %8.2f\n", "elect-loaded", a, b, c);
        stat(17, 1, &a, &b, &c);
        std::fprintf(stderr, "  %-16s %8.2f %8.2f %8.2f\n", "3/4-ckpt(w0)", a, b, c);
        stat(23, 1, &a, &b, &c);
        std::fprintf(stderr, "  %-16s %8.2f %8.2f %8.2f\n", "sel:elected", a, b, c);
        stat(22, 1, &a, &b, &c);
        if (c > 0) std::fprintf(stderr, "  %-16s %8.2f %8.2f %8.2f\n", "sel:elected-2nd", a, b, c);
        stat(12, 4, &a, &b, &c);
        std::fprintf(stderr, "  %-16s %8.2f %8.2f %8.2f\n  end %.2f\n", "wave scan-end", a, b, c, (t[nwg * 24] - t0) / 100.0);
        { // streaming end per workgroup class: blockIdx % 8 (the XCD a block lands on) and blockIdx / 32 (dispatch order)
            double sx[8] = {}, sq[8] = {};
            int nx[8] = {}, nqd[8] = {};
            for (size_t g = 0; g < nwg; g++) {
                const unsigned long long v = t[g * 24 + 3];
                if (v < t0 || v - t0 > 100000000ull) continue;
                sx[g % 8] += (v - t0) / 100.0, nx[g % 8]++;
                const size_t oct = g * 8 / nwg;
                sq[oct] += (v - t0) / 100.0, nqd[oct]++;
            }
            for (int slot : {18, 19, 20, 21, 22, 17, 1}) { // checkpoints after 4 ... 1024 trips, the 3/4 checkpoint, the end of streaming
                double s8[8] = {};
                int n8[8] = {};
                for (size_t g = 0; g < nwg; g++) {
                    const unsigned long long v = t[g * 24 + slot];
                    if (v < t0 || v - t0 > 100000000ull) continue;
                    s8[g % 8] += (v - t0) / 100.0, n8[g % 8]++;
                }
                static const char* const what[] = {"4 trips", "16 trips", "64 trips", "256 trips", "1024 trips"};
                std::fprintf(stderr, "  %-14s mean by blockIdx %% 8:", slot == 17 ? "3/4 checkpoint" : slot == 1 ? "scan end" : what[slot - 18]);
                for (int i = 0; i < 8; i++) std::fprintf(stderr, " %7.1f", n8[i] ? s8[i] / n8[i] : 0.0);
                std::fprintf(stderr, "\n");
            }
            std::fprintf(stderr, "  arrived, mean by blockIdx %% 8:");
            for (int i = 0; i < 8; i++) std::fprintf(stderr, " %7.1f", nx[i] ? sx[i] / nx[i] : 0.0);
            std::fprintf(stderr, "\n  arrived, mean by blockIdx octile:");
            for (int i = 0; i < 8; i++) std::fprintf(stderr, " %7.1f", nqd[i] ? sq[i] / nqd[i] : 0.0);
            std::fprintf(stderr, "\n");
        }
        (void) hipMemset(s.d_dbg, 0, t.size() * 8);
    }
    if (done && !(h->flags & 2u)) {
        s.redo_streak = 0;
        return GSIM_OK;
    }
    if (done) {
        s.redo_streak = s.redo_streak < 6 ? s.redo_streak + 1 : 6;
        if (s.redo_streak >= 2) s.fused_skip = 1u << s.redo_streak;
    }
    if (!done) s.state_dirty = true; // the launch ended without closing the query: the state is re-zeroed
    // handed back: the per-query state is zero again (the last selector reset it), `redo` is set
    int rc = enqueue_query(db, s, query, k, cutoff, metric, alpha, beta, row_base, out, true, kClassic, pipe_slot);
    if (rc != GSIM_OK) return rc;
    return wait_stream(s.stream);
}

bool hit_before(const gsim_hit& x, const gsim_hit& y)
{
    if (x.score > y.score) return true;
    if (x.score < y.score) return false;
    return x.row < y.row;
}

// FingerprintDB::search's merge (fingerprintdb_cuda.cu:363-380: std::sort of all storages' results, first k kept).  The
// shards' lists arrive in canonical order and their keys are unique, so the first k of the sorted union are the first
// k of a k-way merge: O(k log #lists) instead of sorting #lists x k hits (8 x 1000: ~20 us instead of ~0.4 ms per query).
// `lists` holds the concatenated lists, `ends[i]` the end of list i in it.  Returns the number of hits written.
uint32_t merge_canonical_lists(const std::vector<gsim_hit>& lists, const std::vector<size_t>& ends, uint32_t k, gsim_hit* out)
{
    struct Head {
        size_t pos, end;
    };
    std::vector<Head> heads;
    size_t begin = 0;
    for (size_t e : ends) {
        if (e > begin) heads.push_back({begin, e});
        begin = e;
    }
    auto later = [&](const Head& x, const Head& y) { return hit_before(lists[y.pos], lists[x.pos]); }; // (a max-heap on "comes first")
    std::make_heap(heads.begin(), heads.end(), later);
    uint32_t n = 0;
    while (n < k && !heads.empty()) {
        std::pop_heap(heads.begin(), heads.end(), later);
        Head& h = heads.back();
        out[n++] = lists[h.pos++];
        if (h.pos < h.end) std::push_heap(heads.begin(), heads.end(), later);
        else heads.pop_back();
    }
    return n;
}

// FoldFingerprintFunctorCPU (calculation_functors.cpp:22-41): bit `pos` of the fingerprint is
// OR-ed into bit `pos % (32 * Wf)`; since 32 * Wf is a multiple of 32 that is word (w % Wf), same
// bit -- i.e. the F consecutive blocks of Wf words are OR-ed together.
void fold_row(const uint32_t* row, uint32_t W, uint32_t F, uint32_t* out)
{
    const uint32_t Wf = W / F;
    for (uint32_t j = 0; j < Wf; j++) out[j] = 0;
    for (uint32_t w = 0; w < W; w++) out[w % Wf] |= row[w];
}

// fold_data (fingerprintdb_cuda.cpp:56-69) over a row range, on all host threads
void fold_rows_mt(const uint32_t* rows, uint64_t nrows, uint32_t W, uint32_t F, uint32_t* out)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > nrows) nt = nrows ? static_cast<unsigned>(nrows) : 1;
    const uint32_t Wf = W / F;
    const uint64_t per = (nrows + nt - 1) / nt;
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; t++) {
        const uint64_t lo = per * t, hi = std::min<uint64_t>(lo + per, nrows);
        if (lo >= hi) break;
        pool.emplace_back([=] {
            for (uint64_t r = lo; r < hi; r++) fold_row(rows + r * W, W, F, out + r * Wf);
        });
    }
    for (auto& th : pool) th.join();
}

// memcpy on several host threads (a single thread moves ~7 GB/s here; table loading is startup
// time, not the hot path, but a 128 GB table should not take half a minute)
void parallel_memcpy(void* dst, const void* src, size_t bytes)
{
    const size_t kMin = size_t(8) << 20;
    unsigned nt = std::min<unsigned>(8, std::max<unsigned>(1, std::thread::hardware_concurrency()));
    if (bytes < 2 * kMin || nt < 2) {
        std::memcpy(dst, src, bytes);
        return;
    }
    nt = static_cast<unsigned>(std::min<size_t>(nt, bytes / kMin));
    const size_t per = (bytes / nt + 4095) & ~size_t(4095);
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nt; i++) {
        const size_t off = per * i;
        if (off >= bytes) break;
        const size_t n = std::min(per, bytes - off);
        th.emplace_back([=] { std::memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, n); });
    }
    for (auto& t : th) t.join();
}

// Pageable host memory -> device through two pinned staging buffers: the (threaded) copy into one
// buffer overlaps the DMA of the other (pageable hipMemcpy: 10 GB/s).
int upload_rows(void* d_dst, const void* h_src, size_t bytes, hipStream_t stream)
{
    const size_t kChunk = size_t(64) << 20;
    if (bytes <= kChunk) {
        GSIM_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
        return GSIM_OK;
    }
    void* stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    int rc = GSIM_OK;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipHostMalloc(&stage[i], kChunk, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    }
    size_t off = 0;
    for (int i = 0; e == hipSuccess && off < bytes; i ^= 1) {
        const size_t n = std::min(kChunk, bytes - off);
        e = hipEventSynchronize(done[i]); // the previous DMA out of this buffer (no-op the first time)
        if (e != hipSuccess) break;
        parallel_memcpy(stage[i], static_cast<const char*>(h_src) + off, n);
        e = hipMemcpyAsync(static_cast<char*>(d_dst) + off, stage[i], n, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipEventRecord(done[i], stream);
        off += n;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) rc = fail_hip(e, "upload_rows");
    for (int i = 0; i < 2; i++) {
        if (done[i]) (void) hipEventDestroy(done[i]);
        if (stage[i]) (void) hipHostFree(stage[i]);
    }
    return rc;
}


std::mutex g_rr_mutex;
int g_next_device = 0;

constexpr uint32_t kBatchMaxQ = 256; // queries per batch call on a shard (larger requests are split)

// Buffers of the multi-query path, sized for kBatchMaxQ queries and result blocks of k hits.
int ensure_batch_buffers(gsim_db* db, Shard& s, uint32_t k)
{
    GSIM_HIP(set_device(s.device));
    if (s.bq_cap == 0) {
        const int wpc = env_int("GSIM_BATCH_WAVES_PER_CU", 12);
        s.bgeo = gsim::scan_geometry(s.nrows, s.W, s.num_cus, wpc, 8);
        const uint64_t nchunks = (s.nrows + 63) / 64;
        uint64_t nw = static_cast<uint64_t>(s.num_cus) * static_cast<uint64_t>(wpc);
        if (nw > nchunks) nw = nchunks ? nchunks : 1;
        nw = (nw + 3) / 4 * 4;
        s.bgeo.nwaves = static_cast<uint32_t>(nw);
        // candidate slots per wave: the worst case (every pair a candidate) when that is small,
        // else 64 Ki entries; a wave that needs more sets the overflow flag and the host falls back
        const uint64_t rows_per_wave = ((nchunks + nw - 1) / nw) * 64 + 256; // chunks are 64 x (1..4) rows
        uint64_t cap = rows_per_wave * gsim::kBQ;
        // a wave of the matrix-core pass meets 32 queries and every row of its workgroup
        const uint64_t mfma_waves = gsim::batch_mfma_waves(s.num_cus);
        if (gsim::batch_mfma_supported(s.W)) {
            cap = std::max<uint64_t>(cap, (s.nrows / static_cast<uint64_t>(s.num_cus) + 512) * 32);
            nw = std::max<uint64_t>(nw, mfma_waves);
        }
        const uint64_t lim = static_cast<uint64_t>(env_int("GSIM_BATCH_SEG_CAP", 65536));
        if (cap > lim) cap = lim;
        if (cap < 256) cap = 256;
        s.bseg_cap = static_cast<uint32_t>(cap);
        const size_t slots = static_cast<size_t>(nw) * cap;
        GSIM_HIP(hipMalloc(&s.d_bqueries, static_cast<size_t>(kBatchMaxQ) * s.W * 4));
        GSIM_HIP(hipMalloc(&s.d_bqpop, kBatchMaxQ * 4));
        GSIM_HIP(hipMalloc(&s.d_bstate, sizeof(gsim::BatchQueryState) * kBatchMaxQ));
        GSIM_HIP(hipMalloc(&s.d_bcand, slots * 8));
        GSIM_HIP(hipMalloc(&s.d_bcand_cb, slots * 4));
        GSIM_HIP(hipMalloc(&s.d_bcand_q, slots * 4));
        GSIM_HIP(hipMalloc(&s.d_bseg_count, nw * 4));
        GSIM_HIP(hipMalloc(&s.d_bfin_key, static_cast<size_t>(kBatchMaxQ) * gsim::kSelectCap * 8));
        GSIM_HIP(hipMalloc(&s.d_bfin_cb, static_cast<size_t>(kBatchMaxQ) * gsim::kSelectCap * 4));
        GSIM_HIP(hipMalloc(&s.d_bflags, 64));
        GSIM_HIP(hipMalloc(&s.d_brare, sizeof(gsim::BatchRare)));
        GSIM_HIP(hipHostMalloc(&s.h_brare, sizeof(gsim::BatchRare), hipHostMallocDefault));
        GSIM_HIP(hipHostMalloc(&s.h_bflags, 64, hipHostMallocDefault));
        GSIM_HIP(hipHostMalloc(&s.h_bqueries, static_cast<size_t>(kBatchMaxQ) * (s.W + 1) * 4, hipHostMallocDefault));
        s.bq_cap = kBatchMaxQ;
    }
    const size_t need = gsim_result_block_bytes(k) * kBatchMaxQ;
    if (need > s.h_bresult_bytes) {
        if (s.h_bresult) GSIM_HIP(hipHostFree(s.h_bresult));
        if (s.d_bresult) GSIM_HIP(hipFree(s.d_bresult));
        s.h_bresult = nullptr;
        s.d_bresult = nullptr;
        GSIM_HIP(hipHostMalloc(&s.h_bresult, need, hipHostMallocDefault));
        GSIM_HIP(hipMalloc(&s.d_bresult, need));
        s.h_bresult_bytes = need;
    }
    return GSIM_OK;
}

// Enqueue nq (<= kBatchMaxQ) queries on one shard: ceil(nq / kBQ) passes over the table (one on the
// matrix cores), the result blocks land in `results` (device memory; NULL = the shard's pinned host
// block array s.h_bresult, through s.d_bresult).  No host synchronisation.
int enqueue_batch(gsim_db* db, Shard& s, const uint32_t* queries, uint32_t nq, uint32_t k, float cutoff, int metric,
                  float alpha, float beta, uint32_t row_base, void* results, bool allow_mfma = true)
{
    int rc = ensure_batch_buffers(db, s, k);
    if (rc != GSIM_OK) return rc;
    // NULL: blocks go to device memory and then, in one copy, to the pinned array (256 blocks of
    // 12 KB written by the select kernel straight over PCIe cost ~5 ms per batch)
    const bool to_host = results == nullptr;
    if (to_host) results = s.d_bresult; // (allocated or grown just above)
    GSIM_HIP(set_device(s.device));
    const size_t qbytes = static_cast<size_t>(nq) * s.W * 4;
    std::memcpy(s.h_bqueries, queries, qbytes);
    uint32_t* hp = s.h_bqueries + static_cast<size_t>(nq) * s.W;
    for (uint32_t q = 0; q < nq; q++) hp[q] = popcount_words(queries + static_cast<size_t>(q) * s.W, s.W);
    GSIM_HIP(hipMemcpyAsync(s.d_bqueries, s.h_bqueries, qbytes, hipMemcpyHostToDevice, s.stream));
    GSIM_HIP(hipMemcpyAsync(s.d_bqpop, hp, static_cast<size_t>(nq) * 4, hipMemcpyHostToDevice, s.stream));
    GSIM_HIP(hipMemsetAsync(s.d_bstate, 0, sizeof(gsim::BatchQueryState) * nq, s.stream));
    GSIM_HIP(hipMemsetAsync(s.d_bflags, 0, 64, s.stream));
    gsim::BatchRare& rr = *s.h_brare;
    rr.qstate = s.d_bstate;
    rr.cand = s.d_bcand;
    rr.cand_cb = s.d_bcand_cb;
    rr.cand_q = s.d_bcand_q;
    rr.seg_count = s.d_bseg_count;
    rr.fin_key = s.d_bfin_key;
    rr.fin_cb = s.d_bfin_cb;
    rr.flags = s.d_bflags;
    rr.ticket = s.d_bflags + 1;
    rr.seg_cap = s.bseg_cap;
    rr.pad = 0;
    GSIM_HIP(hipMemcpyAsync(s.d_brare, s.h_brare, sizeof(gsim::BatchRare), hipMemcpyHostToDevice, s.stream));
    gsim::BatchArgs a{};
    a.rows = s.d_rows;
    a.nrows = s.nrows;
    a.W = s.W;
    a.queries = s.d_bqueries;
    a.qpop = s.d_bqpop;
    a.rare = s.d_brare;
    a.k = k;
    a.cutoff = cutoff;
    a.metric = metric;
    a.alpha = alpha;
    a.beta = beta;
    const uint32_t sample = static_cast<uint32_t>(env_int("GSIM_BATCH_SAMPLE_CHUNKS", 8));
    // One contraction pass on the matrix cores for all of them (gsim_batch_mfma.hip: with fewer
    // than 8 x 32 queries the waves of a workgroup share query tiles and split the rows).  With a
    // cutoff it needs the matrix-core sample pass (large tables), which also estimates how many
    // rows the cutoff keeps: a cutoff that keeps many sets bit 3 of the flags and the kernel leaves
    // the batch to the VALU pass (the callers re-enqueue with allow_mfma = false).
    hipEvent_t* bev = nullptr;
    if (db->timing && s.bev_used < kTimingRing) {
        if (s.bev.size() < static_cast<size_t>(2 * (s.bev_used + 1))) {
            for (int i = 0; i < 2; i++) {
                hipEvent_t e;
                GSIM_HIP(hipEventCreate(&e));
                s.bev.push_back(e);
            }
        }
        bev = &s.bev[2 * s.bev_used];
        s.bev_used++;
    }
    static const int mfma_min_q = env_int("GSIM_BATCH_MFMA_MIN_Q", 4);
    if (allow_mfma && mfma_min_q > 0 && nq >= static_cast<uint32_t>(mfma_min_q) &&
        nq <= static_cast<uint32_t>(gsim::kMfmaQueries) && gsim::batch_mfma_supported(s.W) &&
        (!(cutoff > 0.0f) || gsim::batch_mfma_sample_applies(s.W, s.nrows, nq, k, s.num_cus))) {
        a.q0 = 0;
        a.nq = nq;
        // the rows' popcounts: once per table; borrowed rows (gsim_db_attach_device_rows) may have changed since the
        // last call, so theirs are recounted every time (one more read of the table)
        if (!s.d_rowpop) GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_rowpop), gsim::row_popcount_bytes(s.nrows)));
        if (!s.rowpop_valid || !s.owns_rows) {
            GSIM_HIP(gsim::launch_row_popcounts(s.d_rows, s.nrows, s.W, s.d_rowpop, s.stream));
            s.rowpop_valid = true;
        }
        a.rowpop = s.d_rowpop;
        GSIM_HIP(gsim::launch_batch_mfma_pass(a, s.bgeo, s.num_cus, sample, row_base, results,
                                              gsim_result_block_bytes(k), s.stream, bev ? bev[0] : nullptr,
                                              bev ? bev[1] : nullptr));
        if (to_host)
            GSIM_HIP(hipMemcpyAsync(s.h_bresult, s.d_bresult, gsim_result_block_bytes(k) * nq, hipMemcpyDeviceToHost, s.stream));
        GSIM_HIP(hipMemcpyAsync(s.h_bflags, s.d_bflags, 64, hipMemcpyDeviceToHost, s.stream));
        return GSIM_OK;
    }
    if (bev) GSIM_HIP(hipEventRecord(bev[0], s.stream));
    for (uint32_t q0 = 0; q0 < nq; q0 += gsim::kBQ) {
        a.q0 = q0;
        a.nq = std::min<uint32_t>(gsim::kBQ, nq - q0);
        GSIM_HIP(gsim::launch_batch_pass(a, rr, s.bgeo, sample, row_base, results, gsim_result_block_bytes(k),
                                         s.stream));
    }
    if (bev) GSIM_HIP(hipEventRecord(bev[1], s.stream));
    if (to_host)
        GSIM_HIP(hipMemcpyAsync(s.h_bresult, s.d_bresult, gsim_result_block_bytes(k) * nq, hipMemcpyDeviceToHost, s.stream));
    GSIM_HIP(hipMemcpyAsync(s.h_bflags, s.d_bflags, 64, hipMemcpyDeviceToHost, s.stream));
    return GSIM_OK;
}

// One query through the single-query pipeline on every shard, host merge across shards
// (FingerprintDB::search, fingerprintdb_cuda.cu:341-381).
int search_one(gsim_db* db, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha, float beta,
               gsim_hit* hits, uint32_t* count, uint64_t* approx, std::vector<gsim_hit>& merged)
{
    const size_t nsh = db->shards.size();
    for (auto& s : db->shards) {
        int rc = ensure_result_capacity(s, k);
        if (rc != GSIM_OK) return rc;
        // the select kernel writes the block straight into pinned host memory
        rc = enqueue_query(db, s, query, k, cutoff, metric, alpha, beta,
                           db->row_base + static_cast<uint32_t>(s.first_row), s.h_result, true);
        if (rc != GSIM_OK) return rc;
    }
    uint64_t ap = 0;
    merged.clear();
    std::vector<size_t> ends;
    for (auto& s : db->shards) {
        GSIM_HIP(set_device(s.device));
        int rc = finish_query_sync(db, s, query, k, cutoff, metric, alpha, beta,
                                   db->row_base + static_cast<uint32_t>(s.first_row), s.h_result);
        if (rc != GSIM_OK) return rc;
        const gsim_result_header* h = reinterpret_cast<const gsim_result_header*>(s.h_result);
        const gsim_hit* hh = reinterpret_cast<const gsim_hit*>(h + 1);
        ap += h->approx;
        if (nsh == 1) {
            std::memcpy(hits, hh, sizeof(gsim_hit) * h->count);
            *count = h->count;
        } else {
            merged.insert(merged.end(), hh, hh + h->count);
            ends.push_back(merged.size());
        }
    }
    if (nsh > 1) *count = merge_canonical_lists(merged, ends, k, hits); // fingerprintdb_cuda.cu:363-380
    if (approx) *approx = ap;
    return GSIM_OK;
}

// gsim_db_search_each on a single-shard handle: the queries still run strictly one after the other on the GPU (one
// stream, one per-query state), but up to kPipe of them are enqueued ahead of the one the host is waiting for, each with
// its own pinned result block and completion word -- the next kernel starts when the previous one retires instead of
// after a host round trip (flag seen, hits copied, next launch: ~8 us per query).
int search_each_pipelined(gsim_db* db, Shard& s, const uint32_t* queries, uint32_t nq, uint32_t k, uint32_t kout, float cutoff,
                          int metric, float alpha, float beta, gsim_hit* hits, uint32_t* counts, uint64_t* approx)
{
    GSIM_HIP(set_device(s.device));
    const size_t blk = gsim_result_block_bytes(k);
    if (blk > s.h_pipe_block) {
        if (s.h_pipe) GSIM_HIP(hipHostFree(s.h_pipe));
        s.h_pipe = nullptr;
        s.h_pipe_block = 0;
        GSIM_HIP(hipHostMalloc(reinterpret_cast<void**>(&s.h_pipe), blk * kPipe, kHostPolled));
        s.h_pipe_block = blk;
    }
    const uint32_t row_base = db->row_base + static_cast<uint32_t>(s.first_row);
    uint32_t issued = 0;
    for (uint32_t done = 0; done < nq; done++) {
        for (; issued < nq && issued - done < static_cast<uint32_t>(kPipe); issued++) {
            const int rc = enqueue_query(db, s, queries + static_cast<size_t>(issued) * db->W, k, cutoff, metric, alpha, beta, row_base,
                                         s.h_pipe + (issued % kPipe) * s.h_pipe_block, true, kAuto, issued % kPipe);
            if (rc != GSIM_OK) return rc;
        }
        void* out = s.h_pipe + (done % kPipe) * s.h_pipe_block;
        const int rc = finish_query_sync(db, s, queries + static_cast<size_t>(done) * db->W, k, cutoff, metric, alpha, beta, row_base,
                                         out, done % kPipe);
        if (rc != GSIM_OK) return rc;
        const gsim_result_header* h = static_cast<const gsim_result_header*>(out);
        std::memcpy(hits + static_cast<size_t>(done) * kout, h + 1, sizeof(gsim_hit) * h->count);
        counts[done] = h->count;
        if (approx) approx[done] = h->approx;
    }
    return GSIM_OK;
}

// Search of a folded table, fingerprintdb_cuda.cu:228-339 with m_fold_factor > 1, per storage:
// folded query vs folded rows on the GPU for the k*F*(int)log2(2F) best FOLDED scores (:284-287),
// re-score those with the full fingerprints on the host (:307-314), stable partial bubble sort
// (:315), keep min(k, .) and stop at the first re-scored value below the cutoff (:317-331);
// then FingerprintDB::search's merge over the storages (:363-380).
int search_folded(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k, float cutoff, gsim_hit* hits,
                  uint32_t* counts, uint64_t* approx)
{
    const uint32_t F = db->fold, W = db->W, Wf = W / F;
    int lg = 0;
    while ((1u << (lg + 1)) <= 2 * F) lg++;
    const uint64_t want = static_cast<uint64_t>(k) * F * static_cast<uint64_t>(lg);
    std::vector<uint32_t> fq(Wf);
    std::vector<gsim_hit> merged;
    std::vector<int> idx;
    std::vector<float> sc;
    // The re-score runs on the device when every storage also holds its full fingerprints in HBM (gsim_db_finalize puts
    // them there when they fit) and the candidate list fits the device sort; GSIM_FOLD_RESCORE=host forces the host path.
    static const bool force_host = std::getenv("GSIM_FOLD_RESCORE") && std::string(std::getenv("GSIM_FOLD_RESCORE")) == "host";
    bool on_device = !force_host && want <= 65536 && k > 0;
    for (auto& s : db->shards) on_device = on_device && s.d_full != nullptr;
    // the host path for one storage: its folded candidates (in s.h_result) re-scored with the host copy of the full rows
    auto rescore_on_host = [&](Shard& s, const uint32_t* query) {
        const gsim_result_header* h = reinterpret_cast<const gsim_result_header*>(s.h_result);
        const gsim_hit* hh = reinterpret_cast<const gsim_hit*>(h + 1);
        const uint32_t n = h->count;
        idx.resize(n);
        sc.resize(n);
        std::vector<uint16_t> cm(n), pc(n);
        for (uint32_t j = 0; j < n; j++) { // tanimoto_similarity_cpu on the FULL fingerprints (:387-399)
            const uint32_t* d = db->host_rows.data() + (s.first_row + hh[j].row) * W;
            int total = 0, common = 0, pd = 0;
            for (uint32_t w = 0; w < W; w++) {
                const int p2 = __builtin_popcount(d[w]);
                pd += p2;
                total += __builtin_popcount(query[w]) + p2;
                common += __builtin_popcount(query[w] & d[w]);
            }
            idx[j] = static_cast<int>(j);
            sc[j] = static_cast<float>(common) / static_cast<float>(total - common);
            cm[j] = static_cast<uint16_t>(common);
            pc[j] = static_cast<uint16_t>(pd);
        }
        // top_results_bubble_sort(indices, scores, k) (fingerprintdb_cuda.cpp:92-103): k passes of a bubble sort with
        // a strict '>' -- stable, so its first k entries are the first k of a stable descending sort.  That sort is
        // what runs here (O(n log n) instead of O(k n): k = 1000, F = 8 means 32 k candidates x 1000 passes per
        // storage and query); the literal bubble sort only when a NaN score (0/0: two empty fingerprints) is
        // present, for which '>' is not an order and the two would differ.
        bool has_nan = false;
        for (uint32_t j = 0; j < n; j++) has_nan = has_nan || sc[j] != sc[j];
        if (!has_nan) {
            std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return sc[x] > sc[y]; });
            std::vector<float> sorted(n);
            for (uint32_t j = 0; j < n; j++) sorted[j] = sc[idx[j]];
            sc.swap(sorted);
        } else {
            for (uint32_t a = 0; a < k && a < n; a++) {
                for (uint32_t b = n - 1; b > a; b--) {
                    if (sc[b] > sc[b - 1]) {
                        std::swap(idx[b], idx[b - 1]);
                        std::swap(sc[b], sc[b - 1]);
                    }
                }
            }
        }
        const uint32_t keep = std::min(k, n);
        for (uint32_t a = 0; a < keep; a++) {
            if (sc[a] < cutoff) break;
            gsim_hit o;
            o.row = db->row_base + static_cast<uint32_t>(s.first_row) + hh[idx[a]].row;
            o.score = sc[a];
            o.common = cm[idx[a]];
            o.popc_db = pc[idx[a]];
            merged.push_back(o);
        }
        return h->approx;
    };
    for (uint32_t q = 0; q < nq; q++) {
        const uint32_t* query = queries + static_cast<size_t>(q) * W;
        fold_row(query, W, F, fq.data());
        const uint32_t qa = popcount_words(query, W);
        std::vector<uint32_t> kshard(db->shards.size());
        for (size_t i = 0; i < db->shards.size(); i++) {
            Shard& s = db->shards[i];
            kshard[i] = static_cast<uint32_t>(std::min<uint64_t>(want, s.nrows));
            int rc = ensure_result_capacity(s, std::max(kshard[i], k));
            if (rc != GSIM_OK) return rc;
            if (!on_device) {
                rc = enqueue_query(db, s, fq.data(), kshard[i], cutoff, GSIM_METRIC_TANIMOTO, 0.f, 0.f, 0, s.h_result, true);
                if (rc != GSIM_OK) return rc;
                continue;
            }
            // device route, all enqueued on the storage's stream: folded search -> candidates' block in device memory ->
            // re-score with the full rows, sort, first k at or above the cutoff -> pinned host block
            GSIM_HIP(set_device(s.device));
            const uint32_t npad = next_pow2_u32(kshard[i] ? kshard[i] : 1);
            if (!s.d_fq) {
                GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_fq), static_cast<size_t>(W) * 4 + 64));
                GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_key2), static_cast<size_t>(65536) * 8));
                GSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_cb2), static_cast<size_t>(65536) * 4));
                GSIM_HIP(hipHostMalloc(reinterpret_cast<void**>(&s.h_fq), static_cast<size_t>(W) * 4 + 64, hipHostMallocDefault));
            }
            GSIM_HIP(hipStreamSynchronize(s.stream)); // (the pinned staging of the previous query's fingerprint is free)
            std::memcpy(s.h_fq, query, static_cast<size_t>(W) * 4);
            s.h_fq[W] = 0;
            GSIM_HIP(hipMemcpyAsync(s.d_fq, s.h_fq, static_cast<size_t>(W) * 4 + 4, hipMemcpyHostToDevice, s.stream)); // (+ the NaN flag, cleared)
            rc = enqueue_query(db, s, fq.data(), kshard[i], cutoff, GSIM_METRIC_TANIMOTO, 0.f, 0.f, 0, s.d_result, false);
            if (rc != GSIM_OK) return rc;
            GSIM_HIP(gsim::launch_fold_rescore(s.d_result, s.d_full, s.d_fq, W, qa, s.d_key2, s.d_cb2, npad, s.d_fq + W, k, cutoff,
                                               db->row_base + static_cast<uint32_t>(s.first_row), s.h_result, s.stream));
            GSIM_HIP(hipMemcpyAsync(s.h_fq + W, s.d_fq + W, 4, hipMemcpyDeviceToHost, s.stream));
        }
        uint64_t ap = 0;
        merged.clear();
        for (size_t i = 0; i < db->shards.size(); i++) {
            Shard& s = db->shards[i];
            GSIM_HIP(set_device(s.device));
            if (on_device) {
                int rc = wait_stream(s.stream);
                if (rc != GSIM_OK) return rc;
                if (s.h_fq[W] == 0) { // (no NaN among the re-scored candidates: the block in s.h_result is the storage's answer)
                    const gsim_result_header* h = reinterpret_cast<const gsim_result_header*>(s.h_result);
                    const gsim_hit* hh = reinterpret_cast<const gsim_hit*>(h + 1);
                    ap += h->approx;
                    merged.insert(merged.end(), hh, hh + h->count);
                    continue;
                }
                rc = enqueue_query(db, s, fq.data(), kshard[i], cutoff, GSIM_METRIC_TANIMOTO, 0.f, 0.f, 0, s.h_result, true);
                if (rc != GSIM_OK) return rc;
            }
            int rc = finish_query_sync(db, s, fq.data(), kshard[i], cutoff, GSIM_METRIC_TANIMOTO, 0.f, 0.f, 0, s.h_result);
            if (rc != GSIM_OK) return rc;
            ap += rescore_on_host(s, query);
        }
        if (db->shards.size() > 1) std::stable_sort(merged.begin(), merged.end(), hit_before);
        const uint32_t n = static_cast<uint32_t>(std::min<size_t>(merged.size(), k));
        if (n) std::memcpy(hits + static_cast<size_t>(q) * k, merged.data(), sizeof(gsim_hit) * n);
        counts[q] = n;
        if (approx) approx[q] = ap;
    }
    return GSIM_OK;
}

} // namespace

extern "C" {

const char* gsim_last_error(void)
{
    return g_last_error.c_str();
}

const char* gsim_version(void)
{
    return "gpusimilarity_amd 0.1 (gfx950)";
}

int gsim_device_count(int* count)
{
    if (!count) return fail(GSIM_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        // reference get_gpu_count() reports 0 GPUs rather than failing (fingerprintdb_cuda.cu:40-52)
        (void) hipGetLastError();
        n = 0;
    }
    if (n > 0 && alias_devices()) n = alias_devices(); // test hook, see alias_devices()
    *count = n;
    return GSIM_OK;
}

int gsim_device_free_bytes(int device, size_t* free_bytes)
{
    if (!free_bytes) return fail(GSIM_ERR_INVALID, "free_bytes is NULL");
    int n = 0;
    gsim_device_count(&n);
    if (device < 0 || device >= n) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
    GSIM_HIP(set_device(device));
    size_t fr = 0, tot = 0;
    GSIM_HIP(hipMemGetInfo(&fr, &tot));
    *free_bytes = fr;
    return GSIM_OK;
}

int gsim_available_device_bytes(size_t* total_free_bytes)
{
    if (!total_free_bytes) return fail(GSIM_ERR_INVALID, "total_free_bytes is NULL");
    int n = 0;
    gsim_device_count(&n);
    size_t sum = 0;
    for (int d = 0; d < n; d++) {
        size_t fr = 0;
        int rc = gsim_device_free_bytes(d, &fr);
        if (rc != GSIM_OK) return rc;
        sum += fr;
    }
    *total_free_bytes = sum;
    return GSIM_OK;
}

int gsim_next_device(size_t required_bytes, int* device)
{
    if (!device) return fail(GSIM_ERR_INVALID, "device is NULL");
    int n = 0;
    gsim_device_count(&n);
    if (n == 0) return fail(GSIM_ERR_NO_DEVICE, "no GPU available");
    std::lock_guard<std::mutex> lock(g_rr_mutex);
    for (int i = 0; i < n; i++) {
        const int d = g_next_device++ % n;
        size_t fr = 0;
        int rc = gsim_device_free_bytes(d, &fr);
        if (rc != GSIM_OK) return rc;
        if (fr > required_bytes) {
            *device = d;
            return GSIM_OK;
        }
    }
    return fail(GSIM_ERR_NOMEM, "Can't find a GPU with enough memory to copy data.");
}

int gsim_db_create(uint32_t fp_bits, gsim_db** out)
{
    if (!out) return fail(GSIM_ERR_INVALID, "out is NULL");
    if (fp_bits == 0 || fp_bits % 32 != 0 || fp_bits > 32768)
        return fail(GSIM_ERR_INVALID, "fp_bits must be a positive multiple of 32, <= 32768");
    gsim_db* db = new (std::nothrow) gsim_db;
    if (!db) return fail(GSIM_ERR_NOMEM, "out of host memory");
    db->fp_bits = fp_bits;
    db->W = fp_bits / 32;
    *out = db;
    return GSIM_OK;
}

int gsim_db_add_rows(gsim_db* db, const uint32_t* rows, uint64_t nrows)
{
    if (!db || (!rows && nrows)) return fail(GSIM_ERR_INVALID, "NULL argument");
    if (db->finalized) return fail(GSIM_ERR_STATE, "table already finalized");
    if (db->nrows + nrows > 0xFFFFFFFFull) return fail(GSIM_ERR_INVALID, "more than 2^32-1 rows");
    try {
        db->slice_first.push_back(db->nrows);
        const size_t old_words = db->host_rows.size();
        db->host_rows.resize(old_words + static_cast<size_t>(nrows) * db->W);
        parallel_memcpy(db->host_rows.data() + old_words, rows, static_cast<size_t>(nrows) * db->W * 4);
    } catch (const std::bad_alloc&) {
        return fail(GSIM_ERR_NOMEM, "out of host memory");
    }
    db->nrows += nrows;
    db->has_host_copy = true;
    return GSIM_OK;
}

int gsim_db_set_fold_factor(gsim_db* db, uint32_t fold_factor)
{
    if (!db) return fail(GSIM_ERR_INVALID, "db is NULL");
    if (db->finalized) return fail(GSIM_ERR_STATE, "table already finalized");
    if (fold_factor == 0 || fold_factor > db->W) return fail(GSIM_ERR_INVALID, "fold factor out of range");
    db->fold_requested = fold_factor;
    return GSIM_OK;
}

uint32_t gsim_db_fold_factor(const gsim_db* db)
{
    return db ? db->fold : 0;
}

int gsim_fold_fingerprint(const uint32_t* fingerprint, uint32_t words, uint32_t fold_factor, uint32_t* out)
{
    if (!fingerprint || !out) return fail(GSIM_ERR_INVALID, "NULL argument");
    if (fold_factor == 0 || words == 0 || words % fold_factor != 0)
        return fail(GSIM_ERR_INVALID, "fold factor must divide the word count");
    fold_row(fingerprint, words, fold_factor, out);
    return GSIM_OK;
}

int gsim_db_finalize(gsim_db* db, int device, int ndevices)
{
    if (!db) return fail(GSIM_ERR_INVALID, "db is NULL");
    if (db->finalized) return fail(GSIM_ERR_STATE, "table already finalized");
    int ndev = 0;
    gsim_device_count(&ndev);
    if (ndev == 0) return fail(GSIM_ERR_NO_DEVICE, "no GPU available");
    if (ndevices == 0) { // all devices from `device` on
        if (device < 0) device = 0;
        if (device >= ndev) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
        ndevices = ndev - device;
    }
    if (ndevices < 0) return fail(GSIM_ERR_INVALID, "ndevices < 0");
    if (device >= 0 && device + ndevices > ndev) return fail(GSIM_ERR_NO_DEVICE, "device range exceeds the GPUs present");
    // copyToGPU's factor adjustment, fingerprintdb_cuda.cu:170-173
    db->fold = db->fold_requested ? db->fold_requested : 1;
    while (db->W % db->fold != 0) db->fold++;
    if (db->fold > 1) {
        // Folded table (fingerprintdb_cuda.cu:184-194): every add_rows slice is one storage with
        // its own candidate list, placed round-robin like get_next_gpu (:54-68, :186-188).
        if (!db->has_host_copy) return fail(GSIM_ERR_STATE, "folding needs the host copy of the rows");
        const uint32_t Wf = db->W / db->fold;
        const size_t nsl = db->slice_first.size();
        db->shards.resize(nsl);
        std::vector<uint32_t> folded;
        for (size_t i = 0; i < nsl; i++) {
            Shard& s = db->shards[i];
            s.first_row = db->slice_first[i];
            s.nrows = (i + 1 < nsl ? db->slice_first[i + 1] : db->nrows) - s.first_row;
            s.W = Wf;
            const size_t bytes = static_cast<size_t>(s.nrows) * Wf * 4;
            if (device < 0 || ndevices != 1) {
                int d = 0;
                int rc = gsim_next_device(bytes, &d);
                if (rc != GSIM_OK) return rc;
                s.device = (device >= 0 && ndevices > 1) ? device + (d % ndevices) : d;
            } else {
                s.device = device;
            }
            folded.resize(static_cast<size_t>(s.nrows) * Wf);
            fold_rows_mt(db->host_rows.data() + s.first_row * db->W, s.nrows, db->W, db->fold, folded.data());
            GSIM_HIP(set_device(s.device));
            GSIM_HIP(hipMalloc(&s.d_rows, bytes ? bytes : 16));
            s.owns_rows = true;
            if (bytes) GSIM_HIP(hipMemcpy(s.d_rows, folded.data(), bytes, hipMemcpyHostToDevice));
            int rc = setup_shard(db, s);
            if (rc != GSIM_OK) return rc;
            // The full fingerprints as well, when the device has room for them (on a 288 GB MI355X it practically always
            // has: folding is then a speed device, not a capacity one): the candidates are re-scored on the GPU.  Without
            // them the re-score runs on the host, as in the reference (fingerprintdb_cuda.cu:307-331).
            static const int full_on_device = env_int("GSIM_FOLD_FULL_ON_DEVICE", 1);
            const size_t full_bytes = static_cast<size_t>(s.nrows) * db->W * 4;
            size_t fr = 0, tot = 0;
            if (full_on_device && full_bytes && hipMemGetInfo(&fr, &tot) == hipSuccess && fr > full_bytes + (size_t(2) << 30)) {
                if (hipMalloc(reinterpret_cast<void**>(&s.d_full), full_bytes) == hipSuccess) {
                    const int urc = upload_rows(s.d_full, db->host_rows.data() + s.first_row * db->W, full_bytes, nullptr);
                    if (urc != GSIM_OK) return urc;
                } else {
                    (void) hipGetLastError();
                    s.d_full = nullptr;
                }
            }
        }
        db->finalized = true;
        return GSIM_OK;
    }
    const size_t row_bytes = static_cast<size_t>(db->W) * 4;
    if (ndevices == 1 && device < 0) {
        int rc = gsim_next_device(static_cast<size_t>(db->nrows) * row_bytes, &device);
        if (rc != GSIM_OK) return rc;
    }
    if (device < 0) device = 0;
    if (device + ndevices > ndev) return fail(GSIM_ERR_NO_DEVICE, "device range exceeds the GPUs present");
    if (static_cast<uint64_t>(ndevices) > db->nrows && db->nrows > 0) ndevices = static_cast<int>(db->nrows);
    const uint64_t per = (db->nrows + ndevices - 1) / (ndevices ? ndevices : 1);
    db->shards.resize(static_cast<size_t>(ndevices));
    for (int i = 0; i < ndevices; i++) {
        Shard& s = db->shards[i];
        s.device = device + i;
        s.first_row = std::min<uint64_t>(per * i, db->nrows);
        s.nrows = std::min<uint64_t>(per, db->nrows - s.first_row);
        GSIM_HIP(set_device(s.device));
        const size_t bytes = static_cast<size_t>(s.nrows) * row_bytes;
        GSIM_HIP(hipMalloc(&s.d_rows, bytes ? bytes : 16));
        s.owns_rows = true;
        if (bytes) {
            const int urc = upload_rows(s.d_rows, db->host_rows.data() + s.first_row * db->W, bytes, nullptr);
            if (urc != GSIM_OK) return urc;
        }
        int rc = setup_shard(db, s);
        if (rc != GSIM_OK) return rc;
    }
    db->finalized = true;
    return GSIM_OK;
}

int gsim_db_generate(gsim_db* db, uint64_t seed, int kind, uint64_t first_row, uint64_t nrows, int device)
{
    if (!db) return fail(GSIM_ERR_INVALID, "db is NULL");
    if (db->finalized || db->nrows) return fail(GSIM_ERR_STATE, "table already holds rows");
    if (kind != GSIM_SYNTH_SPARSE && kind != GSIM_SYNTH_DENSE && kind != GSIM_SYNTH_MORGAN)
        return fail(GSIM_ERR_INVALID, "unknown synthetic kind");
    if (kind == GSIM_SYNTH_MORGAN && db->W > 12288) return fail(GSIM_ERR_INVALID, "Morgan-shaped rows: fp_bits too large");
    if (nrows > 0xFFFFFFFFull) return fail(GSIM_ERR_INVALID, "more than 2^32-1 rows");
    int ndev = 0;
    gsim_device_count(&ndev);
    if (ndev == 0) return fail(GSIM_ERR_NO_DEVICE, "no GPU available");
    if (device < 0 || device >= ndev) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
    db->nrows = nrows;
    db->shards.resize(1);
    Shard& s = db->shards[0];
    s.device = device;
    s.first_row = 0;
    s.nrows = nrows;
    GSIM_HIP(set_device(device));
    const size_t bytes = static_cast<size_t>(nrows) * db->W * 4;
    GSIM_HIP(hipMalloc(&s.d_rows, bytes ? bytes : 16));
    s.owns_rows = true;
    int rc = setup_shard(db, s);
    if (rc != GSIM_OK) return rc;
    GSIM_HIP(gsim::launch_generate(s.d_rows, seed, kind, first_row, nrows, db->W, s.stream));
    GSIM_HIP(hipStreamSynchronize(s.stream));
    db->finalized = true;
    return GSIM_OK;
}

int gsim_synth_row(uint64_t seed, int kind, uint64_t row, uint32_t fp_bits, uint32_t* out_words)
{
    if (!out_words) return fail(GSIM_ERR_INVALID, "out_words is NULL");
    if (fp_bits == 0 || fp_bits % 32 != 0 || fp_bits > 32768) return fail(GSIM_ERR_INVALID, "fp_bits must be a positive multiple of 32, <= 32768");
    const uint32_t W = fp_bits / 32;
    if (kind == GSIM_SYNTH_MORGAN) {
        gsim::synth_row_morgan(out_words, seed, row, W);
    } else if (kind == GSIM_SYNTH_SPARSE || kind == GSIM_SYNTH_DENSE) {
        for (uint32_t j = 0; j < W; j++) out_words[j] = gsim::synth_word_iid(seed, kind == GSIM_SYNTH_DENSE, row * W + j);
    } else {
        return fail(GSIM_ERR_INVALID, "unknown synthetic kind");
    }
    return GSIM_OK;
}

int gsim_db_attach_device_rows(gsim_db* db, const void* d_rows, uint64_t nrows, int device)
{
    if (!db || (!d_rows && nrows)) return fail(GSIM_ERR_INVALID, "NULL argument");
    if (db->finalized || db->nrows) return fail(GSIM_ERR_STATE, "table already holds rows");
    if (reinterpret_cast<uintptr_t>(d_rows) % 16 != 0) return fail(GSIM_ERR_INVALID, "device rows must be 16-byte aligned");
    if (nrows > 0xFFFFFFFFull) return fail(GSIM_ERR_INVALID, "more than 2^32-1 rows");
    int ndev = 0;
    gsim_device_count(&ndev);
    if (device < 0 || device >= ndev) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
    db->nrows = nrows;
    db->shards.resize(1);
    Shard& s = db->shards[0];
    s.device = device;
    s.nrows = nrows;
    s.d_rows = const_cast<void*>(d_rows);
    s.owns_rows = false;
    int rc = setup_shard(db, s);
    if (rc != GSIM_OK) return rc;
    db->finalized = true;
    return GSIM_OK;
}

int gsim_db_destroy(gsim_db* db)
{
    if (!db) return GSIM_OK;
    for (auto& s : db->shards) free_shard(s);
    delete db;
    return GSIM_OK;
}

uint64_t gsim_db_count(const gsim_db* db)
{
    return db ? db->nrows : 0;
}

uint32_t gsim_db_fp_bits(const gsim_db* db)
{
    return db ? db->fp_bits : 0;
}

size_t gsim_db_data_bytes(const gsim_db* db)
{
    return db ? static_cast<size_t>(db->nrows) * db->W * 4 : 0;
}

int gsim_db_shard_count(const gsim_db* db)
{
    return db ? static_cast<int>(db->shards.size()) : 0;
}

int gsim_db_row(const gsim_db* db, uint64_t row, uint32_t* out_words)
{
    if (!db || !out_words) return fail(GSIM_ERR_INVALID, "NULL argument");
    if (row >= db->nrows) return fail(GSIM_ERR_INVALID, "row index out of range");
    if (db->has_host_copy) {
        std::memcpy(out_words, db->host_rows.data() + row * db->W, static_cast<size_t>(db->W) * 4);
        return GSIM_OK;
    }
    for (const auto& s : db->shards) {
        if (row >= s.first_row && row < s.first_row + s.nrows) {
            GSIM_HIP(set_device(s.device));
            const unsigned char* src = static_cast<const unsigned char*>(s.d_rows) +
                                       static_cast<size_t>(row - s.first_row) * db->W * 4;
            GSIM_HIP(hipMemcpy(out_words, src, static_cast<size_t>(db->W) * 4, hipMemcpyDeviceToHost));
            return GSIM_OK;
        }
    }
    return fail(GSIM_ERR_STATE, "row not resident");
}

int gsim_db_set_stream(gsim_db* db, void* hip_stream)
{
    if (!db || !db->finalized) return fail(GSIM_ERR_STATE, "table not finalized");
    if (db->shards.size() != 1) return fail(GSIM_ERR_STATE, "set_stream needs a single-shard handle");
    std::lock_guard<std::mutex> guard(db->search_mutex); // (not under a running search)
    Shard& s = db->shards[0];
    s.stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : s.own_stream;
    return GSIM_OK;
}

int gsim_db_set_row_base(gsim_db* db, uint32_t row_base)
{
    if (!db) return fail(GSIM_ERR_INVALID, "db is NULL");
    db->row_base = row_base;
    return GSIM_OK;
}

size_t gsim_result_block_bytes(uint32_t k)
{
    const size_t raw = sizeof(gsim_result_header) + static_cast<size_t>(k) * sizeof(gsim_hit);
    return (raw + 15) / 16 * 16;
}

static int check_search_args(gsim_db* db, const uint32_t* queries, int metric)
{
    if (!db || !queries) return fail(GSIM_ERR_INVALID, "NULL argument");
    if (!db->finalized) return fail(GSIM_ERR_STATE, "table not finalized (no rows on a GPU)");
    if (metric != GSIM_METRIC_TANIMOTO && metric != GSIM_METRIC_TVERSKY) return fail(GSIM_ERR_INVALID, "unknown metric");
    return GSIM_OK;
}

int gsim_db_search(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t kout, float cutoff, int metric,
                   float alpha, float beta, gsim_hit* hits, uint32_t* counts, uint64_t* approx)
{
    int rc = check_search_args(db, queries, metric);
    if (rc != GSIM_OK) return rc;
    if ((!hits && kout && nq) || (!counts && nq)) return fail(GSIM_ERR_INVALID, "NULL output");
    std::lock_guard<std::mutex> guard(db->search_mutex);
    // kout is the stride of the caller's hits array; the search itself never asks for more hits than the
    // table has rows (result blocks, pinned buffers and the select's capacity scale with k: a wild count
    // from a client must not size them)
    const uint32_t k = static_cast<uint32_t>(std::min<uint64_t>(kout, db->nrows));
    const size_t nsh = db->shards.size();
    std::vector<gsim_hit> merged;
    if (db->fold > 1) {
        if (metric != GSIM_METRIC_TANIMOTO) return fail(GSIM_ERR_INVALID, "folded tables support Tanimoto only");
        return search_folded(db, queries, nq, kout, cutoff, hits, counts, approx);
    }
    const bool batched = !g_force_each && nq >= 4 && k <= static_cast<uint32_t>(gsim::kSelectCap) && k > 0 &&
                         gsim::batch_supported(db->W) && env_int("GSIM_BATCH", 1) != 0;
    if (batched) {
        // Multi-query path: kBQ queries share each pass over the table (VALU-bound).
        const size_t blk = gsim_result_block_bytes(k);
        for (uint32_t base = 0; base < nq; base += kBatchMaxQ) {
            const uint32_t nb = std::min<uint32_t>(kBatchMaxQ, nq - base);
            const uint32_t* qb = queries + static_cast<size_t>(base) * db->W;
            std::vector<char> redo(nb, 0);
            for (auto& s : db->shards) {
                if (s.nrows == 0) continue;
                rc = enqueue_batch(db, s, qb, nb, k, cutoff, metric, alpha, beta,
                                   db->row_base + static_cast<uint32_t>(s.first_row), nullptr);
                if (rc != GSIM_OK) return rc;
            }
            bool overflow = false, dense_cutoff = false;
            for (auto& s : db->shards) {
                if (s.nrows == 0) continue;
                GSIM_HIP(set_device(s.device));
                rc = wait_stream(s.stream);
                if (rc != GSIM_OK) return rc;
                if ((s.h_bflags[0] & 24u) == 8u) dense_cutoff = true; // (8: a dense cutoff; 16: the matrix-core pass counted it itself)
                if (s.h_bflags[0] & 16u) db->dense_batches++;
            }
            if (dense_cutoff) { // the cutoff keeps too many rows for the matrix-core pass: VALU pass
                for (auto& s : db->shards) {
                    if (s.nrows == 0) continue;
                    rc = enqueue_batch(db, s, qb, nb, k, cutoff, metric, alpha, beta,
                                       db->row_base + static_cast<uint32_t>(s.first_row), nullptr, false);
                    if (rc != GSIM_OK) return rc;
                }
            }
            for (auto& s : db->shards) {
                if (s.nrows == 0) continue;
                GSIM_HIP(set_device(s.device));
                rc = wait_stream(s.stream);
                if (rc != GSIM_OK) return rc;
                if (s.h_bflags[0] & 1u) overflow = true;
                if (std::getenv("GSIM_DEBUG_BATCH")) { // counters of instrumented builds (GSIM_MF_TIMING)
                    std::fprintf(stderr, "batch flags %u dbg", s.h_bflags[0]);
                    for (int d = 2; d < 16; d++) std::fprintf(stderr, " %u", s.h_bflags[d]);
                    std::fprintf(stderr, "\n");
                }
            }
            for (uint32_t q = 0; q < nb; q++) {
                uint64_t ap = 0;
                merged.clear();
                std::vector<size_t> ends;
                bool bad = overflow;
                for (auto& s : db->shards) {
                    if (s.nrows == 0) continue;
                    const gsim_result_header* h = reinterpret_cast<const gsim_result_header*>(s.h_bresult + q * blk);
                    if (h->flags & 2u) bad = true;
                    ap += h->approx;
                    const gsim_hit* hh = reinterpret_cast<const gsim_hit*>(h + 1);
                    merged.insert(merged.end(), hh, hh + h->count);
                    ends.push_back(merged.size());
                }
                if (bad) {
                    redo[q] = 1;
                    continue;
                }
                const uint32_t n = merge_canonical_lists(merged, ends, k, hits + static_cast<size_t>(base + q) * kout);
                counts[base + q] = n;
                if (approx) approx[base + q] = ap;
            }
            // heavy ties / candidate overflow: those queries go through the single-query path
            for (uint32_t q = 0; q < nb; q++) {
                if (!redo[q]) continue;
                rc = search_one(db, qb + static_cast<size_t>(q) * db->W, k, cutoff, metric, alpha, beta,
                                hits + static_cast<size_t>(base + q) * kout, &counts[base + q],
                                approx ? &approx[base + q] : nullptr, merged);
                if (rc != GSIM_OK) return rc;
            }
        }
        return GSIM_OK;
    }
    static const int pipelined = env_int("GSIM_EACH_PIPELINE", 1);
    if (g_force_each && pipelined && nsh == 1 && nq > 1 && k > 0 && db->shards[0].nrows > 0 && !db->shards[0].d_dbg &&
        !std::getenv("GSIM_FUSED_DEBUG"))
        return search_each_pipelined(db, db->shards[0], queries, nq, k, kout, cutoff, metric, alpha, beta, hits, counts, approx);
    for (uint32_t q = 0; q < nq; q++) {
        const uint32_t* query = queries + static_cast<size_t>(q) * db->W;
        rc = search_one(db, query, k, cutoff, metric, alpha, beta, hits + static_cast<size_t>(q) * kout, &counts[q],
                        approx ? &approx[q] : nullptr, merged);
        if (rc != GSIM_OK) return rc;
    }
    return GSIM_OK;
}

int gsim_db_search_each(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k, float cutoff, int metric,
                        float alpha, float beta, gsim_hit* hits, uint32_t* counts, uint64_t* approx)
{
    g_force_each = true;
    const int rc = gsim_db_search(db, queries, nq, k, cutoff, metric, alpha, beta, hits, counts, approx);
    g_force_each = false;
    return rc;
}

int gsim_db_search_device(gsim_db* db, const uint32_t* query, uint32_t k, float cutoff, int metric, float alpha,
                          float beta, void* d_result)
{
    int rc = check_search_args(db, query, metric);
    if (rc != GSIM_OK) return rc;
    if (!d_result) return fail(GSIM_ERR_INVALID, "d_result is NULL");
    if (db->shards.size() != 1) return fail(GSIM_ERR_STATE, "search_device needs a single-shard handle");
    if (db->fold > 1) return fail(GSIM_ERR_STATE, "search_device does not support folded tables");
    std::lock_guard<std::mutex> guard(db->search_mutex);
    Shard& s = db->shards[0];
    return enqueue_query(db, s, query, k, cutoff, metric, alpha, beta, db->row_base, d_result, false);
}

int gsim_db_search_batch_device(gsim_db* db, const uint32_t* queries, uint32_t nq, uint32_t k, float cutoff, int metric,
                                float alpha, float beta, void* d_results)
{
    int rc = check_search_args(db, queries, metric);
    if (rc != GSIM_OK) return rc;
    if (!d_results) return fail(GSIM_ERR_INVALID, "d_results is NULL");
    if (db->shards.size() != 1) return fail(GSIM_ERR_STATE, "search_batch_device needs a single-shard handle");
    if (db->fold > 1) return fail(GSIM_ERR_STATE, "search_batch_device does not support folded tables");
    std::lock_guard<std::mutex> guard(db->search_mutex);
    Shard& s = db->shards[0];
    const size_t blk = gsim_result_block_bytes(k);
    unsigned char* out = static_cast<unsigned char*>(d_results);
    const bool batched = nq >= 4 && k <= static_cast<uint32_t>(gsim::kSelectCap) && k > 0 && gsim::batch_supported(db->W) &&
                         s.nrows > 0 && env_int("GSIM_BATCH", 1) != 0;
    for (uint32_t base = 0; base < nq; base += kBatchMaxQ) {
        const uint32_t nb = std::min<uint32_t>(kBatchMaxQ, nq - base);
        const uint32_t* qb = queries + static_cast<size_t>(base) * db->W;
        bool redo = !batched;
        if (batched) {
            rc = enqueue_batch(db, s, qb, nb, k, cutoff, metric, alpha, beta, db->row_base, out + base * blk);
            if (rc != GSIM_OK) return rc;
            // the one host synchronisation of a batch: did a query overflow its candidate segment or
            // collect too many ties for the multi-query select (bit 2, set by batch_select_kernel)?
            GSIM_HIP(set_device(s.device));
            rc = wait_stream(s.stream);
            if (rc != GSIM_OK) return rc;
            if (s.h_bflags[0] & 16u) db->dense_batches++;
            if ((s.h_bflags[0] & 24u) == 8u) { // the cutoff keeps too many rows for the exact path and has no band: VALU pass
                rc = enqueue_batch(db, s, qb, nb, k, cutoff, metric, alpha, beta, db->row_base, out + base * blk, false);
                if (rc != GSIM_OK) return rc;
                rc = wait_stream(s.stream);
                if (rc != GSIM_OK) return rc;
            }
            redo = (s.h_bflags[0] & 5u) != 0;
        }
        if (redo) { // those cases are rare: the whole chunk goes through the single-query pipeline
            for (uint32_t q = 0; q < nb; q++) {
                rc = enqueue_query(db, s, qb + static_cast<size_t>(q) * db->W, k, cutoff, metric, alpha, beta,
                                   db->row_base, out + (base + q) * blk, false);
                if (rc != GSIM_OK) return rc;
            }
        }
    }
    return GSIM_OK;
}

int gsim_merge_device(int device, void* hip_stream, const void* d_blocks, uint32_t nblocks, size_t block_bytes,
                      uint32_t k, void* d_result)
{
    if (!d_blocks || !d_result || nblocks == 0) return fail(GSIM_ERR_INVALID, "NULL / empty argument");
    if (block_bytes < gsim_result_block_bytes(k)) return fail(GSIM_ERR_INVALID, "block_bytes too small for k");
    GSIM_HIP(set_device(device));
    GSIM_HIP(gsim::launch_merge_batch(d_blocks, nblocks, 1, block_bytes, k, d_result,
                                      static_cast<hipStream_t>(hip_stream)));
    return GSIM_OK;
}

int gsim_merge_device_batch(int device, void* hip_stream, const void* d_blocks, uint32_t nranks, uint32_t nq,
                            size_t block_bytes, uint32_t k, void* d_results)
{
    if (!d_blocks || !d_results || nranks == 0) return fail(GSIM_ERR_INVALID, "NULL / empty argument");
    if (block_bytes < gsim_result_block_bytes(k)) return fail(GSIM_ERR_INVALID, "block_bytes too small for k");
    if (nq == 0) return GSIM_OK;
    GSIM_HIP(set_device(device));
    GSIM_HIP(gsim::launch_merge_batch(d_blocks, nranks, nq, block_bytes, k, d_results,
                                      static_cast<hipStream_t>(hip_stream)));
    return GSIM_OK;
}

int gsim_merge_host(const void* blocks, uint32_t nblocks, size_t block_bytes, uint32_t k, void* result)
{
    if (!blocks || !result || nblocks == 0) return fail(GSIM_ERR_INVALID, "NULL / empty argument");
    if (block_bytes < sizeof(gsim_result_header)) return fail(GSIM_ERR_INVALID, "block_bytes too small");
    std::vector<gsim_hit> all;
    std::vector<size_t> ends;
    bool canonical = true;
    uint64_t approx = 0;
    uint32_t flags = 0;
    for (uint32_t i = 0; i < nblocks; i++) {
        const unsigned char* b = static_cast<const unsigned char*>(blocks) + static_cast<size_t>(i) * block_bytes;
        gsim_result_header h;
        std::memcpy(&h, b, sizeof(h));
        if (sizeof(h) + static_cast<size_t>(h.count) * sizeof(gsim_hit) > block_bytes)
            return fail(GSIM_ERR_INVALID, "result block count exceeds block_bytes");
        const size_t old = all.size();
        all.resize(old + h.count);
        if (h.count) std::memcpy(all.data() + old, b + sizeof(h), static_cast<size_t>(h.count) * sizeof(gsim_hit));
        canonical = canonical && std::is_sorted(all.begin() + static_cast<std::ptrdiff_t>(old), all.end(), hit_before);
        ends.push_back(all.size());
        approx += h.approx;
        flags |= h.flags;
    }
    gsim_result_header out;
    if (canonical) { // blocks made by the search kernels are in canonical order: a k-way merge is the sorted union's head
        std::vector<gsim_hit> head(std::min<size_t>(all.size(), k));
        const uint32_t n = merge_canonical_lists(all, ends, static_cast<uint32_t>(head.size()), head.data());
        head.resize(n);
        all.swap(head);
    } else {
        std::sort(all.begin(), all.end(), hit_before);
    }
    out.count = static_cast<uint32_t>(std::min<size_t>(all.size(), k));
    out.flags = flags;
    out.approx = approx;
    std::memcpy(result, &out, sizeof(out));
    if (out.count)
        std::memcpy(static_cast<unsigned char*>(result) + sizeof(out), all.data(), sizeof(gsim_hit) * out.count);
    return GSIM_OK;
}

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

int gsim_db_enable_timing(gsim_db* db, int enable)
{
    if (!db) return fail(GSIM_ERR_INVALID, "db is NULL");
    std::lock_guard<std::mutex> guard(db->search_mutex);
    db->timing = enable != 0;
    db->acc = gsim_timing{};
    for (auto& s : db->shards) {
        int rc = drain_timing(db, s);
        if (rc != GSIM_OK) return rc;
        unsigned long long c = 0, f = 0;
        unsigned long long r = 0;
        rc = read_totals(s, &c, &f, &r);
        if (rc != GSIM_OK) return rc;
        s.base_ncand = c;
        s.base_nfinal = f;
        s.base_nredo = r;
    }
    db->acc = gsim_timing{};
    return GSIM_OK;
}

int gsim_db_get_timing(gsim_db* db, gsim_timing* out)
{
    if (!db || !out) return fail(GSIM_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> guard(db->search_mutex);
    db->acc.candidates_sum = 0;
    db->acc.finalists_sum = 0;
    db->acc.handed_back = 0;
    db->acc.handed_back_why = 0;
    for (auto& s : db->shards) {
        int rc = drain_timing(db, s);
        if (rc != GSIM_OK) return rc;
        unsigned long long c = 0, f = 0;
        unsigned long long r = 0;
        rc = read_totals(s, &c, &f, &r);
        if (rc != GSIM_OK) return rc;
        db->acc.candidates_sum += c - s.base_ncand;
        db->acc.finalists_sum += f - s.base_nfinal;
        db->acc.handed_back += r - s.base_nredo;
        db->acc.handed_back_why |= s.h_state->redo_why & 31u; // (bit 5 = "a selector saw it fail": not a reason of its own)
    }
    db->acc.batches_dense_cutoff = db->dense_batches;
    *out = db->acc;
    return GSIM_OK;
}

int gsim_debug_score_table(int device, int metric, float alpha, float beta, uint32_t a, uint32_t max_b,
                           uint32_t max_c, float* out)
{
    if (!out) return fail(GSIM_ERR_INVALID, "out is NULL");
    int ndev = 0;
    gsim_device_count(&ndev);
    if (device < 0 || device >= ndev) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
    GSIM_HIP(set_device(device));
    const size_t n = static_cast<size_t>(max_b + 1) * (max_c + 1);
    float* d = nullptr;
    GSIM_HIP(hipMalloc(&d, n * sizeof(float)));
    hipError_t e = gsim::launch_score_table(metric, alpha, beta, a, max_b, max_c, d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d, n * sizeof(float), hipMemcpyDeviceToHost);
    (void) hipFree(d);
    if (e != hipSuccess) return fail_hip(e, "score table");
    return GSIM_OK;
}

int gsim_debug_prefilter_constants(int device, int metric, float alpha, float beta, uint32_t max_qa, int has_cutoff,
                                   float cutoff, float* out)
{
    if (!out) return fail(GSIM_ERR_INVALID, "out is NULL");
    if (max_qa > 32768) return fail(GSIM_ERR_INVALID, "max_qa too large");
    const int tv = metric == GSIM_METRIC_TVERSKY ? 1 : 0;
    if (device < 0) {
        gsim::prefilter_table_host(tv, alpha, beta, max_qa, has_cutoff, cutoff, out);
        return GSIM_OK;
    }
    int ndev = 0;
    gsim_device_count(&ndev);
    if (device >= ndev) return fail(GSIM_ERR_NO_DEVICE, "device index out of range");
    GSIM_HIP(set_device(device));
    const size_t n = static_cast<size_t>(max_qa + 1) * (has_cutoff ? 1 : gsim::kBBins) * 4;
    float* d = nullptr;
    GSIM_HIP(hipMalloc(&d, n * sizeof(float)));
    hipError_t e = gsim::launch_prefilter_table(tv, alpha, beta, max_qa, has_cutoff, cutoff, d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d, n * sizeof(float), hipMemcpyDeviceToHost);
    (void) hipFree(d);
    if (e != hipSuccess) return fail_hip(e, "prefilter table");
    return GSIM_OK;
}

} // extern "C"
