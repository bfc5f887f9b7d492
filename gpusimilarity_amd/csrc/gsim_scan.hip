// gsim_scan.hip -- gfx950 (MI355X / CDNA4) kernels of the fingerprint scan engine.
//
// Replaces the Thrust pipeline of the reference's FingerprintDB::search_storage
// (fingerprintdb_cuda.cu:228-339: sequence / transform(TanimotoFunctor) /
// remove_if / sort_by_key over ALL rows).  Two routes, same results bit for bit:
//
// Single-launch path (fused_kernel; k <= kFusedMaxK = 8192, the usual case): ONE persistent
//   launch streams the table, keeps candidates in LDS, exchanges per-wave top-M score
//   summaries to raise a table-wide score threshold, publishes the survivors into
//   per-workgroup regions and lets every workgroup rank its share of them -- one grid-wide
//   wait, no histogram, no per-row scratch in global memory.  See the comment above fused_kernel.
//
// Four-kernel pipeline (the general route: any k, any width, heavy ties, adversarial
//   row orders; also what the single launch hands a query back to):
//   K0 sample_kernel   scores a strided sample, histograms it, publishes a starting
//                      threshold bin for the scan.
//   K1 scan_kernel     one streaming pass over the table, 16 B per lane coalesced
//                      loads (a wave64 load instruction = 1 KiB of consecutive
//                      rows), AND+v_bcnt_u32_b32 popcounts, DPP reduction across
//                      the lanes of a row, the reference's f32 divide, cutoff, and
//                      an in-scan streaming top-k filter: the workgroups share a
//                      table-wide coarse score histogram (device atomics) and a
//                      monotone threshold bin derived from it; only rows at or above
//                      it are written out (candidates, 12 B each, per-wave segments)
//                      -- no per-row score array exists.
//   K2 compact_kernel  finds the coarse bin of the k-th best score from the now
//                      complete histogram and keeps the candidates at or above it.
//   K3 select_kernel   32 workgroups: every finalist's output slot is its rank (the
//                      number of larger unique 64-bit keys), counted from LDS; more
//                      than kSelectCap finalists: one workgroup runs an MSD radix
//                      select; k > kSelectCap: radix select + a two-launch sort.
//
// This file: K0-K2 of the four-kernel pipeline.  gsim_fused.hip: the single launch.  gsim_select.hip: K3, the large-k
// route, folded re-score, merge.  gsim_scan_inl.h: the streaming loop they share.
//
// This is HBM-bound bit arithmetic: no MFMA anywhere (the work is AND + popcount,
// not a contraction).  Wave size is hard-wired to 64.
#include "gsim_device.h"

#include <hip/hip_runtime.h>

#include <type_traits>

#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "../../include/gpusim_hip.h"
#include "gsim_device_common.h"
#include "gsim_prefilter.h"
#include "gsim_scan_inl.h"

namespace gsim
{
namespace
{

// ---------------------------------------------------------------------------
// K1: the scan
// ---------------------------------------------------------------------------

// Streaming top-k filter.
//
// Every workgroup keeps, in LDS, a histogram `hist` of the coarse bins of the rows
// it has EMITTED (written out as candidates).  From time to time a wave pushes the
// not-yet-pushed part of it into the table-wide histogram `ghist` (global memory,
// device-scope atomics), re-reads `ghist` and derives a threshold bin: the largest
// bin B with at least k counted rows at or above it.  The threshold is published
// with atomicMax (`gtau`) and every wave of every workgroup picks it up on its next
// chunk.  A row is emitted only if bin(score) >= the wave's current threshold.
//
// Why this is exact: `ghist` only ever counts distinct rows of the table that have
// really been scanned, so "k counted rows at or above B" implies that the table's
// k-th best score lies in a bin >= B; a row in a lower bin scores strictly less than
// k other rows and cannot be in the top-k.  Everything is monotone (counts and
// thresholds only grow), so there are no barriers and no ordering requirements:
// a stale (lower) threshold only emits more than necessary, a histogram read while
// others add to it only under-counts.  On a random table the number of emitted rows
// falls from N to roughly k * ln(N / k) + (#workgroups * first push).
struct BlockFilter {
    uint32_t hist[kScanBins];    // rows emitted by this workgroup, per coarse bin
    uint32_t flushed[kScanBins]; // part of hist already added to ghist
    uint32_t tau;                // workgroup's copy of the threshold bin (monotone)
    uint32_t nemit;              // candidates emitted by the workgroup so far
    uint32_t trigger;            // nemit value at which the next push / re-read happens
    uint32_t lock;               // one pusher at a time
    // Per-wave staging of emitted candidates.  Candidates go to LDS (ds_write, lgkmcnt) and
    // reach global memory in bursts of >= 64: a global store inside the streaming loop would be
    // waited for by the loop's next s_waitcnt vmcnt(0) (gfx950 has one counter for loads and
    // stores) -- measured at ~0.36 us per emitting iteration.
    u64 stage_key[kScanBlock / 64][kStage];
    uint32_t stage_cb[kScanBlock / 64][kStage];
};

// first push after this many emitted rows per workgroup (then geometrically)
constexpr uint32_t kFirstPush = 64;

// Per-wave view of the filter (members wave-uniform except `kept`).
struct WaveFilter {
    static constexpr bool kFused = false;
    __device__ __forceinline__ void checkpoint(uint32_t, int) {}
    // Narrow rows (128 ... 512 bits): a wave meets 64 ... 512 rows per load and the score (an f32 divide per row) is most of
    // the kernel -- 20 M x 128-bit rows: 149 us against the 40 us the bytes take.  Without a cutoff only rows that can reach
    // the threshold bin need a score: the division-free test of the single launch (gsim_prefilter.h, proven for rows up to
    // 512 bits), at the lower edge of the bin.  (With a cutoff every row at or above it is counted: all are scored.)
    template <int LPR> __device__ __forceinline__ void offer_counts(bool active, uint32_t row, uint32_t val, const ScanArgs& a, int lane)
    {
        if constexpr (LPR >= 1 && LPR <= 4) {
            if (!has_cutoff && k) { // (wave-uniform)
                if (tau != pk_tau) { // (wave-uniform; the threshold moves a few times per query)
                    pk_tau = tau;
                    const PrefilterConstants pk = prefilter_constants(a.metric == GSIM_METRIC_TVERSKY, a.alpha, a.beta, a.qpop,
                                                                      prefilter_level(true, static_cast<float>(tau) * (1.0f / kScanBins), 0u), true);
                    pk_ka = pk.ka;
                    pk_kb = pk.kb;
                }
                const bool maybe = active && static_cast<float>(val >> 16) >= __builtin_fmaf(pk_kb, static_cast<float>(val & 0xFFFFu), pk_ka);
                if (__ballot(maybe) == 0) return; // no row of this round can reach the threshold bin: none is scored
                active = maybe;                   // (a row the test rejects lies below the bin: not a candidate, and nothing counts it)
            }
        }
        offer_scored(*this, active, row, val, a, lane);
    }
    uint32_t pk_tau;
    float pk_ka, pk_kb;
    BlockFilter* sh;
    QueryState* st;
    u64* seg;         // this wave's private candidate segment (keys)
    uint32_t* seg_cb; // ... and the popcounts the score came from (common << 16 | popc_db)
    u64* stg_key;     // this wave's LDS staging area
    uint32_t* stg_cb;
    uint32_t k, tau, step, cursor, staged, kept;
    float cutoff;
    bool has_cutoff;

    __device__ __forceinline__ void init(BlockFilter* b, QueryState* state, u64* s, uint32_t* scb, uint32_t kk,
                                         float cut)
    {
        sh = b;
        stg_key = b->stage_key[threadIdx.x >> 6];
        stg_cb = b->stage_cb[threadIdx.x >> 6];
        staged = 0;
        st = state;
        seg = s;
        seg_cb = scb;
        k = kk;
        cutoff = cut;
        has_cutoff = cut > 0.0f; // fingerprintdb_cuda.cu:263: compaction only if cutoff > 0
        pk_tau = 0; // (no threshold yet: everything passes)
        pk_ka = 0.0f;
        pk_kb = 0.0f;
        tau = kk ? state->gtau : static_cast<uint32_t>(kScanBins); // gtau: 0, or set by sample_kernel
        step = kk / 8 > 32 ? kk / 8 : 32;
        cursor = 0;
        kept = 0;
    }

    // device-coherent read of the table-wide threshold (issued a chunk ahead of its use)
    __device__ __forceinline__ uint32_t load_gtau() const
    {
        return __hip_atomic_load(&st->gtau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // pick up a threshold raised by another wave (same workgroup: LDS; any workgroup: g)
    __device__ __forceinline__ void refresh(uint32_t g, int lane)
    {
        const uint32_t t = __hip_atomic_load(&sh->tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        tau = t > tau ? t : tau;
        if (g > tau) { // raised by another workgroup: hand it to the other waves of this one through LDS
            tau = g;
            if (lane == 0) atomicMax(&sh->tau, g);
        }
    }

    // Push this workgroup's new counts into ghist, derive the threshold from ghist.
    __device__ __forceinline__ void push_and_rethreshold(int lane)
    {
        constexpr int PER = kScanBins / 64;
        uint32_t locked = 0;
        if (lane == 0) locked = atomicExch(&sh->lock, 1u);
        locked = __builtin_amdgcn_readfirstlane(locked);
        if (locked == 0) {
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const uint32_t b = static_cast<uint32_t>(lane * PER + i);
                const uint32_t h = __hip_atomic_load(&sh->hist[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t fl = sh->flushed[b];
                if (b >= tau && h > fl) {
                    atomicAdd(&st->ghist[b], h - fl);
                    sh->flushed[b] = h;
                }
            }
            if (lane == 0) __hip_atomic_store(&sh->lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // threshold from the table-wide histogram: device-coherent (sc1) 16-byte buffer
        // loads, 4 per lane -- 1024 separate 4-byte sc1 loads cost ~20 us per push
        uint32_t h[PER];
        uint32_t s = 0;
        {
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc(st->ghist, 0, kScanBins * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < PER / 4; i++) {
                const u32x4 v4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * PER * 4 + i * 16, 0, /*sc1*/ 16);
                h[4 * i + 0] = v4.x;
                h[4 * i + 1] = v4.y;
                h[4 * i + 2] = v4.z;
                h[4 * i + 3] = v4.w;
                s += v4.x + v4.y + v4.z + v4.w;
            }
        }
        uint32_t bin_k, cnt;
        threshold_from_counts<PER>(h, s, k, lane, bin_k, cnt);
        if (cnt >= k) {
            if (lane == 0) {
                atomicMax(&st->gtau, bin_k);
                atomicMax(&sh->tau, bin_k);
            }
            tau = bin_k > tau ? bin_k : tau;
        }
    }

    // staged candidates -> this wave's global segment, coalesced
    __device__ __forceinline__ void flush_stage(int lane)
    {
        for (uint32_t i = lane; i < staged; i += 64) {
            seg[cursor + i] = stg_key[i];
            seg_cb[cursor + i] = stg_cb[i];
        }
        cursor += staged;
        staged = 0;
    }

    // One row per lane (or an inactive lane).
    __device__ __forceinline__ void offer(bool active, uint32_t row, float raw_score, uint32_t cb, int lane)
    {
        const float s = apply_cutoff(raw_score, cutoff);
        const bool keep = active && (!has_cutoff || s != 0.0f);
        kept += keep ? 1u : 0u;
        const uint32_t bin = coarse_bin(s);
        const bool cand = keep && bin >= tau;
        const u64 m = __ballot(cand);
        if (m != 0) {
            if (cand) {
                const uint32_t slot = staged + lane_rank(m);
                stg_key[slot] = make_key(s, row);
                stg_cb[slot] = cb;
                atomicAdd(&sh->hist[bin], 1u); // ds_add_u32
            }
            const uint32_t n = static_cast<uint32_t>(__popcll(m));
            staged += n;
            if (staged > 64) flush_stage(lane);
            uint32_t old = 0;
            if (lane == 0) old = atomicAdd(&sh->nemit, n);
            old = __builtin_amdgcn_readfirstlane(old);
            const uint32_t trig = __hip_atomic_load(&sh->trigger, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old < trig && old + n >= trig) { // exactly one wave crosses a given trigger
                push_and_rethreshold(lane);
                if (lane == 0) {
                    // next push after 50 % more emitted rows (at least `step`): a handful of pushes per
                    // workgroup and query; the emission rate falls as the threshold rises
                    const uint32_t now = __hip_atomic_load(&sh->nemit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t inc = now / 2 > step ? now / 2 : step;
                    __hip_atomic_store(&sh->trigger, now + inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }

    __device__ __forceinline__ void finish(uint32_t w, const ScanArgs& a, int lane)
    {
        if (staged) flush_stage(lane);
        if (lane == 0) {
            a.seg_count[w] = cursor;
            if (cursor) atomicAdd(&a.state->ncand, static_cast<u64>(cursor));
        }
        if (has_cutoff) {
            const uint32_t tot = wave_sum(kept);
            if (lane == 0 && tot) atomicAdd(&a.state->kept, static_cast<u64>(tot));
        }
    }
};

__device__ __forceinline__ void block_filter_init(BlockFilter* sh, uint32_t k, uint32_t tau0)
{
    for (int i = threadIdx.x; i < kScanBins; i += kScanBlock) {
        sh->hist[i] = 0;
        sh->flushed[i] = 0;
    }
    if (threadIdx.x == 0) {
        sh->tau = k ? tau0 : static_cast<uint32_t>(kScanBins);
        sh->nemit = 0;
        sh->trigger = k ? (k < kFirstPush ? k : kFirstPush) : 0xFFFFFFFFu;
        sh->lock = 0;
    }
    __syncthreads();
}

// After every wave of the workgroup is done: whatever has not been pushed yet goes
// into the table-wide histogram, for the bins at or above the final threshold.
// ghist is then exact for every bin >= the largest threshold any wave used, which is
// all K2 needs (see compact_kernel).
__device__ __forceinline__ void block_filter_flush(BlockFilter* sh, const ScanArgs& a)
{
    __syncthreads();
    const uint32_t tau = sh->tau;
    for (int i = threadIdx.x; i < kScanBins; i += kScanBlock) {
        const uint32_t h = sh->hist[i], fl = sh->flushed[i];
        if (static_cast<uint32_t>(i) >= tau && h > fl) atomicAdd(&a.state->ghist[i], h - fl);
    }
}

template <int LPR, int U> __global__ __launch_bounds__(kScanBlock) void scan_kernel(ScanArgs a, ScanGeometry g)
{
    __shared__ BlockFilter s_filter;
    if (a.gate && *a.gate == 0) return; // enqueued as the fallback of the single-launch path, which succeeded
    const int lane = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * (kScanBlock / 64) + (threadIdx.x >> 6));
    block_filter_init(&s_filter, a.k, a.state->gtau);

    const u32x4 q = reinterpret_cast<const u32x4*>(a.query)[lane % LPR];
    if (w == 0 && lane < LPR && a.query_dev != a.query) reinterpret_cast<u32x4*>(a.query_dev)[lane] = q;

    WaveFilter f;
    f.init(&s_filter, a.state, a.cand + static_cast<u64>(w) * g.seg_cap,
           a.cand_cb + static_cast<u64>(w) * g.seg_cap, a.k, a.cutoff);
    scan_rows<LPR, U>(a, g, f, q, w, lane);
    f.finish(w, a, lane);
    block_filter_flush(&s_filter, a);
}

// K0 sample_kernel: a valid starting threshold for the scan.
//
// The scan's filter starts from "emit everything" and needs a few exchanges through
// the table-wide histogram before it prunes; with every workgroup in that state at
// once, the start-up costs ~60 us.  This kernel scores a strided sample of the table
// (nsample chunks, evenly spaced), histograms ALL sampled rows (no emission), and its
// last workgroup publishes tau0 = the largest bin with >= k sampled rows at or above
// it.  The sample is a subset of the table, so tau0 is a valid lower bound of the
// table's k-th best bin.  The histogram is zeroed again: the scan re-reads the
// sampled rows (<0.3 % extra traffic) and counts them itself.
struct SampleFilter {
    uint32_t* hist; // workgroup's LDS histogram
    float cutoff;
    bool has_cutoff;
    template <int LPR> __device__ __forceinline__ void offer_counts(bool active, uint32_t row, uint32_t val, const ScanArgs& a, int lane)
    {
        offer_scored(*this, active, row, val, a, lane);
    }
    __device__ __forceinline__ void offer(bool active, uint32_t, float raw_score, uint32_t, int)
    {
        const float s = apply_cutoff(raw_score, cutoff);
        if (active && (!has_cutoff || s != 0.0f)) atomicAdd(&hist[coarse_bin(s)], 1u);
    }
};

// The end of a sample kernel: the workgroup's histogram into the table-wide one; the last workgroup turns that into
// tau0 and clears it.
__device__ __forceinline__ void sample_publish(const ScanArgs& a, uint32_t* s_hist, uint32_t& s_last, int lane)
{
    __syncthreads();
    // No fences: a __threadfence() on this part writes the XCD's L2 back and invalidates it -- two per workgroup were most
    // of a 20 us sample.  The adds are agent-scope atomics (performed at the coherence point) and RETURN: once the values are
    // back they have been performed, the ticket after the barrier is ordered behind them; the last workgroup reads the
    // table-wide histogram with agent-scope atomic loads.
    uint32_t sink = 0;
    for (int i = threadIdx.x; i < kScanBins; i += blockDim.x)
        if (s_hist[i]) sink += __hip_atomic_fetch_add(&a.state->ghist[i], s_hist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(sink));
    __syncthreads();
    if (threadIdx.x == 0) s_last = (__hip_atomic_fetch_add(&a.state->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < 64) {
        constexpr int PER = kScanBins / 64;
        uint32_t h[PER];
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            h[i] = __hip_atomic_load(&a.state->ghist[lane * PER + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s += h[i];
        }
        uint32_t bin_k, cnt;
        threshold_from_counts<PER>(h, s, a.k, lane, bin_k, cnt);
        if (lane == 0) {
            a.state->gtau = (a.k && cnt >= a.k) ? bin_k : 0u;
            a.state->done = 0;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kScanBins; i += blockDim.x) a.state->ghist[i] = 0;
}

template <int LPR, int U>
__global__ __launch_bounds__(kScanBlock) void sample_kernel(ScanArgs a, uint32_t nsample, u64 stride_chunks)
{
    __shared__ uint32_t s_hist[kScanBins];
    __shared__ uint32_t s_last;
    if (a.gate && *a.gate == 0) return;
    const int lane = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * (kScanBlock / 64) + (threadIdx.x >> 6));
    for (int i = threadIdx.x; i < kScanBins; i += kScanBlock) s_hist[i] = 0;
    __syncthreads();
    constexpr int CH = U * (64 / LPR);
    const u32x4 q = reinterpret_cast<const u32x4*>(a.query)[lane % LPR];
    const u32x4* __restrict__ db = reinterpret_cast<const u32x4*>(a.rows);
    SampleFilter f;
    f.hist = s_hist;
    f.cutoff = a.cutoff;
    f.has_cutoff = a.cutoff > 0.0f;
    const uint32_t nw = gridDim.x * (kScanBlock / 64);
    for (uint32_t i = w; i < nsample; i += nw) {
        const u64 c = static_cast<u64>(i) * stride_chunks; // a full chunk by construction
        const u32x4* p = db + c * (CH * LPR) + lane;
        u32x4 d[U];
#pragma unroll
        for (int j = 0; j < U; j++) d[j] = p[j * 64]; // plain loads: the scan re-reads these lines
        reduce_chunk<LPR, U, true>(d, q, c * CH, a, f, lane);
    }
    sample_publish(a, s_hist, s_last, lane);
}

// Any fingerprint width (W words, not a power-of-two number of 16-byte lanes).  The R rows of a wave's chunk
// (ScanGeometry::chunk_rows: 64, fewer for very wide rows) are R W consecutive words, a multiple of 16 bytes at a
// 16-byte boundary: the wave copies them verbatim into its LDS region with 16-byte global_load_lds (no registers, one
// address computation per 16 bytes), then every lane reads back ITS row -- 16 bytes per read when the rows are
// 16-byte multiples.  (One row per lane straight from global memory -- the reference's access pattern,
// fingerprintdb_cuda.cu:98 -- touches 64 different 128-byte lines per load instruction: 0.18-0.32 of the HBM peak.)
constexpr uint32_t kGenericLdsBytes = 96 * 1024; // dynamic LDS: the query + four wave regions

__host__ __device__ inline uint32_t generic_query_words(uint32_t W) { return (W + 3u) & ~3u; }
__host__ __device__ inline uint32_t generic_lds_bytes(uint32_t W, uint32_t R) { return (generic_query_words(W) + (kScanBlock / 64) * R * W) * 4u; }

struct GenericChunk {
    const uint32_t* db;
    uint32_t* srow;       // this wave's LDS region: R x W words
    const uint32_t* sq;   // the query in LDS
    uint32_t W, R;
    u64 total_words;

    __device__ __forceinline__ void init(const ScanArgs& a, uint32_t R_, uint32_t* s_words, uint32_t wv)
    {
        db = reinterpret_cast<const uint32_t*>(a.rows);
        W = a.W, R = R_;
        sq = s_words;
        srow = s_words + generic_query_words(W) + wv * R * W;
        total_words = a.nrows * W;
        for (uint32_t i = threadIdx.x; i < W; i += kScanBlock) s_words[i] = a.query[i]; // (a workgroup barrier follows in the caller)
    }
    // chunk c -> LDS; returns when it is there
    __device__ __forceinline__ void load(u64 c, int lane) const
    {
        const uint32_t units = R * W / 4u; // 16-byte units per chunk
        const u64 base = c * (static_cast<u64>(R) * W);
        for (uint32_t u0 = 0; u0 < units; u0 += 64u) {
            const uint32_t u = u0 + static_cast<uint32_t>(lane);
            const u64 gi = base + 4ull * u;
            if (u < units && gi + 4u <= total_words) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (db + gi),
                                                 (__attribute__((address_space(3))) void*) (srow + 4u * u0), 16, 0, 0);
            } else if (u < units) { // the table's last words (and what lies behind them in the last chunk)
#pragma unroll
                for (uint32_t t = 0; t < 4; t++) srow[4u * u + t] = gi + t < total_words ? db[gi + t] : 0u;
            }
        }
        __builtin_amdgcn_s_waitcnt(0); // vmcnt(0): the words are in LDS
        __builtin_amdgcn_wave_barrier();
    }
    // popc(row & query), popc(row) of this lane's row of the chunk in LDS
    __device__ __forceinline__ void count(int lane, uint32_t& cc, uint32_t& bb) const
    {
        const uint32_t* mine = srow + (static_cast<uint32_t>(lane) < R ? static_cast<uint32_t>(lane) : 0u) * W;
        cc = 0, bb = 0;
        if (W % 4u == 0) { // rows are 16-byte multiples: ds_read_b128
            const u32x4* m4 = reinterpret_cast<const u32x4*>(mine);
            const u32x4* q4 = reinterpret_cast<const u32x4*>(sq);
            for (uint32_t j = 0; j < W / 4u; j++) {
                const u32x4 x = m4[j], q = q4[j];
                cc = bcnt_acc(x.x & q.x, bcnt_acc(x.y & q.y, bcnt_acc(x.z & q.z, bcnt_acc(x.w & q.w, cc))));
                bb = bcnt_acc(x.x, bcnt_acc(x.y, bcnt_acc(x.z, bcnt_acc(x.w, bb))));
            }
        } else {
            uint32_t j = 0;
            for (; j + 4 <= W; j += 4) {
                uint32_t xr[4], qr[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    xr[u] = mine[j + u];
                    qr[u] = sq[j + u];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    cc = bcnt_acc(xr[u] & qr[u], cc);
                    bb = bcnt_acc(xr[u], bb);
                }
            }
            for (; j < W; j++) {
                const uint32_t xr = mine[j];
                cc = bcnt_acc(xr & sq[j], cc);
                bb = bcnt_acc(xr, bb);
            }
        }
        __builtin_amdgcn_wave_barrier(); // the next chunk's words overwrite the region
    }
};

__global__ __launch_bounds__(kScanBlock) void scan_generic_kernel(ScanArgs a, ScanGeometry g)
{
    __shared__ BlockFilter s_filter;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_words[]; // [query, padded to 4 words][4 waves x R rows x W words]
    if (a.gate && *a.gate == 0) return;
    const int lane = threadIdx.x & 63;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t w = blockIdx.x * (kScanBlock / 64) + wv;
    GenericChunk ch;
    ch.init(a, g.chunk_rows, s_words, wv);
    block_filter_init(&s_filter, a.k, a.state->gtau); // (ends with a workgroup barrier: the query is in place)
    WaveFilter f;
    f.init(&s_filter, a.state, a.cand + static_cast<u64>(w) * g.seg_cap,
           a.cand_cb + static_cast<u64>(w) * g.seg_cap, a.k, a.cutoff);
    for (u64 c = w; c < g.nchunks; c += g.nwaves) {
        ch.load(c, lane);
        uint32_t cc, bb;
        ch.count(lane, cc, bb);
        const u64 row = c * ch.R + lane;
        const bool active = static_cast<uint32_t>(lane) < ch.R && row < a.nrows;
        f.refresh((c / g.nwaves) % 8 == 0 ? f.load_gtau() : 0u, lane);
        const float s = score_of(a.metric, a.alpha, a.beta, a.qpop, bb, cc);
        f.offer(active, static_cast<uint32_t>(row), s, (cc << 16) + bb, lane);
    }
    f.finish(w, a, lane);
    block_filter_flush(&s_filter, a);
}

// K0 for narrow rows (128 / 256 bits) and every width without a register-streaming template: one row per lane, 64
// consecutive rows per sampled chunk, the row's words read by its lane (16-byte loads where rows are whole units) -- a
// sample is 64 Ki ... 1 Mi rows (launch_sample), its access pattern does not matter.  What matters is the end of the kernel:
// every workgroup adds its histogram to the table-wide one and takes a ticket (with the two fences sample_publish once had:
// 1024 workgroups of four waves 67 us, 256: 21 us) -- sixteen waves per workgroup, at most 64 workgroups, no fences: 8.6 us.
constexpr int kSampleRowsBlock = 1024;
__global__ __launch_bounds__(kSampleRowsBlock) void sample_rows_kernel(ScanArgs a, uint32_t nsample, u64 stride_chunks)
{
    __shared__ uint32_t s_hist[kScanBins];
    __shared__ uint32_t s_last;
    if (a.gate && *a.gate == 0) return;
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * (kSampleRowsBlock / 64) + (threadIdx.x >> 6);
    for (int i = threadIdx.x; i < kScanBins; i += kSampleRowsBlock) s_hist[i] = 0;
    __syncthreads();
    SampleFilter f;
    f.hist = s_hist;
    f.cutoff = a.cutoff;
    f.has_cutoff = a.cutoff > 0.0f;
    const uint32_t nw = gridDim.x * (kSampleRowsBlock / 64);
    const uint32_t W = a.W;
    constexpr int NC = 4; // chunks in flight per wave (one row per lane: latency-bound)
    for (uint32_t i = w; i < nsample; i += NC * nw) {
        const uint32_t* p[NC];
        bool valid[NC];
        uint32_t cc[NC], bb[NC];
#pragma unroll
        for (int u = 0; u < NC; u++) {
            valid[u] = i + u * nw < nsample;
            const u64 chunk = static_cast<u64>(valid[u] ? i + u * nw : i) * stride_chunks; // a full chunk by construction
            p[u] = static_cast<const uint32_t*>(a.rows) + (chunk * 64u + static_cast<u64>(lane)) * W;
            cc[u] = bb[u] = 0;
        }
        if (W % 4 == 0) {
            const u32x4* q4 = reinterpret_cast<const u32x4*>(a.query);
            for (uint32_t t = 0; t < W / 4; t++) {
                const u32x4 q = q4[t];
#pragma unroll
                for (int u = 0; u < NC; u++) {
                    const u32x4 d = reinterpret_cast<const u32x4*>(p[u])[t];
                    cc[u] += __popc(d.x & q.x) + __popc(d.y & q.y) + __popc(d.z & q.z) + __popc(d.w & q.w);
                    bb[u] += __popc(d.x) + __popc(d.y) + __popc(d.z) + __popc(d.w);
                }
            }
        } else {
            auto words = [&](auto ww) { // (a compile-time width: every load of the four rows issued before the first count)
                constexpr uint32_t WW = decltype(ww)::value;
                if constexpr (WW != 0) {
                    uint32_t d[NC][WW];
#pragma unroll
                    for (int u = 0; u < NC; u++)
#pragma unroll
                        for (uint32_t t = 0; t < WW; t++) d[u][t] = p[u][t];
#pragma unroll
                    for (uint32_t t = 0; t < WW; t++) {
                        const uint32_t q = a.query[t];
#pragma unroll
                        for (int u = 0; u < NC; u++) {
                            cc[u] += __popc(d[u][t] & q);
                            bb[u] += __popc(d[u][t]);
                        }
                    }
                } else {
                    for (uint32_t t = 0; t < W; t++) {
                        const uint32_t q = a.query[t];
#pragma unroll
                        for (int u = 0; u < NC; u++) {
                            const uint32_t x = p[u][t];
                            cc[u] += __popc(x & q);
                            bb[u] += __popc(x);
                        }
                    }
                }
            };
            switch (W) {
            case 3: words(std::integral_constant<uint32_t, 3>{}); break;
            case 5: words(std::integral_constant<uint32_t, 5>{}); break;
            case 6: words(std::integral_constant<uint32_t, 6>{}); break;
            case 7: words(std::integral_constant<uint32_t, 7>{}); break;
            case 10: words(std::integral_constant<uint32_t, 10>{}); break;
            case 14: words(std::integral_constant<uint32_t, 14>{}); break;
            case 9: words(std::integral_constant<uint32_t, 9>{}); break;
            case 11: words(std::integral_constant<uint32_t, 11>{}); break;
            default: words(std::integral_constant<uint32_t, 0>{}); break;
            }
        }
#pragma unroll
        for (int u = 0; u < NC; u++) f.offer(valid[u], 0u, score_of(a.metric, a.alpha, a.beta, a.qpop, bb[u], cc[u]), 0u, lane);
    }
    sample_publish(a, s_hist, s_last, lane);
}

// ---------------------------------------------------------------------------
// K1 for rows of L = W / 4 sixteen-byte units where L is not a power of two (896-, 1536-, 768-, 640-bit rows ...)
// ---------------------------------------------------------------------------
// The power-of-two widths stream through registers (scan_rows); scan_generic_kernel stages its chunks through LDS and
// tops out at 0.6-0.7 of the HBM rate: with the LDS as the only buffer a CU has ~half of it in flight.  Rows that are
// whole 16-byte units can stream through registers as well.  P = the odd part of L consecutive wave loads (P KB) hold
// exactly 64 / g whole rows (g = gcd(L, 64)): unit u = 64 j + lane of a chunk belongs to row u / L, position u % L.
// Every lane counts its unit against the query unit of that position (P query units per lane, loaded once), an inclusive
// prefix sum over the chunk's units in unit order (DPP scan per load, the running total carried from load to load)
// turns "sum over a row's units" into E(row) - E(row - 1), E = the prefix at the row's last unit, fetched with one
// ds_bpermute per load -- packed (common << 16 | popc): a chunk holds at most 8192 P <= 57 344 bits, neither field
// overflows.  Lane r then scores row r of the chunk.  P <= 7: L in {3, 5, 7} x 2^i; other widths keep the LDS route.
// C = sub-chunks per trip: a wave keeps 2 x C P KB in flight (register double buffer); C P ~ 7-9 loads, as the template path's 8.
// (the loop itself: scan_rows_ragged, gsim_scan_inl.h -- the single launch runs it too)
template <int P, int C> __global__ __launch_bounds__(kScanBlock) void scan_ragged_kernel(ScanArgs a, ScanGeometry g)
{
    __shared__ BlockFilter s_filter;
    if (a.gate && *a.gate == 0) return;
    const int lane = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * (kScanBlock / 64) + (threadIdx.x >> 6));
    block_filter_init(&s_filter, a.k, a.state->gtau);
    WaveFilter f;
    f.init(&s_filter, a.state, a.cand + static_cast<u64>(w) * g.seg_cap, a.cand_cb + static_cast<u64>(w) * g.seg_cap, a.k, a.cutoff);
    scan_rows_ragged<P, C>(a, g, f, w, lane);
    f.finish(w, a, lane);
    block_filter_flush(&s_filter, a);
}

// ---------------------------------------------------------------------------
// K2: compaction at the k-th best coarse bin
// ---------------------------------------------------------------------------
//
// After the scan, ghist[b] is the exact number of table rows in bin b for every
// b >= T, T = the largest threshold any wave used, and an under-count below T.  The
// table's k-th best bin B* is >= T (every threshold is a lower bound for it), so the
// largest B with sum_{b>=B} ghist[b] >= k is exactly B*; every top-k row has
// bin >= B* >= the threshold its wave compared it with and was therefore emitted.
constexpr int kCompactStage = 1024; // finalists staged in LDS per workgroup

// One wavefront per candidate segment, several loads in flight per lane.  The
// survivors of a workgroup are staged in LDS and appended to `finalists` with ONE
// global atomic per workgroup (a single hot word only sustains ~90 returning
// atomics per microsecond); entries beyond the staging area (heavy ties) are
// appended directly.
__global__ __launch_bounds__(kScanBlock) void compact_kernel(ScanArgs a, ScanGeometry g, u64* finalists,
                                                             uint32_t* finalists_cb, uint32_t cap)
{
    __shared__ u64 s_stage[kCompactStage];
    __shared__ uint32_t s_stage_cb[kCompactStage];
    __shared__ uint32_t s_n, s_base;
    if (a.gate && *a.gate == 0) return;
    const int lane = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * (kScanBlock / 64) + (threadIdx.x >> 6));
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t n = a.k ? a.seg_count[w] : 0;
    if (n != 0) {
        uint32_t bstar, cnt;
        find_threshold(a.state->ghist, a.k, lane, bstar, cnt);
        const u64* seg = a.cand + static_cast<u64>(w) * g.seg_cap;
        const uint32_t* seg_cb = a.cand_cb + static_cast<u64>(w) * g.seg_cap;
        constexpr int UN = 4; // independent loads in flight per lane
        constexpr uint32_t STAGE = kCompactStage;
        for (uint32_t base = 0; base < n; base += 64 * UN) {
            u64 key[UN];
            uint32_t cb[UN];
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const uint32_t i = base + u * 64 + lane;
                key[u] = i < n ? seg[i] : 0ull;
                cb[u] = i < n ? seg_cb[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const uint32_t i = base + u * 64 + lane;
                const bool ok = i < n && coarse_bin(key_score(static_cast<uint32_t>(key[u] >> 32))) >= bstar;
                const u64 m = __ballot(ok);
                if (m != 0) {
                    const uint32_t cntw = static_cast<uint32_t>(__popcll(m));
                    uint32_t pos = 0;
                    if (lane == 0) pos = atomicAdd(&s_n, cntw);
                    pos = __builtin_amdgcn_readfirstlane(pos);
                    const uint32_t e = pos + lane_rank(m); // slot in the workgroup's reservation order
                    if (pos + cntw <= STAGE) {
                        if (ok) {
                            s_stage[e] = key[u];
                            s_stage_cb[e] = cb[u];
                        }
                    } else {
                        const uint32_t first_over = pos > STAGE ? pos : STAGE;
                        uint32_t gpos = 0;
                        if (lane == 0) gpos = atomicAdd(&a.state->nfinal, pos + cntw - first_over);
                        gpos = __builtin_amdgcn_readfirstlane(gpos);
                        if (ok) {
                            if (e < STAGE) {
                                s_stage[e] = key[u];
                                s_stage_cb[e] = cb[u];
                            } else {
                                const uint32_t idx = gpos + (e - first_over);
                                if (idx < cap) {
                                    finalists[idx] = key[u];
                                    finalists_cb[idx] = cb[u];
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    const uint32_t reserved = s_n;
    const uint32_t staged = reserved < static_cast<uint32_t>(kCompactStage) ? reserved
                                                                             : static_cast<uint32_t>(kCompactStage);
    if (staged == 0) return;
    if (threadIdx.x == 0) s_base = atomicAdd(&a.state->nfinal, staged);
    __syncthreads();
    const uint32_t gbase = s_base;
    for (uint32_t i = threadIdx.x; i < staged; i += kScanBlock) {
        if (gbase + i < cap) {
            finalists[gbase + i] = s_stage[i];
            finalists_cb[gbase + i] = s_stage_cb[i];
        }
    }
}

template <int LPR, int U> hipError_t launch_scan_t(const ScanArgs& a, const ScanGeometry& g, hipStream_t s)
{
    const uint32_t nblocks = g.nwaves / (kScanBlock / 64);
    hipLaunchKernelGGL((scan_kernel<LPR, U>), dim3(nblocks), dim3(kScanBlock), 0, s, a, g);
    return hipGetLastError();
}

} // namespace

// ---------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------

static bool is_pow2(uint32_t x)
{
    return x && !(x & (x - 1));
}

// loads per chunk of scan_ragged_kernel for rows of L sixteen-byte units: the odd part of L when that is 3, 5, ... 15, a
// chunk holds at least one whole row (64 P / L >= 1) and a row's bits fit the packed counts' 16-bit fields, else 0
static uint32_t ragged_loads_of(uint32_t L, bool enabled)
{
    if (!enabled || L == 0) return 0;
    uint32_t odd = L;
    while (odd % 2 == 0) odd /= 2;
    if (odd < 3 || odd > 15 || L * 128u > 65535u) return 0;
    return (64u * odd) % L == 0 && 64u * odd / L >= 1 ? odd : 0;
}

ScanGeometry scan_geometry(uint64_t nrows, uint32_t W, int num_cus, int waves_per_cu, int unroll, bool ragged)
{
    ScanGeometry g{};
    const uint32_t lpr = (W % 4 == 0 && is_pow2(W / 4) && W / 4 <= 64) ? W / 4 : 0;
    g.lanes_per_row = lpr;
    if (lpr) {
        unroll = 8; // (the only unroll built: 4 loads per chunk cost 12 %, 16 bought nothing -- profiles/r01_sweeps.txt)
        g.unroll = static_cast<uint32_t>(unroll);
        g.chunk_rows = g.unroll * (64 / lpr);
    } else if (W % 4 == 0 && ragged_loads_of(W / 4, ragged) != 0) {
        // whole 16-byte units per row, odd part 3 ... 15: streamed through registers (scan_ragged_kernel)
        g.ragged_loads = ragged_loads_of(W / 4, ragged);
        g.unroll = g.ragged_loads == 3 ? 3 : (g.ragged_loads == 5 ? 2 : 1); // sub-chunks per trip: 9, 10 or P loads in flight
        g.chunk_rows = g.unroll * (64u * g.ragged_loads / (W / 4));
    } else {
        // generic widths: the four waves' chunks live in LDS (scan_generic_kernel), fewer rows per chunk when they are wide
        g.unroll = 1;
        g.chunk_rows = 64;
        while (g.chunk_rows > 4 && generic_lds_bytes(W, g.chunk_rows) > kGenericLdsBytes) g.chunk_rows /= 2;
        // a wave has one chunk in flight and computes between loads: as many workgroups per CU as the LDS holds (up to four)
        const uint32_t lds = generic_lds_bytes(W, g.chunk_rows) + static_cast<uint32_t>(sizeof(BlockFilter));
        const int per_cu = std::max(1, std::min(4, static_cast<int>(150u * 1024u / lds)));
        waves_per_cu = std::max(waves_per_cu, per_cu * (kScanBlock / 64));
    }
    g.nchunks = (nrows + g.chunk_rows - 1) / g.chunk_rows;
    uint64_t nw = static_cast<uint64_t>(num_cus) * static_cast<uint64_t>(waves_per_cu);
    if (nw > g.nchunks) nw = g.nchunks;
    if (nw < 1) nw = 1;
    const uint32_t wpb = kScanBlock / 64;
    nw = (nw + wpb - 1) / wpb * wpb;
    g.nwaves = static_cast<uint32_t>(nw);
    const uint64_t per = (g.nchunks + g.nwaves - 1) / g.nwaves;
    g.seg_cap = static_cast<uint32_t>((per ? per : 1) * g.chunk_rows);
    return g;
}

template <int LPR, int U>
hipError_t launch_sample_t(const ScanArgs& a, uint32_t nsample, uint64_t stride, uint32_t nblocks, hipStream_t s)
{
    hipLaunchKernelGGL((sample_kernel<LPR, U>), dim3(nblocks), dim3(kScanBlock), 0, s, a, nsample, stride);
    return hipGetLastError();
}

// Starting threshold from a strided sample (large tables only).
hipError_t launch_sample(const ScanArgs& a, const ScanGeometry& g, uint32_t chunks_per_wave, hipStream_t s, bool* launched, int sample_shift)
{
    if (launched) *launched = false;
    if (a.k == 0 || chunks_per_wave == 0) return hipSuccess;
    if (g.lanes_per_row == 0 || g.lanes_per_row <= 2) {
        // sample_rows_kernel, chunks of 64 rows.  The k-th best of a sample of S rows out of N leaves ~k N / S rows above it:
        // S = k N / 2^15 keeps that at ~32 Ki rows (a few dozen per scan wave) -- at least 64 Ki rows, at most 1 Mi, never
        // more than 1/8 of the table (down to 16 Ki rows, tables of 131 k rows: sparse 128-bit tables of 0.5 M rows were
        // handed back one query in ten without a seed); under that the scan's own warm-up is cheaper.
        const int shift = sample_shift > 0 && sample_shift < 40 ? sample_shift : 15;
        uint64_t want = (static_cast<uint64_t>(a.k) * a.nrows) >> shift;
        if (want < 65536) want = 65536;
        if (want > (1u << 20)) want = 1u << 20;
        if (want > a.nrows / 8) want = a.nrows / 8;
        if (want < 16384) return hipSuccess;
        const uint32_t nsample = static_cast<uint32_t>(want / 64);
        const uint64_t stride = (a.nrows / 64) / nsample;
        uint32_t nblocks = 64;
        if (nblocks > nsample / (kSampleRowsBlock / 64)) nblocks = nsample / (kSampleRowsBlock / 64);
        hipLaunchKernelGGL(sample_rows_kernel, dim3(nblocks), dim3(kSampleRowsBlock), 0, s, a, nsample, stride);
        if (launched) *launched = true;
        return hipGetLastError();
    }
    const uint32_t chunk_rows = g.chunk_rows;
    const uint64_t nfull = a.nrows / chunk_rows;
    // never sample more than 1/8 of the table; under one chunk per wave the scan's own warm-up is cheaper
    const uint64_t fit = nfull / (8ull * g.nwaves);
    if (fit < chunks_per_wave) chunks_per_wave = static_cast<uint32_t>(fit);
    if (chunks_per_wave == 0) return hipSuccess;
    if (launched) *launched = true;
    const uint64_t want = static_cast<uint64_t>(g.nwaves) * chunks_per_wave;
    const uint64_t stride = nfull / want;
    const uint32_t nsample = static_cast<uint32_t>(want);
    const uint32_t nblocks = g.nwaves / (kScanBlock / 64);
#define GSIM_CASE(L, UU) \
    if (g.lanes_per_row == L && g.unroll == UU) return launch_sample_t<L, UU>(a, nsample, stride, nblocks, s);
    GSIM_CASE(8, 8)
    GSIM_CASE(16, 8)
    GSIM_CASE(4, 8)
    GSIM_CASE(32, 8)
    GSIM_CASE(64, 8)
#undef GSIM_CASE
    return hipSuccess;
}

hipError_t launch_scan(const ScanArgs& a, const ScanGeometry& g, hipStream_t s)
{
#define GSIM_CASE(L, UU) \
    if (g.lanes_per_row == L && g.unroll == UU) return launch_scan_t<L, UU>(a, g, s);
    GSIM_CASE(8, 8)
    GSIM_CASE(16, 8)
    GSIM_CASE(1, 8)
    GSIM_CASE(2, 8)
    GSIM_CASE(4, 8)
    GSIM_CASE(32, 8)
    GSIM_CASE(64, 8)
#undef GSIM_CASE
    if (g.lanes_per_row != 0) return hipErrorInvalidValue;
    const uint32_t nblocks = g.nwaves / (kScanBlock / 64);
    if (g.ragged_loads == 3) hipLaunchKernelGGL((scan_ragged_kernel<3, 3>), dim3(nblocks), dim3(kScanBlock), 0, s, a, g);
    else if (g.ragged_loads == 5) hipLaunchKernelGGL((scan_ragged_kernel<5, 2>), dim3(nblocks), dim3(kScanBlock), 0, s, a, g);
    else if (g.ragged_loads == 7) hipLaunchKernelGGL((scan_ragged_kernel<7, 1>), dim3(nblocks), dim3(kScanBlock), 0, s, a, g);
    else if (g.ragged_loads == 9) hipLaunchKernelGGL((scan_ragged_kernel<9, 1>), dim3(nblocks), dim3(kScanBlock), 0, s, a, g);
    else if (g.ragged_loads == 11) hipLaunchKernelGGL((scan_ragged_kernel<11, 1>), dim3(nblocks), dim3(kScanBlock), 0, s, a, g);
    else if (g.ragged_loads == 13) hipLaunchKernelGGL((scan_ragged_kernel<13, 1>), dim3(nblocks), dim3(kScanBlock), 0, s, a, g);
    else if (g.ragged_loads == 15) hipLaunchKernelGGL((scan_ragged_kernel<15, 1>), dim3(nblocks), dim3(kScanBlock), 0, s, a, g);
    if (g.ragged_loads) return hipGetLastError();
    static DynLdsOnce once;
    const hipError_t e = once.ensure(reinterpret_cast<const void*>(scan_generic_kernel), kGenericLdsBytes);
    if (e != hipSuccess) return e;
    const uint32_t lds = generic_lds_bytes(a.W, g.chunk_rows);
    if (lds > kGenericLdsBytes) return hipErrorInvalidValue;
    hipLaunchKernelGGL(scan_generic_kernel, dim3(nblocks), dim3(kScanBlock), lds, s, a, g);
    return hipGetLastError();
}

hipError_t launch_compact(const ScanArgs& a, const ScanGeometry& g, unsigned long long* finalists,
                          uint32_t* finalists_cb, uint32_t finalists_cap, hipStream_t s)
{
    const uint32_t nblocks = g.nwaves / (kScanBlock / 64);
    hipLaunchKernelGGL(compact_kernel, dim3(nblocks), dim3(kScanBlock), 0, s, a, g, finalists, finalists_cb,
                       finalists_cap);
    return hipGetLastError();
}

} // namespace gsim
