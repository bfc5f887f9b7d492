// gsim_device.hip -- gfx950 (MI355X / CDNA4) kernels of the fingerprint scan engine.
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
//                      select; k > kSelectCap: global-memory bitonic sort.
//
// This is HBM-bound bit arithmetic: no MFMA anywhere (the work is AND + popcount,
// not a contraction).  Wave size is hard-wired to 64.
#include "gsim_device.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include "../../include/gpusim_hip.h"
#include "gsim_device_common.h"
#include "gsim_synth.h"

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
        tau = kk ? state->gtau : static_cast<uint32_t>(kScanBins); // gtau: 0, or set by sample_kernel
#if defined(GSIM_ABLATE) && GSIM_ABLATE >= 11
        tau = GSIM_FIXED_TAU; // ablations 11-13: fixed threshold bin, no pushes; 14: fixed start, then adaptive
#endif
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
#if defined(GSIM_ABLATE) && GSIM_ABLATE == 4
        asm volatile("" ::"s"(m)); // ablation 4: the filter's fast path only (no emission code)
        return;
#endif
        if (m != 0) {
            if (cand) {
                const uint32_t slot = staged + lane_rank(m);
                stg_key[slot] = make_key(s, row);
                stg_cb[slot] = cb;
#if !(defined(GSIM_ABLATE) && (GSIM_ABLATE == 5 || GSIM_ABLATE == 11))
                atomicAdd(&sh->hist[bin], 1u); // ds_add_u32
#endif
            }
            const uint32_t n = static_cast<uint32_t>(__popcll(m));
            staged += n;
            if (staged > 64) flush_stage(lane);
#if defined(GSIM_ABLATE) && (GSIM_ABLATE == 5 || GSIM_ABLATE == 6 || GSIM_ABLATE == 11 || GSIM_ABLATE == 12)
            return;
#endif
            uint32_t old = 0;
            if (lane == 0) old = atomicAdd(&sh->nemit, n);
            old = __builtin_amdgcn_readfirstlane(old);
            const uint32_t trig = __hip_atomic_load(&sh->trigger, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#if defined(GSIM_ABLATE) && (GSIM_ABLATE == 7 || GSIM_ABLATE == 13)
            asm volatile("" ::"s"(old), "s"(trig));
            return;
#endif
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

__device__ __forceinline__ u32x4 stream_load(const u32x4* p)
{
    // read once: keep it out of the caches' way (+13 % on tables far larger than the caches; default-policy loads
    // for tables that fit the 256 MB Infinity Cache were tried on repeated queries over 1 M rows: no gain)
    return __builtin_nontemporal_load(p);
}

// LPR = 16-byte lanes per fingerprint (fp_bits / 128), U = loads per lane per chunk.
// A chunk is CH = U * 64 / LPR consecutive rows = U KiB of the table; wave w takes
// chunks w, w + nwaves, ...  The loads of the next chunk are issued before the
// current one is reduced (register double buffer): 2U KiB in flight per wave.
// The loop body over full chunks is branch-free up to the (rare) emit path, so the
// compiler's s_waitcnt placement leaves the prefetch in flight during the reduce;
// the table's last, partial chunk is handled once, outside the loop.
template <int LPR, int U, bool FULL, typename Filter>
__device__ __forceinline__ void reduce_chunk(const u32x4 (&d)[U], const u32x4& q, u64 row0, const ScanArgs& a,
                                             Filter& f, int lane)
{
    constexpr int RPL = 64 / LPR;
    constexpr int ROUNDS = (U + LPR - 1) / LPR;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
#if defined(GSIM_ABLATE) && GSIM_ABLATE == 1
    // ablation 1: loads only (one XOR per dword keeps them alive)
    u32x4 x = d[0];
#pragma unroll
    for (int j = 1; j < U; j++) x ^= d[j];
    asm volatile("" ::"v"(x.x ^ x.y ^ x.z ^ x.w ^ q.x));
    (void) row0; (void) a; (void) f; (void) grp; (void) sub;
    return;
#endif
    uint32_t v[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
        // v_and + v_bcnt_u32_b32 (popcount with accumulate)
        const uint32_t cc =
            __popc(d[j].x & q.x) + __popc(d[j].y & q.y) + __popc(d[j].z & q.z) + __popc(d[j].w & q.w);
        const uint32_t bb = __popc(d[j].x) + __popc(d[j].y) + __popc(d[j].z) + __popc(d[j].w);
        v[j] = group_sum<LPR>((cc << 16) + bb); // both sums < 2^16 (fp_bits <= 32768)
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        // lane (grp, sub) takes the row of load j = r*LPR + sub
        uint32_t val = 0;
#pragma unroll
        for (int jj = 0; jj < U; jj++) {
            if (jj / LPR == r) val = (sub == jj % LPR) ? v[jj] : val;
        }
#if defined(GSIM_ABLATE) && GSIM_ABLATE == 2
        asm volatile("" ::"v"(val)); // ablation 2: + popcounts and the DPP reduction
        (void) row0; (void) a; (void) f; (void) grp;
        continue;
#endif
        const int j = r * LPR + sub;
        const u64 row = row0 + static_cast<u64>(j * RPL + grp);
        const bool active = (j < U) && (FULL || row < a.nrows);
        const float s = score_of(a.metric, a.alpha, a.beta, a.qpop, val & 0xFFFFu, val >> 16);
#if defined(GSIM_ABLATE) && GSIM_ABLATE == 3
        asm volatile("" ::"v"(s), "v"(active)); // ablation 3: + the score
        (void) f;
        continue;
#endif
        f.offer(active, static_cast<uint32_t>(row), s, val, lane);
    }
}

// The streaming loop of one wavefront: chunks w, w + nwaves, ... of the table through filter f.
template <int LPR, int U, typename Filter>
__device__ __forceinline__ void scan_rows(const ScanArgs& a, const ScanGeometry& g, Filter& f, const u32x4& q, uint32_t w,
                                          int lane)
{
    constexpr int RPL = 64 / LPR; // rows per load instruction
    constexpr int CH = U * RPL;   // rows per chunk
    const u32x4* __restrict__ db = reinterpret_cast<const u32x4*>(a.rows);
    uint32_t gt = 0; // table-wide threshold, loaded ahead of its use
    uint32_t trip = 0;
    const uint32_t wib = w % (kScanBlock / 64);

    const u64 nfull = a.nrows / CH; // chunks with all CH rows present
    if (w < nfull) {
        const u64 last = w + (nfull - 1 - w) / g.nwaves * g.nwaves; // this wave's last full chunk
        u32x4 nxt[U];
        {
            const u32x4* p = db + static_cast<u64>(w) * (CH * LPR) + lane;
#pragma unroll
            for (int j = 0; j < U; j++) nxt[j] = stream_load(p + j * 64);
        }
        for (u64 c = w;; c += g.nwaves) {
            u32x4 d[U];
#pragma unroll
            for (int j = 0; j < U; j++) d[j] = nxt[j];
            // prefetch; on the final trip it re-reads the last chunk (no branch in the body)
            const u64 cn = c + g.nwaves <= last ? c + g.nwaves : last;
            const u32x4* p = db + cn * (CH * LPR) + lane;
#pragma unroll
            for (int j = 0; j < U; j++) nxt[j] = stream_load(p + j * 64);
            f.refresh(gt, lane);
            // The workgroup polls the table-wide threshold every 8th chunk while it moves fast
            // (first 64 chunks), then every 32nd, then every 128th; the waves take turns so that
            // no single wave pays for all polls.  A poll is one more entry in the loop's vmcnt
            // queue: its latency is exposed whenever it exceeds the prefetch's (~1 us each).
            // (Single-launch path: every chunk of the first 32, in turns -- a small table is over
            // after 16 trips and its first threshold arrives around the 10th.)
            {
                const uint32_t period = (Filter::kFused && trip < 32u) ? 1u : (trip < 64u ? 8u : (trip < 512u ? 32u : 128u));
                if ((trip & (period - 1u)) == 0 && ((trip / period) & (kScanBlock / 64 - 1)) == wib) gt = f.load_gtau();
                trip++;
            }
            reduce_chunk<LPR, U, true>(d, q, c * CH, a, f, lane);
            if (Filter::kFused) f.checkpoint(trip, lane);
            if (c == last) break;
        }
    }
    if (nfull < g.nchunks && w == nfull % g.nwaves) { // the table's partial last chunk
        const u64 row0 = nfull * CH;
        const u32x4* p = db + row0 * LPR + lane;
        const int grp = lane / LPR;
        u32x4 d[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            const u64 row = row0 + static_cast<u64>(j * RPL + grp);
            d[j] = row < a.nrows ? stream_load(p + j * 64) : u32x4{0, 0, 0, 0};
        }
        f.refresh(f.load_gtau(), lane);
        reduce_chunk<LPR, U, false>(d, q, row0, a, f, lane);
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

// ---------------------------------------------------------------------------
// The single-launch path: scan, publish, select and the result block in ONE kernel
// ---------------------------------------------------------------------------
//
// The four-kernel pipeline costs ~110 us of launches, boundaries and host turn-around on
// top of the streaming time (1 M x 1024-bit rows stream in 18 us), and its table-wide
// histogram costs ~12 ns per device atomic, serialised per cache line: pushing a 256-row
// histogram from 256 workgroups is tens of microseconds.  This kernel does the whole query
// in one persistent launch with NO grid barrier, NO histogram and NO global atomics beyond
// tickets:
//
//   1. every wave streams its chunks (scan_rows) and keeps the rows at or above the current
//      threshold in its own LDS store (kFusedWaveCap slots; compacted in place when the
//      threshold has risen).  The threshold is an exact 32-bit score key (order_key), 0 at
//      the start: until the first one arrives every row is stored -- LDS writes only;
//   2. in-loop checkpoints (after 1, 4, 16, ... trips and after 3/4 of them): every streaming wave leaves ONE
//      score key in LDS, its M-th best (M ~ 2k / #waves); the workgroup's forwarder wave copies the four keys to a
//      table-wide array (plain write-through stores) and takes a two-level ticket; the last arriver's poller wave
//      elects the r-th largest report, r = ceil(k / M), and publishes it as the new threshold (atomicMax).  Valid
//      because each of the r largest reports stands for M distinct rows really scanned at or above it: at least k
//      rows score at or above the threshold, so no top-k row is below it -- and a slot read early or stale only
//      holds a smaller key, which only lowers the threshold.  The streaming waves never touch global memory for any
//      of this (one in-order vmcnt: a store or atomic would drain their prefetch);
//   3. a workgroup that has finished streaming drops what lies below the freshest in-loop threshold and publishes
//      the rest into ITS OWN fixed region of the list -- no reservation, no exchange before it -- in an order a reader
//      can stop in: canonical order up to kFusedSortCap rows (a local rank count), bucket order above (a counting
//      sort by score key >> shift, highest bucket first); its 16-byte header holds the count, the order, the shift
//      and the workgroup's REPORT, its Mw-th best 64-bit key.  Write-through stores, one wait, a two-level arrival
//      (a counter per group of workgroups, a top counter, one generation word per group);
//   4. every workgroup then becomes a selector.  It waits until all have arrived -- the ONE grid-wide wait of the
//      kernel; bounded by a few scan times of wall clock: on a GPU shared with another queue part of the grid may
//      not have started while the waiters hold their CUs, the query then goes to the four-kernel pipeline, which
//      never waits -- and requests, in one round trip, every workgroup's header and the first 16 entries of every
//      region.  From the reports every selector derives the SAME final threshold (the r-th largest report as a
//      64-bit key -- it carries the row index, so it also cuts through groups of equal scores), keeps the published
//      rows at or above it in LDS (a list is read on, 64 entries at a time, until an entry proves the rest lies below
//      the threshold) and ranks the rows it owns (hash of the row) -- by counting larger keys, or through a histogram
//      of the finalists when there are many -- the output slot of a hit is its rank, keys are unique; the hits of
//      rank < k go straight into the result block, written through at system scope;
//   5. every selector waits for its stores' acknowledgements and takes a (two-level) ticket; the last one writes the
//      header -- for the synchronous API with the query's epoch in the flags word: the caller polls the header of its
//      pinned block, ONE 16-byte store is header and completion signal, no fence anywhere -- and then re-zeroes the
//      per-query state behind the caller's back.
//
// Whatever the path cannot hold (a wave's store that stays full after compaction, more than 16 Ki finalists or 2 Ki
// owned by one selector: extreme ties, rows in ascending score order) sets QueryState::redo and header flag 2; the
// four-kernel pipeline then runs the query.
constexpr int kFusedFinalLds = 16384;   // finalists a selector ranks (LDS)
constexpr int kFusedMineCap = 2048;     // ... of which it owns at most this many
constexpr int kFusedBlock = kScanBlock + 128; // four streaming waves + two service waves (forwarder, poller/elector)
constexpr uint32_t kFusedPrefix = 16;   // entries of every region a selector requests before it knows the region's count
constexpr uint32_t kFusedSortCap = 128; // a workgroup with up to this many rows publishes them in canonical order (the count is
                                        // quadratic: 256 rows that all sit in one wave's store cost 10 us); more: in bucket order
constexpr uint32_t kFusedItems = 1024;  // 64-entry reads beyond the prefixes a selector lists per round (at most 4 per region)
constexpr uint32_t kFusedRankDirect = 2048; // up to this many finalists a selector ranks its rows by comparing each with every finalist
constexpr uint32_t kFusedBins = 1024;   // buckets of the order in which a workgroup with more than kFusedSortCap rows publishes them

struct FusedShared { // (static_assert below: it fits the CU's 160 KB)
    union {
        struct { // while streaming
            u64 key[kScanBlock / 64][kFusedWaveCap];
            uint32_t cb[kScanBlock / 64][kFusedWaveCap];
        } store;
        struct { // selectors
            u64 fkey[kFusedFinalLds];
            union {
                struct {
                    uint32_t idx[kFusedMineCap];
                    uint32_t cb[kFusedMineCap];
                } mine;
                u64 rep[kFusedSelectors]; // the workgroups' end-of-scan reports, during the election only
            } u;
        } sel;
    };
    u64 tauf;                       // the final threshold
    uint32_t tau;       // workgroup's copy of the score-key threshold (monotone; kept fresh by the service wave)
    uint32_t overflow;  // a wave's store overflowed
    uint32_t nemit;     // rows stored by the workgroup (statistics)
    uint32_t scan_done; // streaming waves that have finished
    uint32_t elect_req; // forwarder -> poller: this workgroup took the last ticket of a checkpoint, run the election
    uint32_t fwd_done;  // the forwarder has passed on every in-loop checkpoint
    uint32_t elected;   // the poller's copy of QueryState::elected
    uint32_t abort;     // the poller gave up waiting for the in-loop elections (GPU shared with another queue)
    uint32_t ck_cnt[kFusedCheckpoints];             // streaming waves that have left their summary for checkpoint j
    uint32_t wsum[kScanBlock / 64];                 // ... the summaries (each wave's M-th best score key)
    uint32_t wcount[kScanBlock / 64];
    uint32_t nfin, nmine, ok, ticket;
    uint32_t nitems[4];             // selectors: items listed for round r at [r % 4]
    uint32_t hmin, hmax;            // publish: range of the workgroup's score keys
    uint32_t repbin;                // ... the bucket its report lies in (kFusedBins: none)
    uint32_t repabove;              // ... the rows in higher buckets
    u64 repmin;                     // ... the report
    uint32_t rn[kFusedSelectors];   // selectors: entries | bucket shift << 16 | exact order << 31 of every region
    union {
        uint32_t items[2][kFusedItems]; // selectors: further reads, 64 entries each (round r in [r % 2]):
                                        // region | first entry / 16 << 8 | (entries - 1) << 17 | last item of its region in this round << 23
        uint32_t hist[kFusedBins];      // publish: rows per bucket, then each bucket's next position in the list
        struct {                        // selectors, ranking many finalists by bucket:
            uint32_t hist[kFusedBins];  //   finalists per bucket of the 64-bit key, then the finalists in higher buckets
            uint32_t head[kFusedBins];  //   the first of this selector's rows in the bucket (+ 1); they are chained
            uint32_t queue[kScanBlock / 64][128]; // per wave: (finalist, row of this selector in its bucket) pairs to compare
        } rk;
    };
};

static_assert(sizeof(FusedShared) <= 160 * 1024, "FusedShared exceeds the LDS of a CU");
// the packed words of the select phase
static_assert(kFusedSelectors <= 256 && kFusedRegion <= 8192, "item = region (8 bits) | first entry / 16 (9 bits) | entries - 1 (6 bits) | last (1 bit)");
static_assert(kFusedRegion <= 0xFFFF, "rn = entries (16 bits) | bucket shift (5 bits) << 16 | exact order << 31");
static_assert(kFusedFinalLds <= (1 << 14) && kFusedMineCap < (1 << 18), "slot of a finalist: 14 bits (| count << 14 in a node, | node << 14 in a queue entry)");
static_assert(kFusedItems >= 4 * kFusedSelectors, "a round lists at most four items per region");

__device__ __forceinline__ uint32_t agent_load(const uint32_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave-wide max, DPP within the 16-lane rows and four readlanes (a shuffle chain costs ~700 cycles)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    uint32_t o;
    o = dpp<0xB1>(v);  v = o > v ? o : v;
    o = dpp<0x4E>(v);  v = o > v ? o : v;
    o = dpp<0x141>(v); v = o > v ? o : v;
    o = dpp<0x140>(v); v = o > v ? o : v;
    const uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// wave-wide sum, same shape
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v)
{
    v += dpp<0xB1>(v);
    v += dpp<0x4E>(v);
    v += dpp<0x141>(v);
    v += dpp<0x140>(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

// checkpoint j is due after 4^j trips; the last one after 3/4 of the trips every wave makes
struct FusedSchedule {
    uint32_t min_trips, last_ck;
    __device__ __forceinline__ void init(uint32_t min_trips_)
    {
        min_trips = min_trips_;
        uint32_t np = 0;
        while (np < 14 && (1u << (2 * np)) < min_trips) np++;
        last_ck = np;
    }
    __device__ __forceinline__ uint32_t trip(uint32_t j) const
    {
        if (j < last_ck) return 1u << (2 * j);
        if (j == last_ck) {
            const uint32_t t = min_trips - min_trips / 4;
            return (min_trips >= 32 && t > (1u << (2 * (last_ck - 1)))) ? t : 0xFFFFFFFFu;
        }
        return 0xFFFFFFFFu;
    }
    __device__ __forceinline__ uint32_t inloop() const // number of checkpoints inside the streaming loop
    {
        return last_ck + (trip(last_ck) != 0xFFFFFFFFu ? 1u : 0u);
    }
    // Very small tables (up to 8 trips per wave, ~0.5 M 1024-bit rows): the in-loop thresholds come from a quarter of
    // the rows at best and arrive after the scan anyway -- one more checkpoint AFTER the loop, over all rows, costs the
    // same wait and leaves ~1.5 k rows to publish instead of 4-8 k (which few workgroups would share).
    __device__ __forceinline__ bool end_ck() const { return min_trips <= 8; }
    __device__ __forceinline__ uint32_t count() const { return inloop() + (end_ck() ? 1u : 0u); } // checkpoints in all
    // the checkpoint whose threshold a small table's workgroups wait for before they publish: the one after the loop
    // where there is one, else the last but one in the loop (9 ... 63 trips: the last one's election ends about when the
    // scan does -- waiting for it cost 3 us at 1 M rows, and the one before already leaves few enough rows)
    __device__ __forceinline__ uint32_t need() const { return end_ck() ? count() : (inloop() > 1u ? inloop() - 1u : inloop()); }
    // few trips: the scan may end before the last in-loop threshold has been elected (see fused_poller)
    __device__ __forceinline__ bool late() const { return min_trips < 64; }
    // (The workgroups do not finish together: the classes blockIdx % 8 = {0,1,2,7} and {3,4,5,6} -- two halves of
    // the chip -- end 3-4 % apart at 100 M rows, 10 % at 10 M, and WHICH half is the slow one changes from query to
    // query: contention, not a property of an XCD.  Remedies that were built and measured, none kept: per-class
    // shares of the table steered by the previous queries' times; handing out the table's tail dynamically; a shared
    // last quarter.  DESIGN.md 7.)
};

// A streaming wave's view.  Its loop touches global memory only through the table loads: the
// threshold comes from LDS (the service wave keeps it fresh), summaries go to LDS.  gfx950 counts
// loads, stores and atomics in ONE in-order counter, so a single global store or atomic inside
// the loop would drain the prefetch at the next wait (1-4 us each).
struct FusedFilter {
    static constexpr bool kFused = true;
    FusedShared* sh;
    QueryState* st;
    u64* skey;      // this wave's LDS store
    uint32_t* scb;
    uint32_t M, wv, w;
    uint32_t k, tau, staged, kept, emitted;
    float cutoff;
    bool has_cutoff, store_off;
    FusedSchedule sched;
    uint32_t next_ck, ck_j;
    u64* dbg;

    __device__ __forceinline__ uint32_t load_gtau() const { return 0u; } // (no polls from the streaming loop)

    __device__ __forceinline__ void refresh(uint32_t g, int lane)
    {
        const uint32_t t = __hip_atomic_load(&sh->tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (dbg && tau == 0 && (t | g) != 0 && lane == 0 && wv == 0) dbg[8] = wall_clock64();
        tau = t > tau ? t : tau;
        if (g > tau) {
            tau = g;
            if (lane == 0) atomicMax(&sh->tau, g);
        }
    }

    // The M-th best 64-bit key of this wave's store: "this wave holds M distinct rows at or above this key in the
    // canonical order" (0: fewer than M rows).  Every lane keeps the best four of the entries it visits, then M
    // rounds of wave-wide max + pop (a lane that holds more than four of the wave's M best under-reports: a
    // smaller key, for which the statement still holds).
    __device__ __forceinline__ u64 mth_best(int lane) const
    {
        u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        for (uint32_t i = lane; i < staged; i += 64) {
            const u64 v = skey[i];
            if (v > t3) {
                t3 = v;
                if (t3 > t2) { const u64 x = t2; t2 = t3; t3 = x; }
                if (t2 > t1) { const u64 x = t1; t1 = t2; t2 = x; }
                if (t1 > t0) { const u64 x = t0; t0 = t1; t1 = x; }
            }
        }
        u64 mth = 0;
        for (uint32_t r = 0; r < M; r++) {
            const uint32_t hi = wave_max_u32(static_cast<uint32_t>(t0 >> 32));
            const uint32_t lo = wave_max_u32(static_cast<uint32_t>(t0 >> 32) == hi ? static_cast<uint32_t>(t0) : 0u);
            mth = (static_cast<u64>(hi) << 32) | lo;
            const u64 b = __ballot(t0 == mth);
            if (lane == __builtin_ctzll(b)) {
                t0 = t1;
                t1 = t2;
                t2 = t3;
                t3 = 0;
            }
        }
        return mth;
    }

    // called once per trip of the streaming loop with the number of chunks this wave has finished
    __device__ __forceinline__ void checkpoint(uint32_t trips_done, int lane)
    {
        if (trips_done != next_ck) return;
        const u64 mth = mth_best(lane);
        if (lane == 0) {
            sh->wsum[wv] = static_cast<uint32_t>(mth >> 32); // the score key: 0 = fewer than M rows so far
            atomicAdd(&sh->ck_cnt[ck_j], 1u); // (LDS, after the summary: a wave's LDS operations execute in order)
        }
        if (dbg && lane == 0 && wv == 0 && ck_j == 0) dbg[9] = wall_clock64();
        if (dbg && lane == 0 && wv == 0) {
            if (ck_j + 1 == sched.inloop()) dbg[17] = wall_clock64(); // the last in-loop checkpoint (3/4 of the trips)
            else if (ck_j >= 1 && ck_j <= 5) dbg[17 + ck_j] = wall_clock64(); // after 4, 16, 64, 256, 1024 trips
        }
        ck_j++;
        next_ck = M ? sched.trip(ck_j) : 0xFFFFFFFFu;
    }

    // drop the stored rows below the current threshold, in place.  One wave; LDS operations of a
    // wave execute in order: a batch is read completely before its survivors are written at or
    // below the positions just read.
    __device__ __forceinline__ void compact_store(int lane)
    {
        uint32_t out = 0;
        for (uint32_t base = 0; base < staged; base += 64) {
            const uint32_t i = base + lane;
            const bool in = i < staged;
            const u64 key = in ? skey[i] : 0ull;
            const uint32_t cb = in ? scb[i] : 0u;
            const bool keep = in && static_cast<uint32_t>(key >> 32) >= tau;
            const u64 m = __ballot(keep);
            if (keep) {
                const uint32_t slot = out + lane_rank(m);
                skey[slot] = key;
                scb[slot] = cb;
            }
            out += static_cast<uint32_t>(__popcll(m));
        }
        staged = out;
    }

    // One row per lane (or an inactive lane).
    __device__ __forceinline__ void offer(bool active, uint32_t row, float raw_score, uint32_t cb, int lane)
    {
        const float s = apply_cutoff(raw_score, cutoff);
        const bool keep = active && (!has_cutoff || s != 0.0f);
        kept += keep ? 1u : 0u;
        const uint32_t okey = order_key(s);
        const bool cand = keep && okey >= tau;
        const u64 m = __ballot(cand);
        if (m == 0) return;
        const uint32_t n = static_cast<uint32_t>(__popcll(m));
        emitted += n;
        if (store_off) return;
        if (cand) {
            const uint32_t slot = staged + lane_rank(m);
            skey[slot] = (static_cast<u64>(okey) << 32) | static_cast<u64>(~row);
            scb[slot] = cb;
        }
        staged += n;
        if (staged > static_cast<uint32_t>(kFusedWaveCap - 64)) {
            refresh(agent_load(&st->gtau), lane);
            compact_store(lane);
            if (staged > static_cast<uint32_t>(kFusedWaveCap - 64)) {
                store_off = true; // ties / rows in ascending score order: the four-kernel pipeline takes the query
                if (lane == 0) __hip_atomic_store(&sh->overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
};

// The in-loop election: every wave reported its M-th best score key, so each of the r = ceil(k / M) largest
// reports stands for M distinct rows at or above it: at least k rows score at or above the r-th
// largest report, which is published as the threshold (to 15 leading bits, rounded down).  One wave.
__device__ __forceinline__ void fused_elect(FusedShared& sh, QueryState* st, uint32_t* summ, uint32_t nvals, uint32_t k,
                                            int lane, u64* dbg)
{
    if (dbg && lane == 0) dbg[10] = wall_clock64();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(summ, 0, nvals * 4u, 0x00020000);
    uint32_t v[64];
#pragma unroll
    for (int i = 0; i < 16; i++) { // (reads past nvals return 0)
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (i * 64 + lane) * 16, 0, /*sc1*/ 16);
        v[4 * i + 0] = x.x;
        v[4 * i + 1] = x.y;
        v[4 * i + 2] = x.z;
        v[4 * i + 3] = x.w;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (dbg && lane == 0) dbg[16] = wall_clock64();
    // (k here is the rank r.)  Keys of kept rows with a score in [0, 2) have bit 31 set and bit 30 clear; the selection runs
    // on bits 29..15 (the exponent and 8 bits of the mantissa), two 15-bit values per register.
    // Anything else is reported smaller than it is (negative scores as absent, scores >= 2 clamped):
    // under-reporting only lowers the threshold.  y >= c  <=>  bit 15 of (y + 0x8000 - c).
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    u16x2 y[32];
#pragma unroll
    for (int i = 0; i < 32; i++) {
        uint32_t q[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t x = v[2 * i + h];
            const uint32_t m = (x & 0x7FFFFFFFu) >> 15;
            q[h] = (x & 0x80000000u) ? (m > 0x7FFFu ? 0x7FFFu : m) : 0u;
        }
        y[i] = u16x2{static_cast<unsigned short>(q[0]), static_cast<unsigned short>(q[1])};
    }
    const uint32_t pairs = (nvals + 127) / 128; // registers in use per lane (values past nvals are 0)
    uint32_t p15 = 0;
#pragma unroll 1
    for (int bit = 14; bit >= 0; bit--) { // rolled: this code runs once per checkpoint, from a cold instruction cache
        const uint32_t cand = p15 | (1u << bit);
        const unsigned short kk = static_cast<unsigned short>(0x8000u - cand);
        const u16x2 kv{kk, kk};
        u16x2 c0{0, 0}, c1{0, 0};
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            if (static_cast<uint32_t>(gq * 8) < pairs) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    c0 += (y[8 * gq + i] + kv) >> 15;
                    c1 += (y[8 * gq + i + 1] + kv) >> 15;
                }
            }
        }
        const u16x2 cs = c0 + c1;
        const uint32_t c = static_cast<uint32_t>(cs.x) + static_cast<uint32_t>(cs.y);
        if (wave_sum_dpp(c) >= k) p15 = cand;
    }
    const uint32_t prefix = p15 ? (0x80000000u | (p15 << 15)) : 0u; // p15 == 0: fewer than r reports so far
    if (prefix != 0 && lane == 0) {
        atomicMax(&st->gtau, prefix);
        atomicMax(&sh.tau, prefix);
    }
    if (dbg && lane == 0) dbg[11] = wall_clock64();
}

// The service waves: everything of the in-loop threshold protocol that touches global memory.  Both leave when the
// workgroup's streaming waves are done (scan_done), so neither can outlive the scan.
//
// Wave 4 (forwarder): when the four streaming waves have left their summaries for checkpoint j,
// copies them (4 keys) to the table-wide array and takes the checkpoint's ticket -- two
// levels, one counter per XCD-sized group of workgroups (b % 8) and one on top, 128 bytes apart:
// 256 arrivals on one word serialise at ~12 ns each.  The last arriver hands the election to its poller.  The
// stores are not waited for: a slot read before its store lands holds smaller keys (older or
// zero), which only lowers the threshold.
// Wave 5 (poller): keeps the workgroup's LDS copy of the table-wide threshold fresh -- it polls
// every microsecond at first (a small table is over in 20) and backs off to one poll per ~60 us -- and runs the
// elections its forwarder wins.
__device__ __forceinline__ void fused_forwarder(FusedShared& sh, const FusedArgs& fa, const FusedSchedule& sched, int lane)
{
    const bool active = fa.summ_keys != 0 && !(fa.xflags & 2u);
    const uint32_t nck = active ? sched.count() : 0u;
    const uint32_t nwg = gridDim.x;
    const uint32_t x = blockIdx.x % 8u;
    const uint32_t group_size = (nwg - x + 7u) / 8u, ngroups = nwg < 8u ? nwg : 8u;
    // every wave of the grid reaches every scheduled checkpoint (the schedule is made from the FEWEST trips any wave
    // makes), so every checkpoint's ticket completes: this wave passes all of them on, also after the streaming loop
    for (uint32_t j = 0; j < nck;) {
        if (__hip_atomic_load(&sh.ck_cnt[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != kScanBlock / 64) {
            // (the early checkpoints are a few microseconds apart; from the fourth on this wave naps ~1.7 us at a time:
            // a wave that polls LDS every 64 clocks takes issue slots from the streaming wave on its SIMD)
            if (j >= 3) __builtin_amdgcn_s_sleep(64);
            else __builtin_amdgcn_s_sleep(1);
            continue;
        }
        if (lane < kScanBlock / 64)
            __hip_atomic_store(&fa.summ[static_cast<u64>(blockIdx.x) * (kScanBlock / 64) + lane], sh.wsum[lane], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        uint32_t* tk = fa.tickets + static_cast<size_t>(j) * 9 * 32;
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&tk[x * 32], 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t == group_size - 1) {
            if (lane == 0) t = atomicAdd(&tk[8 * 32], 1u);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t == ngroups - 1 && lane == 0) // the poller wave runs the election: this wave stays free for the next checkpoint
                atomicMax(&sh.elect_req, j + 1);
        }
        j++;
    }
    if (lane == 0) __hip_atomic_store(&sh.fwd_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Small tables (sched.late()): the last in-loop threshold takes ~12 us from its checkpoint to every workgroup and the
// scan may be over before that.  A workgroup that published against no threshold would publish all its rows, and
// every selector would wade through the whole table: on such tables the streaming waves wait, after their loop, until
// every in-loop election has been held (QueryState::elected) -- the poller stays and keeps the count fresh in LDS, for
// at most fa.wait_ticks (then the query is handed back).
__device__ __forceinline__ void fused_poller(FusedShared& sh, QueryState* st, const FusedArgs& fa, const FusedSchedule& sched,
                                             uint32_t nwaves, uint32_t k, int lane, u64* dbg)
{
    const bool active = fa.summ_keys != 0 && !(fa.xflags & 2u);
    const bool stay = active && sched.late();
    const unsigned long long t0 = wall_clock64();
    for (uint32_t spins = 0;; spins++) {
        const u64 ge = __hip_atomic_load(reinterpret_cast<const u64*>(&st->gtau), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // {gtau, elected}: one poll
        const uint32_t g = static_cast<uint32_t>(ge), el = static_cast<uint32_t>(ge >> 32);
        if (lane == 0) {
            if (g > __hip_atomic_load(&sh.tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) atomicMax(&sh.tau, g);
            __hip_atomic_store(&sh.elected, el, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); // (after the threshold it belongs to)
        }
        // a poll every ~2 us at first, every ~5 us from the 64th on, every ~60 us from the 512th on
        const uint32_t naps = spins < 512u ? 1u : 16u;
        for (uint32_t i = 0; i < naps; i++) {
            uint32_t req = 0; // (consumed with an exchange: a request stored between a plain load and a plain clear would be lost)
            if (lane == 0 && __hip_atomic_load(&sh.elect_req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) req = atomicExch(&sh.elect_req, 0u);
            req = __builtin_amdgcn_readfirstlane(req);
            if (req && active) {
                fused_elect(sh, st, fa.summ, nwaves, (k + fa.summ_keys - 1) / fa.summ_keys, lane, dbg);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the threshold is out before the count
                // (the HIGHEST checkpoint whose election has been held: one election may serve two requests that the same
                // workgroup won back to back, and only the last checkpoint's matters to those who wait)
                if (lane == 0) atomicMax(&st->elected, req);
                break; // (poll at once: this workgroup's own waves want the count too)
            }
            if (__hip_atomic_load(&sh.scan_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == kScanBlock / 64 &&
                __hip_atomic_load(&sh.fwd_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0 &&
                __hip_atomic_load(&sh.elect_req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
                // streaming over, every checkpoint forwarded, no election owed by this workgroup
                if (!stay || el >= sched.need()) return;
                if (wall_clock64() - t0 > fa.wait_ticks) { // (only when part of the grid cannot start: a shared GPU)
                    if (lane == 0) {
                        atomicOr(&st->redo, kRedoElectionWait);
                        __hip_atomic_store(&sh.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    return;
                }
            }
            if (spins < 64u) __builtin_amdgcn_s_sleep(8); // (units of 64 clocks: ~0.2 us; a small table is over in 20-50 us)
            else __builtin_amdgcn_s_sleep(127);                    // ~3.4 us
        }
    }
}

template <int LPR, int U>
__global__ __launch_bounds__(kFusedBlock) void fused_kernel(ScanArgs a, ScanGeometry g, FusedArgs fa)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fused_smem[];
    FusedShared& sh = *reinterpret_cast<FusedShared*>(fused_smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    QueryState* st = a.state;
    u64* dbg = fa.dbg ? fa.dbg + static_cast<u64>(blockIdx.x) * 24 : nullptr;
#define GSIM_STAMP(i) do { if (dbg && tid == 0) dbg[i] = wall_clock64(); } while (0)
    GSIM_STAMP(0);
    if (tid == 0) {
        sh.tau = 0;
        sh.overflow = 0;
        sh.hmin = ~0u;
        sh.hmax = 0u;
        sh.nemit = 0;
        sh.scan_done = 0;
        sh.elect_req = 0;
        sh.fwd_done = 0;
        sh.elected = 0;
        sh.abort = 0;
    }
    if (tid < kFusedCheckpoints) sh.ck_cnt[tid] = 0;
    __syncthreads();
    FusedSchedule sched;
    constexpr int CHR = U * (64 / LPR); // rows per chunk
    const u64 nfull = a.nrows / CHR;    // full chunks
    sched.init(static_cast<uint32_t>(nfull / g.nwaves));
    if (wv == kScanBlock / 64) {
        fused_forwarder(sh, fa, sched, lane);
        return;
    }
    if (wv == kScanBlock / 64 + 1) {
        fused_poller(sh, st, fa, sched, g.nwaves, a.k, lane, dbg);
        return;
    }
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * (kScanBlock / 64) + wv);
    const uint32_t nwg = gridDim.x;

    const u32x4 q = reinterpret_cast<const u32x4*>(a.query)[lane % LPR];
    FusedFilter f;
    f.sh = &sh;
    f.st = st;
    f.skey = sh.store.key[wv];
    f.scb = sh.store.cb[wv];
    f.M = fa.summ_keys;
    f.wv = static_cast<uint32_t>(wv);
    f.w = w;
    f.k = a.k;
    f.tau = 0;
    f.staged = 0;
    f.kept = 0;
    f.emitted = 0;
    f.cutoff = a.cutoff;
    f.has_cutoff = a.cutoff > 0.0f; // fingerprintdb_cuda.cu:263: compaction only if cutoff > 0
    f.store_off = false;
    f.sched = sched;
    f.ck_j = 0;
    f.next_ck = (f.M && !(fa.xflags & 2u)) ? sched.trip(0) : 0xFFFFFFFFu;
    f.dbg = dbg;
    scan_rows<LPR, U>(a, g, f, q, w, lane);
    if (sched.end_ck() && f.M && !(fa.xflags & 2u)) { // the checkpoint after the loop: this wave's M-th best over all its rows
        const u64 mth = f.mth_best(lane);
        if (lane == 0) {
            sh.wsum[wv] = static_cast<uint32_t>(mth >> 32);
            atomicAdd(&sh.ck_cnt[sched.inloop()], 1u);
        }
    }
    if (lane == 0) atomicAdd(&sh.scan_done, 1u); // (the service waves leave)
    if (f.has_cutoff) {
        const uint32_t tot = wave_sum(f.kept);
        if (lane == 0 && tot) atomicAdd(&st->kept, static_cast<u64>(tot));
    }
    if (dbg && lane == 0) dbg[12 + wv] = wall_clock64();
    if (dbg && lane == 0 && wv == 0) dbg[1] = wall_clock64();

    // ---- 3. publish: this workgroup's survivors and its end-of-scan report ---------------------
    // No exchange precedes it: the rows at or above the freshest in-loop threshold the workgroup has seen go into its
    // own region of the list (no reservation), in canonical order when there are few -- each row's position is the
    // number of larger keys in the workgroup -- and the row at position Mw - 1 is the workgroup's REPORT: "Mw distinct
    // rows of mine are at or above this 64-bit key".  The selectors derive the final threshold from the reports.
    if (sched.late() && fa.summ_keys != 0 && !(fa.xflags & 2u)) { // small table: the in-loop thresholds may still be on their way
        const uint32_t nck = sched.need();
        while (__hip_atomic_load(&sh.elected, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < nck &&
               __hip_atomic_load(&sh.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
            __builtin_amdgcn_s_sleep(4);
    }
    f.refresh(0u, lane); // (the service wave kept the workgroup's LDS copy of the threshold fresh: no global load here)
    if (!f.store_off) f.compact_store(lane);
    if (lane == 0) {
        sh.wcount[wv] = f.store_off ? 0u : f.staged;
        if (f.emitted) atomicAdd(&sh.nemit, f.emitted);
    }
    __syncthreads(); // (released once the service waves have exited too)
    GSIM_STAMP(2);
    const bool bad = __hip_atomic_load(&sh.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
    uint32_t ntot = 0;
#pragma unroll
    for (int i = 0; i < kScanBlock / 64; i++) {
        const uint32_t c = sh.wcount[i];
        ntot += c;
    }
    if (bad) ntot = 0;
    // Up to kFusedSortCap rows: canonical order, each row's position is the number of larger keys in the workgroup.  More
    // (a late threshold on a short table, a large k, series of analogs, ties): BUCKET order -- a counting sort by
    // (score key >> shift), 1024 buckets over the workgroup's range of score keys, highest bucket first, any order
    // inside a bucket.  Either way a selector reads a list from its head and stops at the first entry that proves the
    // rest lies below the final threshold: whatever a workgroup publishes beyond the finalists costs nobody a read
    // (unordered lists were read in full by every selector: k = 8192 on 1 M rows published 180 k rows, 200 us).
    const bool sorted = ntot <= kFusedSortCap;
    const uint32_t Mw = fa.final_keys; // rows a workgroup's report stands for (fused_final_keys)
    const __amdgpu_buffer_rsrc_t hrsrc_w = __builtin_amdgcn_make_buffer_rsrc(fa.hdr, 0, nwg * kFusedHeaderBytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        static_cast<unsigned char*>(fa.pub) + static_cast<size_t>(blockIdx.x) * (kFusedRegion * 16u), 0, kFusedRegion * 16u, 0x00020000);
    const uint32_t mine_n = bad ? 0u : f.staged;
    uint32_t shift = 0;
    if (sorted) {
        for (uint32_t i = lane; i < mine_n; i += 64) {
            const u64 key = f.skey[i];
            uint32_t pos = 0; // the number of larger keys (keys are unique)
#pragma unroll 1
            for (int w2 = 0; w2 < kScanBlock / 64; w2++) {
                const uint32_t cnt = sh.wcount[w2];
                const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(sh.store.key[w2]);
                for (uint32_t j = 0; j < cnt; j += 2) {
                    const ulonglong2 kk = k2[j >> 1];
                    pos += kk.x > key ? 1u : 0u;
                    pos += (j + 1 < cnt && kk.y > key) ? 1u : 0u;
                }
            }
            if (Mw && pos == Mw - 1u) // the workgroup's report: straight into its header (bytes 8..15)
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{static_cast<uint32_t>(key), static_cast<uint32_t>(key >> 32)}, hrsrc_w,
                                                      blockIdx.x * kFusedHeaderBytes + 8u, 0, /*sc1*/ 16);
            const u32x4 e{static_cast<uint32_t>(key), static_cast<uint32_t>(key >> 32), f.scb[i], 0u};
            __builtin_amdgcn_raw_buffer_store_b128(e, rsrc, pos * 16u, 0, /*sc1: write-through*/ 16);
        }
    } else {
        { // the range of the workgroup's score keys; the buckets' counters
            uint32_t lo = ~0u, hi = 0u;
            for (uint32_t i = lane; i < mine_n; i += 64) {
                const uint32_t h = static_cast<uint32_t>(f.skey[i] >> 32);
                lo = h < lo ? h : lo;
                hi = h > hi ? h : hi;
            }
            hi = wave_max_u32(hi);
            lo = ~wave_max_u32(~lo);
            if (lane == 0 && mine_n) {
                atomicMax(&sh.hmax, hi);
                atomicMin(&sh.hmin, lo);
            }
            for (uint32_t i = static_cast<uint32_t>(tid); i < kFusedBins; i += kScanBlock) sh.hist[i] = 0;
        }
        __syncthreads();
        const uint32_t hmin = sh.hmin, hmax = sh.hmax, span = hmax - hmin;
        const uint32_t bits = span ? 32u - static_cast<uint32_t>(__clz(static_cast<int>(span))) : 0u;
        shift = bits > 10u ? bits - 10u : 0u;
        if ((hmax >> shift) - (hmin >> shift) >= kFusedBins) shift++; // (span >> shift < 1024, the difference of the quotients may be one more)
        const uint32_t binbase = hmin >> shift;
        for (uint32_t i = lane; i < mine_n; i += 64) atomicAdd(&sh.hist[(static_cast<uint32_t>(f.skey[i] >> 32) >> shift) - binbase], 1u);
        __syncthreads();
        if (wv == 0) { // a bucket's rows follow those of every higher bucket; the report's bucket: where the count reaches Mw
            constexpr int PER = static_cast<int>(kFusedBins) / 64;
            uint32_t h[PER];
            uint32_t sm = 0;
#pragma unroll
            for (int i = 0; i < PER; i++) {
                h[i] = sh.hist[lane * PER + i];
                sm += h[i];
            }
            uint32_t incl = sm; // rows in the buckets of lanes >= lane
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = static_cast<uint32_t>(__shfl_down(static_cast<int>(incl), d, 64));
                if (lane + d < 64) incl += t;
            }
            uint32_t acc = incl - sm, rb = kFusedBins;
#pragma unroll
            for (int i = PER - 1; i >= 0; i--) {
                sh.hist[lane * PER + i] = acc;
                if (Mw && acc < Mw && acc + h[i] >= Mw) rb = static_cast<uint32_t>(lane * PER + i);
                acc += h[i];
            }
            const u64 m = __ballot(rb != kFusedBins);
            if (m == 0 ? lane == 0 : lane == __builtin_ctzll(m)) {
                sh.repbin = rb;
                sh.repabove = rb != kFusedBins ? sh.hist[rb] : 0u; // (this lane wrote it: the rows in higher buckets)
            }
        }
        __syncthreads();
        const uint32_t rb = sh.repbin;
        for (uint32_t i = lane; i < mine_n; i += 64) {
            const u64 key = f.skey[i];
            const uint32_t b = (static_cast<uint32_t>(key >> 32) >> shift) - binbase;
            const uint32_t pos = atomicAdd(&sh.hist[b], 1u);
            const u32x4 e{static_cast<uint32_t>(key), static_cast<uint32_t>(key >> 32), f.scb[i], 0u};
            __builtin_amdgcn_raw_buffer_store_b128(e, rsrc, pos * 16u, 0, /*sc1: write-through*/ 16);
        }
        if (wv == 0 && rb != kFusedBins) {
            // the report, the workgroup's Mw-th best key: the count of rows, from the top bucket down, reaches Mw in bucket
            // rb -- the (Mw - rows above)-th best of THAT bucket's rows (usually one or two; a table-wide tie: all of them)
            const uint32_t need = Mw - sh.repabove;
            u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
            for (int w2 = 0; w2 < kScanBlock / 64; w2++) {
                const uint32_t cnt = sh.wcount[w2];
                for (uint32_t i = lane; i < cnt; i += 64) {
                    const u64 v = sh.store.key[w2][i];
                    if ((static_cast<uint32_t>(v >> 32) >> shift) - binbase == rb && v > t3) {
                        t3 = v;
                        if (t3 > t2) { const u64 x = t2; t2 = t3; t3 = x; }
                        if (t2 > t1) { const u64 x = t1; t1 = t2; t2 = x; }
                        if (t1 > t0) { const u64 x = t0; t0 = t1; t1 = x; }
                    }
                }
            }
            u64 mth = 0;
            for (uint32_t rr = 0; rr < need; rr++) { // (a lane holding more than four of the best under-reports: still valid)
                const uint32_t hi = wave_max_u32(static_cast<uint32_t>(t0 >> 32));
                const uint32_t lo = wave_max_u32(static_cast<uint32_t>(t0 >> 32) == hi ? static_cast<uint32_t>(t0) : 0u);
                mth = (static_cast<u64>(hi) << 32) | lo;
                const u64 bm = __ballot(t0 == mth);
                if (lane == __builtin_ctzll(bm)) {
                    t0 = t1;
                    t1 = t2;
                    t2 = t3;
                    t3 = 0;
                }
            }
            if (lane == 0) sh.repmin = mth;
        }
        __syncthreads();
    }
    // the header: {entries | exact order << 31, bucket shift, report}.  In exact order the report was written above by the
    // thread that held it -- and not at all when the workgroup holds fewer than Mw rows: the selectors take a report as
    // present only if entries >= Mw, so a stale one from an earlier query is never read.
    if (tid == 0) {
        if (sorted) {
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{ntot | 0x80000000u, 0u}, hrsrc_w, blockIdx.x * kFusedHeaderBytes, 0, /*sc1*/ 16);
        } else {
            const u64 rep = sh.repmin;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{ntot, shift, static_cast<uint32_t>(rep), static_cast<uint32_t>(rep >> 32)}, hrsrc_w,
                                                   blockIdx.x * kFusedHeaderBytes, 0, /*sc1*/ 16);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave: its entries (and header parts) are out
    __syncthreads();
    if (tid == 0) {
        if (bad) atomicOr(&st->redo, kRedoStore);
        // The arrival, two levels (MI355X_MICROARCH.md "barrier-xcd"): a counter per group of workgroups b % 8 (the XCD a
        // block lands on, as observed -- only speed depends on it), the group's last arriver adds to the top counter,
        // the last of those raises one generation word per group.  Every workgroup then polls ITS group's word: 32
        // pollers per line instead of 256 on every line (a flat count polled by all cost 7-10 us after the last arrival).
        const uint32_t x = blockIdx.x % 8u;
        const uint32_t group_size = (nwg - x + 7u) / 8u, ngroups = nwg < 8u ? nwg : 8u;
        if (atomicAdd(&fa.arrive[x * 32u], 1u) == group_size - 1u && atomicAdd(&fa.arrive[8u * 32u], 1u) == ngroups - 1u) {
            for (uint32_t gq = 0; gq < 8u; gq++)
                __hip_atomic_store(&fa.arrive[(9u + gq) * 32u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        atomicAdd(&st->ncand, static_cast<u64>(sh.nemit)); // (statistics: nobody waits for it)
    }
    GSIM_STAMP(3);

    // ---- 4. select: every workgroup of the grid (fused_supported: at most kFusedSelectors) ------
    const uint32_t nsel = nwg, r = blockIdx.x;
    if (wv == 0) {
        uint32_t ok = 1;
        // On a GPU this kernel has to itself the wait is the spread of the streaming end times.  When another queue
        // holds part of the CUs, workgroups of this grid may not have started yet and will not while the waiters keep
        // theirs: after fa.wait_ticks (a few scan times) without the last arrival the query goes to the classic
        // kernels, which never wait.
        const unsigned long long t_wait = wall_clock64();
        const uint32_t* gen = &fa.arrive[(9u + blockIdx.x % 8u) * 32u];
        for (uint32_t spins = 0;; spins++) {
            if (agent_load(gen) != 0) break;
            __builtin_amdgcn_s_sleep(2);
            if ((spins & 255u) == 255u && wall_clock64() - t_wait > fa.wait_ticks) {
                ok = 0;
                if (lane == 0) atomicOr(&st->redo, kRedoArrivalWait);
                break;
            }
        }
        if (lane == 0) {
            sh.ok = (ok && agent_load(&st->redo) == 0) ? 1u : 0u; // (one reader: the value is the same for the whole workgroup)
            sh.nfin = 0;
            sh.nmine = 0;
            sh.nitems[0] = 0;
            sh.nitems[1] = 0;
            sh.repmin = 0ull; // (from here on: the finalists' summed distance from the threshold)
            sh.tauf = 0ull;
        }
    }
    // (no acquire fence: everything read below was stored write-through and is read with sc1 loads, past the L1)
    __syncthreads();
    GSIM_STAMP(4);
    // ONE round trip: the header of region `tid` and the first kFusedPrefix entries of every region are requested
    // together, before anything is known about them (regions hold last query's rows beyond their count).  The entries
    // go straight into LDS (global_load_lds, 16 B per lane, no registers), so that the code that filters them stays
    // small: it is fetched cold in every launch.  Slot s = 16 g + p of the staging area receives entry (p - g) mod 16
    // of region g: thread g later walks ITS region's entries, and the rotation spreads the 64 lanes over all banks.
    // The staging area is the upper half of the finalist array: at most 4096 staged entries become finalists.
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(fa.pub, 0, nwg * (kFusedRegion * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(fa.hdr, 0, nwg * kFusedHeaderBytes, 0x00020000);
    constexpr int PL = static_cast<int>(kFusedPrefix);
    u32x4* staging = reinterpret_cast<u32x4*>(&sh.sel.fkey[kFusedFinalLds / 2]);
    // A grid of fewer than 129 workgroups (small tables) leaves threads to spare: S = 2, 4, ... threads share a region,
    // each taking sixteen consecutive entries of it ("virtual region" v = S g + part), so that the requested prefix is
    // 16 S entries -- the finalists per region grow as the grid shrinks.
    uint32_t lgS = 0;
    while ((nwg << (lgS + 1u)) <= static_cast<uint32_t>(kFusedSelectors)) lgS++;
    const uint32_t my_region = static_cast<uint32_t>(tid) >> lgS, my_part = static_cast<uint32_t>(tid) & ((1u << lgS) - 1u);
    const u32x4 hd = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, my_region * kFusedHeaderBytes, 0, /*sc1*/ 16); // (zeros past the grid)
    {
        const unsigned char* pubc = static_cast<const unsigned char*>(fa.pub);
#pragma unroll
        for (int u = 0; u < PL; u++) {
            const uint32_t slot = static_cast<uint32_t>(u * kScanBlock + tid);
            const uint32_t v = slot / kFusedPrefix, j = (slot - v) % kFusedPrefix; // virtual region, entry inside it
            const uint32_t gi = v >> lgS, ent = ((v & ((1u << lgS) - 1u)) * kFusedPrefix) + j;
            if (gi < nwg)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*) (pubc + static_cast<size_t>(gi) * (kFusedRegion * 16u) + ent * 16u),
                    (__attribute__((address_space(3))) void*) (staging + u * kScanBlock + wv * 64), 16, 0, /*sc1*/ 16);
        }
    }
    // (the header was requested first and loads return in order: the election below runs while the prefixes land)
    const uint32_t n_mine = (hd.x & 0x7FFFFFFFu) < kFusedRegion ? (hd.x & 0x7FFFFFFFu) : kFusedRegion; // entries of region my_region
    const bool sorted_mine = (hd.x >> 31) != 0;
    const u64 rep_mine = (Mw && n_mine >= Mw && my_part == 0) ? ((static_cast<u64>(hd.w) << 32) | hd.z) : 0ull; // (one thread per region holds its report)
    sh.sel.u.rep[tid] = rep_mine;
    __syncthreads(); // the reports of all regions
    const bool good0 = sh.ok != 0; // (not good: headers and regions may be stale -- nothing below is used, the query is handed back)
    // The final threshold: the r-th largest of the workgroups' reports, r = ceil(k / Mw).  Each of the r largest
    // reports stands for Mw distinct rows at or above it in the canonical order, so at least k rows are at or above
    // the r-th largest: no row of the top k lies below it.  The keys carry the row index: the threshold also cuts
    // through a group of equal scores.  Every thread ranks its region's report by counting larger ones (256 broadcast
    // reads); the thread whose report has rank r - 1 publishes it.  Every selector finds the same value.
    if (good0 && Mw) {
        const uint32_t rr = (a.k + Mw - 1u) / Mw;
        const ulonglong2* r2 = reinterpret_cast<const ulonglong2*>(sh.sel.u.rep);
        uint32_t rank = 0;
#pragma unroll 8
        for (int j = 0; j < kFusedSelectors / 2; j++) {
            const ulonglong2 kk = r2[j];
            rank += kk.x > rep_mine ? 1u : 0u;
            rank += kk.y > rep_mine ? 1u : 0u;
        }
        if (rep_mine != 0ull && rank == rr - 1u) sh.tauf = rep_mine;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's part of the prefixes is in LDS
    __syncthreads();                                 // ... and everybody's; the threshold is known
    const u64 tauf = good0 ? sh.tauf : ~0ull;
    if (dbg && tid == 0) dbg[23] = wall_clock64();
    // finalists = the published rows at or above the final threshold -> LDS.  Thread g takes region g's staged entries
    // (all sixteen reads issued at once); the rows this selector owns (a hash of the row) are noted with their popcounts.
    bool good = good0;
    // Every list is in order -- exact (canonical) or by bucket: an entry that lies below the threshold (exact order), or
    // in a lower bucket than the threshold does (bucket order), proves that everything behind it is below the threshold.
    const uint32_t shift_mine = sorted_mine ? 0u : (hd.y & 31u);
    const uint32_t tauf_hi = static_cast<uint32_t>(tauf >> 32);
    auto stops = [&](u64 key, bool exact, uint32_t shift) -> bool {
        return exact ? key < tauf : (static_cast<uint32_t>(key >> 32) >> shift) < (tauf_hi >> shift);
    };
    const uint32_t pre_all = kFusedPrefix << lgS; // entries of a region that were requested
    u64 dacc = 0; // sum over the finalists this thread lists of (score key - the threshold's): scales the ranking's buckets
    auto take = [&](bool in, const u32x4& ent) { // one published row per lane -> the finalists, if it is at or above the threshold
        const u64 key = (static_cast<u64>(ent.y) << 32) | ent.x;
        const bool pass = in && key >= tauf;
        dacc += pass ? (key - tauf) >> 32 : 0ull;
        const u64 m = __ballot(pass);
        if (m == 0) return;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&sh.nfin, static_cast<uint32_t>(__popcll(m)));
        base = __builtin_amdgcn_readfirstlane(base);
        const uint32_t slot = base + lane_rank(m);
        if (pass && slot < static_cast<uint32_t>(kFusedFinalLds)) {
            sh.sel.fkey[slot] = key;
            if ((((~ent.x * 2654435761u) >> 16) * nsel) >> 16 == r) {
                const uint32_t mp = atomicAdd(&sh.nmine, 1u);
                if (mp < static_cast<uint32_t>(kFusedMineCap)) {
                    sh.sel.u.mine.idx[mp] = slot;
                    sh.sel.u.mine.cb[mp] = ent.z;
                }
            }
        }
    };
    {
        // finalists = the published rows at or above the final threshold -> LDS.  Thread g takes region g's staged entries
        // (all sixteen reads issued at once); the rows this selector owns (a hash of the row) are noted with their popcounts.
        const uint32_t g16 = static_cast<uint32_t>(tid) * kFusedPrefix;
        const uint32_t first = my_part * kFusedPrefix; // this thread's sixteen entries of the region: first .. first + 15
        const uint32_t npre = (good0 && n_mine > first) ? (n_mine - first < kFusedPrefix ? n_mine - first : kFusedPrefix) : 0u;
        u32x4 ev[PL];
#pragma unroll
        for (int j = 0; j < PL; j++) ev[j] = staging[g16 + ((static_cast<uint32_t>(j) + static_cast<uint32_t>(tid)) % kFusedPrefix)];
        uint32_t passm = 0, stopm = 0; // bit j: entry j is a finalist / ends the list's part at or above the threshold
#pragma unroll
        for (int j = 0; j < PL; j++) {
            const u64 key = (static_cast<u64>(ev[j].y) << 32) | ev[j].x;
            const bool pass = static_cast<uint32_t>(j) < npre && key >= tauf;
            passm |= pass ? (1u << j) : 0u;
            dacc += pass ? (key - tauf) >> 32 : 0ull;
            stopm |= (static_cast<uint32_t>(j) < npre && stops(key, sorted_mine, shift_mine)) ? (1u << j) : 0u;
        }
        const uint32_t cnt = static_cast<uint32_t>(__popc(passm));
        auto wave_scan = [&](uint32_t v, uint32_t& tot) -> uint32_t { // inclusive prefix sum over the wave, and the total
            uint32_t incl = v;
            { uint32_t o; o = dpp_shr<1>(incl); incl += o; o = dpp_shr<2>(incl); incl += o; o = dpp_shr<4>(incl); incl += o; o = dpp_shr<8>(incl); incl += o; }
            const uint32_t row_tot0 = __builtin_amdgcn_readlane(incl, 15), row_tot1 = __builtin_amdgcn_readlane(incl, 31),
                           row_tot2 = __builtin_amdgcn_readlane(incl, 47), row_tot3 = __builtin_amdgcn_readlane(incl, 63);
            const int rowi = lane >> 4;
            incl += (rowi > 0 ? row_tot0 : 0u) + (rowi > 1 ? row_tot1 : 0u) + (rowi > 2 ? row_tot2 : 0u);
            tot = row_tot0 + row_tot1 + row_tot2 + row_tot3;
            return incl;
        };
        // more rows of this region may qualify: its list is longer than the requested prefix and the prefix's last part
        // holds no entry that ends it.  The next 64 entries become an item of round 0 (below).
        const bool more = good0 && my_part == (1u << lgS) - 1u && n_mine > pre_all && stopm == 0;
        uint32_t wtot, wtot2;
        const uint32_t incl = wave_scan(cnt, wtot), incl2 = wave_scan(more ? 1u : 0u, wtot2);
        uint32_t base = 0, base2 = 0;
        if (lane == 0 && wtot) base = atomicAdd(&sh.nfin, wtot); // (one LDS atomic per wave and list, not one per lane)
        if (lane == 0 && wtot2) base2 = atomicAdd(&sh.nitems[0], wtot2);
        base = __builtin_amdgcn_readfirstlane(base);
        base2 = __builtin_amdgcn_readfirstlane(base2);
        if (my_part == 0) sh.rn[my_region] = n_mine | (shift_mine << 16) | (sorted_mine ? 0x80000000u : 0u);
        if (more) {
            const uint32_t left = n_mine - pre_all;
            sh.items[0][base2 + incl2 - 1u] = my_region | ((pre_all / 16u) << 8) | (((left < 64u ? left : 64u) - 1u) << 17) | (1u << 23);
        }
        const uint32_t slot0 = base + incl - cnt;
#pragma unroll
        for (int j = 0; j < PL; j++) {
            if (passm & (1u << j)) {
                const uint32_t slot = slot0 + static_cast<uint32_t>(__popc(passm & ((1u << j) - 1u)));
                sh.sel.fkey[slot] = (static_cast<u64>(ev[j].y) << 32) | ev[j].x; // (< 4096: below the staging area)
                if ((((~ev[j].x * 2654435761u) >> 16) * nsel) >> 16 == r) { // this selector ranks it
                    const uint32_t mp = atomicAdd(&sh.nmine, 1u);
                    if (mp < static_cast<uint32_t>(kFusedMineCap)) {
                        sh.sel.u.mine.idx[mp] = slot;
                        sh.sel.u.mine.cb[mp] = ev[j].z;
                    }
                }
            }
        }
    }
    {
        // Lists read beyond their prefix (a large k, series of analogs in neighbouring rows, ties), in rounds.  An item is 64
        // entries of one region (one per lane); every wave takes every fourth item of the round's list, eight at a time
        // with the eight loads in flight together: 32 items per round trip, whichever regions they belong to.  A region's
        // last item of a round, if it holds no entry that ends the list, lists the region's items of the next round:
        // as many entries again as have been read beyond the prefix, at most 4 items (the list holds 4 per region).
        // item = region | first entry / 16 << 8 | (entries - 1) << 17 | last of its region << 23.
        constexpr int IF = 8;
#pragma unroll 1
        for (uint32_t round = 0;; round++) {
            if (tid == 0) sh.nitems[(round + 2u) % 4u] = 0; // (last read two rounds ago -- every wave is past that --, appended to in the next round)
            __syncthreads(); // this round's items and their number (the first time: and the finalists of the prefixes)
            const uint32_t nit = sh.nitems[round % 4u];
            if (nit == 0) break;
            const uint32_t* cur = sh.items[round & 1u];
            uint32_t* nxt = sh.items[(round + 1u) & 1u];
#pragma unroll 1
            for (uint32_t i0 = static_cast<uint32_t>(wv); i0 < nit; i0 += 4u * IF) {
                u32x4 x[IF];
                uint32_t itm[IF];
                uint32_t lim = 0;
#pragma unroll
                for (int u = 0; u < IF; u++) {
                    const uint32_t idx = i0 + 4u * static_cast<uint32_t>(u);
                    itm[u] = cur[idx < nit ? idx : i0]; // (past the list: this wave's first item again, not taken)
                    const uint32_t start = ((itm[u] >> 8) & 0x1FFu) * 16u, cnt = ((itm[u] >> 17) & 63u) + 1u;
                    lim |= (idx < nit && static_cast<uint32_t>(lane) < cnt) ? (1u << u) : 0u;
                    x[u] = __builtin_amdgcn_raw_buffer_load_b128(prsrc, (itm[u] & 0xFFu) * (kFusedRegion * 16u) + (start + static_cast<uint32_t>(lane)) * 16u, 0, /*sc1*/ 16);
                }
#pragma unroll
                for (int u = 0; u < IF; u++) {
                    const bool in = ((lim >> u) & 1u) != 0;
                    take(in, x[u]);
                    if (i0 + 4u * static_cast<uint32_t>(u) < nit && (itm[u] >> 23) != 0) { // (wave-uniform) the region's last item of this round
                        const uint32_t reg = itm[u] & 0xFFu, rnv = sh.rn[reg];
                        const uint32_t n_g = rnv & 0xFFFFu, end = ((itm[u] >> 8) & 0x1FFu) * 16u + ((itm[u] >> 17) & 63u) + 1u;
                        const u64 key = (static_cast<u64>(x[u].y) << 32) | x[u].x;
                        const bool stop = __ballot(in && stops(key, (rnv >> 31) != 0, (rnv >> 16) & 31u)) != 0;
                        if (!stop && end < n_g) {
                            uint32_t ni = (end - pre_all) / 64u; // as many entries again as read so far beyond the prefix
                            const uint32_t left = (n_g - end + 63u) / 64u;
                            ni = ni < 1u ? 1u : (ni > 4u ? 4u : ni);
                            ni = ni < left ? ni : left;
                            uint32_t at = 0;
                            if (lane == 0) at = atomicAdd(&sh.nitems[(round + 1u) % 4u], ni);
                            at = __builtin_amdgcn_readfirstlane(at);
                            if (static_cast<uint32_t>(lane) < ni) {
                                const uint32_t st0 = end + static_cast<uint32_t>(lane) * 64u;
                                const uint32_t c = n_g - st0 < 64u ? n_g - st0 : 64u;
                                nxt[at + lane] = reg | ((st0 / 16u) << 8) | ((c - 1u) << 17) | (static_cast<uint32_t>(lane) == ni - 1u ? (1u << 23) : 0u);
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    const uint32_t nfin = sh.nfin;
    uint32_t why = good ? 0u : kRedoSeen;
    if (good && nfin > static_cast<uint32_t>(kFusedFinalLds)) why = kRedoFinalists;
    good = good && nfin <= static_cast<uint32_t>(kFusedFinalLds);
    if (good) {
        if (tid == 0 && (nfin & 1u)) sh.sel.fkey[nfin] = 0ull; // pad to a pair for the b128 reads (nfin < kFusedFinalLds or even)
        __syncthreads();
        GSIM_STAMP(5);
        const uint32_t nmine = sh.nmine;
        good = nmine <= static_cast<uint32_t>(kFusedMineCap);
        if (!good) why = kRedoOwned;
        if (good) {
            gsim_result_header* hdr = reinterpret_cast<gsim_result_header*>(fa.result);
            gsim_hit* hits = reinterpret_cast<gsim_hit*>(hdr + 1);
            // (system-scope write-through stores: nothing of the block stays behind in this XCD's L2, no write-back is owed
            // before the ticket -- the wait for their acknowledgement is the release)
            const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(hits, 0, a.k * 12u, 0x00020000);
            auto write_hit = [&](u64 mine, uint32_t rank, uint32_t cb) {
                const uint32_t w0 = ~static_cast<uint32_t>(mine) + fa.row_base;
                const uint32_t w1 = __float_as_uint(key_score(static_cast<uint32_t>(mine >> 32)));
                const uint32_t w2 = (cb >> 16) | (cb << 16); // {common, popc_db}
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{w0, w1}, hrs, rank * 12u, 0, /*sc0 sc1*/ 17);
                __builtin_amdgcn_raw_buffer_store_b32(w2, hrs, rank * 12u + 8u, 0, /*sc0 sc1*/ 17);
            };
            // (the bucket route keeps a 16-byte node per row of this selector in the unused end of the finalist array)
            const bool by_bucket = nfin > kFusedRankDirect && nfin + 2u * nmine + 2u <= static_cast<uint32_t>(kFusedFinalLds);
            if (by_bucket) {
                // Many finalists (a large k): comparing each of this selector's rows with every finalist is nfin^2 / #selectors
                // 64-bit compares per selector (k = 8192: 40 us).  Instead: a histogram of the finalists over 1024 buckets of
                // the 64-bit key between the threshold and the largest key; rank = finalists in higher buckets + larger keys
                // in the row's own bucket, the latter counted in ONE pass over the finalists -- each looks up whether its
                // bucket holds rows of this selector (chained per bucket) and is compared with those only.
                // (the buckets: 1023 equal steps of the key from the threshold to four times the finalists' mean distance
                // from it, and one for everything above -- the scores thin out quickly above the threshold, and the
                // largest key, the query's own row, is far away: steps up to IT left 95 % of the finalists in 60 buckets)
                const u64 base = tauf; // (every finalist is at or above the threshold)
                { // the summed distance, wave by wave (three 16-bit slices: each sums to less than 2^22 over the wave)
                    const u64 tot = static_cast<u64>(wave_sum(static_cast<uint32_t>(dacc) & 0xFFFFu)) + (static_cast<u64>(wave_sum(static_cast<uint32_t>(dacc >> 16) & 0xFFFFu)) << 16) +
                                    (static_cast<u64>(wave_sum(static_cast<uint32_t>(dacc >> 32) & 0xFFFFu)) << 32);
                    if (lane == 0) atomicAdd(&sh.repmin, tot); // (zero since the selectors' start)
                }
                for (uint32_t i = static_cast<uint32_t>(tid); i < kFusedBins; i += kScanBlock) { // (the items are done with)
                    sh.rk.hist[i] = 0;
                    sh.rk.head[i] = 0;
                }
                __syncthreads();
                // 4 x the mean distance of the score keys from the threshold's, in 1023 steps of 2^(shift - 32)
                const u64 reach = (sh.repmin << 2) / nfin + 1ull;
                const uint32_t rbits = 64u - static_cast<uint32_t>(__clzll(static_cast<long long>(reach)));
                const uint32_t shift = 32u + (rbits > 10u ? rbits - 10u : 0u);
                auto bucket = [&](u64 key) -> uint32_t {
                    const u64 d = (key - base) >> shift;
                    return d < kFusedBins - 1u ? static_cast<uint32_t>(d) : kFusedBins - 1u;
                };
                // node t, 16 bytes from the array's end downwards: {the row's key, the bucket's next row + 1, larger keys in the bucket}
                u32x4* nodes = reinterpret_cast<u32x4*>(&sh.sel.fkey[kFusedFinalLds]);
                for (uint32_t t = static_cast<uint32_t>(tid); t < nmine; t += kScanBlock) {
                    const u64 key = sh.sel.fkey[sh.sel.u.mine.idx[t]];
                    const uint32_t before = atomicExch(&sh.rk.head[bucket(key)], t + 1u);
                    *(nodes - 1 - static_cast<int>(t)) = u32x4{static_cast<uint32_t>(key), static_cast<uint32_t>(key >> 32), before, 0u};
                }
                __syncthreads(); // the chains
                // A finalist whose bucket holds rows of this selector -- one in twenty -- is compared with them.  Walking the
                // chains where they are met kept whole waves waiting on a few lanes' dependent reads (18 us); the (finalist,
                // node) pairs go through a queue of the wave instead and are taken 64 at a time, every lane busy.
                {
                    uint32_t* q = sh.rk.queue[wv];
                    uint32_t qn = 0; // (wave-uniform)
                    auto drain = [&](bool all) {
                        while (qn >= 64u || (all && qn != 0u)) {
                            const uint32_t n = qn < 64u ? qn : 64u;
                            const bool has = static_cast<uint32_t>(lane) < n;
                            const uint32_t e = has ? q[qn - n + static_cast<uint32_t>(lane)] : 0u; // finalist | node << 14
                            __builtin_amdgcn_wave_barrier();
                            qn -= n;
                            uint32_t onward = 0;
                            if (has) {
                                u32x4* nd = nodes - static_cast<int>(e >> 14);
                                const u32x4 node = *nd;
                                if (sh.sel.fkey[e & 0x3FFFu] > ((static_cast<u64>(node.y) << 32) | node.x)) atomicAdd(reinterpret_cast<uint32_t*>(nd) + 3, 1u);
                                onward = node.z;
                            }
                            const u64 m = __ballot(onward != 0u);
                            if (onward) q[qn + lane_rank(m)] = (e & 0x3FFFu) | (onward << 14);
                            qn += static_cast<uint32_t>(__popcll(m));
                            __builtin_amdgcn_wave_barrier();
                        }
                    };
                    for (uint32_t j0 = static_cast<uint32_t>(wv) * 64u; j0 < nfin; j0 += kScanBlock) {
                        const uint32_t j = j0 + static_cast<uint32_t>(lane);
                        uint32_t at = 0;
                        if (j < nfin) {
                            const uint32_t bk = bucket(sh.sel.fkey[j]);
                            atomicAdd(&sh.rk.hist[bk], 1u);
                            at = sh.rk.head[bk];
                        }
                        const u64 m = __ballot(at != 0u);
                        if (at) q[qn + lane_rank(m)] = j | (at << 14);
                        qn += static_cast<uint32_t>(__popcll(m));
                        __builtin_amdgcn_wave_barrier();
                        drain(false);
                    }
                    drain(true);
                }
                __syncthreads();
                if (wv == 0) { // hist[b] <- the finalists in buckets above b
                    constexpr int PER = static_cast<int>(kFusedBins) / 64;
                    uint32_t h[PER];
                    uint32_t sm = 0;
#pragma unroll
                    for (int i = 0; i < PER; i++) {
                        h[i] = sh.rk.hist[lane * PER + i];
                        sm += h[i];
                    }
                    uint32_t incl = sm;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t t = static_cast<uint32_t>(__shfl_down(static_cast<int>(incl), d, 64));
                        if (lane + d < 64) incl += t;
                    }
                    uint32_t acc = incl - sm;
#pragma unroll
                    for (int i = PER - 1; i >= 0; i--) {
                        sh.rk.hist[lane * PER + i] = acc;
                        acc += h[i];
                    }
                }
                __syncthreads();
                for (uint32_t t = static_cast<uint32_t>(tid); t < nmine; t += kScanBlock) {
                    const u32x4 node = *(nodes - 1 - static_cast<int>(t));
                    const u64 mine = (static_cast<u64>(node.y) << 32) | node.x;
                    const uint32_t rank = sh.rk.hist[bucket(mine)] + node.w;
                    if (rank < a.k) write_hit(mine, rank, sh.sel.u.mine.cb[t]);
                }
            }
            const uint32_t npair = (nfin + 1u) >> 1;
            const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(sh.sel.fkey);
            // RG lanes share one row: each counts the larger keys among every RG-th pair
            // (ds_read_b128, two keys per read, several reads in flight), then a shuffle sum
            constexpr int RG = 16;
            const uint32_t sub = static_cast<uint32_t>(tid % RG);
            for (uint32_t t0 = 0; t0 < nmine && !by_bucket; t0 += kScanBlock / RG) {
                const uint32_t t = t0 + static_cast<uint32_t>(tid / RG);
                const bool have = t < nmine;
                const u64 mine = have ? sh.sel.fkey[sh.sel.u.mine.idx[t]] : ~0ull;
                uint32_t rank = 0;
                for (uint32_t j0 = sub; j0 < npair; j0 += RG * 8) { // eight reads in flight
                    ulonglong2 kk[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t j = j0 + u * RG;
                        kk[u] = k2[j < npair ? j : npair - 1];
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const bool in = j0 + u * RG < npair;
                        rank += (in && kk[u].x > mine) ? 1u : 0u;
                        rank += (in && kk[u].y > mine) ? 1u : 0u;
                    }
                }
#pragma unroll
                for (int d = RG / 2; d > 0; d >>= 1) rank += static_cast<uint32_t>(__shfl_xor(static_cast<int>(rank), d, 64));
                if (have && sub == 0 && rank < a.k) write_hit(mine, rank, sh.sel.u.mine.cb[t]);
            }
        }
    }
    if (!good && tid == 0) atomicOr(&st->redo, why);
    // ---- 5. the last selector closes the query -----------------------------------------------
    GSIM_STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every wave: its hits have left the CU
    __syncthreads();
    if (tid == 0) {
        // The hits were stored write-through at system scope (sc0 sc1) and every wave has waited for their
        // acknowledgements: they are in memory, there is nothing for a release fence to write back.  (Plain stores need
        // the fence -- 16 of 600 k queries came back incomplete without it, DESIGN.md 7 (g) -- and it cost 1.3 us per
        // query.  GSIM_FUSED_FLAGS=1024 puts it back.)
        if (fa.xflags & 1024u) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // the ticket also carries "this selector saw the query fail" (bit 16 up): the closer learns it without another
        // round trip (a workgroup that set QueryState::redo while publishing did so before the grid-wide wait: every
        // selector read it after the wait and is not `good`)
        // Two levels, as the arrival: 256 atomics on ONE word queue up behind each other (the last ticket came 3.3 us
        // after the average selector was ready); a ticket per group b % 8 first, the group's last adds to the top.
        const uint32_t x = blockIdx.x % 8u;
        const uint32_t group_size = (nwg - x + 7u) / 8u, ngroups = nwg < 8u ? nwg : 8u;
        const uint32_t tg = atomicAdd(&fa.arrive[(17u + x) * 32u], good ? 1u : 0x10001u);
        uint32_t closing = 0, failed = 0;
        if ((tg & 0xFFFFu) == group_size - 1u) {
            const bool gfail = (tg >> 16) != 0 || !good;
            const uint32_t tt = atomicAdd(&st->sel_done, gfail ? 0x10001u : 1u);
            closing = (tt & 0xFFFFu) == ngroups - 1u ? 1u : 0u;
            failed = ((tt >> 16) != 0 || gfail) ? 1u : 0u;
        }
        sh.ticket = closing | (failed << 1);
    }
    __syncthreads();
    GSIM_STAMP(7);
    if (!(sh.ticket & 1u)) return;
    const uint32_t redo = ((sh.ticket & 2u) != 0 || !good) ? 1u : 0u;
    if (redo && tid == 0) atomicOr(&st->redo, kRedoSeen); // (the gated classic kernels behind an enqueue-only launch read it)
    if (tid == 0) {
        { // the header, write-through as the hits
            const u64 approx = a.cutoff > 0.0f ? __hip_atomic_load(&st->kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.nrows;
            const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(fa.result, 0, 16, 0x00020000);
            // The block is complete (every selector waited for its hits before its ticket): for a synchronous caller the
            // header carries the query's epoch -- the host polls it, one 16-byte write tells it everything -- and the
            // tidying up below happens behind the caller's back.
            const uint32_t flags = (redo ? 2u : 0u) | (fa.done_flag ? fa.epoch << 8 : 0u);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{redo ? 0u : (nfin < a.k ? nfin : a.k), flags, static_cast<uint32_t>(approx), static_cast<uint32_t>(approx >> 32)},
                                                   rrs, 0, 0, /*sc0 sc1*/ 17);
        }
        // re-zero the per-query state for the next launch (stream-ordered behind this one)
        st->ncand_sum += __hip_atomic_load(&st->ncand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st->nfinal_sum += redo ? 0u : nfin;
        st->queries += redo ? 0u : 1u;
        st->redo_sum += redo ? 1u : 0u;
        if (redo) st->redo_why |= __hip_atomic_load(&st->redo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->kept, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->ncand, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->gtau, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->elected, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->sel_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // synchronous callers read the hand-back from the header (no gated kernels behind this launch): the next
        // launch, possibly already enqueued, starts clean
        if (fa.done_flag) __hip_atomic_store(&st->redo, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < kFusedCheckpoints * 9) fa.tickets[tid * 32] = 0;
    if (tid < static_cast<int>(kFusedArriveWords)) fa.arrive[tid * 32] = 0;
    { // the in-loop summaries: zero again for the next query (16-byte stores)
        uint4* sm = reinterpret_cast<uint4*>(fa.summ);
        const uint32_t n16 = (g.nwaves + 3) / 4;
        for (uint32_t i = tid; i < n16; i += kScanBlock) sm[i] = uint4{0, 0, 0, 0};
    }
    if (dbg && tid == 0) fa.dbg[static_cast<u64>(gridDim.x) * 24] = wall_clock64(); // the very end
#undef GSIM_STAMP
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
    for (int i = threadIdx.x; i < kScanBins; i += kScanBlock)
        if (s_hist[i]) atomicAdd(&a.state->ghist[i], s_hist[i]);
    // ticket: the last workgroup turns the histogram into tau0 and clears it
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(&a.state->done, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
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
    for (int i = threadIdx.x; i < kScanBins; i += kScanBlock) a.state->ghist[i] = 0;
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

// K0 for the generic widths: as sample_kernel, chunks through GenericChunk
__global__ __launch_bounds__(kScanBlock) void sample_generic_kernel(ScanArgs a, uint32_t R, uint32_t nsample, u64 stride_chunks)
{
    __shared__ uint32_t s_hist[kScanBins];
    __shared__ uint32_t s_last;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_words[];
    if (a.gate && *a.gate == 0) return;
    const int lane = threadIdx.x & 63;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t w = blockIdx.x * (kScanBlock / 64) + wv;
    GenericChunk ch;
    ch.init(a, R, s_words, wv);
    for (int i = threadIdx.x; i < kScanBins; i += kScanBlock) s_hist[i] = 0;
    __syncthreads();
    SampleFilter f;
    f.hist = s_hist;
    f.cutoff = a.cutoff;
    f.has_cutoff = a.cutoff > 0.0f;
    const uint32_t nw = gridDim.x * (kScanBlock / 64);
    for (uint32_t i = w; i < nsample; i += nw) {
        const u64 c = static_cast<u64>(i) * stride_chunks; // a full chunk by construction
        ch.load(c, lane);
        uint32_t cc, bb;
        ch.count(lane, cc, bb);
        f.offer(static_cast<uint32_t>(lane) < R, 0u, score_of(a.metric, a.alpha, a.beta, a.qpop, bb, cc), 0u, lane);
    }
    sample_publish(a, s_hist, s_last, lane);
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

// ---------------------------------------------------------------------------
// K3: exact select + sort of the finalists, result emission
// ---------------------------------------------------------------------------

__device__ __forceinline__ void emit_hit(const ScanArgs& a, u64 key, uint32_t row_base, gsim_hit* out)
{
    const uint32_t row = ~static_cast<uint32_t>(key);
    const float s = key_score(static_cast<uint32_t>(key >> 32));
    const uint32_t* r = reinterpret_cast<const uint32_t*>(a.rows) + static_cast<u64>(row) * a.W;
    uint32_t cc = 0, bb = 0;
    if ((a.W & 3u) == 0) { // 16-byte loads, all issued before the first use
        const uint4* r4 = reinterpret_cast<const uint4*>(r);
        const uint4* q4 = reinterpret_cast<const uint4*>(a.query_dev);
        const uint32_t n4 = a.W >> 2;
#pragma unroll 8
        for (uint32_t i = 0; i < n4; i++) {
            const uint4 x = r4[i], y = q4[i];
            cc += __popc(x.x & y.x) + __popc(x.y & y.y) + __popc(x.z & y.z) + __popc(x.w & y.w);
            bb += __popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w);
        }
    } else {
        for (uint32_t i = 0; i < a.W; i++) {
            const uint32_t x = r[i];
            cc += __popc(x & a.query_dev[i]);
            bb += __popc(x);
        }
    }
    gsim_hit h;
    h.row = row + row_base;
    h.score = s;
    h.common = static_cast<uint16_t>(cc);
    h.popc_db = static_cast<uint16_t>(bb);
    *out = h;
}

__device__ __forceinline__ u64 approx_count(const ScanArgs& a)
{
    // fingerprintdb_cuda.cu:263-277: survivors when cutoff > 0, else all rows
    return a.cutoff > 0.0f ? a.state->kept : a.nrows;
}

constexpr int kSelectThreads = 256;
constexpr int kSelectBlocks = kSelectCap / kSelectThreads;

// keys[0..n) in LDS, n a power of two: bitonic sort, descending.
__device__ __forceinline__ void bitonic_desc_lds(u64* keys, uint32_t n, int tid, int nthreads)
{
    for (uint32_t size = 2; size <= n; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t t = tid; t < n / 2; t += nthreads) {
                const uint32_t lo = 2 * t - (t & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const u64 x = keys[lo], y = keys[hi];
                if ((x < y) == desc) {
                    keys[lo] = y;
                    keys[hi] = x;
                }
            }
        }
    }
    __syncthreads();
}

// Dynamic LDS layout of select_kernel: kSelectCap keys, then a 256-bin digit
// histogram and control words (heavy-tie path only).
constexpr size_t kSelectLds = static_cast<size_t>(kSelectCap) * sizeof(u64) + 256 * sizeof(uint32_t) + 16;

// Heavy ties (more than kSelectCap finalists): one workgroup runs an MSD radix
// select over the unique 64-bit keys to find the k-th largest key T, gathers the
// exactly-k keys >= T into LDS, sorts them and re-derives the popcounts from the
// table.  k <= kSelectCap.
__device__ void select_heavy(const ScanArgs& a, const u64* finalists, uint32_t m2, uint32_t row_base, u64* keys,
                             uint32_t* dhist, uint32_t* ctl, gsim_result_header* hdr, gsim_hit* hits)
{
    const int tid = threadIdx.x;
    u64 prefix = 0;
    if (tid == 0) ctl[1] = a.k;
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 56 - 8 * pass;
        dhist[tid] = 0; // kSelectThreads == 256 bins
        __syncthreads();
        for (uint32_t i = tid; i < m2; i += kSelectThreads) {
            const u64 key = finalists[i];
            if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&dhist[(key >> shift) & 0xFF], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t remaining = ctl[1], acc = 0;
            int d = 255;
            for (; d > 0; d--) {
                if (acc + dhist[d] >= remaining) break;
                acc += dhist[d];
            }
            ctl[0] = static_cast<uint32_t>(d);
            ctl[1] = remaining - acc;
        }
        __syncthreads();
        prefix = (prefix << 8) | ctl[0];
        __syncthreads();
    }
    // prefix is the k-th largest key; keys are unique -> exactly k keys >= it
    if (tid == 0) ctl[2] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < m2; i += kSelectThreads) {
        const u64 key = finalists[i];
        if (key >= prefix) {
            const uint32_t pos = atomicAdd(&ctl[2], 1u);
            if (pos < static_cast<uint32_t>(kSelectCap)) keys[pos] = key;
        }
    }
    __syncthreads();
    const uint32_t nsel = ctl[2] < static_cast<uint32_t>(kSelectCap) ? ctl[2] : static_cast<uint32_t>(kSelectCap);
    uint32_t n = 1;
    while (n < nsel) n <<= 1;
    for (uint32_t i = nsel + tid; i < n; i += kSelectThreads) keys[i] = 0ull;
    bitonic_desc_lds(keys, n, tid, kSelectThreads);
    const uint32_t nout = nsel < a.k ? nsel : a.k;
    for (uint32_t i = tid; i < nout; i += kSelectThreads) emit_hit(a, keys[i], row_base, hits + i);
    if (tid == 0) {
        hdr->count = nout;
        hdr->flags = 1u;
        hdr->approx = approx_count(a);
    }
}

// K3.  kSelectBlocks workgroups.  Usual case (finalists <= kSelectCap): every
// finalist's output position is its rank = the number of finalists with a larger
// key (keys are unique); each workgroup holds all keys in LDS and ranks 256 of
// them by a broadcast-read counting loop -- no sort, no data movement, the hit
// (row, score, common, popc_db) goes straight from registers to its slot.  The
// result block may live in device memory or in pinned host memory (zero-copy).
// The last workgroup to finish folds the query's counters into the running
// totals and re-zeroes the per-query state for the next query.
__global__ __launch_bounds__(kSelectThreads) void select_kernel(ScanArgs a, const u64* finalists,
                                                                const uint32_t* finalists_cb, uint32_t cap,
                                                                uint32_t row_base, void* d_result)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    uint32_t* dhist = reinterpret_cast<uint32_t*>(smem + static_cast<size_t>(kSelectCap) * sizeof(u64));
    uint32_t* ctl = dhist + 256; // [0] digit, [1] remaining, [2] gather cursor, [3] last-workgroup flag
    const int tid = threadIdx.x;
    if (a.gate && *a.gate == 0) return; // every workgroup reads the gate before the last one can clear it (ticket below)
    gsim_result_header* hdr = reinterpret_cast<gsim_result_header*>(d_result);
    gsim_hit* hits = reinterpret_cast<gsim_hit*>(hdr + 1);
    uint32_t m2 = a.k ? a.state->nfinal : 0;
    if (m2 > cap) m2 = cap; // cannot happen: cap covers every candidate slot
    if (m2 <= static_cast<uint32_t>(kSelectCap)) {
        const uint32_t first = blockIdx.x * kSelectThreads;
        if (first < m2) {
            const uint32_t npad = (m2 + 1u) & ~1u;
            for (uint32_t i = tid; i < npad; i += kSelectThreads) keys[i] = i < m2 ? finalists[i] : 0ull;
            __syncthreads();
            const uint32_t i = first + tid;
            if (i < m2) {
                const u64 mine = keys[i];
                const uint32_t cb = finalists_cb[i]; // issued before the counting loop
                uint32_t rank = 0;
                const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(keys);
#pragma unroll 4
                for (uint32_t j = 0; j < npad / 2; j++) { // ds_read_b128 broadcast: two keys per read
                    const ulonglong2 kk = k2[j];
                    rank += (kk.x > mine) ? 1u : 0u;
                    rank += (kk.y > mine) ? 1u : 0u;
                }
                if (rank < a.k) {
                    gsim_hit h;
                    h.row = ~static_cast<uint32_t>(mine) + row_base;
                    h.score = key_score(static_cast<uint32_t>(mine >> 32));
                    h.common = static_cast<uint16_t>(cb >> 16);
                    h.popc_db = static_cast<uint16_t>(cb & 0xFFFFu);
                    hits[rank] = h;
                }
            }
        }
        if (blockIdx.x == 0 && tid == 0) {
            hdr->count = m2 < a.k ? m2 : a.k;
            hdr->flags = 0;
            hdr->approx = approx_count(a);
        }
    } else if (blockIdx.x == 0) {
        select_heavy(a, finalists, m2, row_base, keys, dhist, ctl, hdr, hits);
    }
    // ticket: the last workgroup resets the state (all others are done reading it)
    __threadfence();
    __syncthreads();
    if (tid == 0) ctl[3] = (atomicAdd(&a.state->done, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (ctl[3]) {
        if (tid == 0) {
            a.state->ncand_sum += a.state->ncand;
            a.state->nfinal_sum += a.state->nfinal;
            a.state->queries += 1;
            a.state->kept = 0;
            a.state->ncand = 0;
            a.state->nfinal = 0;
            a.state->done = 0;
            a.state->gtau = 0;
            a.state->redo = 0;
        }
        for (int i = tid; i < kScanBins; i += kSelectThreads) a.state->ghist[i] = 0;
    }
}

__global__ __launch_bounds__(256) void reset_state_kernel(QueryState* st, LargeKState* lk)
{
    if (threadIdx.x == 0 && lk) {
        lk->prefix = 0;
        lk->remaining = 0;
        lk->ticket = 0;
        lk->count = 0;
        lk->all = 0;
    }
    if (threadIdx.x == 0) {
        st->ncand_sum += st->ncand;
        st->nfinal_sum += st->nfinal;
        st->queries += 1;
        st->kept = 0;
        st->ncand = 0;
        st->nfinal = 0;
        st->done = 0;
        st->gtau = 0;
        st->redo = 0;
    }
    for (int i = threadIdx.x; i < kScanBins; i += 256) st->ghist[i] = 0;
}

// ---------------------------------------------------------------------------
// large-k path (k > kSelectCap): bitonic sort of ALL finalists in global memory
// (multi-launch), then emission of the first k.  Exact for any input.
// ---------------------------------------------------------------------------

// The k-th largest finalist key by an MSD radix descent, one launch per byte, the finalist count read ON THE DEVICE:
// nothing of the large-k path is sized by the host from a value it would have to wait for.  Pass p histograms byte
// (7 - p) of the keys that match the prefix found so far (LDS histogram per workgroup, one global atomic per non-empty
// bin); the last workgroup (ticket) picks the digit that holds the wanted rank, extends the prefix and clears the
// histogram.  Fewer finalists than k: `all` is set and every finalist is taken.
__global__ __launch_bounds__(256) void largek_pass_kernel(ScanArgs a, const u64* finalists, uint32_t cap, LargeKState* lk, int pass)
{
    __shared__ uint32_t s_h[256];
    __shared__ uint32_t s_last;
    const int tid = threadIdx.x;
    uint32_t nfinal = a.state->nfinal;
    if (nfinal > cap) nfinal = cap;
    if (pass > 0 && lk->all) return;
    const int shift = 56 - 8 * pass;
    const u64 prefix = lk->prefix;
    const uint32_t want = pass == 0 ? a.k : lk->remaining;
    s_h[tid] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256 + tid; i < nfinal; i += gridDim.x * 256) {
        const u64 key = finalists[i];
        if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&s_h[(key >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    if (s_h[tid]) atomicAdd(&lk->hist[tid], s_h[tid]);
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = (atomicAdd(&lk->ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (tid < 64) {
        uint32_t h[4];
        uint32_t s4 = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            h[i] = __hip_atomic_load(&lk->hist[tid * 4 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s4 += h[i];
        }
        uint32_t bin, cnt;
        threshold_from_counts<4>(h, s4, want, tid, bin, cnt);
        if (tid == 0) {
            if (cnt < want) { // (pass 0 only: fewer finalists than k)
                lk->all = 1;
                lk->prefix = 0;
            } else {
                const uint32_t pop = __hip_atomic_load(&lk->hist[bin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lk->prefix = (prefix << 8) | bin;
                lk->remaining = want - (cnt - pop);
            }
            lk->ticket = 0;
        }
    }
    __syncthreads();
    lk->hist[tid] = 0;
}

// the keys at or above the k-th largest (exactly min(k, #finalists) of them: keys are unique) -> out[0 ..)
__global__ __launch_bounds__(256) void largek_gather_kernel(ScanArgs a, const u64* finalists, uint32_t cap, LargeKState* lk, u64* out,
                                                            uint32_t out_cap)
{
    const int lane = threadIdx.x & 63;
    uint32_t nfinal = a.state->nfinal;
    if (nfinal > cap) nfinal = cap;
    const u64 kth = lk->all ? 0ull : lk->prefix;
    const uint32_t n64 = (nfinal + 63u) & ~63u;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n64; i += gridDim.x * 256) {
        const u64 key = i < nfinal ? finalists[i] : 0ull;
        const bool take = i < nfinal && key >= kth;
        const u64 m = __ballot(take);
        if (m == 0) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&lk->count, static_cast<uint32_t>(__popcll(m)));
        base = __builtin_amdgcn_readfirstlane(base);
        const uint32_t pos = base + lane_rank(m);
        if (take && pos < out_cap) out[pos] = key;
    }
}

// ---------------------------------------------------------------------------
// folded tables: the candidates' re-score with the full fingerprints, on the device
// ---------------------------------------------------------------------------
// fingerprintdb_cuda.cu:307-331: the R = k F (int)log2(2F) best FOLDED scores of a storage are re-scored with the full
// fingerprints (tanimoto_similarity_cpu, :387-399), stably sorted by the new score (top_results_bubble_sort: strict '>',
// so ties keep the order of the folded list) and the first min(k, R) kept up to the first one below the cutoff.  The
// reference does this on the host (slide 19 lists it as future GPU work); here the full rows are resident as well
// (288 GB hold both) and three small launches do it: re-score into keys (score key << 32 | ~position), a bitonic sort
// of the <= 64 Ki keys, emission.  A NaN score (0 / 0: two empty fingerprints) is not ordered by '>': it raises a flag
// and the host path, which has the literal bubble sort for that case, answers the query.
__global__ __launch_bounds__(256) void fold_rescore_kernel(const void* folded_block, const uint32_t* full_rows, const uint32_t* full_query,
                                                           uint32_t W, uint32_t qpop, u64* keys, uint32_t* cbs, uint32_t npad,
                                                           uint32_t* nan_flag)
{
    const gsim_result_header* hdr = reinterpret_cast<const gsim_result_header*>(folded_block);
    const gsim_hit* cand = reinterpret_cast<const gsim_hit*>(hdr + 1);
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= npad) return;
    if (j >= hdr->count) {
        keys[j] = 0ull; // padding of the sort: below every real key
        return;
    }
    const uint32_t* r = full_rows + static_cast<u64>(cand[j].row) * W;
    uint32_t cc = 0, bb = 0;
    for (uint32_t i = 0; i < W; i++) {
        const uint32_t x = r[i];
        cc += __popc(x & full_query[i]);
        bb += __popc(x);
    }
    const float s = score_of(GSIM_METRIC_TANIMOTO, 0.f, 0.f, qpop, bb, cc);
    if (s != s) atomicOr(nan_flag, 1u);
    keys[j] = (static_cast<u64>(order_key(s)) << 32) | static_cast<u64>(~j);
    cbs[j] = (cc << 16) | bb;
}

__global__ __launch_bounds__(256) void fold_emit_kernel(const void* folded_block, const u64* sorted_keys, const uint32_t* cbs, uint32_t k,
                                                        float cutoff, uint32_t row_base, void* out_block)
{
    const gsim_result_header* fh = reinterpret_cast<const gsim_result_header*>(folded_block);
    const gsim_hit* cand = reinterpret_cast<const gsim_hit*>(fh + 1);
    gsim_result_header* oh = reinterpret_cast<gsim_result_header*>(out_block);
    gsim_hit* out = reinterpret_cast<gsim_hit*>(oh + 1);
    const uint32_t keep = fh->count < k ? fh->count : k;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // the list is in descending score order: "up to the first one below the cutoff" = the entries at or above it
    if (i < keep) {
        const u64 key = sorted_keys[i];
        const float s = key_score(static_cast<uint32_t>(key >> 32));
        if (!(s < cutoff)) {
            const uint32_t j = ~static_cast<uint32_t>(key);
            const uint32_t cb = cbs[j];
            gsim_hit h;
            h.row = cand[j].row + row_base;
            h.score = s;
            h.common = static_cast<uint16_t>(cb >> 16);
            h.popc_db = static_cast<uint16_t>(cb & 0xFFFFu);
            out[i] = h;
        }
    }
    if (i == 0) { // the count: how many of the first `keep` are at or above the cutoff (they form a prefix)
        uint32_t lo = 0, hi = keep;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (!(key_score(static_cast<uint32_t>(sorted_keys[mid] >> 32)) < cutoff)) lo = mid + 1;
            else hi = mid;
        }
        oh->count = lo;
        oh->flags = fh->flags;
        oh->approx = fh->approx;
    }
}

__global__ __launch_bounds__(256) void fill_keys_kernel(u64* keys, u64 from, u64 to)
{
    const u64 i = from + static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < to) keys[i] = 0;
}

__global__ __launch_bounds__(256) void bitonic_step_kernel(u64* keys, uint32_t n, uint32_t size, uint32_t stride)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n / 2) return;
    const uint32_t lo = 2 * t - (t & (stride - 1));
    const uint32_t hi = lo + stride;
    const bool desc = (lo & size) == 0;
    const u64 x = keys[lo], y = keys[hi];
    if ((x < y) == desc) {
        keys[lo] = y;
        keys[hi] = x;
    }
}

__global__ __launch_bounds__(256) void emit_hits_kernel(ScanArgs a, const u64* sorted_keys, const LargeKState* lk,
                                                        uint32_t row_base, u64 approx_if_no_cutoff, uint32_t flags,
                                                        void* d_result)
{
    gsim_result_header* hdr = reinterpret_cast<gsim_result_header*>(d_result);
    gsim_hit* hits = reinterpret_cast<gsim_hit*>(hdr + 1);
    const uint32_t nkeys = lk->count < a.k ? lk->count : a.k;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nkeys) emit_hit(a, sorted_keys[i], row_base, hits + i);
    if (i == 0) {
        hdr->count = nkeys;
        hdr->flags = flags;
        hdr->approx = a.cutoff > 0.0f ? a.state->kept : approx_if_no_cutoff;
    }
}

// ---------------------------------------------------------------------------
// merge of per-shard result blocks (fingerprintdb_cuda.cu:363-380)
// ---------------------------------------------------------------------------

__device__ __forceinline__ const gsim_result_header* block_hdr(const void* blocks, size_t block_bytes, uint32_t i)
{
    return reinterpret_cast<const gsim_result_header*>(reinterpret_cast<const unsigned char*>(blocks) +
                                                       static_cast<size_t>(i) * block_bytes);
}

// Every list is in canonical order and keys are unique across lists, so the
// output position of an element is the number of elements that precede it:
// its own index plus, for every other list, a binary search.
// blockIdx.y = query: its lists are the blocks q, q + nq, q + 2 nq, ... of the gathered buffer
// (rank-major, as an all-gather of per-rank [nq] block arrays leaves them).
__global__ __launch_bounds__(256) void merge_kernel(const void* all_blocks, uint32_t nblocks, uint32_t nq,
                                                    size_t block_bytes, uint32_t k, void* d_results)
{
    const uint32_t q = blockIdx.y;
    const void* blocks = static_cast<const unsigned char*>(all_blocks) + static_cast<size_t>(q) * block_bytes;
    const size_t list_stride = static_cast<size_t>(nq) * block_bytes;
    gsim_result_header* ohdr =
        reinterpret_cast<gsim_result_header*>(static_cast<unsigned char*>(d_results) + static_cast<size_t>(q) * block_bytes);
    gsim_hit* out = reinterpret_cast<gsim_hit*>(ohdr + 1);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        u64 approx = 0, total = 0;
        uint32_t flags = 0;
        for (uint32_t i = 0; i < nblocks; i++) {
            const gsim_result_header* h = block_hdr(blocks, list_stride, i);
            approx += h->approx;
            total += h->count;
            flags |= h->flags;
        }
        ohdr->count = total < k ? static_cast<uint32_t>(total) : k;
        ohdr->flags = flags;
        ohdr->approx = approx;
    }
    const uint32_t li = t / k, e = t % k;
    if (li >= nblocks) return;
    const gsim_result_header* mh = block_hdr(blocks, list_stride, li);
    if (e >= mh->count) return;
    const gsim_hit* mine = reinterpret_cast<const gsim_hit*>(mh + 1);
    const gsim_hit me = mine[e];
    const u64 mykey = make_key(me.score, me.row);
    uint32_t rank = e;
    for (uint32_t j = 0; j < nblocks; j++) {
        if (j == li) continue;
        const gsim_result_header* h = block_hdr(blocks, list_stride, j);
        const gsim_hit* lst = reinterpret_cast<const gsim_hit*>(h + 1);
        uint32_t lo = 0, hi = h->count; // first index whose key < mykey
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (make_key(lst[mid].score, lst[mid].row) > mykey) lo = mid + 1;
            else hi = mid;
        }
        rank += lo;
    }
    if (rank < k) out[rank] = me;
}

// ---------------------------------------------------------------------------
// synthetic table generator (twin of oracle gso_synth_word)
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void generate_kernel(uint32_t* rows, u64 seed, int kind, u64 first_row,
                                                       u64 nwords, uint32_t W)
{
    const u64 stride = static_cast<u64>(gridDim.x) * blockDim.x;
    for (u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x; i < nwords; i += stride)
        rows[i] = synth_word_iid(seed, kind == GSIM_SYNTH_DENSE, first_row * W + i);
}

// GSIM_SYNTH_MORGAN (gsim_synth.h): a row is made whole, by one thread, in LDS; the workgroup's rows
// then leave with coalesced stores.  R rows per workgroup (host: kMorganLdsWords / W, at most 256).
constexpr uint32_t kMorganLdsWords = 12288;

__global__ __launch_bounds__(256) void generate_morgan_kernel(uint32_t* rows, u64 seed, u64 first_row, u64 nrows,
                                                              uint32_t W, uint32_t R)
{
    __shared__ uint32_t s_rows[kMorganLdsWords];
    for (u64 r0 = static_cast<u64>(blockIdx.x) * R; r0 < nrows; r0 += static_cast<u64>(gridDim.x) * R) {
        const uint32_t n = static_cast<uint32_t>(nrows - r0 < R ? nrows - r0 : R);
        if (threadIdx.x < n) synth_row_morgan(s_rows + threadIdx.x * W, seed, first_row + r0 + threadIdx.x, W);
        __syncthreads();
        uint32_t* dst = rows + r0 * W;
        for (uint32_t i = threadIdx.x; i < n * W; i += 256) dst[i] = s_rows[i];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void score_table_kernel(int metric, float alpha, float beta, uint32_t a,
                                                          uint32_t max_b, uint32_t max_c, float* out)
{
    const u64 n = static_cast<u64>(max_b + 1) * (max_c + 1);
    const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = static_cast<uint32_t>(i / (max_b + 1)), b = static_cast<uint32_t>(i % (max_b + 1));
    out[i] = score_of(metric, alpha, beta, a, b, c);
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

// hipFuncAttributeMaxDynamicSharedMemorySize, set once per device and kernel (whether the runtime keeps the attribute
// per function or per device is its business; a multi-device handle launches the same kernel on several devices).
struct DynLdsOnce {
    std::atomic<bool> done[64] = {};
    hipError_t ensure(const void* fn, size_t bytes)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64 && done[dev].load(std::memory_order_acquire)) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);
        return hipSuccess;
    }
};

static bool is_pow2(uint32_t x)
{
    return x && !(x & (x - 1));
}

ScanGeometry scan_geometry(uint64_t nrows, uint32_t W, int num_cus, int waves_per_cu, int unroll)
{
    ScanGeometry g{};
    const uint32_t lpr = (W % 4 == 0 && is_pow2(W / 4) && W / 4 <= 64) ? W / 4 : 0;
    g.lanes_per_row = lpr;
    if (lpr) {
        if (unroll != 4 && unroll != 8 && unroll != 16) unroll = 8;
        g.unroll = static_cast<uint32_t>(unroll);
        g.chunk_rows = g.unroll * (64 / lpr);
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
hipError_t launch_sample(const ScanArgs& a, const ScanGeometry& g, uint32_t chunks_per_wave, hipStream_t s)
{
    if (a.k == 0 || chunks_per_wave == 0) return hipSuccess;
    const uint64_t nfull = a.nrows / g.chunk_rows;
    // never sample more than 1/8 of the table; under one chunk per wave the scan's own warm-up is cheaper
    const uint64_t fit = nfull / (8ull * g.nwaves);
    if (fit < chunks_per_wave) chunks_per_wave = static_cast<uint32_t>(fit);
    if (chunks_per_wave == 0) return hipSuccess;
    const uint64_t want = static_cast<uint64_t>(g.nwaves) * chunks_per_wave;
    const uint64_t stride = nfull / want;
    const uint32_t nsample = static_cast<uint32_t>(want);
    const uint32_t nblocks = g.nwaves / (kScanBlock / 64);
    if (g.lanes_per_row == 0) {
        const uint32_t lds = generic_lds_bytes(a.W, g.chunk_rows);
        static DynLdsOnce once;
        const hipError_t e = once.ensure(reinterpret_cast<const void*>(sample_generic_kernel), kGenericLdsBytes);
        if (e != hipSuccess) return e;
        if (lds > kGenericLdsBytes) return hipErrorInvalidValue;
        hipLaunchKernelGGL(sample_generic_kernel, dim3(nblocks), dim3(kScanBlock), lds, s, a, g.chunk_rows, nsample, stride);
        return hipGetLastError();
    }
#define GSIM_CASE(L, UU) \
    if (g.lanes_per_row == L && g.unroll == UU) return launch_sample_t<L, UU>(a, nsample, stride, nblocks, s);
    GSIM_CASE(8, 8)
    GSIM_CASE(8, 4)
    GSIM_CASE(8, 16)
    GSIM_CASE(16, 8)
    GSIM_CASE(16, 4)
    GSIM_CASE(16, 16)
    GSIM_CASE(1, 8)
    GSIM_CASE(2, 8)
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
    GSIM_CASE(8, 4)
    GSIM_CASE(8, 16)
    GSIM_CASE(16, 8)
    GSIM_CASE(16, 4)
    GSIM_CASE(16, 16)
    GSIM_CASE(1, 8)
    GSIM_CASE(2, 8)
    GSIM_CASE(4, 8)
    GSIM_CASE(32, 8)
    GSIM_CASE(64, 8)
#undef GSIM_CASE
    if (g.lanes_per_row != 0) return hipErrorInvalidValue;
    const uint32_t nblocks = g.nwaves / (kScanBlock / 64);
    static DynLdsOnce once;
    const hipError_t e = once.ensure(reinterpret_cast<const void*>(scan_generic_kernel), kGenericLdsBytes);
    if (e != hipSuccess) return e;
    const uint32_t lds = generic_lds_bytes(a.W, g.chunk_rows);
    if (lds > kGenericLdsBytes) return hipErrorInvalidValue;
    hipLaunchKernelGGL(scan_generic_kernel, dim3(nblocks), dim3(kScanBlock), lds, s, a, g);
    return hipGetLastError();
}

template <int LPR, int U>
hipError_t launch_fused_t(const ScanArgs& a, const ScanGeometry& g, const FusedArgs& f, hipStream_t s)
{
    const uint32_t nblocks = g.nwaves / (kScanBlock / 64);
    static DynLdsOnce once;
    const hipError_t e = once.ensure(reinterpret_cast<const void*>(fused_kernel<LPR, U>), sizeof(FusedShared));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((fused_kernel<LPR, U>), dim3(nblocks), dim3(kFusedBlock), sizeof(FusedShared), s, a, g, f);
    return hipGetLastError();
}

bool fused_supported(const ScanGeometry& g)
{
    // every workgroup of the grid is a selector and reads every workgroup's header with one thread
    return g.lanes_per_row != 0 && g.unroll == 8 && g.nwaves <= static_cast<uint32_t>(kFusedSelectors) * (kScanBlock / 64);
}

// M of the checkpoint summaries ("my M-th best key"): about 2k / nwaves, so that the election's rank
// r = ceil(k / M) sits in the middle of the reports; 0 when even M = 16 leaves r above the number of
// waves (tiny grids: no thresholds, every row is published) or the reports would not fit the
// electing wave's registers (64 x 64 keys).
uint32_t fused_summary_keys(uint32_t nwaves, uint32_t k)
{
    if (nwaves == 0 || nwaves > static_cast<uint32_t>(kFusedSelectors) * (kScanBlock / 64)) return 0;
    uint32_t m = (2 * k + nwaves - 1) / nwaves;
    if (m < 1) m = 1;
    if (m > 16) m = 16;
    if ((k + m - 1) / m > nwaves) return 0;
    return m;
}

// Mw of the end-of-scan reports ("my Mw-th best key", one per workgroup): the final threshold is the r-th largest
// report, r = ceil(k / Mw).  With rows spread evenly the number of rows above it is nwg x lambda, lambda solving
// P(Poisson(lambda) >= Mw) = r / nwg: ~2.0 k at Mw = 2 k / nwg (r in the middle of the reports), ~1.4 k around
// Mw = 1.25 k / nwg (r at 0.8 of them), rising again beyond -- and the selectors hold 16 Ki finalists, k up to 8 Ki.
// 0: the grid has fewer workgroups than r would need (tiny tables: every published row is a finalist).
uint32_t fused_final_keys(uint32_t nwg, uint32_t k)
{
    if (nwg == 0 || k == 0) return 0;
    uint32_t m = (5 * k + 4 * nwg - 1) / (4 * nwg);
    if (m < 1) m = 1;
    if (m > 64) return 0;
    return m;
}

hipError_t launch_fused(const ScanArgs& a, const ScanGeometry& g, const FusedArgs& f, hipStream_t s)
{
#define GSIM_CASE(L) \
    if (g.lanes_per_row == L && g.unroll == 8) \
        return launch_fused_t<L, 8>(a, g, f, s);
    GSIM_CASE(1)
    GSIM_CASE(2)
    GSIM_CASE(4)
    GSIM_CASE(8)
    GSIM_CASE(16)
    GSIM_CASE(32)
    GSIM_CASE(64)
#undef GSIM_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_compact(const ScanArgs& a, const ScanGeometry& g, unsigned long long* finalists,
                          uint32_t* finalists_cb, uint32_t finalists_cap, hipStream_t s)
{
    const uint32_t nblocks = g.nwaves / (kScanBlock / 64);
    hipLaunchKernelGGL(compact_kernel, dim3(nblocks), dim3(kScanBlock), 0, s, a, g, finalists, finalists_cb,
                       finalists_cap);
    return hipGetLastError();
}

hipError_t launch_select(const ScanArgs& a, const unsigned long long* finalists, const uint32_t* finalists_cb,
                         uint32_t finalists_cap, uint32_t row_base, void* d_result, hipStream_t s)
{
    static DynLdsOnce once;
    const hipError_t e = once.ensure(reinterpret_cast<const void*>(select_kernel), kSelectLds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(select_kernel, dim3(kSelectBlocks), dim3(kSelectThreads), kSelectLds, s, a, finalists,
                       finalists_cb, finalists_cap, row_base, d_result);
    return hipGetLastError();
}

hipError_t launch_fold_rescore(const void* folded_block, const uint32_t* full_rows, const uint32_t* full_query, uint32_t W, uint32_t qpop,
                               unsigned long long* keys, uint32_t* cbs, uint32_t npad, uint32_t* nan_flag, uint32_t k, float cutoff,
                               uint32_t row_base, void* out_block, hipStream_t s)
{
    hipLaunchKernelGGL(fold_rescore_kernel, dim3((npad + 255) / 256), dim3(256), 0, s, folded_block, full_rows, full_query, W, qpop, keys, cbs,
                       npad, nan_flag);
    hipError_t e = launch_bitonic_global(keys, npad, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fold_emit_kernel, dim3((k + 255) / 256 ? (k + 255) / 256 : 1), dim3(256), 0, s, folded_block, keys, cbs, k, cutoff, row_base,
                       out_block);
    return hipGetLastError();
}

hipError_t launch_reset_state(QueryState* state, LargeKState* lk, hipStream_t s)
{
    hipLaunchKernelGGL(reset_state_kernel, dim3(1), dim3(256), 0, s, state, lk);
    return hipGetLastError();
}

// k > kSelectCap: the k-th largest finalist key by eight radix passes, then the keys at or above it into `out`
// (out_cap >= k entries; the caller zero-fills it and sorts it afterwards).  Nothing here is sized by the finalist count.
hipError_t launch_largek_select(const ScanArgs& a, const unsigned long long* finalists, uint32_t finalists_cap, LargeKState* lk,
                                unsigned long long* out, uint32_t out_cap, hipStream_t s)
{
    for (int pass = 0; pass < 8; pass++)
        hipLaunchKernelGGL(largek_pass_kernel, dim3(256), dim3(256), 0, s, a, finalists, finalists_cap, lk, pass);
    hipLaunchKernelGGL(largek_gather_kernel, dim3(256), dim3(256), 0, s, a, finalists, finalists_cap, lk, out, out_cap);
    return hipGetLastError();
}

hipError_t launch_bitonic_global(unsigned long long* keys, uint32_t n_pow2, hipStream_t s)
{
    if (n_pow2 < 2) return hipSuccess;
    const uint32_t nb = (n_pow2 / 2 + 255) / 256;
    for (uint32_t size = 2; size <= n_pow2 && size != 0; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            hipLaunchKernelGGL(bitonic_step_kernel, dim3(nb), dim3(256), 0, s, keys, n_pow2, size, stride);
        }
    }
    return hipGetLastError();
}

hipError_t launch_fill_zero_keys(unsigned long long* keys, uint64_t from, uint64_t to, hipStream_t s)
{
    if (to <= from) return hipSuccess;
    const uint64_t nb = (to - from + 255) / 256;
    hipLaunchKernelGGL(fill_keys_kernel, dim3(static_cast<uint32_t>(nb)), dim3(256), 0, s, keys, from, to);
    return hipGetLastError();
}

hipError_t launch_emit_hits(const ScanArgs& a, const unsigned long long* sorted_keys, const LargeKState* lk,
                            uint32_t row_base, uint64_t approx_if_no_cutoff, uint32_t flags, void* d_result,
                            hipStream_t s)
{
    const uint32_t nb = a.k ? (a.k + 255) / 256 : 1; // (the kernel emits min(k, gathered) hits)
    hipLaunchKernelGGL(emit_hits_kernel, dim3(nb), dim3(256), 0, s, a, sorted_keys, lk, row_base,
                       approx_if_no_cutoff, flags, d_result);
    return hipGetLastError();
}

hipError_t launch_merge_batch(const void* d_blocks, uint32_t nblocks, uint32_t nq, size_t block_bytes, uint32_t k,
                              void* d_results, hipStream_t s)
{
    const uint64_t nthreads = static_cast<uint64_t>(nblocks) * (k ? k : 1);
    const uint32_t nb = static_cast<uint32_t>((nthreads + 255) / 256);
    hipLaunchKernelGGL(merge_kernel, dim3(nb ? nb : 1, nq), dim3(256), 0, s, d_blocks, nblocks, nq, block_bytes,
                       k ? k : 1, d_results);
    return hipGetLastError();
}

hipError_t launch_generate(void* rows, uint64_t seed, int kind, uint64_t first_row, uint64_t nrows, uint32_t W,
                           hipStream_t s)
{
    const uint64_t nwords = nrows * W;
    if (nwords == 0) return hipSuccess;
    if (kind == GSIM_SYNTH_MORGAN) {
        if (W > kMorganLdsWords) return hipErrorInvalidValue;
        const uint32_t R = std::min<uint32_t>(256u, kMorganLdsWords / W);
        uint64_t nb = (nrows + R - 1) / R;
        if (nb > 65536) nb = 65536;
        hipLaunchKernelGGL(generate_morgan_kernel, dim3(static_cast<uint32_t>(nb)), dim3(256), 0, s,
                           reinterpret_cast<uint32_t*>(rows), seed, first_row, nrows, W, R);
        return hipGetLastError();
    }
    uint64_t nb = (nwords + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(generate_kernel, dim3(static_cast<uint32_t>(nb)), dim3(256), 0, s,
                       reinterpret_cast<uint32_t*>(rows), seed, kind, first_row, nwords, W);
    return hipGetLastError();
}

hipError_t launch_score_table(int metric, float alpha, float beta, uint32_t a, uint32_t max_b, uint32_t max_c,
                              float* d_out, hipStream_t s)
{
    const uint64_t n = static_cast<uint64_t>(max_b + 1) * (max_c + 1);
    const uint32_t nb = static_cast<uint32_t>((n + 255) / 256);
    hipLaunchKernelGGL(score_table_kernel, dim3(nb), dim3(256), 0, s, metric, alpha, beta, a, max_b, max_c, d_out);
    return hipGetLastError();
}

} // namespace gsim
