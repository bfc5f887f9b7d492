// gsim_scan_inl.h -- the streaming loop of one wavefront over the packed-uint32 table, shared by the kernels that scan
// it (scan_kernel, sample_kernel: gsim_scan.hip; fused_kernel: gsim_fused.hip).  Replaces the reference's thread-per-row
// TanimotoFunctor (fingerprintdb_cuda.cu:76-104).  Internal; device code only.
#pragma once

#include <atomic>

#include "gsim_device_common.h"

namespace gsim
{
namespace
{

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
// What most filters do with a row's counts (val = popc(q & row) << 16 | popc(row)): score it, offer the score.
template <typename Filter>
__device__ __forceinline__ void offer_scored(Filter& f, bool active, uint32_t row, uint32_t val, const ScanArgs& a, int lane)
{
    f.offer(active, row, score_of(a.metric, a.alpha, a.beta, a.qpop, val & 0xFFFFu, val >> 16), val, lane);
}

template <int LPR, int U, bool FULL, typename Filter>
__device__ __forceinline__ void reduce_chunk(const u32x4 (&d)[U], const u32x4& q, u64 row0, const ScanArgs& a,
                                             Filter& f, int lane)
{
    constexpr int RPL = 64 / LPR;
    constexpr int ROUNDS = (U + LPR - 1) / LPR;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
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
        const int j = r * LPR + sub;
        const u64 row = row0 + static_cast<u64>(j * RPL + grp);
        const bool active = (j < U) && (FULL || row < a.nrows);
        // (the filter scores the row: the reference's arithmetic, score_of -- or, where it can prove from the counts alone
        // that the row lies below its threshold, nothing at all: FusedFilter::offer_counts)
        f.template offer_counts<LPR>(active, static_cast<uint32_t>(row), val, a, lane);
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

// The same loop for rows of L = W / 4 sixteen-byte units where L is not a power of two but 3, 5 or 7 times one (see
// scan_ragged_kernel, gsim_scan.hip, for the derivation): P consecutive wave loads hold 64 P / L whole rows; an inclusive prefix
// sum over the chunk's units turns a row's counts into the difference of two prefixes; C sub-chunks per trip.
// g.chunk_rows = C * 64 P / L rows per trip.
template <int P, int C, typename Filter>
__device__ __forceinline__ void scan_rows_ragged(const ScanArgs& a, const ScanGeometry& g, Filter& f, uint32_t w, int lane)
{
    const uint32_t L = a.W / 4u;          // units per row
    const uint32_t R = g.chunk_rows / C;  // rows per sub-chunk: 64 P / L
    const u32x4* __restrict__ db = reinterpret_cast<const u32x4*>(a.rows);
    const u32x4* qu = reinterpret_cast<const u32x4*>(a.query);
    u32x4 q[P];
#pragma unroll
    for (int j = 0; j < P; j++) q[j] = qu[(64u * j + static_cast<uint32_t>(lane)) % L];
    // the last unit of this lane's row (lanes >= R have none: they read lane 0's and are inactive)
    const uint32_t end_u = (static_cast<uint32_t>(lane) < R ? static_cast<uint32_t>(lane) + 1u : 1u) * L - 1u;
    const uint32_t je = end_u >> 6;
    const int le4 = static_cast<int>((end_u & 63u) * 4u);
    const u64 total_units = a.nrows * L;
    const u64 nfull = a.nrows / g.chunk_rows; // trips' worth of rows that are all present
    uint32_t gt = 0;

    // one sub-chunk: P loads = R whole rows; row0 = its first row
    auto reduce = [&](const u32x4* d, u64 row0, bool full) {
        uint32_t s[P];
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < P; j++) {
            const uint32_t cc = __popc(d[j].x & q[j].x) + __popc(d[j].y & q[j].y) + __popc(d[j].z & q[j].z) + __popc(d[j].w & q[j].w);
            const uint32_t bb = __popc(d[j].x) + __popc(d[j].y) + __popc(d[j].z) + __popc(d[j].w);
            uint32_t v = (cc << 16) + bb;
            // inclusive prefix sum over the wave: within the rows of 16 lanes by DPP, then the rows' totals
            v += dpp_shr<1>(v);
            v += dpp_shr<2>(v);
            v += dpp_shr<4>(v);
            v += dpp_shr<8>(v);
            const uint32_t t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47),
                           t3 = __builtin_amdgcn_readlane(v, 63);
            const int rowi = lane >> 4;
            v += carry + (rowi > 0 ? t0 : 0u) + (rowi > 1 ? t1 : 0u) + (rowi > 2 ? t2 : 0u);
            carry += t0 + t1 + t2 + t3;
            s[j] = v;
        }
        uint32_t e = 0; // the prefix at this lane's row's last unit
#pragma unroll
        for (int j = 0; j < P; j++) {
            const uint32_t x = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(le4, static_cast<int>(s[j])));
            e = je == static_cast<uint32_t>(j) ? x : e;
        }
        uint32_t prev = static_cast<uint32_t>(__shfl_up(static_cast<int>(e), 1, 64)); // ... and at the previous row's
        prev = lane == 0 ? 0u : prev;
        // (as 32-bit integers: the difference of the two prefixes is the sum over the row's units of common << 16 | bits, exact
        // mod 2^32 whatever the prefixes have grown to -- a row's own bits fit 16 bits, ragged_loads_of)
        const uint32_t val = e - prev;
        const u64 row = row0 + static_cast<u64>(lane);
        const bool active = static_cast<uint32_t>(lane) < R && (full || row < a.nrows);
        f.template offer_counts<64>(active, static_cast<uint32_t>(row), val, a, lane);
    };

    constexpr int NL = P * C; // loads per trip
    if (w < nfull) {
        const u64 last = w + (nfull - 1 - w) / g.nwaves * g.nwaves; // this wave's last full trip
        u32x4 nxt[NL];
        {
            const u32x4* p = db + static_cast<u64>(w) * (64u * NL) + lane;
#pragma unroll
            for (int j = 0; j < NL; j++) nxt[j] = stream_load(p + j * 64);
        }
        uint32_t trip = 0;
        for (u64 c = w;; c += g.nwaves) {
            u32x4 d[NL];
#pragma unroll
            for (int j = 0; j < NL; j++) d[j] = nxt[j];
            const u64 cn = c + g.nwaves <= last ? c + g.nwaves : last; // (the final trip re-reads the last one: no branch in the body)
            const u32x4* p = db + cn * (64u * NL) + lane;
#pragma unroll
            for (int j = 0; j < NL; j++) nxt[j] = stream_load(p + j * 64);
            f.refresh(gt, lane);
            gt = (trip++ & 7u) == 0 ? f.load_gtau() : 0u;
#pragma unroll
            for (int i = 0; i < C; i++) reduce(d + i * P, c * g.chunk_rows + static_cast<u64>(i) * R, true);
            if (Filter::kFused) f.checkpoint(trip, lane);
            if (c == last) break;
        }
    }
    if (nfull < g.nchunks && w == nfull % g.nwaves) { // the table's partial last trip
        const u64 u0 = nfull * (64u * NL) + static_cast<u64>(lane);
        u32x4 d[NL];
#pragma unroll
        for (int j = 0; j < NL; j++) d[j] = u0 + 64u * j < total_units ? stream_load(db + u0 + 64u * j) : u32x4{0, 0, 0, 0};
        f.refresh(f.load_gtau(), lane);
#pragma unroll
        for (int i = 0; i < C; i++) reduce(d + i * P, nfull * g.chunk_rows + static_cast<u64>(i) * R, false);
    }
}

// ... and for rows that are not even whole 16-byte units: WW = 3, 5, 7, 9, 11 or twice that many 32-bit words (96-, 160-, 224-, 288-, 352-, 192-,
// 320-, 448-bit rows).  NL = PP x C consecutive wave loads (PP = the odd part of WW; NL KB = 256 NL words) hold 256 NL / WW
// whole rows, several per lane.  Every lane counts its four words against the query words of their positions (4 NL query
// words per lane, loaded once) and leaves the packed counts (common << 16 | row bits) in the wave's LDS area `wlds` in word
// order; lane l then sums the WW words of row 64 i + l (ds_read_b32 at a stride of WW words: odd, or twice odd read as
// b64 -- conflict-free).  No cross-lane arithmetic at all: with one streaming wave per SIMD the DPP prefix sum this replaced
// cost twice the instructions and all of their latency.
template <int WW, int C, typename Filter>
__device__ __forceinline__ void scan_rows_wragged(const ScanArgs& a, const ScanGeometry& g, Filter& f, uint32_t w, int lane, uint32_t* wlds)
{
    constexpr int PP = WW % 2 ? WW : WW / 2;
    constexpr int NL = PP * C;
    constexpr uint32_t TR = 256u * NL / WW; // rows per trip (= g.chunk_rows)
    static_assert((256 * NL) % WW == 0 && TR % 64 == 0, "a trip holds whole rows, 64 at a time");
    const u32x4* __restrict__ db = reinterpret_cast<const u32x4*>(a.rows);
    u32x4 q[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const uint32_t w0 = 4u * (64u * j + static_cast<uint32_t>(lane));
        q[j] = u32x4{a.query[w0 % WW], a.query[(w0 + 1u) % WW], a.query[(w0 + 2u) % WW], a.query[(w0 + 3u) % WW]};
    }
    const u64 total_words = a.nrows * WW;
    const u64 total_units = (total_words + 3u) / 4u;
    const u64 nfull = a.nrows / TR;
    uint32_t gt = 0;
    u32x4* wl4 = reinterpret_cast<u32x4*>(wlds);

    auto reduce = [&](const u32x4* d, u64 row0, bool full) {
#pragma unroll
        for (int j = 0; j < NL; j++) {
            wl4[64 * j + lane] = u32x4{(static_cast<uint32_t>(__popc(d[j].x & q[j].x)) << 16) + __popc(d[j].x),
                                       (static_cast<uint32_t>(__popc(d[j].y & q[j].y)) << 16) + __popc(d[j].y),
                                       (static_cast<uint32_t>(__popc(d[j].z & q[j].z)) << 16) + __popc(d[j].z),
                                       (static_cast<uint32_t>(__popc(d[j].w & q[j].w)) << 16) + __popc(d[j].w)};
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier(); // (one wave: its LDS operations execute in order)
        uint32_t val[TR / 64];
#pragma unroll
        for (uint32_t rr = 0; rr < TR / 64u; rr++) {
            const uint32_t r = 64u * rr + static_cast<uint32_t>(lane);
            uint32_t v = 0;
            if constexpr (WW % 2 == 0) {
                const uint2* p2 = reinterpret_cast<const uint2*>(wlds) + r * (WW / 2);
#pragma unroll
                for (int t = 0; t < WW / 2; t++) v += p2[t].x + p2[t].y;
            } else {
                const uint32_t* p1 = wlds + r * WW;
#pragma unroll
                for (int t = 0; t < WW; t++) v += p1[t];
            }
            val[rr] = v;
        }
#pragma unroll
        for (uint32_t rr = 0; rr < TR / 64u; rr++) {
            const u64 row = row0 + 64u * rr + static_cast<uint32_t>(lane);
            f.template offer_counts<1>(full || row < a.nrows, static_cast<uint32_t>(row), val[rr], a, lane);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier(); // the next trip overwrites the area
    };

    if (w < nfull) {
        const u64 last = w + (nfull - 1 - w) / g.nwaves * g.nwaves;
        u32x4 nxt[NL];
        {
            const u32x4* p = db + static_cast<u64>(w) * (64u * NL) + lane;
#pragma unroll
            for (int j = 0; j < NL; j++) nxt[j] = stream_load(p + j * 64);
        }
        uint32_t trip = 0;
        for (u64 c = w;; c += g.nwaves) {
            u32x4 d[NL];
#pragma unroll
            for (int j = 0; j < NL; j++) d[j] = nxt[j];
            const u64 cn = c + g.nwaves <= last ? c + g.nwaves : last;
            const u32x4* p = db + cn * (64u * NL) + lane;
#pragma unroll
            for (int j = 0; j < NL; j++) nxt[j] = stream_load(p + j * 64);
            f.refresh(gt, lane);
            gt = (trip++ & 7u) == 0 ? f.load_gtau() : 0u;
            reduce(d, c * TR, true);
            if (Filter::kFused) f.checkpoint(trip, lane);
            if (c == last) break;
        }
    }
    if (nfull < g.nchunks && w == nfull % g.nwaves) { // the table's partial last trip: whole units inside the table, the last one word by word
        const u64 u0 = nfull * (64u * NL) + static_cast<u64>(lane);
        const uint32_t* dbw = reinterpret_cast<const uint32_t*>(a.rows);
        u32x4 d[NL];
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const u64 u = u0 + 64u * j;
            if (4u * u + 4u <= total_words) d[j] = stream_load(db + u);
            else if (u < total_units) d[j] = u32x4{4u * u < total_words ? dbw[4u * u] : 0u, 4u * u + 1u < total_words ? dbw[4u * u + 1u] : 0u,
                                                   4u * u + 2u < total_words ? dbw[4u * u + 2u] : 0u, 0u};
            else d[j] = u32x4{0, 0, 0, 0};
        }
        f.refresh(f.load_gtau(), lane);
        reduce(d, nfull * TR, false);
    }
}

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

} // namespace
} // namespace gsim
