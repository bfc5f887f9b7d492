// gsim_fused.hip -- the single-launch path: scan, publish, select and the result block in ONE gfx950 kernel
// (k <= kFusedMaxK on the specialised widths; replaces fingerprintdb_cuda.cu:228-339 for the usual query).
#include "gsim_device.h"

#include <hip/hip_runtime.h>

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

#include "gsim_fused_protocol.h"
#include "gsim_fused_thresholds.inl"

// LPR > 0: rows of LPR sixteen-byte units (a power of two), U loads per chunk.  LPR < 0: the register-streamed odd widths
// (scan_rows_ragged<-LPR, U>: rows of 3, 5 or 7 x 2^i units, -LPR loads per sub-chunk, U sub-chunks per trip); WORDS: rows of
// -LPR = 3, 5, 7, 9, 11 or twice that many words (scan_rows_wragged<-LPR, U>; its LDS area is FusedShared::store.words).
template <int LPR, int U, bool WORDS = false>
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
        sh.ok = 1u;        // (cleared by a selector wave that gives up waiting or meets a workgroup that failed)
        sh.repmin = 0ull;  // (the workgroup's report: 0 = it holds fewer than Mw rows)
    }
    if (tid < kFusedCheckpoints) sh.ck_cnt[tid] = 0;
    if (tid == 0 && (fa.xflags & 4u)) {
        // seeded by the sample kernel (narrow rows): QueryState::gtau holds a coarse BIN B -- k sampled rows score at least
        // B / 1024 (coarse_bin: exact, a power-of-two scale) -- which becomes the first threshold, as a score key.  Every
        // workgroup converts it for itself and raises the table-wide word (idempotent; a poller that still reads the bin
        // reads a key below every real one: harmless)
        const uint32_t raw = agent_load(&st->gtau);
        if (raw != 0u && raw < static_cast<uint32_t>(kScanBins)) {
            const uint32_t key = order_key(static_cast<float>(raw) * (1.0f / static_cast<float>(kScanBins)));
            sh.tau = key;
            atomicMax(&st->gtau, key);
        } else if (raw >= 0x80000000u) {
            sh.tau = raw; // (another workgroup's conversion)
        }
    }
    __syncthreads();
    FusedSchedule sched;
    const uint32_t CHR = LPR > 0 ? static_cast<uint32_t>(U * (64 / (LPR > 0 ? LPR : 1))) : g.chunk_rows; // rows per chunk (trip)
    const u64 nfull = a.nrows / CHR;    // full chunks
    sched.init(static_cast<uint32_t>(nfull / g.nwaves));
    // Waves 4 .. 7 do not stream: the forwarder, the poller, and two that have nothing to do until the scan is over (they wait at
    // the barrier behind it and cost the streaming waves no issue slot).  A publishing launch (large k) has no selection: they
    // leave when their service is done, as they always did.
    const bool helper = wv >= kScanBlock / 64;
    if (wv == kScanBlock / 64) fused_forwarder(sh, fa, sched, lane);
    if (wv == kScanBlock / 64 + 1) fused_poller(sh, st, fa, sched, g.nwaves, a.k, lane, dbg);
    if (helper && (fa.xflags & kFusedPublishOnly)) return;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * (kScanBlock / 64) + (helper ? 0 : wv));
    const uint32_t nwg = gridDim.x;

    FusedFilter f{};
    if (!helper) {
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
    f.init_prefilter();
    f.sched = sched;
    f.ck_j = 0;
    f.next_ck = (f.M && !(fa.xflags & 2u)) ? sched.trip(0) : 0xFFFFFFFFu;
    f.dbg = dbg;
    if constexpr (LPR > 0) {
        const u32x4 q = reinterpret_cast<const u32x4*>(a.query)[lane % LPR];
        scan_rows<LPR, U>(a, g, f, q, w, lane);
    } else if constexpr (WORDS) {
        scan_rows_wragged<-LPR, U>(a, g, f, w, lane, sh.store.words[wv]);
    } else {
        scan_rows_ragged<-LPR, U>(a, g, f, w, lane);
    }
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

#include "gsim_fused_publish.inl"
#include "gsim_fused_select.inl"
#include "gsim_fused_close.inl"
#undef GSIM_STAMP
}

#include "gsim_fused_largek.inl"

} // namespace

hipError_t launch_fused_binsort(const ScanArgs& a, const FusedArgs& f, uint32_t nwg, unsigned long long* finalists, uint32_t cap, uint32_t* cursors, hipStream_t s)
{
    hipLaunchKernelGGL(fused_binsort_kernel, dim3(nwg < kBinsortGrid ? nwg : kBinsortGrid), dim3(256), 0, s, a, f, nwg, finalists, cap, cursors);
    return hipGetLastError();
}

hipError_t launch_fused_handoff(const ScanArgs& a, const FusedArgs& f, uint32_t nwg, unsigned long long* finalists, uint32_t cap, hipStream_t s)
{
    hipLaunchKernelGGL(fused_handoff_kernel, dim3(nwg), dim3(256), 0, s, a, f, finalists, cap);
    return hipGetLastError();
}

template <int LPR, int U, bool WORDS = false>
hipError_t launch_fused_t(const ScanArgs& a, const ScanGeometry& g, const FusedArgs& f, hipStream_t s)
{
    const uint32_t nblocks = g.nwaves / (kScanBlock / 64);
    const size_t lds = sizeof(FusedShared);
    static DynLdsOnce once;
    const hipError_t e = once.ensure(reinterpret_cast<const void*>(fused_kernel<LPR, U, WORDS>), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((fused_kernel<LPR, U, WORDS>), dim3(nblocks), dim3(kFusedBlock), lds, s, a, g, f);
    return hipGetLastError();
}

// Rows of 3, 5, 7, 9, 11 or twice that many 32-bit words: the single launch streams them through registers at word granularity
// (scan_rows_wragged); the four-kernel pipeline keeps its LDS-staged scan and its own geometry for them.
bool fused_word_geometry(uint64_t nrows, uint32_t W, int num_cus, ScanGeometry* out, bool ragged)
{
    if (!ragged || W % 4 == 0 || W == 0) return false;
    uint32_t odd = W;
    while (odd % 2 == 0) odd /= 2;
    if ((odd != 3 && odd != 5 && odd != 7 && odd != 9 && odd != 11) || W / odd > 2) return false; // (13, 15: no LDS left for their words)
    ScanGeometry g{};
    g.ragged_loads = odd;
    g.ragged_words = 1;
    g.unroll = odd == 3 ? 3 : (odd == 5 ? 2 : 1); // sub-chunks per trip
    g.chunk_rows = g.unroll * (256u * odd / W);
    g.nchunks = (nrows + g.chunk_rows - 1) / g.chunk_rows;
    uint64_t nw = static_cast<uint64_t>(num_cus) * (kScanBlock / 64);
    if (nw > g.nchunks) nw = g.nchunks;
    if (nw < 1) nw = 1;
    nw = (nw + 3) / 4 * 4;
    g.nwaves = static_cast<uint32_t>(nw);
    g.seg_cap = 0; // (no candidate segments: the single launch keeps its candidates in LDS)
    *out = g;
    return true;
}

bool fused_supported(const ScanGeometry& g)
{
    // every workgroup of the grid is a selector and reads every workgroup's header with one thread
    return ((g.lanes_per_row != 0 && g.unroll == 8) || g.ragged_loads != 0) && g.nwaves <= static_cast<uint32_t>(kFusedSelectors) * (kScanBlock / 64);
}

// M of the checkpoint summaries ("my M-th best key"): about 2k / nwaves, so that the election's rank
// r = ceil(k / M) sits in the middle of the reports; 0 when even M = 16 leaves r above the number of
// waves (tiny grids: no thresholds, every row is published) or the reports would not fit the
// electing wave's registers (64 x 64 keys).
uint32_t fused_summary_keys(uint32_t nwaves, uint32_t k, uint32_t max_m)
{
    if (nwaves == 0 || nwaves > static_cast<uint32_t>(kFusedSelectors) * (kScanBlock / 64)) return 0;
    uint32_t m = (2 * k + nwaves - 1) / nwaves;
    if (m < 1) m = 1;
    if (m > max_m) m = max_m; // (16 for the single launch's own k; up to 64 when it only publishes: mth_best's M rounds)
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
    if (g.ragged_words) {
        if (a.W == 3) return launch_fused_t<-3, 3, true>(a, g, f, s);
        if (a.W == 5) return launch_fused_t<-5, 2, true>(a, g, f, s);
        if (a.W == 7) return launch_fused_t<-7, 1, true>(a, g, f, s);
        if (a.W == 6) return launch_fused_t<-6, 3, true>(a, g, f, s);
        if (a.W == 10) return launch_fused_t<-10, 2, true>(a, g, f, s);
        if (a.W == 14) return launch_fused_t<-14, 1, true>(a, g, f, s);
        if (a.W == 9) return launch_fused_t<-9, 1, true>(a, g, f, s);
        if (a.W == 18) return launch_fused_t<-18, 1, true>(a, g, f, s);
        if (a.W == 11) return launch_fused_t<-11, 1, true>(a, g, f, s);
        if (a.W == 22) return launch_fused_t<-22, 1, true>(a, g, f, s);
        return hipErrorInvalidValue;
    }
    if (g.ragged_loads == 3) return launch_fused_t<-3, 3>(a, g, f, s);
    if (g.ragged_loads == 5) return launch_fused_t<-5, 2>(a, g, f, s);
    if (g.ragged_loads == 7) return launch_fused_t<-7, 1>(a, g, f, s);
    if (g.ragged_loads == 9) return launch_fused_t<-9, 1>(a, g, f, s);
    if (g.ragged_loads == 11) return launch_fused_t<-11, 1>(a, g, f, s);
    if (g.ragged_loads == 13) return launch_fused_t<-13, 1>(a, g, f, s);
    if (g.ragged_loads == 15) return launch_fused_t<-15, 1>(a, g, f, s);
    return hipErrorInvalidValue;
}

} // namespace gsim
