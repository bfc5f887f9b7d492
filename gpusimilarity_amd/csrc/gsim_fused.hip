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
//      sort by score key >> shift, highest bucket first); its 16-byte header holds the count, the order, the shift,
//      the workgroup's REPORT, its Mw-th best 64-bit key -- and the launch's TAG, which every entry carries too.  Write-
//      through stores and nothing else: no wait for their acknowledgements, no counter -- the header IS the arrival;
//   4. every workgroup then becomes a selector -- all EIGHT of its waves: the two service waves and two that waited out the
//      scan at a barrier join the four streaming waves (the phases below are bound by instruction issue and LDS round
//      trips).  Two threads look after a region; they poll its header until it carries the launch's tag -- the ONE grid-wide
//      wait of the kernel; bounded by a few scan times of wall clock: on a GPU shared with another queue part of the grid
//      may not have started while the waiters hold their CUs, the query then goes to the four-kernel pipeline, which never
//      waits -- and the first 16 entries of a region are requested as soon as its header has shown up (four regions per
//      load): the lists of the workgroups that finished early are in LDS before the last one has published.  An entry
//      without the tag was overtaken by its header and is read again.  From the reports every selector derives the SAME
//      final threshold (a report with at least r - 1 larger ones, found through 32 sampled reports; a 64-bit key -- it
//      carries the row index, so it also cuts through groups of equal scores), keeps the published rows at or above it in
//      LDS (a list is read on, up to 256 entries in the first round, until an entry proves the rest lies below the
//      threshold) and ranks the rows it owns (hash of the row) -- by counting larger keys, or through a histogram of the
//      finalists when there are many -- the output slot of a hit is its rank, keys are unique; the hits of
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
constexpr int kFusedBlock = 2 * kScanBlock; // four streaming waves, two service waves (forwarder, poller/elector) and two that only wait: all
                                            // EIGHT are selectors -- the phases behind the scan are bound by instruction issue and LDS
                                            // round trips, and a SIMD with two waves issues while one of them waits
constexpr uint32_t kFusedPrefix = 8;    // entries of a region ONE thread takes before the region's count is known (two threads per region on
                                        // a full grid: sixteen entries of every region are requested)
constexpr uint32_t kFusedSortCap = 128; // a workgroup with up to this many rows publishes them in canonical order (the count is
                                        // quadratic: 256 rows that all sit in one wave's store cost 10 us); more: in bucket order
constexpr uint32_t kFusedItems = 1024;  // 64-entry reads beyond the prefixes a selector lists per round (at most 4 per region)
constexpr uint32_t kFusedRankDirect = 3072; // up to this many finalists a selector ranks its rows by comparing each with every finalist
constexpr uint32_t kFusedBins = 1024;   // buckets of the order in which a workgroup with more than kFusedSortCap rows publishes them

struct FusedShared { // (static_assert below: it fits the CU's 160 KB)
    union {
        struct { // while streaming
            u64 key[kScanBlock / 64][kFusedWaveCap];
            uint32_t cb[kScanBlock / 64][kFusedWaveCap];
            uint32_t words[kScanBlock / 64][256 * 12]; // scan_rows_wragged's per-word counts (the streaming part of the union: 144 KB, as the selectors')
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
    uint32_t exact;                 // selectors: no sampled report qualified as the final threshold -- every report is ranked
    uint32_t cks, cks_total;        // selectors: sum of the words of the hits this workgroup wrote / of all hits (the closer)
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
            uint32_t queue[kFusedBlock / 64][128]; // per wave: (finalist, row of this selector in its bucket) pairs to compare
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
    // Narrow rows (up to 512 bits: a lane scores a row for every one or two 16-byte loads) are bound by the per-row
    // arithmetic, not by HBM: the reference's f32 divide, the order key and the compare cost ~30 vector instructions per
    // row, the popcounts 12.  "score >= threshold" is linear in the counts (gsim_prefilter.h: c >= ka + kb b, conservative
    // under f32 rounding, proven exhaustively by tests/cpp/prefilter_check.cpp for every (a, b, c) of these widths and any
    // achievable score as the level), so a row is scored only when it may reach the wave's current threshold -- with a
    // threshold in place: a handful per thousand.  Without a cutoff only (a cutoff needs every row's exact score for `approx`).
    float pk_ka, pk_kb;  // the pair test at the level of `pk_tau`
    uint32_t pk_tau;

    __device__ __forceinline__ uint32_t load_gtau() const { return 0u; } // (no polls from the streaming loop)

    __device__ __forceinline__ void init_prefilter()
    {
        pk_tau = 0;
        pk_ka = 0.0f; // (no threshold yet: everything passes)
        pk_kb = 0.0f;
    }

    __device__ __forceinline__ void update_prefilter(const ScanArgs& a)
    {
        if (tau == pk_tau) return; // (wave-uniform; the threshold moves a few times per query)
        pk_tau = tau;
        const PrefilterConstants pk = prefilter_constants(a.metric == GSIM_METRIC_TVERSKY, a.alpha, a.beta, a.qpop,
                                                          prefilter_level(true, key_score(tau), 0u), true);
        pk_ka = pk.ka;
        pk_kb = pk.kb;
    }

    template <int LPR> __device__ __forceinline__ void offer_counts(bool active, uint32_t row, uint32_t val, const ScanArgs& a, int lane)
    {
        if constexpr (LPR >= 1 && LPR <= 4) { // (the register-streamed odd widths pass LPR = 64: their rows are wider than the proof covers)
            if (!has_cutoff) { // (wave-uniform)
                update_prefilter(a);
                const bool maybe = active && static_cast<float>(val >> 16) >= __builtin_fmaf(pk_kb, static_cast<float>(val & 0xFFFFu), pk_ka);
                if (__ballot(maybe) == 0) return; // no row of this round can reach the threshold: none is scored
                active = maybe; // (a row the test rejects scores below the threshold: not a candidate, and nothing counts it)
            }
        }
        offer_scored(*this, active, row, val, a, lane);
    }

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
        if (M > 64u) return mth_best_deep(lane); // (wave-uniform; only the publishing launch of k above 65 536)
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

    // ... with eight keys per lane (512 per wave) for M up to 256: k above 65 536 through the publishing launch.  Rare and long
    // (M rounds).  Scalars, not an array, and inlined: a call or an indexed array put the kernel on scratch memory.
    __device__ __forceinline__ u64 mth_best_deep(int lane) const
    {
        u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0;
        for (uint32_t i = lane; i < staged; i += 64) {
            const u64 v = skey[i];
            if (v > t7) {
                t7 = v;
                if (t7 > t6) { const u64 x = t6; t6 = t7; t7 = x; }
                if (t6 > t5) { const u64 x = t5; t5 = t6; t6 = x; }
                if (t5 > t4) { const u64 x = t4; t4 = t5; t5 = x; }
                if (t4 > t3) { const u64 x = t3; t3 = t4; t4 = x; }
                if (t3 > t2) { const u64 x = t2; t2 = t3; t3 = x; }
                if (t2 > t1) { const u64 x = t1; t1 = t2; t2 = x; }
                if (t1 > t0) { const u64 x = t0; t0 = t1; t1 = x; }
            }
        }
        u64 mth = 0;
#pragma unroll 1
        for (uint32_t r = 0; r < M; r++) {
            const uint32_t hi = wave_max_u32(static_cast<uint32_t>(t0 >> 32));
            const uint32_t lo = wave_max_u32(static_cast<uint32_t>(t0 >> 32) == hi ? static_cast<uint32_t>(t0) : 0u);
            mth = (static_cast<u64>(hi) << 32) | lo;
            const u64 b = __ballot(t0 == mth);
            if (lane == __builtin_ctzll(b)) {
                t0 = t1; t1 = t2; t2 = t3; t3 = t4; t4 = t5; t5 = t6; t6 = t7; t7 = 0;
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
        // four batches of 64 entries per trip, all eight LDS reads in flight before the first write (a wave's LDS
        // operations execute in order and the survivors land at or below positions already read: one read at a time cost
        // ~2 us at the end of a 1 M-row scan -- a thousand entries per wave, a dependent LDS round trip per batch)
        uint32_t out = 0;
        for (uint32_t base = 0; base < staged; base += 256) {
            u64 key[4];
            uint32_t cb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = base + 64u * u + lane;
                const bool in = i < staged;
                key[u] = in ? skey[i] : 0ull;
                cb[u] = in ? scb[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = base + 64u * u + lane;
                const bool keep = i < staged && static_cast<uint32_t>(key[u] >> 32) >= tau;
                const u64 m = __ballot(keep);
                if (keep) {
                    const uint32_t slot = out + lane_rank(m);
                    skey[slot] = key[u];
                    scb[slot] = cb[u];
                }
                out += static_cast<uint32_t>(__popcll(m));
            }
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
    // an election this workgroup's forwarder won (consumed with an exchange: a request stored between a plain load and a plain
    // clear would be lost); the HIGHEST checkpoint whose election has been held is recorded: one election may serve two requests
    // that the same workgroup won back to back, and only the last checkpoint's matters to those who wait
    auto serve = [&]() -> bool {
        uint32_t req = 0;
        if (lane == 0 && __hip_atomic_load(&sh.elect_req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) req = atomicExch(&sh.elect_req, 0u);
        req = __builtin_amdgcn_readfirstlane(req);
        if (!req || !active) return false;
        fused_elect(sh, st, fa.summ, nwaves, (k + fa.summ_keys - 1) / fa.summ_keys, lane, dbg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the threshold is out before the count
        if (lane == 0) atomicMax(&st->elected, req);
        return true;
    };
    for (uint32_t spins = 0;; spins++) {
        (void) serve(); // (before the poll as well: the poll is a ~1.5 us round trip, and an election is the longest step of a checkpoint)
        const u64 ge = __hip_atomic_load(reinterpret_cast<const u64*>(&st->gtau), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // {gtau, elected}: one poll
        const uint32_t g = static_cast<uint32_t>(ge), el = static_cast<uint32_t>(ge >> 32);
        if (lane == 0) {
            if (g > __hip_atomic_load(&sh.tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) atomicMax(&sh.tau, g);
            __hip_atomic_store(&sh.elected, el, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); // (after the threshold it belongs to)
        }
        // a poll every ~2 us at first, every ~5 us from the 64th on, every ~60 us from the 512th on
        const uint32_t naps = spins < 512u ? 1u : 16u;
        for (uint32_t i = 0; i < naps; i++) {
            // (a request is served at the top of the loop -- one copy of the election code, it is fetched cold in every launch --
            // and the poll right behind it: this workgroup's own waves want the count too)
            if (__hip_atomic_load(&sh.elect_req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) break;
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
    } // (!helper)
    __syncthreads(); // (released once the service waves are here too)
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
    const uint32_t tag = fa.pub_tag; // every entry's fourth word: a reader tells this launch's entries from what the region held before
    // (GSIM_FUSED_FLAGS=4096, the parity suite's way into the selectors' read-again path: the entries leave with the PREVIOUS
    // launch's tag and get their own a few microseconds after the header -- every selector meets entries "still on their way")
    const bool late_tags = (fa.xflags & 4096u) != 0 && !(fa.xflags & kFusedPublishOnly);
    const uint32_t etag = late_tags ? tag - 1u : tag;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        static_cast<unsigned char*>(fa.pub) + static_cast<size_t>(blockIdx.x) * (kFusedRegion * 16u), 0, kFusedRegion * 16u, 0x00020000);
    const uint32_t mine_n = (bad || helper) ? 0u : f.staged; // (the waves that did not stream take part in the barriers only)
    uint32_t shift = 0;
    if (sorted) {
        // Every live thread takes part, the waves that did not stream too (eight of them, four in a publishing launch): SG lanes
        // share a row and each counts the larger keys among every SG-th PAIR of every wave's store -- the reads of a store's first
        // 8 SG entries requested together, the four stores' back to back: one LDS round trip (a lane per row and a read at a time: 2.5 us)
        // -- then a shuffle sum.  Up to kFusedSortCap rows: one pass of the eight waves.
        constexpr uint32_t SG = 4;
        const uint32_t nthr = (fa.xflags & kFusedPublishOnly) ? static_cast<uint32_t>(kScanBlock) : static_cast<uint32_t>(kFusedBlock);
        const uint32_t c0 = sh.wcount[0], c1 = sh.wcount[1], c2 = sh.wcount[2];
        const uint32_t sub = static_cast<uint32_t>(tid) % SG;
        for (uint32_t r0 = 0; r0 < ntot; r0 += nthr / SG) { // (ntot = 0 when a store overflowed)
            const uint32_t rho = r0 + static_cast<uint32_t>(tid) / SG;
            const bool have = rho < ntot;
            uint32_t w2 = 0, i = have ? rho : 0u; // row rho of the workgroup = row i of wave w2's store
            if (have && i >= c0) {
                i -= c0;
                w2 = 1;
                if (i >= c1) {
                    i -= c1;
                    w2 = 2;
                    if (i >= c2) {
                        i -= c2;
                        w2 = 3;
                    }
                }
            }
            const u64 key = have ? sh.store.key[w2][i] : ~0ull;
            uint32_t pos = 0; // the number of larger keys (keys are unique)
#pragma unroll
            for (int v = 0; v < kScanBlock / 64; v++) {
                const uint32_t cnt = sh.wcount[v], npair = (cnt + 1u) >> 1;
                const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(sh.store.key[v]);
                ulonglong2 kk[4];
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) kk[u] = k2[sub + SG * u < npair ? sub + SG * u : 0u];
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) {
                    const uint32_t j = sub + SG * u;
                    pos += (2u * j < cnt && kk[u].x > key) ? 1u : 0u;
                    pos += (2u * j + 1u < cnt && kk[u].y > key) ? 1u : 0u;
                }
                for (uint32_t j = sub + SG * 4u; j < npair; j += SG) { // (a store of more than 32 rows: the rest, a read at a time)
                    const ulonglong2 k1 = k2[j];
                    pos += k1.x > key ? 1u : 0u;
                    pos += (2u * j + 1u < cnt && k1.y > key) ? 1u : 0u;
                }
            }
            pos += static_cast<uint32_t>(__shfl_xor(static_cast<int>(pos), 1, 64));
            pos += static_cast<uint32_t>(__shfl_xor(static_cast<int>(pos), 2, 64));
            if (have && sub == 0) {
                if (Mw && pos == Mw - 1u) sh.repmin = key; // the workgroup's report (one thread holds it)
                const u32x4 e{static_cast<uint32_t>(key), static_cast<uint32_t>(key >> 32), sh.store.cb[w2][i], etag};
                __builtin_amdgcn_raw_buffer_store_b128(e, rsrc, pos * 16u, 0, /*sc1: write-through*/ 16);
            }
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
            const u32x4 e{static_cast<uint32_t>(key), static_cast<uint32_t>(key >> 32), f.scb[i], etag};
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
    }
    // The header, ONE 16-byte store: {entries | exact order << 31, bucket shift | the launch's tag << 5 | "this workgroup failed" << 31,
    // report (0: fewer than Mw rows)}.  It is the workgroup's ARRIVAL: no counter, no wait for the entries' acknowledgements -- a
    // selector takes a region's header for this query's by the tag and every entry for this query's by ITS tag (an entry that is
    // still on its way when the header has landed is read again).  Issued behind a barrier: every wave's entry stores are
    // ahead of it in the memory pipeline (they rarely lose the race), and the LDS store is free for the selectors.
    __syncthreads();
    const bool publish_only = (fa.xflags & kFusedPublishOnly) != 0;
    if (tid == 0) {
        const u64 rep = sh.repmin;
        const bool failed = bad || __hip_atomic_load(&sh.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
        const uint32_t w1 = shift | ((tag & 0x3FFFFFFu) << 5) | (failed && !publish_only ? 0x80000000u : 0u);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{ntot | (sorted ? 0x80000000u : 0u), w1, static_cast<uint32_t>(rep), static_cast<uint32_t>(rep >> 32)},
                                               hrsrc_w, blockIdx.x * kFusedHeaderBytes, 0, /*sc1*/ 16);
        if (!publish_only) { // the selectors' state (first touched behind their next barrier)
            sh.nfin = 0;
            sh.nmine = 0;
            sh.nitems[0] = 0;
            sh.nitems[1] = 0;
            sh.repmin = 0ull; // (from here on: the finalists' summed distance from the threshold)
            sh.tauf = 0ull;
            sh.cks = 0u;
            sh.exact = 0u;
            if (bad) atomicOr(&st->redo, kRedoStore);               // (statistics: the closer adds the reasons up; every selector
            atomicAdd(&st->ncand, static_cast<u64>(sh.nemit));      //  learns of a failure from the headers)
        }
    }
    if (late_tags) {
        for (int i = 0; i < 3 + static_cast<int>(blockIdx.x % 3u); i++) __builtin_amdgcn_s_sleep(127);
        for (uint32_t i = static_cast<uint32_t>(tid); i < ntot; i += kFusedBlock) __builtin_amdgcn_raw_buffer_store_b32(tag, rsrc, i * 16u + 12u, 0, /*sc1*/ 16);
    }
    if (publish_only) {
        // ... and, when the large-k kernels rank the lists, what the four-kernel pipeline's scan leaves for them: the published
        // rows counted per coarse bin in QueryState::ghist (fused_handoff_kernel and largek_one_block_kernel start from it)
        static_assert(kFusedBins >= static_cast<uint32_t>(kScanBins), "the publish phase's bucket counters double as the coarse histogram");
        __syncthreads(); // (the bucket order is done with the counters)
        for (int i = tid; i < kScanBins; i += kScanBlock) sh.hist[i] = 0;
        __syncthreads();
        for (uint32_t i = lane; i < mine_n; i += 64) atomicAdd(&sh.hist[coarse_bin(key_score(static_cast<uint32_t>(f.skey[i] >> 32)))], 1u);
        __syncthreads();
        for (int i = tid; i < kScanBins; i += kScanBlock)
            if (sh.hist[i]) atomicAdd(&st->ghist[i], sh.hist[i]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave: its entries are out
        __syncthreads();
    }
    if (publish_only && tid == 0) {
        if (bad) atomicOr(&st->redo, kRedoStore);
        // A publishing launch counts its workgroups in (nobody waits: the LAST one tidies up), two levels (MI355X_MICROARCH.md
        // "barrier-xcd"): a counter per group of workgroups b % 8 (the XCD a block lands on, as observed -- only speed depends on
        // it), the group's last arriver adds to the top counter.
        const uint32_t x = blockIdx.x % 8u;
        const uint32_t group_size = (nwg - x + 7u) / 8u, ngroups = nwg < 8u ? nwg : 8u;
        atomicAdd(&st->ncand, static_cast<u64>(sh.nemit)); // (before the arrival: the next launch sums it up)
        const bool last = atomicAdd(&fa.arrive[x * 32u], 1u) == group_size - 1u && atomicAdd(&fa.arrive[8u * 32u], 1u) == ngroups - 1u;
        sh.ticket = last ? 1u : 0u;
    }
    GSIM_STAMP(3);
    // The exchange state of the single launch (checkpoint tickets, arrival words, in-loop summaries): zero again for the
    // next query.  One workgroup does it when no other touches it any more.
    auto rezero_exchange = [&]() __attribute__((always_inline)) {
        if (tid < kFusedCheckpoints * 9) fa.tickets[tid * 32] = 0;
        if (tid < static_cast<int>(kFusedArriveWords)) { // (the closing tickets are 64-bit)
            fa.arrive[tid * 32] = 0;
            fa.arrive[tid * 32 + 1] = 0;
        }
        uint4* sm = reinterpret_cast<uint4*>(fa.summ); // (16-byte stores)
        const uint32_t n16 = (g.nwaves + 3) / 4;
        for (uint32_t i = tid; i < n16; i += kScanBlock) sm[i] = uint4{0, 0, 0, 0}; // (the selectors' waves 4 .. 7 repeat some: zeros)
    };
    if (fa.xflags & kFusedPublishOnly) {
        // k above kFusedMaxK: the scan and its thresholds are this launch's, the ranking is the large-k kernels' (they are sized by
        // k, the selectors' LDS is not).  Nobody waits for anybody: the LAST workgroup to arrive -- every other one has
        // published, its service waves are gone -- tidies up; launch_fused_handoff, next on the stream, reads the lists.
        __syncthreads();
        if (!sh.ticket) return;
        if (tid == 0) {
            const uint32_t why = agent_load(&st->redo); // (set before its workgroup's arrival)
            if (why) { // handed back: the gated classic kernels behind this launch start from a clean state
                st->redo_sum += 1u;
                st->redo_why |= why;
                __hip_atomic_store(&st->kept, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                st->ncand_sum += __hip_atomic_load(&st->ncand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&st->ncand, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(&st->gtau, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&st->elected, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh.ok = why ? 0u : 1u;
        }
        rezero_exchange();
        __syncthreads();
        if (!sh.ok) // handed back: the histogram the other workgroups added to is the classic scan's to fill
            for (int i = tid; i < kScanBins; i += kScanBlock) st->ghist[i] = 0;
        return;
    }

    // ---- 4. select: every workgroup of the grid (fused_supported: at most kFusedSelectors) ------
    // There is no arrival to wait for: a selector watches the HEADERS.  Thread t looks after virtual region t (on a full grid:
    // entries 0 .. 7 or 8 .. 15 of region t / 2) -- wave w after virtual regions 64 w .. 64 w + 63 -- and polls its region's
    // header until it carries this launch's tag; as soon as the eight virtual regions of a group (eight lanes fetch the
    // kFusedPrefix entries of one: 64 lanes = eight per load) have shown up, the wave requests their entries straight into LDS
    // (global_load_lds, 16 B per lane, no registers).  The prefixes of the
    // workgroups that finish early arrive while the stragglers are still publishing; behind the last header there is one group's
    // round trip left (before: every selector waited for a counted arrival and then fetched all 64 KB of prefixes, 4.9 us).
    // On a GPU this kernel has to itself the wait is the spread of the streaming end times.  When another queue holds part of
    // the CUs, workgroups of this grid may not have started yet and will not while the waiters keep theirs: after
    // fa.wait_ticks (a few scan times) without a header the query goes to the classic kernels, which never wait.
    const uint32_t nsel = nwg, r = blockIdx.x;
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(fa.pub, 0, nwg * (kFusedRegion * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(fa.hdr, 0, nwg * kFusedHeaderBytes, 0x00020000);
    constexpr int PL = static_cast<int>(kFusedPrefix);
    static_assert(kFusedPrefix == 8, "a group = 64 lanes = eight virtual regions' entries; slot rotation mod 8");
    // Slot s = 8 g + p of the staging area receives entry (p - g) mod 8 of virtual region g: thread g later walks ITS
    // entries, and the rotation spreads the 64 lanes over all banks.  The staging area is the upper half of the finalist
    // array: at most 4096 staged entries become finalists.
    u32x4* staging = reinterpret_cast<u32x4*>(&sh.sel.fkey[kFusedFinalLds / 2]);
    // S = 2 threads share a region on a full grid, S = 4, 8, ... on a grid of fewer than 129 workgroups (small tables), each
    // taking eight consecutive entries of it ("virtual region" v = S g + part), so that the requested prefix is 8 S entries --
    // the finalists per region grow as the grid shrinks.
    uint32_t lgS = 1;
    while ((nwg << (lgS + 1u)) <= static_cast<uint32_t>(kFusedBlock)) lgS++;
    const uint32_t my_region = static_cast<uint32_t>(tid) >> lgS, my_part = static_cast<uint32_t>(tid) & ((1u << lgS) - 1u);
    const uint32_t htag = tag & 0x3FFFFFFu;
    if (tid < 128) sh.hist[tid] = 0u; // (the publish phase is done with its bucket counters: the election's, see below)
    u32x4 hd{0u, 0u, 0u, 0u}; // (zeros past the grid, and for a header that never came)
    {
        const unsigned char* pubc = static_cast<const unsigned char*>(fa.pub);
        bool pend = my_region < nwg;
        u64 issued = 0; // bit 8 u: the entries of this wave's group u have been requested
        const unsigned long long t_wait = wall_clock64();
        for (uint32_t spins = 0;; spins++) {
            if (pend) {
                const u32x4 h = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, my_region * kFusedHeaderBytes, 0, /*sc1*/ 16);
                if (((h.y >> 5) & 0x3FFFFFFu) == htag) {
                    hd = h;
                    pend = false;
                }
            }
            const u64 pm = __ballot(pend);
            u64 any8 = pm | (pm >> 1);
            any8 |= any8 >> 2;
            any8 |= any8 >> 4; // bit 8 u: one of lanes 8 u .. 8 u + 7 still waits for its header
            u64 todo = ~any8 & 0x0101010101010101ull & ~issued;
            issued |= todo;
            while (todo) {
                const uint32_t u = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(__builtin_ctzll(todo) >> 3)));
                todo &= todo - 1ull;
                const uint32_t v = static_cast<uint32_t>(wv) * 64u + 8u * u + (static_cast<uint32_t>(lane) >> 3); // virtual region of this lane's entry
                const uint32_t gi = v >> lgS, ent = ((v & ((1u << lgS) - 1u)) * kFusedPrefix) + ((static_cast<uint32_t>(lane) - v) & (kFusedPrefix - 1u));
                if (gi < nwg)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*) (pubc + static_cast<size_t>(gi) * (kFusedRegion * 16u) + ent * 16u),
                        (__attribute__((address_space(3))) void*) (staging + (static_cast<uint32_t>(wv) * 64u + 8u * u) * kFusedPrefix), 16, 0, /*sc1*/ 16);
            }
            if (pm == 0) break;
            if ((spins & 63u) == 63u && wall_clock64() - t_wait > fa.wait_ticks) {
                if (lane == 0) atomicOr(&st->redo, kRedoArrivalWait);
                break;
            }
            // (the wait is a few polls on a GPU this kernel has to itself; a long one means the rest of the grid cannot start
            // -- another process holds CUs -- and 512 threads per workgroup polling flat out would take ~2 TB/s from ITS scan)
            if (spins < 16u) __builtin_amdgcn_s_sleep(1);
            else if (spins < 256u) __builtin_amdgcn_s_sleep(32); // ~1 us
            else __builtin_amdgcn_s_sleep(127);                  // ~3.4 us
        }
        // a header that never came, or one whose workgroup failed (a store that overflowed, an election it gave up waiting for)
        if (__ballot(pend || (hd.y >> 31) != 0) != 0 && lane == 0) sh.ok = 0u;
    }
    GSIM_STAMP(4);
    // (the election below runs while the last groups' prefixes land)
    const uint32_t n_mine = (hd.x & 0x7FFFFFFFu) < kFusedRegion ? (hd.x & 0x7FFFFFFFu) : kFusedRegion; // entries of region my_region
    const bool sorted_mine = (hd.x >> 31) != 0;
    const u64 rep_mine = (Mw && n_mine >= Mw && my_part == 0) ? ((static_cast<u64>(hd.w) << 32) | hd.z) : 0ull; // (one thread per region holds its report)
    if (my_part == 0) sh.sel.u.rep[my_region] = rep_mine;
    if (static_cast<uint32_t>(tid) >= (static_cast<uint32_t>(kFusedBlock) >> lgS) && tid < kFusedSelectors) sh.sel.u.rep[tid] = 0ull; // (past the grid)
    __syncthreads(); // the reports of all regions
    const bool good0 = sh.ok != 0; // (not good: headers and regions may be stale -- nothing below is used, the query is handed back)
    // The final threshold: a report with at least r - 1 larger ones, r = ceil(k / Mw) -- each of the r largest reports stands for
    // Mw distinct rows at or above it in the canonical order, so at least k rows are at or above such a report: no row of the top
    // k lies below it.  The keys carry the row index: the threshold also cuts through a group of equal scores.  The r-th largest
    // itself is the tightest, and ranking every report against every other (65 k 64-bit compares per selector) took 2.6 us of
    // instruction issue.  Instead: the reports of regions 0 .. 31 are SAMPLES.  Every report counts the samples above it -- its
    // bucket b; a report in a lower bucket is larger than every report in a higher one, and inside a sample's own bucket every
    // other report is larger than the sample -- so the bucket populations give every sample's exact rank:
    // rank(s) = population of buckets 0 .. b(s), minus one.  The threshold is the sample with the smallest rank >= r - 1 (about
    // 256 / 33 reports -- 40 rows -- beyond the r-th largest).  No such sample (all 32 among the r - 1 largest: by (199/256)^32 about 3 in 10 000 queries
    // at k = 1000), or a grid without them: every report is ranked, as before.  Every selector finds the same value.
    constexpr uint32_t kSamples = 32;
    const uint32_t rr = Mw ? (a.k + Mw - 1u) / Mw : 0u;
    // (the region's second thread gets the report from the first: lanes 2 m and 2 m + 1, since S is even)
    const uint32_t nlo = static_cast<uint32_t>(__shfl(static_cast<int>(static_cast<uint32_t>(rep_mine)), lane & ~1, 64));
    const uint32_t nhi = static_cast<uint32_t>(__shfl(static_cast<int>(static_cast<uint32_t>(rep_mine >> 32)), lane & ~1, 64));
    const u64 rep_reg = my_part == 1 ? ((static_cast<u64>(nhi) << 32) | nlo) : rep_mine;
    if (good0 && Mw) {
        uint32_t bkt = 0;
        if (my_part < 2) { // the region's two threads: sixteen samples each
            const ulonglong2* s2 = reinterpret_cast<const ulonglong2*>(sh.sel.u.rep) + (my_part & 1u) * (kSamples / 4);
#pragma unroll
            for (uint32_t j = 0; j < kSamples / 4; j++) {
                const ulonglong2 kk = s2[j];
                bkt += kk.x > rep_reg ? 1u : 0u;
                bkt += kk.y > rep_reg ? 1u : 0u;
            }
        }
        bkt += static_cast<uint32_t>(__shfl_xor(static_cast<int>(bkt), 1, 64));
        if (my_part == 0 && rep_mine != 0ull) {
            atomicAdd(&sh.hist[bkt], 1u); // (zero since the selectors' start; absent reports are not counted)
            if (my_region < kSamples) sh.hist[64u + my_region] = bkt;
        }
    }
    __syncthreads(); // the buckets' populations
    if (good0 && Mw && wv == 0) {
        uint32_t incl = static_cast<uint32_t>(lane) <= kSamples ? sh.hist[lane] : 0u; // lane b: the reports in bucket b ...
        { // ... in buckets 0 .. b (DPP inside the 16-lane rows, readlanes across them: a shuffle chain costs ~700 cycles)
            uint32_t o;
            o = dpp_shr<1>(incl); incl += o;
            o = dpp_shr<2>(incl); incl += o;
            o = dpp_shr<4>(incl); incl += o;
            o = dpp_shr<8>(incl); incl += o;
            const uint32_t row0 = __builtin_amdgcn_readlane(incl, 15), row1 = __builtin_amdgcn_readlane(incl, 31);
            incl += (lane >= 16 ? row0 : 0u) + (lane >= 32 ? row1 : 0u); // (buckets 0 .. 32: rows 0 .. 2)
        }
        const u64 smp = static_cast<uint32_t>(lane) < kSamples ? sh.sel.u.rep[lane] : 0ull; // lane i: sample i, its bucket, its rank
        const uint32_t sb = static_cast<uint32_t>(lane) < kSamples ? sh.hist[64u + static_cast<uint32_t>(lane)] : 0u;
        const uint32_t srank = static_cast<uint32_t>(__shfl(static_cast<int>(incl), static_cast<int>(sb <= kSamples ? sb : 0u), 64)) - 1u;
        const bool cand = smp != 0ull && srank >= rr - 1u;
        const uint32_t best = ~wave_max_u32(cand ? ~((srank << 6) | static_cast<uint32_t>(lane)) : 0u); // the smallest (rank, lane) among them
        if (cand && ((srank << 6) | static_cast<uint32_t>(lane)) == best) sh.tauf = smp;
        if (lane == 0) sh.exact = (best == ~0u) ? 1u : 0u; // no sample qualifies: every report is ranked
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's part of the prefixes is in LDS
    __syncthreads();                                 // ... and everybody's; the threshold is known
    // (GSIM_FUSED_FLAGS=8192: always -- the only way to reach this path on purpose; it then overrides the sample's threshold)
    if (good0 && Mw && (sh.exact != 0u || (fa.xflags & 8192u) != 0u)) { // (rare) no sample had r - 1 reports above it: the r-th largest report, by ranking all
        const ulonglong2* r2 = reinterpret_cast<const ulonglong2*>(sh.sel.u.rep) + (my_part & 1u) * (kFusedSelectors / 4);
        uint32_t rank = 0;
        if (my_part < 2) {
#pragma unroll 8
            for (int j = 0; j < kFusedSelectors / 4; j++) {
                const ulonglong2 kk = r2[j];
                rank += kk.x > rep_reg ? 1u : 0u;
                rank += kk.y > rep_reg ? 1u : 0u;
            }
        }
        rank += static_cast<uint32_t>(__shfl_xor(static_cast<int>(rank), 1, 64));
        if (my_part == 0 && rep_mine != 0ull && rank == rr - 1u) sh.tauf = rep_mine;
        __syncthreads();
    }
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
        {
            // An entry that does not carry this launch's tag was overtaken by its header: read again, from memory, until it has
            // landed (its store was issued before the header's: a matter of a fraction of a microsecond, and rare).
            auto stale = [&]() -> uint32_t {
                uint32_t m = 0;
#pragma unroll
                for (int j = 0; j < PL; j++) m |= (static_cast<uint32_t>(j) < npre && ev[j].w != tag) ? (1u << j) : 0u;
                return m;
            };
            uint32_t stm = stale();
            if (__ballot(stm != 0u) != 0) {
                const unsigned long long t_wait = wall_clock64();
                const uint32_t base = my_region * (kFusedRegion * 16u) + first * 16u;
                do {
#pragma unroll
                    for (int j = 0; j < PL; j++)
                        if ((stm >> j) & 1u) ev[j] = __builtin_amdgcn_raw_buffer_load_b128(prsrc, base + static_cast<uint32_t>(j) * 16u, 0, /*sc1*/ 16);
                    stm = stale();
                    if (wall_clock64() - t_wait > fa.wait_ticks) { // (never seen: the publisher is gone)
                        if (stm) sh.ok = 0u;
                        break;
                    }
                } while (__ballot(stm != 0u) != 0);
            }
        }
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
        // holds no entry that ends it.  The next 256 entries (what the list holds of them) become up to four items of round 0
        // (below): a long list is a series of analogs or a tie -- most of it qualifies -- and a round is a round trip (one
        // item first and "as many again" per round took three rounds, 7 us, for the 222 rows a Morgan-shaped table's
        // workgroup published).
        const bool more = good0 && my_part == (1u << lgS) - 1u && n_mine > pre_all && stopm == 0;
        const uint32_t left0 = more ? n_mine - pre_all : 0u;
        const uint32_t ni0 = (left0 + 63u) / 64u < 4u ? (left0 + 63u) / 64u : 4u;
        uint32_t wtot, wtot2;
        const uint32_t incl = wave_scan(cnt, wtot), incl2 = wave_scan(ni0, wtot2);
        uint32_t base = 0, base2 = 0;
        if (lane == 0 && wtot) base = atomicAdd(&sh.nfin, wtot); // (one LDS atomic per wave and list, not one per lane)
        if (lane == 0 && wtot2) base2 = atomicAdd(&sh.nitems[0], wtot2);
        base = __builtin_amdgcn_readfirstlane(base);
        base2 = __builtin_amdgcn_readfirstlane(base2);
        if (my_part == 0) sh.rn[my_region] = n_mine | (shift_mine << 16) | (sorted_mine ? 0x80000000u : 0u);
        for (uint32_t q = 0; q < ni0; q++) {
            const uint32_t st0 = pre_all + 64u * q, c = n_mine - st0 < 64u ? n_mine - st0 : 64u;
            sh.items[0][base2 + incl2 - ni0 + q] = my_region | ((st0 / 16u) << 8) | ((c - 1u) << 17) | (q == ni0 - 1u ? (1u << 23) : 0u);
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
        // entries of one region (one per lane); every wave takes every eighth item of the round's list, eight at a time
        // with the eight loads in flight together: 64 items per round trip, whichever regions they belong to.  A region's
        // last item of a round, if it holds no entry that ends the list, lists the region's items of the next round:
        // as many entries again as have been read beyond the prefix, at most 4 items (the list holds 4 per region).
        // item = region | first entry / 16 << 8 | (entries - 1) << 17 | last of its region << 23.
        constexpr int IF = 8;
        constexpr uint32_t NW = kFusedBlock / 64; // waves
#pragma unroll 1
        for (uint32_t round = 0;; round++) {
            if (tid == 0) sh.nitems[(round + 2u) % 4u] = 0; // (last read two rounds ago -- every wave is past that --, appended to in the next round)
            __syncthreads(); // this round's items and their number (the first time: and the finalists of the prefixes)
            const uint32_t nit = sh.nitems[round % 4u];
            if (nit == 0) break;
            const uint32_t* cur = sh.items[round & 1u];
            uint32_t* nxt = sh.items[(round + 1u) & 1u];
#pragma unroll 1
            for (uint32_t i0 = static_cast<uint32_t>(wv); i0 < nit; i0 += NW * IF) {
                u32x4 x[IF];
                uint32_t itm[IF];
                uint32_t lim = 0;
#pragma unroll
                for (int u = 0; u < IF; u++) {
                    const uint32_t idx = i0 + NW * static_cast<uint32_t>(u);
                    itm[u] = cur[idx < nit ? idx : i0]; // (past the list: this wave's first item again, not taken)
                    const uint32_t start = ((itm[u] >> 8) & 0x1FFu) * 16u, cnt = ((itm[u] >> 17) & 63u) + 1u;
                    lim |= (idx < nit && static_cast<uint32_t>(lane) < cnt) ? (1u << u) : 0u;
                    x[u] = __builtin_amdgcn_raw_buffer_load_b128(prsrc, (itm[u] & 0xFFu) * (kFusedRegion * 16u) + (start + static_cast<uint32_t>(lane)) * 16u, 0, /*sc1*/ 16);
                }
#pragma unroll
                for (int u = 0; u < IF; u++) {
                    const bool in = ((lim >> u) & 1u) != 0;
                    if (__ballot(in && x[u].w != tag) != 0) { // entries overtaken by their header (see the prefixes): read again
                        const unsigned long long t_wait = wall_clock64();
                        const uint32_t start = ((itm[u] >> 8) & 0x1FFu) * 16u;
                        do {
                            if (in && x[u].w != tag)
                                x[u] = __builtin_amdgcn_raw_buffer_load_b128(prsrc, (itm[u] & 0xFFu) * (kFusedRegion * 16u) + (start + static_cast<uint32_t>(lane)) * 16u, 0, /*sc1*/ 16);
                            if (wall_clock64() - t_wait > fa.wait_ticks) {
                                if (in && x[u].w != tag) sh.ok = 0u;
                                break;
                            }
                        } while (__ballot(in && x[u].w != tag) != 0);
                    }
                    take(in && x[u].w == tag, x[u]);
                    if (i0 + NW * static_cast<uint32_t>(u) < nit && (itm[u] >> 23) != 0) { // (wave-uniform) the region's last item of this round
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
    uint32_t cks = 0; // sum of the words of the hits this thread writes (the block's checksum, see kBlockCheckMul)
    good = good && sh.ok != 0; // (an entry that never arrived)
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
                cks += w0 + w1 + w2;
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
                for (uint32_t i = static_cast<uint32_t>(tid); i < kFusedBins; i += kFusedBlock) { // (the items are done with)
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
                for (uint32_t t = static_cast<uint32_t>(tid); t < nmine; t += kFusedBlock) {
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
                    for (uint32_t j0 = static_cast<uint32_t>(wv) * 64u; j0 < nfin; j0 += kFusedBlock) {
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
                for (uint32_t t = static_cast<uint32_t>(tid); t < nmine; t += kFusedBlock) {
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
            // (a selector owns ~k / 256 of the finalists: with few of them a whole wave shares a row, so that all eight waves work)
            const uint32_t RG = nmine <= static_cast<uint32_t>(kFusedBlock) / 64u ? 64u : (nmine <= static_cast<uint32_t>(kFusedBlock) / 32u ? 32u : 16u);
            const uint32_t sub = static_cast<uint32_t>(tid) % RG;
            for (uint32_t t0 = 0; t0 < nmine && !by_bucket; t0 += kFusedBlock / RG) {
                const uint32_t t = t0 + static_cast<uint32_t>(tid) / RG;
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
                if (RG > 32u) rank += static_cast<uint32_t>(__shfl_xor(static_cast<int>(rank), 32, 64));
                if (RG > 16u) rank += static_cast<uint32_t>(__shfl_xor(static_cast<int>(rank), 16, 64));
#pragma unroll
                for (int d = 8; d > 0; d >>= 1) rank += static_cast<uint32_t>(__shfl_xor(static_cast<int>(rank), d, 64));
                if (have && sub == 0 && rank < a.k) write_hit(mine, rank, sh.sel.u.mine.cb[t]);
            }
        }
    }
    if (!good && tid == 0) atomicOr(&st->redo, why);
    // ---- 5. the last selector closes the query -----------------------------------------------
    GSIM_STAMP(6);
    if (fa.done_flag) { // (wave-uniform; most threads wrote nothing)
        const uint32_t wsum = wave_sum_dpp(cks);
        if (lane == 0 && wsum) atomicAdd(&sh.cks, wsum);
    }
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
        // (64-bit tickets: count in bits 0..15, failures in 16..31, the checksum of the hits written so far in 32..63 -- one
        // atomic carries all three, so the last holder knows the sum without another round trip)
        const u64 mine64 = (static_cast<u64>(sh.cks) << 32) | (good ? 1ull : 0x10001ull);
        const u64 tg = atomicAdd(reinterpret_cast<u64*>(&fa.arrive[(17u + x) * 32u]), mine64);
        uint32_t closing = 0, failed = 0;
        if ((static_cast<uint32_t>(tg) & 0xFFFFu) == group_size - 1u) {
            const bool gfail = ((static_cast<uint32_t>(tg) >> 16) & 0xFFFFu) != 0 || !good;
            const uint32_t gcks = static_cast<uint32_t>((tg + mine64) >> 32);
            const u64 top64 = (static_cast<u64>(gcks) << 32) | (gfail ? 0x10001ull : 1ull);
            const u64 tt = atomicAdd(&st->sel_done, top64);
            closing = (static_cast<uint32_t>(tt) & 0xFFFFu) == ngroups - 1u ? 1u : 0u;
            failed = (((static_cast<uint32_t>(tt) >> 16) & 0xFFFFu) != 0 || gfail) ? 1u : 0u;
            sh.cks_total = static_cast<uint32_t>((tt + top64) >> 32);
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
            // (synchronous callers: the upper half of approx carries the block's checksum, see kBlockCheckMul)
            const uint32_t w3 = fa.done_flag ? sh.cks_total + fa.epoch * kBlockCheckMul : static_cast<uint32_t>(approx >> 32);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{redo ? 0u : (nfin < a.k ? nfin : a.k), flags, static_cast<uint32_t>(approx), w3},
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
        __hip_atomic_store(&st->sel_done, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // synchronous callers read the hand-back from the header (no gated kernels behind this launch): the next
        // launch, possibly already enqueued, starts clean
        if (fa.done_flag) __hip_atomic_store(&st->redo, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    rezero_exchange();
    if (dbg && tid == 0) fa.dbg[static_cast<u64>(gridDim.x) * 24] = wall_clock64(); // the very end
#undef GSIM_STAMP
}

// Behind a kFusedPublishOnly launch (k above kFusedMaxK): what the workgroups published becomes the finalist list of the
// large-k kernels -- what the four-kernel pipeline's scan and compact_kernel leave behind.  The lists hold the rows at or above
// the last IN-LOOP threshold -- taken at 3/4 of the scan from reports in the middle of their range: about 2.8 k rows -- and the
// launch has counted them per coarse bin (QueryState::ghist): this kernel keeps the rows of the bins at or above B*, the bin
// of the k-th best (about k + one bin's rows: what the one-workgroup large-k route is fast for; it starts from the same
// histogram).  Handed back (QueryState::redo): nothing is kept -- the gated classic kernels behind this one produce the
// finalists, or (synchronous callers) the emission reports it and the host runs the query again.
__global__ __launch_bounds__(256) void fused_handoff_kernel(ScanArgs a, FusedArgs fa, u64* finalists, uint32_t cap)
{
    __shared__ uint32_t s_bstar, s_cnt, s_base, s_cur;
    QueryState* st = a.state;
    const int tid = threadIdx.x, lane = tid & 63;
    if (blockIdx.x == 0 && a.query_dev != a.query) // the device copy of the query the emission reads (the classic scan's job otherwise)
        for (uint32_t i = tid; i < a.W; i += 256) a.query_dev[i] = a.query[i];
    if (agent_load(&st->redo) != 0) return; // (set before the launch ended: every workgroup reads the same)
    if (tid < 64) {
        uint32_t bstar, cnt;
        find_threshold(st->ghist, a.k, lane, bstar, cnt); // (fewer than k rows published: bin 0, every row is kept)
        if (tid == 0) {
            s_bstar = bstar;
            s_cnt = 0;
            s_cur = 0;
        }
    }
    __syncthreads();
    const uint32_t bstar = s_bstar;
    const uint32_t n = static_cast<const uint32_t*>(fa.hdr)[blockIdx.x * (kFusedHeaderBytes / 4)] & 0x7FFFFFFFu;
    const u32x4* reg = static_cast<const u32x4*>(fa.pub) + static_cast<size_t>(blockIdx.x) * kFusedRegion;
    const uint32_t n256 = (n + 255u) & ~255u;
    uint32_t mine = 0;
    for (uint32_t i = tid; i < n256; i += 256) mine += (i < n && coarse_bin(key_score(reg[i].y)) >= bstar) ? 1u : 0u;
    mine = wave_sum(mine);
    if (lane == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (tid == 0) s_base = s_cnt ? atomicAdd(&st->nfinal, s_cnt) : 0u;
    __syncthreads();
    const uint32_t base = s_base;
    for (uint32_t i = tid; i < n256; i += 256) {
        u32x4 e{0, 0, 0, 0};
        if (i < n) e = reg[i];
        const bool take = i < n && coarse_bin(key_score(e.y)) >= bstar;
        const u64 m = __ballot(take);
        if (m == 0) continue;
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(&s_cur, static_cast<uint32_t>(__popcll(m)));
        b = __builtin_amdgcn_readfirstlane(b);
        const uint32_t pos = base + b + lane_rank(m);
        if (take && pos < cap) finalists[pos] = (static_cast<u64>(e.y) << 32) | e.x; // (cap >= the table's rows: never short)
    }
}

// The same hand-off BY COARSE BIN (gsim_device.h launch_fused_binsort): every workgroup derives the layout from the histogram
// for itself, then places the rows of its regions.  A device-scope counter per bin hands out the positions inside a bin -- to
// workgroups, not rows: a workgroup counts its rows per bin in LDS first and reserves each bin's share with one atomic (one
// atomic per row queued 11 k of them on ~60 addresses: 40 us at k = 8192), and few workgroups take many regions each so that
// the shares are worth an atomic.  A top bin beyond kBinRankCap rows, or a launch that handed the query back: nothing is placed.
constexpr uint32_t kBinsortGrid = 128;

__global__ __launch_bounds__(256) void fused_binsort_kernel(ScanArgs a, FusedArgs fa, uint32_t nwg, u64* finalists, uint32_t cap, uint32_t* cursors)
{
    __shared__ uint32_t s_base[kScanBins], s_mine[kScanBins], s_off[kScanBins];
    __shared__ uint32_t s_bstar, s_cnt, s_ok;
    QueryState* st = a.state;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && a.query_dev != a.query) // the device copy of the query the emission reads (the classic scan's job otherwise)
        for (uint32_t i = tid; i < a.W; i += 256) a.query_dev[i] = a.query[i];
    if (agent_load(&st->redo) != 0) return; // (set before the launch ended: every workgroup reads the same)
    if (tid < 64) {
        uint32_t bstar, cnt, mx;
        bin_layout(st->ghist, a.k, tid, s_base, bstar, cnt, mx);
        if (tid == 0) {
            s_bstar = bstar;
            s_cnt = cnt;
            s_ok = mx <= kBinRankCap ? 1u : 0u;
        }
    }
    for (int i = tid; i < kScanBins; i += 256) s_mine[i] = 0;
    __syncthreads();
    if (!s_ok) { // (every workgroup finds the same: nobody places anything; the emission reports the hand-back and tidies up)
        if (blockIdx.x == 0 && tid == 0) {
            st->redo_sum += 1u;
            st->redo_why |= kRedoBinTies;
            __hip_atomic_store(&st->redo, kRedoBinTies, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const uint32_t bstar = s_bstar;
    const uint32_t* hdr = static_cast<const uint32_t*>(fa.hdr);
    for (uint32_t r = blockIdx.x; r < nwg; r += gridDim.x) { // this workgroup's rows per bin
        const uint32_t n = hdr[r * (kFusedHeaderBytes / 4)] & 0x7FFFFFFFu;
        const u32x4* reg = static_cast<const u32x4*>(fa.pub) + static_cast<size_t>(r) * kFusedRegion;
        for (uint32_t i = tid; i < n; i += 256) {
            const uint32_t bin = coarse_bin(key_score(reg[i].y));
            if (bin >= bstar) atomicAdd(&s_mine[bin], 1u);
        }
    }
    __syncthreads();
    for (int b = tid; b < kScanBins; b += 256) { // its share of every bin it holds rows of
        const uint32_t c = s_mine[b];
        if (c) s_off[b] = s_base[b] + atomicAdd(&cursors[b], c);
        s_mine[b] = 0; // (from here on: the rows placed so far)
    }
    __syncthreads();
    for (uint32_t r = blockIdx.x; r < nwg; r += gridDim.x) {
        const uint32_t n = hdr[r * (kFusedHeaderBytes / 4)] & 0x7FFFFFFFu;
        const u32x4* reg = static_cast<const u32x4*>(fa.pub) + static_cast<size_t>(r) * kFusedRegion;
        for (uint32_t i = tid; i < n; i += 256) {
            const u32x4 e = reg[i];
            const uint32_t bin = coarse_bin(key_score(e.y));
            if (bin >= bstar) {
                const uint32_t pos = s_off[bin] + atomicAdd(&s_mine[bin], 1u);
                if (pos < cap) finalists[pos] = (static_cast<u64>(e.y) << 32) | e.x; // (cap >= the table's rows: never short)
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) st->nfinal = s_cnt; // rows in the bins >= B* (all published rows if they are fewer than k)
    if (blockIdx.x == 0) // the layout, for the emission: every bin's first position (kScanBins words behind the cursors)
        for (int b = tid; b < kScanBins; b += 256) cursors[kScanBins + b] = s_base[b];
}

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
