// gsim_fused_protocol.h -- the single launch (gsim_fused.hip): what it is, its constants, its LDS layout, its checkpoint schedule, and --
// "THE EDGES", at the end of the opening comment -- the contract of every place where one workgroup depends on another, each written once.
// Included by gsim_fused.hip only, inside namespace gsim { namespace {.  The kernel itself is split by phase:
//   gsim_fused.hip              prologue + phase 1 (the streaming loop, gsim_scan_inl.h) + the launchers
//   gsim_fused_thresholds.inl   phase 2: in-loop thresholds (FusedFilter, the election, the forwarder and poller waves)
//   gsim_fused_publish.inl      phase 3: a workgroup publishes its survivors, its report and its header      (body of fused_kernel)
//   gsim_fused_select.inl       phase 4: every workgroup selects: headers, final threshold, finalists, ranks (body of fused_kernel)
//   gsim_fused_close.inl        phase 5: tickets, the result header, the per-query state's reset              (body of fused_kernel)
//   gsim_fused_largek.inl       behind a publishing launch: fused_handoff_kernel, fused_binsort_kernel
// The .inl files of phases 3-5 are consecutive pieces of ONE function body (fused_kernel): they share its locals.  The split moved
// text, not tokens: the generated code is identical to the one-file form's (checked when it was made, round 6).
#pragma once

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
//
// THE EDGES -- every place where a workgroup depends on what another one did, each contract written once.  "W" = the side that
// writes, "R" = the side that reads; sc1 = write-through / read-through at device scope (cache policy bit of the raw buffer
// instructions), sc0 sc1 = system scope.  The hardware behaviours marked [L1]..[L4] are the litmus tests of
// tests/test_gpu_litmus.py (csrc/gsim_litmus.hip).
//
//   E1  in-loop thresholds                                   gsim_fused_thresholds.inl
//       W  a streaming wave leaves its M-th best score key in LDS (FusedShared::wsum); the workgroup's forwarder wave copies the four
//          keys to FusedArgs::summ[wave] (sc1 stores) and takes the checkpoint's two-level ticket (FusedArgs::tickets, device atomics).
//       R  the LAST arriver's poller wave reads all summ[] (sc1 loads) and raises QueryState::gtau (atomicMax); every poller polls
//          {gtau, elected} with one 64-bit load and hands the key to its streaming waves through LDS (FusedShared::tau).
//       Contract: summ[i] is 0 or a key some wave really held M rows at or above; a stale or early read only yields a SMALLER key;
//          gtau only rises.  Hence any value read at any time is a valid lower bound of the k-th best key.  Nobody waits on this edge
//          except tables of a few trips (FusedSchedule::late), which wait for `need()` elections with a bound (wait_ticks) and hand
//          the query back when it runs out (kRedoElectionWait).
//
//   E2  publish -> select: regions, headers, tags            W: gsim_fused_publish.inl   R: gsim_fused_select.inl
//       W  workgroup b writes its survivors into ITS region (FusedArgs::pub + b * kFusedRegion entries of 16 bytes:
//          {key lo, key hi, popcounts, TAG}) with 16-byte sc1 stores, then -- behind a workgroup barrier, WITHOUT waiting for the
//          entries' acknowledgements -- ONE 16-byte sc1 store of its header (FusedArgs::hdr[b]: {entries | order, bucket shift |
//          tag << 5 | failed << 31, report lo, report hi}).  The tag is new for every launch on the handle and never 0.
//       R  a selector polls hdr[b] (sc1 loads) until it carries the launch's tag: THAT is b's arrival (no counter).  It then reads
//          entries (sc1 loads, also straight into LDS); an entry whose fourth word is not the tag was overtaken by its header and is
//          read again until it is (bounded by wait_ticks: kRedoArrivalWait).
//       Contract: [L1] a 16-byte aligned sc1 store is seen whole by a 16-byte sc1 load -- an entry with the tag is complete, a header
//          with the tag is complete; [L3][L4] a header may become visible before an entry stored earlier -- never assumed otherwise;
//          a region is written by exactly one workgroup per launch and read only after its header; lists are in an order a reader can
//          stop in (canonical up to kFusedSortCap rows, bucket order above).  Regions are zero when allocated (tag 0 = never valid).
//
//   E3  the final threshold                                   gsim_fused_select.inl
//       Every selector computes it FOR ITSELF from the 256 reports in the headers (no exchange): a report with at least r - 1 larger
//       ones, r = ceil(k / Mw).  Contract: the computation is a pure function of the headers' contents, so all selectors agree; the
//       row -> selector assignment is a hash of the row index, so every finalist is ranked by exactly one selector.
//
//   E4  closing: hits, tickets, the result header, the reset  gsim_fused_close.inl
//       W  a selector writes the hits it ranked (rank = output slot) with sc0 sc1 stores, waits for THEIR acknowledgements
//          (s_waitcnt vmcnt(0)), then adds {1 | failed << 16 | checksum of its hits << 32} to its group's ticket and, as the group's
//          last, to QueryState::sel_done (64-bit device atomics, two levels).
//       R  the holder of the last ticket knows every hit is acknowledged and the sum of all hits' words; it writes the result
//          header with ONE 16-byte sc0 sc1 store -- for synchronous callers {count, flags | epoch << 8, approx, checksum + epoch *
//          kBlockCheckMul} -- and then re-zeroes the per-query state and the exchange buffer for the next launch (stream order).
//       Contract: [L2] the 16-byte header store reaches pinned host memory whole: a host that sees the epoch in word 1 reads count
//          and approx of the same store; acknowledged system-scope stores are in host memory on the platforms soaked -- the host
//          does not rely on it: it verifies the checksum against the hits it reads and re-runs a block that stays wrong
//          (capi_query.cpp finish_query_sync; gsim_timing.blocks_rechecked / blocks_torn).
//
//   E5  a publishing launch (large k) -> the kernels behind it    W: gsim_fused_publish.inl   R: gsim_fused_largek.inl, gsim_select.hip
//       W  as E2, plus the published rows counted per coarse bin into QueryState::ghist (device atomics) and a two-level count-in
//          (FusedArgs::arrive); nobody waits: the LAST workgroup to count in tidies up the exchange state.
//       R  fused_handoff_kernel / fused_binsort_kernel run behind the launch on the SAME stream: kernel-boundary ordering makes
//          every region, header and ghist complete and visible -- no tags are needed on this edge (they are still checked).
//
//   E6  forward progress
//       One workgroup per CU (FusedShared fills the CU's LDS), at most kFusedSelectors of them, all co-resident on an idle GPU
//       (two half-grid launches of two lanes: 128 + 128).  The only waits on other workgroups are E1's (small tables) and E2's header
//       polls; both are bounded by FusedArgs::wait_ticks of wall clock (2 ms + four scan times) and end in a hand-back
//       (QueryState::redo, header flag 2: the four-kernel pipeline, which never waits, answers) -- on a GPU shared with another
//       queue part of a grid may not be running while the waiters hold their CUs (tests/test_gpu_cotenancy.py).
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
