// gsim_fused_thresholds.inl -- phase 2 of the single launch: the in-loop threshold protocol (edge E1 of gsim_fused_protocol.h).
// Included by gsim_fused.hip inside namespace gsim { namespace {.

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
