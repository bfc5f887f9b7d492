// gsim_fused_select.inl -- phase 4 of the single launch, a piece of fused_kernel's body (included there): the reader side of edge E2 and
// the final threshold every selector derives for itself (E3) -- gsim_fused_protocol.h.
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
