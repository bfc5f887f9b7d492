// gsim_fused_publish.inl -- phase 3 of the single launch, a piece of fused_kernel's body (included there): edges E2 (regions, headers, tags)
// and, for a publishing launch of a large k, E5 (the count-in) of gsim_fused_protocol.h.
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
