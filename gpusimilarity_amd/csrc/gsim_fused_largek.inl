// gsim_fused_largek.inl -- the kernels behind a publishing launch (k above GSIM_FUSED_SELECT_MAX_K): edge E5 of gsim_fused_protocol.h.
// Included by gsim_fused.hip inside namespace gsim { namespace {.

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
