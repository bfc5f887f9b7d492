// gsim_select.hip -- behind the scan: K3 (exact select of the finalists, result emission), the large-k route (device
// radix select + a two-launch sort), the folded tables' re-score, the merge of per-shard result blocks
// (fingerprintdb_cuda.cu:284-339, 363-380), the synthetic-table generator and the score table of the parity tests.
#include "gsim_device.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include "../../include/gpusim_hip.h"
#include "gsim_device_common.h"
#include "gsim_scan_inl.h"
#include "gsim_synth.h"

namespace gsim
{
namespace
{

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
    // ticket: the last workgroup resets the state (all others are done reading it: what a workgroup read of the state it has
    // used -- the values have arrived -- before its ticket; no fence, which on this part would write the XCD's L2 back)
    __syncthreads();
    if (tid == 0) ctl[3] = (__hip_atomic_fetch_add(&a.state->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1u : 0u;
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

__device__ __forceinline__ void reset_query_state(QueryState* st, LargeKState* lk, int tid, int nthreads)
{
    if (tid == 0 && lk) {
        lk->prefix = 0;
        lk->remaining = 0;
        lk->ticket = 0;
        lk->count = 0;
        lk->all = 0;
    }
    if (tid == 0) {
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
    for (int i = tid; i < kScanBins; i += nthreads) st->ghist[i] = 0;
}

// ---------------------------------------------------------------------------
// large-k path (k > kSelectCap): radix select of the k-th key, gather, sort (launch_sort_desc)
// (multi-launch), then emission of the first k.  Exact for any input.
// ---------------------------------------------------------------------------

// The k-th largest finalist key by an MSD radix descent, one launch per digit of 8 bits (digits of 11 bits, six launches, were
// tried: 8.6 us per pass instead of 5.5 -- a pass is its chain of global round trips, adds -> ticket -> the last workgroup's
// reads, and 2048 bins lengthen it more than two passes fewer save), the finalist count read ON THE DEVICE: nothing of the
// large-k path is sized by the host from a value it would have to wait for.  Pass p histograms bits [lo, hi) = [56 - 8 p, 64 - 8 p)
// of the keys that match the prefix found so far (LDS histogram per workgroup, one global atomic per non-empty bin); the last workgroup
// (ticket) picks the digit that holds the wanted rank, extends the prefix and clears the histogram.  Fewer finalists than k:
// `all` is set and every finalist is taken.
constexpr int kLargeKDigit = 8, kLargeKBins = 1 << kLargeKDigit, kLargeKPasses = 8;
static_assert(kLargeKPasses * kLargeKDigit >= 64 && sizeof(LargeKState::hist) == kLargeKBins * 4, "the passes cover the key");

__global__ __launch_bounds__(256) void largek_pass_kernel(ScanArgs a, const u64* finalists, uint32_t cap, LargeKState* lk, int pass)
{
    __shared__ uint32_t s_h[kLargeKBins];
    __shared__ uint32_t s_last;
    const int tid = threadIdx.x;
    uint32_t nfinal = a.state->nfinal;
    if (nfinal > cap) nfinal = cap;
    if (pass > 0 && lk->all) return;
    const int hi = 64 - kLargeKDigit * pass, lo = hi > kLargeKDigit ? hi - kLargeKDigit : 0;
    const u64 dmask = (1ull << (hi - lo)) - 1ull;
    const u64 prefix = lk->prefix;
    const uint32_t want = pass == 0 ? a.k : lk->remaining;
    for (int i = tid; i < kLargeKBins; i += 256) s_h[i] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256 + tid; i < nfinal; i += gridDim.x * 256) {
        const u64 key = finalists[i];
        if (pass == 0 || (key >> hi) == prefix) atomicAdd(&s_h[(key >> lo) & dmask], 1u);
    }
    __syncthreads();
    // (no fences, as sample_publish: the adds return, so they have been performed before the ticket behind the barrier)
    uint32_t sink = 0;
    for (int i = tid; i < kLargeKBins; i += 256)
        if (s_h[i]) sink += __hip_atomic_fetch_add(&lk->hist[i], s_h[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(sink));
    __syncthreads();
    if (tid == 0) s_last = (__hip_atomic_fetch_add(&lk->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    if (tid < 64) {
        constexpr int PER = kLargeKBins / 64;
        uint32_t h[PER];
        uint32_t sm = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            h[i] = __hip_atomic_load(&lk->hist[tid * PER + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sm += h[i];
        }
        uint32_t bin, cnt;
        threshold_from_counts<PER>(h, sm, want, tid, bin, cnt);
        if (tid == 0) {
            if (cnt < want) { // (pass 0 only: fewer finalists than k)
                lk->all = 1;
                lk->prefix = 0;
            } else {
                const uint32_t pop = __hip_atomic_load(&lk->hist[bin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lk->prefix = (prefix << (hi - lo)) | bin;
                lk->remaining = want - (cnt - pop);
            }
            lk->ticket = 0;
        }
    }
    __syncthreads();
    for (int i = tid; i < kLargeKBins; i += 256) lk->hist[i] = 0;
}

// the keys at or above the k-th largest (exactly min(k, #finalists) of them: keys are unique) -> out[0 ..)
__global__ __launch_bounds__(256) void largek_gather_kernel(ScanArgs a, const u64* finalists, uint32_t cap, LargeKState* lk, u64* out,
                                                            uint32_t out_cap, uint32_t* hint)
{
    const int lane = threadIdx.x & 63;
    uint32_t nfinal = a.state->nfinal;
    if (nfinal > cap) nfinal = cap;
    if (hint && blockIdx.x == 0 && threadIdx.x == 0) *hint = nfinal; // (pinned: the host picks the next large-k query's route by it)
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

// The same selection and gather in ONE launch by one workgroup: nine launches are nine times ~5 us of dependent global round
// trips.  The finalists are the candidates of the bins >= B*, the table's k-th best coarse bin (compact_kernel); the histogram
// that gave B* is still there: every finalist above B* is in the top k, only the r = k - (rows above B*) best keys OF bin B*
// have to be found.  One pass over the finalists collects the keys of bin B* into LDS (up to 16 Ki of them), eight radix passes
// over those (LDS only) find the r-th largest, a second pass over the finalists gathers.  More than 16 Ki keys in bin B* (heavy
// ties): the radix passes read the finalists from global memory, bin B* only -- exact, slow.  Two reads of the finalists by
// one workgroup beat the grid's nine launches up to a few ten thousand finalists (10.7 k: 56 us against 87; 52 k with 20 k of
// them in bin B*: 121 us either way): the host picks the route by the count of the previous large-k query (`hint`, pinned
// memory); either route is exact for any count.
constexpr uint32_t kLargeKLdsKeys = 16384;
constexpr int kLargeKOneThreads = 1024;

__global__ __launch_bounds__(kLargeKOneThreads) void largek_one_block_kernel(ScanArgs a, const u64* finalists, uint32_t cap, LargeKState* lk, u64* out,
                                                                          uint32_t out_cap, uint32_t* hint)
{
    extern __shared__ __attribute__((aligned(16))) u64 skeys[];
    __shared__ uint32_t s_hist[256];
    __shared__ u64 s_prefix;
    __shared__ uint32_t s_want, s_all, s_cursor, s_nb, s_bstar;
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t nfinal = a.state->nfinal;
    if (nfinal > cap) nfinal = cap;
    if (tid < 64) { // B*, and how many of bin B*'s keys the top k takes
        uint32_t bstar, cnt;
        find_threshold(a.state->ghist, a.k, lane, bstar, cnt);
        if (tid == 0) {
            s_all = cnt < a.k ? 1u : 0u; // fewer finalists than k: every finalist is taken
            s_bstar = bstar;
            s_want = cnt < a.k ? 0u : a.k - (cnt - a.state->ghist[bstar]);
            s_prefix = 0;
            s_cursor = 0;
            s_nb = 0;
            if (hint) *hint = nfinal;
        }
    }
    __syncthreads();
    const uint32_t bstar = s_bstar;
    const bool all = s_all != 0;
    auto in_bstar = [&](u64 key) { return coarse_bin(key_score(static_cast<uint32_t>(key >> 32))) == bstar; };
    const uint32_t n64 = (nfinal + 63u) & ~63u;
    if (!all) {
        for (uint32_t i = tid; i < n64; i += kLargeKOneThreads) { // bin B*'s keys -> LDS
            const u64 key = i < nfinal ? finalists[i] : 0ull;
            const bool take = i < nfinal && in_bstar(key);
            const u64 m = __ballot(take);
            if (m == 0) continue;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_nb, static_cast<uint32_t>(__popcll(m)));
            base = __builtin_amdgcn_readfirstlane(base);
            const uint32_t pos = base + lane_rank(m);
            if (take && pos < kLargeKLdsKeys) skeys[pos] = key;
        }
        __syncthreads();
        const uint32_t nb = s_nb;
        const bool in_lds = nb <= kLargeKLdsKeys;
        const uint32_t nscan = in_lds ? nb : nfinal;
        for (int pass = 0; pass < 8; pass++) {
            if (tid < 256) s_hist[tid] = 0;
            __syncthreads();
            const int shift = 56 - 8 * pass;
            const u64 prefix = s_prefix;
            const uint32_t want = s_want;
            for (uint32_t i = tid; i < nscan; i += kLargeKOneThreads) {
                const u64 key = in_lds ? skeys[i] : finalists[i];
                if ((in_lds || in_bstar(key)) && (pass == 0 || (key >> (shift + 8)) == prefix)) atomicAdd(&s_hist[(key >> shift) & 0xFFu], 1u);
            }
            __syncthreads();
            if (tid < 64) {
                uint32_t h[4];
                uint32_t s4 = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    h[i] = s_hist[tid * 4 + i];
                    s4 += h[i];
                }
                uint32_t bin, cnt;
                threshold_from_counts<4>(h, s4, want, tid, bin, cnt); // (want <= the keys that match: bin B* holds at least r keys)
                if (tid == 0) {
                    s_prefix = (prefix << 8) | bin;
                    s_want = want - (cnt - s_hist[bin]);
                }
            }
            __syncthreads();
        }
    }
    const u64 kth = all ? 0ull : s_prefix;
    for (uint32_t i = tid; i < n64; i += kLargeKOneThreads) {
        const u64 key = i < nfinal ? finalists[i] : 0ull;
        const bool take = i < nfinal && key >= kth;
        const u64 m = __ballot(take);
        if (m == 0) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&s_cursor, static_cast<uint32_t>(__popcll(m)));
        base = __builtin_amdgcn_readfirstlane(base);
        const uint32_t pos = base + lane_rank(m);
        if (take && pos < out_cap) out[pos] = key;
    }
    __syncthreads();
    if (tid == 0) {
        lk->count = s_cursor;
        lk->all = s_all;
        lk->prefix = kth;
    }
}

// ---------------------------------------------------------------------------
// folded tables: the candidates' re-score with the full fingerprints, on the device
// ---------------------------------------------------------------------------
// fingerprintdb_cuda.cu:307-331: the R = k F (int)log2(2F) best FOLDED scores of a storage are re-scored with the full
// fingerprints (tanimoto_similarity_cpu, :387-399), stably sorted by the new score (top_results_bubble_sort: strict '>',
// so ties keep the order of the folded list) and the first min(k, R) kept up to the first one below the cutoff.  The
// reference does this on the host (slide 19 lists it as future GPU work); here the full rows are resident as well
// (288 GB hold both) and three small launches do it: re-score into keys (score key << 32 | ~position), a sort (launch_sort_desc)
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

// Sorting 2 ... 64 Ki (and more) unique 64-bit keys, descending -- the large-k path's top-k keys and the folded tables' re-scored
// candidates.  A global bitonic sort with one launch per step was 136 launches for 64 Ki keys (0.6 ms of launches for 0.02 ms of
// work: a launch is ~5 us however little it does); with the steps that fit a tile run in LDS it was still 6 launches and
// 143 us for 32 Ki keys (91 barrier-separated steps in the first).  Now two launches: every tile of kSortTile keys sorted in LDS
// (one compare-exchange per thread and step), then every key's final position by counting -- its position in its own tile + for
// every other tile the keys before it, found by a branch-free binary search (eight tiles' searches in flight per thread).
// Equal keys (only the zero padding) are ordered by tile, so that the positions are a permutation.
constexpr uint32_t kSortTile = 2048;
constexpr int kSortThreads = 1024;

__global__ __launch_bounds__(kSortThreads) void tile_sort_kernel(u64* keys, uint32_t n, const uint32_t* count)
{
    __shared__ u64 t[kSortTile];
    const uint32_t tile = n < kSortTile ? n : kSortTile;
    const uint32_t base = blockIdx.x * tile;
    const uint32_t tid = threadIdx.x;
    const uint32_t valid = count ? (*count < n ? *count : n) : n; // (slots past the gathered keys: padding, whatever the buffer holds)
    for (uint32_t i = tid; i < tile; i += kSortThreads) t[i] = base + i < valid ? keys[base + i] : 0ull;
    // (thread t's pair at strides up to 64 lies among the 128 elements its wave owns: such steps follow each other behind a
    // wave barrier -- a wave's LDS operations execute in order --, only the steps at longer strides, and the first one
    // after them, need the workgroup's: 14 of the 66 steps of a full tile)
    uint32_t prev = 128;
    for (uint32_t sz = 2; sz <= tile; sz <<= 1) {
        for (uint32_t stride = sz >> 1; stride > 0; stride >>= 1) {
            if (stride >= 128 || prev >= 128) __syncthreads();
            else {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            prev = stride;
            if (tid < tile / 2) {
                const uint32_t lo = 2 * tid - (tid & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool desc = (lo & sz) == 0; // (the last stage: sz = tile, every pair descending)
                const u64 x = t[lo], y = t[hi];
                if ((x < y) == desc) {
                    t[lo] = y;
                    t[hi] = x;
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < tile; i += kSortThreads) keys[base + i] = t[i];
}

struct LargeKEmit { // rank_merge_kernel<true>: the hit of every key goes to its position, the header, the state's reset
    ScanArgs a;
    LargeKState* lk;
    uint32_t row_base, flags;
    u64 approx_if_no_cutoff;
    void* d_result;
};

__device__ __forceinline__ void reset_query_state(QueryState* st, LargeKState* lk, int tid, int nthreads);

template <bool EMIT>
__global__ __launch_bounds__(256) void rank_merge_kernel(const u64* __restrict__ keys, u64* __restrict__ out, uint32_t n, LargeKEmit em)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x; // (the grid covers n exactly: n is a multiple of kSortTile)
    const u64 key = keys[i];
    const uint32_t ti = i / kSortTile, ntiles = n / kSortTile;
    uint32_t rank = i % kSortTile;
    constexpr int NC = 8;
    for (uint32_t t0 = 0; t0 < ntiles; t0 += NC) {
        uint32_t cnt[NC];
        const u64* tp[NC];
        bool ge[NC]; // the keys of tile tj that come before `key`: x > key, and in earlier tiles also x == key
#pragma unroll
        for (int u = 0; u < NC; u++) {
            const uint32_t tj = t0 + u < ntiles ? t0 + u : ti;
            tp[u] = keys + static_cast<size_t>(tj) * kSortTile;
            ge[u] = tj < ti;
            cnt[u] = 0;
        }
        for (uint32_t step = kSortTile / 2; step > 0; step >>= 1) {
#pragma unroll
            for (int u = 0; u < NC; u++)
            {
                const u64 x = tp[u][cnt[u] + step - 1u];
                if (x > key || (ge[u] && x == key)) cnt[u] += step;
            }
        }
#pragma unroll
        for (int u = 0; u < NC; u++) {
            const u64 x = tp[u][cnt[u]]; // (cnt <= kSortTile - 1 here)
            if (x > key || (ge[u] && x == key)) cnt[u] += 1u;
            const uint32_t tj = t0 + u;
            rank += (tj < ntiles && tj != ti) ? cnt[u] : 0u;
        }
    }
    if (!EMIT) {
        out[rank] = key;
        return;
    }
    __shared__ uint32_t s_last;
    gsim_result_header* hdr = reinterpret_cast<gsim_result_header*>(em.d_result);
    const uint32_t nkeys = em.lk->count < em.a.k ? em.lk->count : em.a.k;
    if (rank < nkeys) emit_hit(em.a, key, em.row_base, reinterpret_cast<gsim_hit*>(hdr + 1) + rank); // (rank < count: a gathered key, not padding)
    if (i == 0) {
        // (flags bit 31, synchronous callers of the single launch's large-k route: no gated classic kernels ran behind a launch
        // that handed the query back -- QueryState::redo is still set -- so the block says "handed back" (flag 2) and the host
        // runs the query again; read before this workgroup's ticket, cleared by the last one)
        const bool back = (em.flags & 0x80000000u) != 0 && em.a.state->redo != 0;
        hdr->count = back ? 0u : nkeys;
        hdr->flags = (em.flags & 0x7FFFFFFFu) | (back ? 2u : 0u);
        hdr->approx = em.a.cutoff > 0.0f ? em.a.state->kept : em.approx_if_no_cutoff;
    }
    // the last workgroup re-zeroes the per-query state (every workgroup has used what it read of it before its ticket)
    __syncthreads();
    if (threadIdx.x == 0) s_last = (__hip_atomic_fetch_add(&em.lk->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (s_last) reset_query_state(em.a.state, em.lk, static_cast<int>(threadIdx.x), 256);
}

// Large k behind the publishing single launch, for callers that read the block (launch_fused_binsort placed the finalists by
// coarse bin, highest bin first): the list is in order ACROSS bins -- the coarse bin is monotone in the score key -- so a key's
// final position is its bin's first position plus the number of larger keys IN ITS BIN: a few hundred compares against keys
// the whole wave reads together (neighbours in the list share a bin), instead of the radix select over all finalists, the
// gather and the two-launch sort (k = 8192 at 1 M rows: 18 + 16 + 9 us and a 6 us gap -> one launch).  A workgroup takes
// kBinRankKeys list positions; the grid covers k + kBinRankCap of them (fewer than k rows lie above B*, at most kBinRankCap in it).
constexpr uint32_t kBinRankChunk = 2048; // keys staged in LDS at a time
constexpr uint32_t kBinRankKeys = 64;    // list positions per workgroup: its waves share the compares of each
constexpr int kBinRankThreads = 512;

__global__ __launch_bounds__(kBinRankThreads) void binrank_emit_kernel(ScanArgs a, const u64* __restrict__ finalists, uint32_t cap, uint32_t* cursors,
                                                           LargeKState* lk, uint32_t row_base, uint32_t flags, u64 approx_if_no_cutoff, void* d_result)
{
    __shared__ __attribute__((aligned(16))) u64 s_keys[kBinRankChunk];
    __shared__ uint32_t s_rank[kBinRankKeys];
    __shared__ uint32_t s_last, s_rlo, s_rhi;
    QueryState* st = a.state;
    const int tid = threadIdx.x, ki = tid & 63, part = tid >> 6;
    gsim_result_header* hdr = reinterpret_cast<gsim_result_header*>(d_result);
    const bool back = st->redo != 0; // (the launch, or the placement, handed the query back: read before this workgroup's ticket)
    uint32_t nfinal = st->nfinal;
    if (nfinal > cap) nfinal = cap;
    const uint32_t p0 = blockIdx.x * kBinRankKeys;
    if (!back && p0 < nfinal) {
        if (tid < 64) s_rank[tid] = 0;
        // Lane ki of every wave holds list position p0 + ki; wave `part` compares it with every eighth group of eight keys of
        // its bin (alone on its SIMD a wave issues one instruction per ~5 cycles: a bin of 2600 keys, k = 20 000 at 100 M rows, is
        // 13 k instructions per key -- shared by eight waves, on four times as many workgroups, it is a few microseconds).
        const uint32_t p = p0 + ki;
        const bool valid = p < nfinal;
        const u64 key = valid ? finalists[p] : ~0ull;
        const uint32_t bin = valid ? coarse_bin(key_score(static_cast<uint32_t>(key >> 32))) : 0u;
        const uint32_t lo = valid ? cursors[kScanBins + bin] : 0u, hi = valid ? lo + st->ghist[bin] : 0u; // the list positions of this key's bin (the layout launch_fused_binsort left)
        // the workgroup's keys are neighbours in the list: their bins together are ONE stretch of it, from the first key's bin to
        // the last valid one's -- staged in LDS a chunk at a time (coalesced) and compared from there: the lanes of a wave read
        // the same words (straight from memory every thread's loop waited ~0.3 us per four keys: 55 us at k = 8192)
        if (tid == 0) s_rlo = lo;
        if (part == 0 && valid && (p + 1 == nfinal || ki == 63)) s_rhi = hi;
        __syncthreads();
        const uint32_t rlo = s_rlo & ~7u, rhi = s_rhi; // (chunks start at a multiple of eight list positions: 16-byte LDS reads)
        uint32_t rank = 0;
        for (uint32_t c = rlo; c < rhi; c += kBinRankChunk) {
            const uint32_t cn = rhi - c < kBinRankChunk ? rhi - c : kBinRankChunk;
            for (uint32_t i = tid; i < kBinRankChunk; i += kBinRankThreads) s_keys[i] = i < cn ? finalists[c + i] : 0ull; // (past the stretch: below every key)
            __syncthreads();
            const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(s_keys);
            for (uint32_t g = static_cast<uint32_t>(part) * 8u; g < cn; g += 8u * (kBinRankThreads / 64)) { // (wave-uniform)
                const ulonglong2 q0 = k2[g / 2], q1 = k2[g / 2 + 1], q2 = k2[g / 2 + 2], q3 = k2[g / 2 + 3];
                const u64 kk[8] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
                // (no test of the position: the stretch is in order across bins, so the keys of higher bins in front of this key's
                // bin all count and those of lower bins behind it never do -- the count over the whole stretch is the key's place in it)
#pragma unroll
                for (int e = 0; e < 8; e++) rank += kk[e] > key ? 1u : 0u;
            }
            __syncthreads();
        }
        if (rank) atomicAdd(&s_rank[ki], rank);
        __syncthreads();
        const uint32_t pos = rlo + s_rank[ki];
        if (part == 0 && valid && pos < a.k) emit_hit(a, key, row_base, reinterpret_cast<gsim_hit*>(hdr + 1) + pos);
    }
    if (blockIdx.x == 0 && tid == 0) {
        hdr->count = back ? 0u : (nfinal < a.k ? nfinal : a.k);
        hdr->flags = flags | (back ? 2u : 0u);
        hdr->approx = a.cutoff > 0.0f ? st->kept : approx_if_no_cutoff;
    }
    // the last workgroup re-zeroes the per-query state and the bins' cursors (every workgroup is done with both before its ticket)
    __syncthreads();
    if (tid == 0) s_last = (__hip_atomic_fetch_add(&lk->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (s_last) {
        reset_query_state(st, lk, tid, kBinRankThreads);
        for (int i = tid; i < kScanBins; i += kBinRankThreads) cursors[i] = 0;
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

} // namespace

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
    unsigned long long* sorted = nullptr;
    hipError_t e = launch_sort_desc(keys, keys + npad, npad, s, &sorted); // (the caller's buffer holds 2 npad keys)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fold_emit_kernel, dim3((k + 255) / 256 ? (k + 255) / 256 : 1), dim3(256), 0, s, folded_block, sorted, cbs, k, cutoff, row_base,
                       out_block);
    return hipGetLastError();
}

// k > kSelectCap: the k-th largest finalist key by a radix descent, then the keys at or above it into out[0 .. count)
// (out_cap >= k entries; launch_largek_sort_emit sorts them and emits the hits).  Nothing here is sized by the finalist count:
// one workgroup in one launch (`one_block`; the usual case) or eight passes of the whole grid + a gather (many finalists:
// heavy ties at the k-th score) -- both exact for any count; `hint` (pinned, may be null) receives the count.
hipError_t launch_largek_select(const ScanArgs& a, const unsigned long long* finalists, uint32_t finalists_cap, LargeKState* lk,
                                unsigned long long* out, uint32_t out_cap, uint32_t* hint, bool one_block, hipStream_t s)
{
    if (one_block) {
        const size_t lds = static_cast<size_t>(kLargeKLdsKeys) * sizeof(u64);
        static DynLdsOnce once;
        const hipError_t e = once.ensure(reinterpret_cast<const void*>(largek_one_block_kernel), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(largek_one_block_kernel, dim3(1), dim3(kLargeKOneThreads), lds, s, a, finalists, finalists_cap, lk, out, out_cap, hint);
        return hipGetLastError();
    }
    for (int pass = 0; pass < kLargeKPasses; pass++)
        hipLaunchKernelGGL(largek_pass_kernel, dim3(256), dim3(256), 0, s, a, finalists, finalists_cap, lk, pass);
    hipLaunchKernelGGL(largek_gather_kernel, dim3(256), dim3(256), 0, s, a, finalists, finalists_cap, lk, out, out_cap, hint);
    return hipGetLastError();
}

hipError_t launch_sort_desc(unsigned long long* keys, unsigned long long* tmp, uint32_t n_pow2, hipStream_t s, unsigned long long** sorted)
{
    *sorted = keys;
    if (n_pow2 < 2) return hipSuccess;
    const uint32_t tile = n_pow2 < kSortTile ? n_pow2 : kSortTile;
    hipLaunchKernelGGL(tile_sort_kernel, dim3(n_pow2 / tile), dim3(kSortThreads), 0, s, keys, n_pow2, static_cast<const uint32_t*>(nullptr));
    if (n_pow2 > kSortTile) {
        hipLaunchKernelGGL(rank_merge_kernel<false>, dim3(n_pow2 / 256), dim3(256), 0, s, keys, tmp, n_pow2, LargeKEmit{});
        *sorted = tmp;
    }
    return hipGetLastError();
}

hipError_t launch_largek_sort_emit(const ScanArgs& a, unsigned long long* keys, uint32_t n_pow2, LargeKState* lk, uint32_t row_base,
                                   uint64_t approx_if_no_cutoff, uint32_t flags, void* d_result, hipStream_t s)
{
    if (n_pow2 <= kSortTile) return hipErrorInvalidValue; // (k > kSelectCap = 8192: at least eight tiles)
    hipLaunchKernelGGL(tile_sort_kernel, dim3(n_pow2 / kSortTile), dim3(kSortThreads), 0, s, keys, n_pow2, &lk->count);
    LargeKEmit em{a, lk, row_base, flags, approx_if_no_cutoff, d_result};
    hipLaunchKernelGGL(rank_merge_kernel<true>, dim3(n_pow2 / 256), dim3(256), 0, s, keys, static_cast<u64*>(nullptr), n_pow2, em);
    return hipGetLastError();
}

hipError_t launch_binrank_emit(const ScanArgs& a, const unsigned long long* finalists, uint32_t cap, uint32_t* cursors, LargeKState* lk,
                               uint32_t row_base, uint64_t approx_if_no_cutoff, uint32_t flags, void* d_result, hipStream_t s)
{
    const uint32_t nb = (a.k + kBinRankCap + kBinRankKeys - 1u) / kBinRankKeys;
    hipLaunchKernelGGL(binrank_emit_kernel, dim3(nb), dim3(kBinRankThreads), 0, s, a, finalists, cap, cursors, lk, row_base, flags, approx_if_no_cutoff, d_result);
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
