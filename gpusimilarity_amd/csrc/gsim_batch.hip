// gsim_batch.hip -- multi-query scan: Q queries against the table in ceil(Q / kBQ) passes.
//
// A batch of queries is not HBM-bound: with Q queries per table pass the work per
// fingerprint is Q * (2 * W) VALU operations (v_and + v_bcnt_u32_b32 per 32-bit word),
// i.e. the bound is the VALU issue rate, not memory.  This is the
// pass for 128-bit rows and for batches whose cutoff keeps many rows (or any cutoff on small
// tables); 256..2048-bit rows otherwise: gsim_batch_mfma.hip, and its SAMPLE variant sets the starting thresholds of
// both.  The layout is turned around relative to the
// single-query scan:
//   * every lane holds ONE whole fingerprint in registers (W VGPRs) -- no cross-lane
//     reduction at all;
//   * the query words are wave-uniform: they are read with scalar loads (s_load) and
//     used as SGPR operands of v_and_b32, so a (query, row) pair costs exactly
//     2 VALU instructions per word plus ~20 for the score and the filter;
//   * kBQ queries share one pass over the table; each has its own streaming top-k
//     filter (coarse histogram in LDS, pushed into a per-query table-wide histogram,
//     threshold polled back), the same exactness argument as the single-query scan.
// Candidates carry their query id; a compaction kernel routes the survivors to per
// query finalist lists and the rank-select kernel writes one result block per query.
#include "gsim_device.h"

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/gpusim_hip.h"
#include "gsim_device_common.h"
#include "gsim_prefilter.h"

namespace gsim
{
namespace
{

typedef int s16 __attribute__((ext_vector_type(16))); // a 16-dword SGPR tuple
constexpr int kBStage = 96;

// ---- per-workgroup filter state for the kBQ queries of a pass ------------------
struct BatchFilter {
    uint32_t hist[kBQ][kBBins / 2]; // two 16-bit counters per word: un-pushed emitted rows per bin
    uint32_t tau[kBQ];
    uint32_t nemit[kBQ];
    uint32_t trigger[kBQ];
    uint32_t kept[kBQ];
    // 96 staged candidates per wave (flush above 32): keeps the workgroup at 39.5 KB of LDS, i.e.
    // four workgroups = four waves per SIMD on a CU
    u64 stage_key[kScanBlock / 64][kBStage];
    uint32_t stage_cb[kScanBlock / 64][kBStage];
    uint32_t stage_q[kScanBlock / 64][kBStage];
};

// push the un-pushed counts of query slot qs into its table-wide histogram; optionally
// derive the threshold from it (one wavefront)
__device__ __forceinline__ void batch_push(BatchFilter* sh, BatchQueryState* gq, int qs, uint32_t k, int lane,
                                           bool rethreshold)
{
    constexpr int WPL = kBBins / 2 / 64; // packed words per lane (4)
#pragma unroll
    for (int i = 0; i < WPL; i++) {
        const int wd = lane * WPL + i;
        const uint32_t h = atomicExch(&sh->hist[qs][wd], 0u);
        if (h & 0xFFFFu) atomicAdd(&gq->ghist[2 * wd], h & 0xFFFFu);
        if (h >> 16) atomicAdd(&gq->ghist[2 * wd + 1], h >> 16);
    }
    if (!rethreshold) return;
    constexpr int PER = kBBins / 64; // 8 bins per lane
    uint32_t h[PER];
    uint32_t s = 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(gq->ghist, 0, kBBins * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < PER / 4; i++) {
        const u32x4 v4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * PER * 4 + i * 16, 0, /*sc1*/ 16);
        h[4 * i + 0] = v4.x;
        h[4 * i + 1] = v4.y;
        h[4 * i + 2] = v4.z;
        h[4 * i + 3] = v4.w;
        s += v4.x + v4.y + v4.z + v4.w;
    }
    uint32_t bin_k, cnt;
    threshold_from_counts<PER>(h, s, k, lane, bin_k, cnt);
    if (cnt >= k && lane == 0) {
        atomicMax(&gq->gtau, bin_k);
        atomicMax(&sh->tau[qs], bin_k);
    }
}

// SAMPLE = true: histogram every kept row of a strided sample, no emission (K0b).
// RPL = fingerprints held per lane: every scalar-loaded query word is used RPL times.
template <int WORDS, int RPL, bool SAMPLE>
__global__ __launch_bounds__(kScanBlock) void batch_scan_kernel(BatchArgs a, ScanGeometry g, uint32_t nsample,
                                                                u64 stride_chunks)
{
    __shared__ BatchFilter sh;
    __shared__ uint32_t s_last;
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * (kScanBlock / 64) + wib);
    const int nq = static_cast<int>(a.nq);
    for (int i = threadIdx.x; i < kBQ * (kBBins / 2); i += kScanBlock) (&sh.hist[0][0])[i] = 0;
    if (threadIdx.x < kBQ) {
        const int q = threadIdx.x;
        sh.tau[q] = (q < nq && a.k) ? a.rare->qstate[a.q0 + q].gtau : static_cast<uint32_t>(kBBins);
        sh.nemit[q] = 0;
        sh.trigger[q] = a.k < 64u ? (a.k ? a.k : 0xFFFFFFFFu) : 64u;
        sh.kept[q] = 0;
    }
    __syncthreads();

    const bool has_cutoff = a.cutoff > 0.0f;
    const uint32_t step = a.k / 8 > 32 ? a.k / 8 : 32;
    u64 seg_off = static_cast<u64>(w) * a.rare->seg_cap; // the pointers themselves are fetched on use (rare)
    uint32_t cursor = 0, staged = 0;
    u64* stg_key = sh.stage_key[wib];
    uint32_t* stg_cb = sh.stage_cb[wib];
    uint32_t* stg_q = sh.stage_q[wib];
    // queries are read-only for the whole kernel: the constant address space makes hipcc fetch
    // them with s_load_dwordx4/x8 into SGPRs (one v_and_b32 v, s, v per word, no VGPR copy)
    typedef const __attribute__((address_space(4))) uint32_t* const_u32p;
    const const_u32p qbase = (const_u32p) (a.queries + static_cast<size_t>(a.q0) * WORDS);
    const const_u32p qpops = (const_u32p) (a.qpop + a.q0);
    const u32x4* __restrict__ db = reinterpret_cast<const u32x4*>(a.rows);

    auto flush_stage = [&]() {
        const BatchRare* rr = a.rare;
        for (uint32_t i = lane; i < staged; i += 64) {
            if (cursor + i < rr->seg_cap) {
                rr->cand[seg_off + cursor + i] = stg_key[i];
                rr->cand_cb[seg_off + cursor + i] = stg_cb[i];
                rr->cand_q[seg_off + cursor + i] = stg_q[i];
            }
        }
        cursor += staged;
        staged = 0;
    };

    // query popcounts: lane q holds popc(query q) (read with v_readlane in the query loop)
    const uint32_t vqpop = lane < nq ? qpops[lane] : 0u;
    uint32_t vtau = 0, keptv = 0;
    float vtm = 0.0f;
    // (the filter arithmetic lives in gsim_prefilter.h: shared with the host-side proof test)
    const float cut_hi = valu_cutoff_hi(a.cutoff); // cutoff * (1 + 2^-21)
    const float cut_lo = valu_cutoff_lo(a.cutoff);
    constexpr int CHR = 64 * RPL; // rows per wave iteration
    const u64 nchunks = SAMPLE ? nsample : (a.nrows + CHR - 1) / CHR;
    uint32_t trip = 0;
    for (u64 ci = w; ci < nchunks; ci += g.nwaves) {
        const u64 c = SAMPLE ? ci * stride_chunks : ci;
        // RPL whole fingerprints per lane (lane-per-row loads: each lane streams its own row)
        u32x4 r4[RPL][WORDS / 4];
        uint32_t bb[RPL];
        bool active[RPL];
        u64 rowi[RPL];
#pragma unroll
        for (int r = 0; r < RPL; r++) {
            rowi[r] = c * CHR + r * 64 + lane;
            active[r] = rowi[r] < a.nrows;
            const u32x4* p = db + rowi[r] * (WORDS / 4);
#pragma unroll
            for (int j = 0; j < WORDS / 4; j++) r4[r][j] = active[r] ? p[j] : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int r = 0; r < RPL; r++) {
            bb[r] = 0;
#pragma unroll
            for (int j = 0; j < WORDS / 4; j++)
                bb[r] += __popc(r4[r][j].x) + __popc(r4[r][j].y) + __popc(r4[r][j].z) + __popc(r4[r][j].w);
        }

        if (!SAMPLE) {
            if ((trip & 15u) == 0 && lane < nq) {
                // poll the table-wide thresholds of this pass's queries (one lane per query)
                const uint32_t t = __hip_atomic_load(&a.rare->qstate[a.q0 + lane].gtau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t > sh.tau[lane]) atomicMax(&sh.tau[lane], t);
            }
            // Thresholds live in a VGPR for the query loop (lane q = query q), read with
            // v_readlane: no LDS access inside the loop -- LDS and scalar loads share lgkmcnt, a
            // ds_read per query would make every query wait for its own s_loads.
            vtau = lane < kBQ ? sh.tau[lane] : 0u;
            // pre-filter constant T- = tau/kBBins * (1 - 2^-21): cf < RN(T- * den) implies
            // RN(cf / den) < tau/kBBins, i.e. bin < tau (multi-query filter)
            vtm = valu_filter_level(vtau);
        }
        trip++;

        // ---- inner products of this wave's rows with the pass's queries ------------------
        // Query words are wave-uniform: s_load_dwordx16 into SGPRs, used directly as the scalar
        // operand of v_and_b32 (no VGPR copy, no LDS traffic).  hipcc leaves these loads
        // single-buffered (load, wait, 8 words of work, load, ...), so for 1024/2048-bit rows the
        // 16-word blocks are double-buffered by hand: the next block (or the next query's first
        // block) is in flight while the current one is reduced.  Scalar loads return out of
        // order, hence the full lgkmcnt(0) before a buffer is used; sched_barrier keeps the
        // compiler from regrouping the phases.
        // Measured limit of this form (scripts/valu_op_rate_probe.hip, scripts/qbcast_probe.hip,
        // DESIGN.md): v_bcnt_u32_b32 (VOP3) and a v_and with a scalar operand both issue in 4 cycles
        // per wave64, so the loop runs at the 8 cycles per word-pair ceiling of the instruction
        // pair; bringing the query words into VGPRs (LDS broadcast, v_mov) only trades the 4-cycle
        // v_and for a 2-cycle one plus the delivery cost.  Batches of 256..2048-bit rows therefore use the
        // matrix-core pass (gsim_batch_mfma.hip).
        constexpr bool MANUAL = (WORDS % 32 == 0);
        constexpr int NB = WORDS / 16;
        s16 qA, qB;
        if (MANUAL && nq > 0) {
            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(qA) : "s"(qbase));
        }
        for (int q = 0; q < nq; q++) {
            const const_u32p qw = qbase + q * WORDS;
            uint32_t acc[RPL][4];
#pragma unroll
            for (int r = 0; r < RPL; r++) acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0;
            if (MANUAL) {
#pragma unroll
                for (int blk = 0; blk < NB; blk++) {
                    s16& cur = (blk & 1) ? qB : qA; // NB is even: block 0 of every query is qA
                    s16& nxt = (blk & 1) ? qA : qB;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    if (blk + 1 < NB) {
                        asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(nxt) : "s"(qw), "n"((blk + 1) * 64));
                    } else if (q + 1 < nq) {
                        asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(nxt) : "s"(qw), "n"(WORDS * 4));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
#pragma unroll
                        for (int r = 0; r < RPL; r++) {
                            const u32x4 x = r4[r][blk * 4 + j];
                            acc[r][0] = bcnt_acc(x.x & static_cast<uint32_t>(cur[4 * j + 0]), acc[r][0]);
                            acc[r][1] = bcnt_acc(x.y & static_cast<uint32_t>(cur[4 * j + 1]), acc[r][1]);
                            acc[r][2] = bcnt_acc(x.z & static_cast<uint32_t>(cur[4 * j + 2]), acc[r][2]);
                            acc[r][3] = bcnt_acc(x.w & static_cast<uint32_t>(cur[4 * j + 3]), acc[r][3]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < WORDS / 4; j++) {
                    const uint32_t q0w = qw[4 * j + 0], q1w = qw[4 * j + 1], q2w = qw[4 * j + 2], q3w = qw[4 * j + 3];
#pragma unroll
                    for (int r = 0; r < RPL; r++) {
                        acc[r][0] = bcnt_acc(r4[r][j].x & q0w, acc[r][0]);
                        acc[r][1] = bcnt_acc(r4[r][j].y & q1w, acc[r][1]);
                        acc[r][2] = bcnt_acc(r4[r][j].z & q2w, acc[r][2]);
                        acc[r][3] = bcnt_acc(r4[r][j].w & q3w, acc[r][3]);
                    }
                }
            }
            const uint32_t qa = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(vqpop), q));
            const uint32_t tau = SAMPLE ? 0u : static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(vtau), q));
            const float tm = SAMPLE ? 0.0f : __uint_as_float(static_cast<uint32_t>(
                                                 __builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(vtm)), q)));
#pragma unroll
            for (int r = 0; r < RPL; r++) {
                const uint32_t cc = (acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3]);
                const float den = score_den(a.metric, a.alpha, a.beta, qa, bb[r], cc);
                const float cf = static_cast<float>(cc);
                // division-free pre-filter: almost every pair ends here
                const bool maybe = active[r] && !valu_filter_rejects(tm, cf, den);
                bool decided_kept = false;
                if (!SAMPLE && has_cutoff) {
                    // rows counted as "kept" need RN(cf/den) >= cutoff: decided without the divide
                    // unless cf/den is within 2^-21 of the cutoff
                    const bool surely = active[r] && valu_surely_kept(cut_hi, cf, den, cc);
                    const bool surely_not = !active[r] || valu_surely_not_kept(cut_lo, cf, den, cc);
                    const bool unsure = !(surely || surely_not);
                    const u64 mu = __ballot(unsure || maybe);
                    if (mu == 0) { // nobody needs the exact score
                        const u64 ms = __ballot(surely);
                        if (lane == q) keptv += static_cast<uint32_t>(__popcll(ms));
                        continue;
                    }
                    decided_kept = false;
                } else if (!SAMPLE) {
                    if (__ballot(maybe) == 0) continue;
                }
                (void) decided_kept;
                float s = __fdiv_rn(cf, den); // == score_of(...)
                s = apply_cutoff(s, a.cutoff);
                const bool keep = active[r] && (!has_cutoff || s != 0.0f);
                const uint32_t bin = batch_bin(s);
                if (SAMPLE) {
                    if (keep) atomicAdd(&sh.hist[q][bin >> 1], (bin & 1u) ? 65536u : 1u);
                    continue;
                }
                if (has_cutoff) {
                    const u64 mk = __ballot(keep);
                    if (lane == q) keptv += static_cast<uint32_t>(__popcll(mk));
                }
                const bool cand = keep && bin >= tau;
                const u64 m = __ballot(cand);
                if (m != 0) {
                    if (cand) {
                        const uint32_t slot = staged + lane_rank(m);
                        stg_key[slot] = make_key(s, static_cast<uint32_t>(rowi[r]));
                        stg_cb[slot] = (cc << 16) + bb[r];
                        stg_q[slot] = static_cast<uint32_t>(q);
                        atomicAdd(&sh.hist[q][bin >> 1], (bin & 1u) ? 65536u : 1u);
                    }
                    const uint32_t n = static_cast<uint32_t>(__popcll(m));
                    staged += n;
                    if (staged > 32) flush_stage();
                    uint32_t old = 0;
                    if (lane == 0) old = atomicAdd(&sh.nemit[q], n);
                    old = __builtin_amdgcn_readfirstlane(old);
                    const uint32_t trig = sh.trigger[q];
                    if (old < trig && old + n >= trig) {
                        batch_push(&sh, &a.rare->qstate[a.q0 + q], q, a.k, lane, true);
                        if (lane == 0) {
                            const uint32_t now = sh.nemit[q];
                            uint32_t inc = now / 2 > step ? now / 2 : step;
                            if (inc > 32768u) inc = 32768u; // the un-pushed counters are 16 bits wide
                            sh.trigger[q] = now + inc;
                        }
                        // this wave's register copy of the thresholds
                        vtau = lane < kBQ ? sh.tau[lane] : 0u;
                        vtm = __fmul_rn(static_cast<float>(vtau) * (1.0f / kBBins), 1.0f - 4.76837158203125e-7f);
                    }
                }
            }
        }
    }
    if (!SAMPLE) {
        if (staged) flush_stage();
        if (lane == 0) {
            const BatchRare* rr = a.rare;
            rr->seg_count[w] = cursor < rr->seg_cap ? cursor : rr->seg_cap;
            if (cursor > rr->seg_cap) { // segment overflow: the host grows the segments to what was needed (flags[15]) and runs the batch again
                atomicOr(rr->flags, 1u);
                atomicMax(&rr->flags[15], cursor);
            }
        }
    }
    if (!SAMPLE && has_cutoff && lane < nq && keptv) atomicAdd(&a.rare->qstate[a.q0 + lane].kept, static_cast<u64>(keptv));
    __syncthreads();
    for (int q = wib; q < nq; q += kScanBlock / 64) batch_push(&sh, &a.rare->qstate[a.q0 + q], q, a.k, lane, false);
    if (!SAMPLE) return;
    // sample pass: the last workgroup turns every histogram into a starting threshold
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(a.rare->ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    for (int q = wib; q < nq; q += kScanBlock / 64) {
        BatchQueryState* gq = &a.rare->qstate[a.q0 + q];
        constexpr int PER = kBBins / 64;
        uint32_t h[PER];
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            h[i] = __hip_atomic_load(&gq->ghist[lane * PER + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s += h[i];
        }
        uint32_t bin_k, cnt;
        threshold_from_counts<PER>(h, s, a.k, lane, bin_k, cnt);
        if (lane == 0) gq->gtau = (a.k && cnt >= a.k) ? bin_k : 0u;
#pragma unroll
        for (int i = 0; i < PER; i++) gq->ghist[lane * PER + i] = 0;
    }
    if (threadIdx.x == 0) *a.rare->ticket = 0;
}

// B* per query, then route the surviving candidates to the per-query finalist lists.
__global__ __launch_bounds__(64) void batch_bstar_kernel(BatchArgs a)
{
    const int lane = threadIdx.x;
    BatchQueryState* gq = &a.rare->qstate[a.q0 + blockIdx.x];
    constexpr int PER = kBBins / 64;
    uint32_t h[PER];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        h[i] = gq->ghist[lane * PER + i];
        s += h[i];
    }
    uint32_t bin_k = 0, cnt = 0;
    if (a.k) threshold_from_counts<PER>(h, s, a.k, lane, bin_k, cnt);
    if (lane == 0) gq->bstar = bin_k;
}

__global__ __launch_bounds__(kScanBlock) void batch_compact_kernel(BatchArgs a, ScanGeometry g)
{
    const int lane = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * (kScanBlock / 64) + (threadIdx.x >> 6));
    const BatchRare r = *a.rare;
    const uint32_t n = a.k ? r.seg_count[w] : 0;
    const u64* seg = r.cand + static_cast<u64>(w) * r.seg_cap;
    const uint32_t* seg_cb = r.cand_cb + static_cast<u64>(w) * r.seg_cap;
    const uint32_t* seg_q = r.cand_q + static_cast<u64>(w) * r.seg_cap;
    for (uint32_t i = lane; i < n; i += 64) {
        const u64 key = seg[i];
        const uint32_t q = a.q0 + seg_q[i];
        BatchQueryState* gq = &a.rare->qstate[q];
        if (batch_bin(key_score(static_cast<uint32_t>(key >> 32))) >= gq->bstar) {
            const uint32_t pos = atomicAdd(&gq->nfinal, 1u);
            if (pos < static_cast<uint32_t>(kSelectCap)) {
                r.fin_key[static_cast<size_t>(q) * kSelectCap + pos] = key;
                r.fin_cb[static_cast<size_t>(q) * kSelectCap + pos] = seg_cb[i];
            }
        }
    }
}

// One result block per query: rank select exactly as select_kernel's usual case.
// More than kSelectCap finalists (heavy ties): flags = 2, the host re-runs that query
// through the single-query path.
__global__ __launch_bounds__(256) void batch_select_kernel(BatchArgs a, uint32_t row_base, unsigned char* results,
                                                           size_t block_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    const int tid = threadIdx.x;
    const uint32_t q = a.q0 + blockIdx.y;
    BatchQueryState* gq = &a.rare->qstate[q];
    gsim_result_header* hdr = reinterpret_cast<gsim_result_header*>(results + static_cast<size_t>(q) * block_bytes);
    gsim_hit* hits = reinterpret_cast<gsim_hit*>(hdr + 1);
    const uint32_t m2 = a.k ? gq->nfinal : 0;
    const u64 approx = a.cutoff > 0.0f ? gq->kept : a.nrows;
    if (m2 > static_cast<uint32_t>(kSelectCap)) {
        if (blockIdx.x == 0 && tid == 0) {
            hdr->count = 0;
            hdr->flags = 2;
            hdr->approx = approx;
            atomicOr(a.rare->flags, 4u); // one word tells the host that some query needs the fallback
        }
        return;
    }
    if (blockIdx.x * 256 < m2) {
        const u64* fk = a.rare->fin_key + static_cast<size_t>(q) * kSelectCap;
        const uint32_t npad = (m2 + 1u) & ~1u;
        for (uint32_t i = tid; i < npad; i += 256) keys[i] = i < m2 ? fk[i] : 0ull;
        __syncthreads();
        // the workgroups of a query stride over its finalists (a few hundred to a few thousand: four
        // workgroups per query instead of kSelectCap / 256 keep the launch small)
        for (uint32_t i = blockIdx.x * 256 + tid; i < m2; i += gridDim.x * 256) {
            const u64 mine = keys[i];
            const uint32_t cb = a.rare->fin_cb[static_cast<size_t>(q) * kSelectCap + i];
            uint32_t rank = 0;
            const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(keys);
#pragma unroll 4
            for (uint32_t j = 0; j < npad / 2; j++) {
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
        hdr->approx = approx;
    }
}

template <int WORDS, int RPL>
hipError_t launch_batch_t(const BatchArgs& a, const ScanGeometry& g, uint32_t sample_chunks, hipStream_t s,
                          bool sample_only)
{
    const uint32_t nblocks = g.nwaves / (kScanBlock / 64);
    const u64 nchunks = (a.nrows + 64 * RPL - 1) / (64 * RPL);
    u64 want = static_cast<u64>(g.nwaves) * sample_chunks;
    if (sample_chunks > 96u / RPL) { // 16-bit un-pushed counters: <= 96 x 256 rows per workgroup
        sample_chunks = 96u / RPL;
        want = static_cast<u64>(g.nwaves) * sample_chunks;
    }
    if (sample_chunks && a.k && nchunks >= want * 16) {
        hipLaunchKernelGGL((batch_scan_kernel<WORDS, RPL, true>), dim3(nblocks), dim3(kScanBlock), 0, s, a, g,
                           static_cast<uint32_t>(want), (nchunks - 1) / want);
    }
    if (sample_only) return hipGetLastError();
    hipLaunchKernelGGL((batch_scan_kernel<WORDS, RPL, false>), dim3(nblocks), dim3(kScanBlock), 0, s, a, g, 0u, u64(0));
    return hipGetLastError();
}

hipError_t launch_batch_scan(const BatchArgs& a, const ScanGeometry& g, uint32_t sample_chunks, hipStream_t s,
                             bool sample_only)
{
    const int rpl_env = static_cast<int>((a.opts >> 8) & 255u);
    const int rpl = rpl_env ? rpl_env : (a.W >= 64 ? 1 : (a.W >= 32 ? 2 : 4)); // rows per lane (scripts/bench_batch.py)
    switch (a.W) {
    case 4: return launch_batch_t<4, 4>(a, g, sample_chunks, s, sample_only);
    case 8: return launch_batch_t<8, 4>(a, g, sample_chunks, s, sample_only);
    case 16:
        return rpl == 1 ? launch_batch_t<16, 1>(a, g, sample_chunks, s, sample_only)
                        : rpl == 2 ? launch_batch_t<16, 2>(a, g, sample_chunks, s, sample_only)
                                   : launch_batch_t<16, 4>(a, g, sample_chunks, s, sample_only);
    case 32:
        return rpl == 1 ? launch_batch_t<32, 1>(a, g, sample_chunks, s, sample_only)
                        : rpl == 2 ? launch_batch_t<32, 2>(a, g, sample_chunks, s, sample_only)
                                   : launch_batch_t<32, 4>(a, g, sample_chunks, s, sample_only);
    case 64:
        return rpl == 2 ? launch_batch_t<64, 2>(a, g, sample_chunks, s, sample_only)
                        : launch_batch_t<64, 1>(a, g, sample_chunks, s, sample_only);
    default: return hipErrorInvalidValue;
    }
}

// B* -> compact -> select for the a.nq queries from a.q0 whose candidates sit in nwaves segments.
hipError_t launch_batch_finish(const BatchArgs& a, uint32_t nwaves, uint32_t row_base, void* results,
                               size_t block_bytes, hipStream_t s)
{
    hipLaunchKernelGGL(batch_bstar_kernel, dim3(a.nq), dim3(64), 0, s, a);
    const uint32_t nblocks = nwaves / (kScanBlock / 64);
    ScanGeometry g{};
    g.nwaves = nwaves;
    hipLaunchKernelGGL(batch_compact_kernel, dim3(nblocks), dim3(kScanBlock), 0, s, a, g);
    const size_t lds = static_cast<size_t>(kSelectCap) * sizeof(u64);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(batch_select_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(batch_select_kernel, dim3(4, a.nq), dim3(256), lds, s, a, row_base,
                       reinterpret_cast<unsigned char*>(results), block_bytes);
    return hipGetLastError();
}

} // namespace

bool batch_supported(uint32_t W)
{
    return W == 4 || W == 8 || W == 16 || W == 32 || W == 64;
}

// One pass: sample -> scan -> B* -> compact -> select for the a.nq (<= kBQ) queries starting at a.q0.
hipError_t launch_batch_pass(const BatchArgs& a, const BatchRare& /*rare_host*/, const ScanGeometry& g,
                             uint32_t sample_chunks, uint32_t row_base, void* results, size_t block_bytes, hipStream_t s)
{
    const hipError_t e = launch_batch_scan(a, g, sample_chunks, s, false);
    if (e != hipSuccess) return e;
    return launch_batch_finish(a, g.nwaves, row_base, results, block_bytes, s);
}

// The same pipeline with the scan on the matrix cores (gsim_batch_mfma.hip): sample passes per
// kBQ queries (they set the starting thresholds), ONE contraction pass for all a.nq
// (<= kMfmaQueries) queries, then B* / compact / select for all of them.
hipError_t launch_batch_mfma_pass(const BatchArgs& a, const ScanGeometry& g, int num_cus, uint32_t sample_chunks,
                                  uint32_t row_base, void* results, size_t block_bytes, hipStream_t s, hipEvent_t ev0,
                                  hipEvent_t ev1)
{
    hipError_t es = hipSuccess;
    const bool sampled = launch_batch_mfma_sample(a, num_cus, s, &es); // large tables: one launch for all queries
    if (es != hipSuccess) return es;
    for (uint32_t q0 = 0; !sampled && q0 < a.nq; q0 += kBQ) {
        BatchArgs as = a;
        as.q0 = a.q0 + q0;
        as.nq = a.nq - q0 < static_cast<uint32_t>(kBQ) ? a.nq - q0 : static_cast<uint32_t>(kBQ);
        const hipError_t e = launch_batch_scan(as, g, sample_chunks, s, true);
        if (e != hipSuccess) return e;
    }
    if (ev0) (void) hipEventRecord(ev0, s); // timing: the contraction kernel alone
    const hipError_t e = launch_batch_mfma_scan(a, num_cus, s);
    if (e != hipSuccess) return e;
    if (ev1) (void) hipEventRecord(ev1, s);
    return launch_batch_finish(a, batch_mfma_waves(num_cus), row_base, results, block_bytes, s);
}

} // namespace gsim
