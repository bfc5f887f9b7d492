// gsim_batch_mfma.hip -- the multi-query pass as a binary contraction on the matrix cores.
//
// With Q queries per table pass the intersection counts form a (Q x K) . (K x N) product of 0/1
// matrices (K = fingerprint bits).  On the vector ALU that costs two instructions per 32-bit
// word and (query, row) pair (gsim_batch.hip: 8 cycles per word-pair per SIMD, which IS the
// VALU issue ceiling for v_and + v_bcnt); gfx950's block-scaled MFMA
// (v_mfma_scale_f32_32x32x64_f8f6f4, both operands FP4/E2M1) does 32 x 32 pairs x 64 bits per
// instruction -- one eighth of the VALU time per bit -- and accumulates exactly in f32 (the
// counts are < 2^24).  Only the batch path uses it; the single-query scan stays a
// streaming HBM-bound kernel.
//
// Packed bits -> FP4 operands with ONE v_and per operand dword (scripts/mfma_fp4_probe.hip):
//   x & 0x11111111 -> nibbles {0, 0.5}    block scale 2^1
//   x & 0x22222222 -> nibbles {0, 1.0}    block scale 2^0
//   x & 0x44444444 -> nibbles {0, 2.0}    block scale 2^-1
//   (x >> 3) & 0x11111111                 (0x8 is the FP4 sign bit: -0, so that class is shifted)
// i.e. each 256-bit group of a row (8 words: 4 per lane half) feeds four MFMAs, one per class.
// Which bit lands in which k slot is irrelevant as long as queries and rows use the same map.
//
// Work split: one 512-thread workgroup per CU; wave w owns query tile w (32 queries, the A
// operand), expanded ONCE into registers (2 W VGPRs); all eight waves stream the same table
// rows, staged through LDS in 64 KB blocks by global_load_lds (double-buffered, XOR-swizzled
// through the global address so the per-lane 16-byte fragment reads are conflict-free).  Per
// (32 queries x 32 rows) tile: W/2 MFMAs.  The epilogue is a division-free linear pre-filter:
// the candidate condition score >= tau/kBBins is c >= ka[q] + kb[q] * popc(row) (conservative
// by 2^-12), tested as max_q(c u[q] + v[q]) >= popc(row) with two vector instructions per pair;
// only tiles with a passing pair stage the raw pairs in LDS, and those are scored exactly
// (the reference's f32 divide) 64 at a time.  The streaming top-k filter is the table-wide one
// of the other scans (per-query histogram + threshold, monotone updates), kept in global
// memory here because emissions are rare after the sample pass.
//
// popc(row) comes from a 2 B/row side array (row_popcount_kernel) staged with the row block.
//
// Measured: 125 M x 2048-bit x 256 queries in 22 ms on one MI355X = 0.60 of
// the dense FP4 peak, MFMA pipe ~60 % busy; what limits it is the VALU work in the MFMA shadow
// (7.6 instructions per MFMA) and the stalls of the epilogue, not HBM (1.4 TB/s) and not LDS
// (no bank conflicts); the kernel has no registers left (256, 2048-bit rows) to restructure either.
#include "gsim_device.h"

#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/gpusim_hip.h"
#include "gsim_device_common.h"
#include "gsim_prefilter.h"

namespace gsim
{
namespace
{

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kMWaves = 8;             // waves per workgroup = query tiles per pass
constexpr int kMBlock = kMWaves * 64;  // threads
constexpr int kMChunks = 4096;         // 16-byte chunks per LDS row block (64 KB, two buffers)
constexpr int kMaxMT = 2;              // query tiles per wave (2 for 1024-bit rows: halves the operand work)
constexpr int kMStage = 96;            // raw candidates staged per wave (processed in bulk above 32)

// GSIM_MF_TIMING: per-phase cycle counters of wave 0 of every workgroup, summed into flags[2..]
// (units of 64 cycles; printed by the host under GSIM_DEBUG_BATCH)
#ifndef GSIM_MF_TIMING
#define GSIM_MF_TIMING 0
#endif
#if GSIM_MF_TIMING
#define MF_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define MF_ACC(slot, t0, t1) tacc[slot] += (t1) - (t0)
#else
#define MF_T(var)
#define MF_ACC(slot, t0, t1)
#endif

constexpr int kScale1 = 0x80808080;  // E8M0 2^1
constexpr int kScale0 = 0x7F7F7F7F;  // 2^0
constexpr int kScaleM = 0x7E7E7E7E;  // 2^-1

// LDS of the sample kernel (its own geometry: one 64 KB row block, 64 KB of histograms)
struct MfmaShared {
    u32x4 rows[2][kMChunks];
    uint32_t qpop[kMWaves][32 * kMaxMT];
};

// LDS of the contraction kernel.  BC = 16-byte chunks per row block; `rows` is a ring of RING blocks: the DMA fills slot
// (n + RING - 1) % RING while the waves read slot n % RING.
template <int WORDS, int BC, int RING> struct MfmaRing {
    static constexpr int CPR = WORDS / 4, RB = BC / CPR, RP = RING;
    static constexpr int HB = (WORDS <= 16 && BC == kMChunks) ? 1 : 2; // histogram staging buffers (512-bit rows in 64 KB blocks: the LDS is full)
    u32x4 rows[RING][BC];
    uint16_t rpop[RP][RB];            // popc(row) of a block's rows, from the table's side array
    uint32_t tau_poll[kMWaves][64];   // the queries' table-wide thresholds, polled by DMA (no register, no wait)
    uint32_t hist[HB][kBBins];         // a query's table-wide histogram, fetched by DMA for the threshold refresh
    // pre-filter constants in accumulator order: [query tile of the wave][lane half][acc register]
    float kap_a[kMWaves][kMaxMT][2][16];
    float kap_b[kMWaves][kMaxMT][2][16];
    float kap_u[kMWaves][kMaxMT][2][16]; // the same bound solved for popc(row): c * u + v >= popc(row)
    float kap_v[kMWaves][kMaxMT][2][16];
    uint32_t tau[kMWaves][32 * kMaxMT];
    uint32_t qpop[kMWaves][32 * kMaxMT];
    uint32_t kept[kMWaves][32 * kMaxMT]; // rows at or above the cutoff seen by the wave, per query
    uint32_t stage_row[kMWaves][kMStage]; // pairs that passed the pre-filter: row, (common << 16) + popc(row),
    uint32_t stage_cb[kMWaves][kMStage];  // query of the tile -- scored exactly in bulk (drain_stage)
    uint32_t stage_q[kMWaves][kMStage];
};

// The class masks are passed in VGPRs: v_and_b32 with two vector operands issues in 2 cycles,
// with a literal or scalar operand in 4 (scripts/valu_op_rate_probe.hip).
struct ClassMasks {
    uint32_t m1, m2, m4;
};

template <int CLS> __device__ __forceinline__ uint32_t fp4_word(uint32_t x, const ClassMasks& k)
{
    return CLS == 0 ? (x & k.m1) : CLS == 1 ? (x & k.m2) : CLS == 2 ? (x & k.m4) : ((x >> 3) & k.m1);
}

// element by element: a vector AND with a splat mask makes hipcc keep four copies of every mask
template <int CLS> __device__ __forceinline__ v4i fp4_class(u32x4 x, const ClassMasks& k)
{
    return v4i{static_cast<int>(fp4_word<CLS>(x.x, k)), static_cast<int>(fp4_word<CLS>(x.y, k)),
               static_cast<int>(fp4_word<CLS>(x.z, k)), static_cast<int>(fp4_word<CLS>(x.w, k))};
}

template <int CLS> __device__ __forceinline__ v16f mfma_class(v4i qa, v4i rb, v16f acc)
{
    const v8i A = {qa.x, qa.y, qa.z, qa.w, 0, 0, 0, 0};
    const v8i B = {rb.x, rb.y, rb.z, rb.w, 0, 0, 0, 0};
    constexpr int sc = CLS == 1 ? kScale0 : (CLS == 2 ? kScaleM : kScale1);
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, /*A fp4*/ 4, /*B fp4*/ 4, 0, sc, 0, sc);
}

// element r (wave-uniform) of an accumulator, for the rare path: through a COPY -- a dynamic index into the accumulators
// themselves made hipcc keep them in scratch memory throughout the kernel
__device__ __forceinline__ float pick16(const v16f& v, int r)
{
    v16f t = v;
    asm volatile("" : "+v"(t));
    return t[r];
}

// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8)
template <int N> __device__ __forceinline__ void wait_vm_outstanding()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

// (the pre-filter arithmetic lives in gsim_prefilter.h: shared with the host-side proof test)

// MT = query tiles per wave: the expanded row operand of a tile is used by MT MFMAs per class, so
// the operand work per MFMA is 5 / MT instructions; 2 W MT registers hold the queries.
// NT = row tiles in flight per wave (accumulators: 16 MT NT registers; two independent MFMA chains
// per wave are worth having).  BC / RING: the row block and the ring (MfmaRing).
// DN = the dense-cutoff variant (MT = 1): the rows at or above a cutoff that keeps a sizeable part of the table are COUNTED
// from the accumulators (gsim_prefilter.h cutoff_band: two fused multiply-adds and two compares per pair, sixteen counters
// per lane) instead of going through the exact path one by one; only pairs inside the band, and the top-k candidates at the
// queries' current thresholds, are staged.  Its constants share the LDS arrays of the plain variant's two query tiles:
// kap_u/kap_v[.][0] = tile test at the top-k threshold, kap_u/kap_v[.][1] = "surely kept", kap_a[.][0]/[1] = u, v of "surely not".
template <int WORDS, int MT, int NT, int BC, int RING, bool DN>
__global__ __launch_bounds__(kMBlock) void batch_mfma_kernel(BatchArgs a, u64 nblocks)
{
    static_assert(!DN || MT == 1, "the dense-cutoff variant holds one query tile per wave");
    static_assert(RING >= 2 && RING <= 4, "ring geometry");
    using Ring = MfmaRing<WORDS, BC, RING>;
    constexpr int WV = kMWaves;
    constexpr int kWBlock = kMBlock;
    constexpr int QW = 32 * MT;         // queries per wave
    constexpr int KG = WORDS / 8;       // 256-bit groups per row
    constexpr int CPR = WORDS / 4;      // 16-byte chunks per row
    constexpr int RPLN = 16 / CPR;      // rows per 256-byte LDS line
    constexpr int RB = BC / CPR;        // rows per LDS block
    constexpr int NTB = RB / 32;        // 32-row tiles per block
    constexpr int NA = NT;              // accumulator sets per query tile
    constexpr int LPW = BC / kWBlock;   // row loads per wave and block
    constexpr int PCH = RB / 8;         // 16-byte chunks of row popcounts per block
    constexpr int NPW = (PCH + 63) / 64; // waves that load them
    constexpr int AHEAD = RING - 1;     // blocks requested ahead of the one being consumed
    static_assert(WORDS % 8 == 0 && CPR <= 16 && NTB >= 1 && NTB % NT == 0 && BC % kWBlock == 0 && RB % 8 == 0, "unsupported row width");
    static_assert(2 * WORDS * MT <= 128, "query operands must fit in registers");
    static_assert((AHEAD - 1) * (LPW + 2) < 64, "vmcnt");
    static_assert(sizeof(Ring) <= 160 * 1024, "LDS");
    __shared__ Ring sh;

    const int lane = threadIdx.x & 63;
    const int wq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const uint32_t w = blockIdx.x * kMWaves + wq; // candidate segment of this wave
    const int nq = static_cast<int>(a.nq);
    // Fewer than eight query tiles: several waves share a tile and split the row tiles of a block
    // between them (tile = wave % p2, row group = wave / p2, p2 = query tiles rounded up to 2^n).
    const int ntiles = (nq + QW - 1) / QW;
    const int p2 = ntiles <= 1 ? 1 : (ntiles <= 2 ? 2 : (ntiles <= 4 ? 4 : 8));
    const int tile = wq % p2, rgroup = wq / p2, ngroups = WV / p2;
    const int q0t = tile * QW; // first query of this wave's tile(s)
    const bool wave_has_queries = q0t < nq && rgroup * NT < NTB;
    // The rare arguments once, into scalar registers; device pointers carry the global address
    // space so that stores and atomics are global_* instructions (a flat_* access may alias LDS and
    // would have to wait for the row block in flight to LDS).
    typedef __attribute__((address_space(1))) u64* g_u64p;
    typedef __attribute__((address_space(1))) uint32_t* g_u32p;
    typedef const __attribute__((address_space(1))) void* g_cvoidp;
    typedef __attribute__((address_space(3))) void* l_voidp;
    const BatchRare rr = *a.rare;
    BatchQueryState* qstate = rr.qstate + a.q0;
    const u32x4* __restrict__ db = reinterpret_cast<const u32x4*>(a.rows);
    const bool has_cutoff = a.cutoff > 0.0f;
    // A cutoff that keeps a sizeable fraction of the table (estimated by the sample pass, bit 3 of
    // the flags) would send that fraction of all pairs through the exact path: the plain variant leaves such a
    // batch alone, the dense variant -- launched right behind it -- every other one.
    const bool dense_batch = has_cutoff && (*((g_u32p) rr.flags) & 8u) != 0;
    if (DN ? !dense_batch : dense_batch) {
        if (lane == 0 && !DN) rr.seg_count[w] = 0;
        return;
    }

    // ---- A operand: this wave's 32 queries, expanded once ----------------------------------
    ClassMasks km{0x11111111u, 0x22222222u, 0x44444444u};
    asm volatile("" : "+v"(km.m1), "+v"(km.m2), "+v"(km.m4)); // keep them in VGPRs
    v4i aexp[MT][KG][4];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        const int ql = q0t + m * 32 + i;
        const u32x4* qp = reinterpret_cast<const u32x4*>(a.queries + static_cast<size_t>(a.q0 + ql) * WORDS);
#pragma unroll
        for (int g = 0; g < KG; g++) {
            const u32x4 x = ql < nq ? qp[2 * g + h] : u32x4{0, 0, 0, 0};
            aexp[m][g][0] = fp4_class<0>(x, km);
            aexp[m][g][1] = fp4_class<1>(x, km);
            aexp[m][g][2] = fp4_class<2>(x, km);
            aexp[m][g][3] = fp4_class<3>(x, km);
        }
    }
    // per-query constants of this wave (wave-private LDS: no workgroup barrier needed)
    auto set_query_constants = [&](uint32_t tau) { // lanes 0..QW-1: query `lane` of the wave
        const int m = lane >> 5;
        const bool valid = q0t + lane < nq;
        const float level = prefilter_level(has_cutoff, a.cutoff, tau);
        const PrefilterConstants pk = prefilter_constants(a.metric == GSIM_METRIC_TVERSKY, a.alpha, a.beta, sh.qpop[wq][lane],
                                                          level, valid);
        const int hh = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3); // accumulator slot of query i of a tile
        if (DN) {
            // the candidates' level: the query's top-k threshold, whatever the cutoff (the cutoff itself is applied to
            // every staged pair by the exact path); the band of the cutoff for the count
            const PrefilterConstants pt = prefilter_constants(a.metric == GSIM_METRIC_TVERSKY, a.alpha, a.beta, sh.qpop[wq][lane],
                                                              prefilter_level(false, 0.0f, tau), valid);
            const CutoffBand cb = cutoff_band(a.metric == GSIM_METRIC_TVERSKY, a.alpha, a.beta, sh.qpop[wq][lane], a.cutoff, valid);
            sh.kap_u[wq][0][hh][r] = pt.u;
            sh.kap_v[wq][0][hh][r] = pt.v;
            sh.kap_b[wq][0][hh][r] = pt.kb; // (pair test of the candidates: c >= fma(kb, b, ka))
            sh.kap_b[wq][1][hh][r] = pt.ka;
            sh.kap_u[wq][1][hh][r] = cb.us;
            sh.kap_v[wq][1][hh][r] = cb.vs;
            sh.kap_a[wq][0][hh][r] = cb.un;
            sh.kap_a[wq][1][hh][r] = cb.vn;
            sh.tau[wq][lane] = tau;
            return;
        }
        sh.kap_a[wq][m][hh][r] = pk.ka;
        sh.kap_b[wq][m][hh][r] = pk.kb;
        sh.kap_u[wq][m][hh][r] = pk.u;
        sh.kap_v[wq][m][hh][r] = pk.v;
        sh.tau[wq][lane] = tau;
    };
    sh.tau_poll[wq][lane] = 0;
    if (lane < QW) {
        const bool valid = q0t + lane < nq;
        sh.kept[wq][lane] = 0;
        sh.qpop[wq][lane] = valid ? a.qpop[a.q0 + q0t + lane] : 0u;
        const uint32_t t0 = valid ? *((g_u32p) &qstate[q0t + lane].gtau) : static_cast<uint32_t>(kBBins);
        sh.tau_poll[wq][lane] = valid ? t0 : 0u;
        set_query_constants(t0);
    }

    // ---- candidate staging (as in batch_scan_kernel) ---------------------------------------
    const u64 seg_off = static_cast<u64>(w) * rr.seg_cap;
    const g_u64p seg_key = (g_u64p) (rr.cand + seg_off);
    const g_u32p seg_cb = (g_u32p) (rr.cand_cb + seg_off);
    const g_u32p seg_q = (g_u32p) (rr.cand_q + seg_off);
    uint32_t cursor = 0, staged = 0;
    uint32_t* stg_row = sh.stage_row[wq];
    uint32_t* stg_cb = sh.stage_cb[wq];
    uint32_t* stg_q = sh.stage_q[wq];
    // Exact scores of the staged pairs, 64 at a time: the survivors of the current thresholds go to
    // this wave's candidate segment and into the table-wide histograms.  Done in bulk because every
    // global access here is followed, at the end of the row block, by a wait of the whole workgroup.
    auto drain_stage = [&]() {
        for (uint32_t s0 = 0; s0 < staged; s0 += 64) {
            const uint32_t e = s0 + lane;
            const bool have = e < staged;
            const uint32_t row = have ? stg_row[e] : 0u;
            const uint32_t cb = have ? stg_cb[e] : 0u;
            const uint32_t qraw = have ? stg_q[e] : 0u; // query of the tile | candidate << 30 | to be counted << 31 (dense variant)
            const uint32_t qi = qraw & 0x3FFFFFFFu;
            float sc = score_of(a.metric, a.alpha, a.beta, sh.qpop[wq][qi], cb & 0xFFFFu, cb >> 16);
            sc = apply_cutoff(sc, a.cutoff);
            const uint32_t bin = batch_bin(sc);
            const bool keep = have && (!has_cutoff || sc != 0.0f); // fingerprintdb_cuda.cu:265-271
            if (has_cutoff && keep && (!DN || (qraw >> 31) != 0)) atomicAdd(&sh.kept[wq][qi], 1u); // LDS; one global add per wave and query at the end
            const bool cand = keep && bin >= sh.tau[wq][qi] && (!DN || ((qraw >> 30) & 1u) != 0);
            const u64 mc = __ballot(cand);
            if (cand) {
                const uint32_t pos = cursor + lane_rank(mc);
                if (pos < rr.seg_cap) {
                    seg_key[pos] = make_key(sc, row);
                    seg_cb[pos] = cb;
                    seg_q[pos] = static_cast<uint32_t>(q0t) + qi;
                }
                // no return value: fire and forget
                __hip_atomic_fetch_add((g_u32p) &qstate[q0t + qi].ghist[bin], 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            cursor += static_cast<uint32_t>(__popcll(mc));
        }
        staged = 0;
    };

    // ---- row block staging: global -> LDS, swizzled ----------------------------------------
    // LDS chunk p of a block (p = (j * 8 + wave) * 64 + lane for the wave's j-th load) holds chunk
    // c = (p % CPR') ^ swizzle of row p / CPR'; both the row offset inside a load group and c are the
    // same for every j, so a lane keeps two values and does 32-bit row arithmetic per load (rows per
    // device < 2^31).  Blocks past the table's end are requested like any other (their rows clamp to the last one and
    // nothing of them is used): every wave issues the same number of loads per block, which is what the counted waits need.
    constexpr int RPJ = kWBlock / CPR; // rows covered by one load of the whole workgroup
    const int line0 = wq * 4 + (lane >> 4);
    const uint32_t rowl0 = static_cast<uint32_t>(line0 * RPLN + (lane & 15) / CPR);
    const uint32_t chunk0 = static_cast<uint32_t>(((lane & 15) % CPR) ^ (line0 % CPR));
    const u32x4* dbc = db + chunk0;
    const uint32_t last_row = static_cast<uint32_t>(a.nrows - 1);
    const u32x4* rowpop16 = reinterpret_cast<const u32x4*>(a.rowpop);
    const u64 last_pop_chunk = (a.nrows - 1) / 8;
    const uint32_t first_past = static_cast<uint32_t>(nblocks) * RB; // (blocks past the end: rows clamp anyway)
    auto issue_block = [&](u64 blk, int slot, int pslot) {
        uint32_t first = static_cast<uint32_t>(blk < nblocks ? blk * RB : first_past) + rowl0;
#pragma unroll
        for (int j = 0; j < LPW; j++) {
            uint32_t grow = first + j * RPJ;
            grow = grow < last_row ? grow : last_row;
            __builtin_amdgcn_global_load_lds((g_cvoidp) (dbc + static_cast<u64>(grow) * CPR), (l_voidp) (&sh.rows[slot][(j * WV + wq) * 64]), 16,
                                             0, 0);
        }
        // the rows' popcounts: RB 16-bit counts = PCH 16-byte chunks, one per thread of the first waves
        if (wq < NPW) {
            if (wq * 64 + lane < PCH) {
                u64 pc = (blk < nblocks ? blk : nblocks) * PCH + static_cast<u64>(wq * 64 + lane);
                pc = pc < last_pop_chunk ? pc : last_pop_chunk;
                __builtin_amdgcn_global_load_lds((g_cvoidp) (rowpop16 + pc), (l_voidp) (&sh.rpop[pslot][wq * 64 * 8]), 16, 0, 0);
            }
        }
    };
    // The queries' table-wide thresholds, polled by DMA into LDS: a load into a register would have to be waited for
    // with everything requested before it, and would drain the ring.  What lands is read a few blocks later; thresholds
    // only rise, an old one is a valid one.
    const bool polls = wave_has_queries;
    auto issue_poll = [&]() {
        if (polls && lane < QW && q0t + lane < nq)
            __builtin_amdgcn_global_load_lds((g_cvoidp) (&qstate[q0t + lane].gtau), (l_voidp) (&sh.tau_poll[wq][0]), 4, 0, /*sc1*/ 16);
    };
    // all but the youngest AHEAD - 1 groups of requests (a group = one iteration's poll + block) have landed
    auto wait_for_next_block = [&]() {
        if constexpr (AHEAD == 1) {
            wait_vm_outstanding<0>();
        } else {
            const int extra = (polls ? 1 : 0) + (wq < NPW ? 1 : 0);
            if (extra == 0) wait_vm_outstanding<(AHEAD - 1) * LPW>();
            else if (extra == 1) wait_vm_outstanding<(AHEAD - 1) * (LPW + 1)>();
            else wait_vm_outstanding<(AHEAD - 1) * (LPW + 2)>();
        }
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    uint32_t kcnt[DN ? 16 : 1]; // dense variant: rows surely at or above the cutoff, per accumulator slot (= query) of this lane
#pragma unroll
    for (int r = 0; r < (DN ? 16 : 1); r++) kcnt[r] = 0;
    u64 blk = blockIdx.x;
    // ---- prologue: the first AHEAD blocks are requested, the first one has landed ----
#pragma unroll
    for (int d = 0; d < AHEAD; d++) issue_block(blk + static_cast<u64>(d) * gridDim.x, d % RING, d % Ring::RP);
    if constexpr (AHEAD == 1) {
        wait_vm_outstanding<0>();
    } else {
        if (wq < NPW) wait_vm_outstanding<(AHEAD - 1) * (LPW + 1)>();
        else wait_vm_outstanding<(AHEAD - 1) * LPW>();
    }
    lds_barrier();

#if GSIM_MF_TIMING
    unsigned long long tacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    constexpr int PER = kBBins / 64; // histogram bins per lane in a threshold update
    // Threshold refresh: every 2^RS blocks ONE wave of the workgroup (in rotation; the workgroups staggered) requests the
    // table-wide histogram of one of its queries by DMA and, LAT blocks later (when the waits at the block ends have seen it
    // land), derives the query's threshold from it.  HB staging buffers: event e + HB may only be requested after event e has
    // been read, HB 2^RS > LAT.  As often as that allows -- every block with two buffers and one block in flight: the rate is
    // worth more than the thresholds it produces (125 M x 2048-bit rows, 256 queries, thresholds already final: every other
    // block 25.8 ms, every fourth 27.2, every eighth 30.3; 512-bit rows, every fourth block: 20.9 ms against 15.3 --
    // profiles/r05_batch_mfma_experiments.txt).
    uint32_t turn = blockIdx.x;
    constexpr int LAT = AHEAD;                                   // blocks from a histogram's request to its use
    constexpr int RS = (LAT + Ring::HB) / Ring::HB <= 2 ? 1 : 2; // (every block -- possible with two buffers -- was slower than every other one: 27.5 ms)
    int slot = 0, pslot = 0;                   // ring slot / popcount slot of the block being consumed
    int pend_q = -1, pend_buf = 0;             // refresh under way: the query whose histogram was requested, its staging buffer
    uint32_t pend_at = 0;
    uint32_t it = 0;

    // ---- the work on one group of row tiles (NT tiles from t2 on), in pieces the main loop arranges ----
    v16f acc[MT][NA];
    // contraction over the rows of NT tiles
    auto contract = [&](const u32x4* cbase, int t2) __attribute__((always_inline)) {
        MF_T(tk0);
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int c = 0; c < NA; c++) acc[m][c] = v16f{};
        {
            const u32x4* lrow[NT];
            int xr[NT];
#pragma unroll
            for (int tt = 0; tt < NT; tt++) {
                const int row = (t2 + tt) * 32 + i;
                const int line = row / RPLN;
                lrow[tt] = cbase + line * 16 + (row % RPLN) * CPR;
                xr[tt] = line % CPR;
            }
#pragma unroll
            for (int g = 0; g < KG; g++) {
                u32x4 b[NT];
#pragma unroll
                for (int tt = 0; tt < NT; tt++) {
                    b[tt] = lrow[tt][(2 * g + h) ^ xr[tt]];
                }
                // the expanded row operand of a class is used by all MT query tiles
#define GSIM_MFMA_CLASS(C)                                                                                     \
{                                                                                                          \
    v4i e[NT];                                                                                             \
    _Pragma("unroll") for (int tt = 0; tt < NT; tt++) e[tt] = fp4_class<C>(b[tt], km);                     \
    _Pragma("unroll") for (int m = 0; m < MT; m++)                                                         \
        _Pragma("unroll") for (int tt = 0; tt < NT; tt++)                                                  \
            acc[m][tt] = mfma_class<C>(aexp[m][g][C], e[tt], acc[m][tt]);                                  \
}
                GSIM_MFMA_CLASS(0)
                GSIM_MFMA_CLASS(1)
                GSIM_MFMA_CLASS(2)
                GSIM_MFMA_CLASS(3)
#undef GSIM_MFMA_CLASS
                // keep the 256-bit groups apart: with MT = 2 hipcc otherwise expands the row operands
                // of all groups first (64 more live registers, spills inside the tile loop)
                if (MT > 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MT > 1) {
            // all accumulators are complete HERE: without this hipcc sinks the MFMA chain of the
            // second query tile below the epilogue of the first and keeps every expanded row
            // operand alive for it
#pragma unroll
            for (int m = 0; m < MT; m++) asm volatile("" : "+v"(acc[m][0]));
        }
        MF_T(tk1);
        MF_ACC(0, tk0, tk1);
    };
#define ACC(m, tt) acc[m][tt]
    // the tiles' epilogue: linear pre-filter, exact path only for tiles with a passing pair (blk: the block the tiles belong
    // to, ps: its slot of row popcounts)
    auto tile_epilogue = [&](u64 blk, int t2, int ps) __attribute__((always_inline)) {
        MF_T(tk1);
        uint32_t pb[NT];
#pragma unroll
        for (int tt = 0; tt < NT; tt++) pb[tt] = sh.rpop[ps][(t2 + tt) * 32 + i];
        float pbf[NT];
        bool active[NT];
        u64 rowi[NT];
#pragma unroll
        for (int tt = 0; tt < NT; tt++) {
            pbf[tt] = static_cast<float>(pb[tt]);
            rowi[tt] = blk * RB + (t2 + tt) * 32 + i;
            active[tt] = rowi[tt] < a.nrows;
        }

        // ---- epilogue: linear pre-filter, exact path only for tiles with a passing pair ----
#pragma unroll
        for (int m = 0; m < MT; m++) {
            // Fast test, vector ALU only (a compare per pair into a scalar mask would stall on
            // the VALU->SALU dependency 32 times), two instructions per pair: the bound solved
            // for popc(row), maximum over the 16 queries of the lane, one compare per tile.
            u64 bits = 0, rmask = 0, bits_cand = 0, bits_count = 0;
            if (DN) {
                // dense cutoff: every pair is classified against the cutoff's band -- counted, dismissed, or (rarely)
                // left to the exact path -- and, as in the plain variant, against the query's top-k threshold
                const f32x4* kut = reinterpret_cast<const f32x4*>(sh.kap_u[wq][0][h]);
                const f32x4* kvt = reinterpret_cast<const f32x4*>(sh.kap_v[wq][0][h]);
                const f32x4* kus = reinterpret_cast<const f32x4*>(sh.kap_u[wq][1][h]);
                const f32x4* kvs = reinterpret_cast<const f32x4*>(sh.kap_v[wq][1][h]);
                const f32x4* kun = reinterpret_cast<const f32x4*>(sh.kap_a[wq][0][h]);
                const f32x4* kvn = reinterpret_cast<const f32x4*>(sh.kap_a[wq][1][h]);
                float mxt[NT], ph[NT], pl[NT];
                uint32_t tsure[NT], tmaybe[NT];
#pragma unroll
                for (int tt = 0; tt < NT; tt++) {
                    mxt[tt] = -3.0e38f;
                    ph[tt] = active[tt] ? pbf[tt] + 0.01f : __builtin_inff(); // (rows past the table's end: nothing passes)
                    pl[tt] = active[tt] ? pbf[tt] - 0.01f : __builtin_inff();
                    tsure[tt] = 0;
                    tmaybe[tt] = 0;
                }
                // (three passes over the accumulators, kept apart: unrolled together the constants of all of them are
                // loaded ahead and the expanded queries spill)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const f32x4 vut = kut[r4], vvt = kvt[r4];
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2 ut2{vut[e], vut[e + 1]}, vt2{vvt[e], vvt[e + 1]};
#pragma unroll
                        for (int tt = 0; tt < NT; tt++) {
                            const f32x2 c2{ACC(m, tt)[4 * r4 + e], ACC(m, tt)[4 * r4 + e + 1]};
                            const f32x2 t2v = __builtin_elementwise_fma(c2, ut2, vt2);
                            mxt[tt] = fmaxf(fmaxf(mxt[tt], t2v.x), t2v.y);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const f32x4 vus = kus[r4], vvs = kvs[r4];
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2 us2{vus[e], vus[e + 1]}, vs2{vvs[e], vvs[e + 1]};
#pragma unroll
                        for (int tt = 0; tt < NT; tt++) {
                            const f32x2 c2{ACC(m, tt)[4 * r4 + e], ACC(m, tt)[4 * r4 + e + 1]};
                            const f32x2 s2v = __builtin_elementwise_fma(c2, us2, vs2);
                            const uint32_t s0 = s2v.x >= ph[tt] ? 1u : 0u, s1 = s2v.y >= ph[tt] ? 1u : 0u;
                            kcnt[4 * r4 + e] += s0;
                            kcnt[4 * r4 + e + 1] += s1;
                            tsure[tt] += s0 + s1;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const f32x4 vun = kun[r4], vvn = kvn[r4];
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2 un2{vun[e], vun[e + 1]}, vn2{vvn[e], vvn[e + 1]};
#pragma unroll
                        for (int tt = 0; tt < NT; tt++) {
                            const f32x2 c2{ACC(m, tt)[4 * r4 + e], ACC(m, tt)[4 * r4 + e + 1]};
                            const f32x2 n2v = __builtin_elementwise_fma(c2, un2, vn2);
                            tmaybe[tt] += (n2v.x >= pl[tt] ? 1u : 0u) + (n2v.y >= pl[tt] ? 1u : 0u);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                bool anyd = false;
#pragma unroll
                for (int tt = 0; tt < NT; tt++) anyd = anyd || (active[tt] && mxt[tt] >= pbf[tt] - 0.01f) || tmaybe[tt] != tsure[tt];
                if (__builtin_expect(__ballot(anyd) != 0, 0)) {
                    const f32x4* kbt = reinterpret_cast<const f32x4*>(sh.kap_b[wq][0][h]);
                    const f32x4* kat = reinterpret_cast<const f32x4*>(sh.kap_b[wq][1][h]);
#pragma unroll
                    for (int r4 = 0; r4 < 4; r4++) {
                        const f32x4 vb = kbt[r4], va = kat[r4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int r = 4 * r4 + e;
#pragma unroll
                            for (int tt = 0; tt < NT; tt++)
                                bits_cand |= (active[tt] && ACC(m, tt)[r] >= __builtin_fmaf(vb[e], pbf[tt], va[e])) ? (1ull << (16 * tt + r)) : 0ull;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int r4 = 0; r4 < 4; r4++) {
                        const f32x4 vus = kus[r4], vvs = kvs[r4], vun = kun[r4], vvn = kvn[r4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int r = 4 * r4 + e;
#pragma unroll
                            for (int tt = 0; tt < NT; tt++) {
                                const float c = ACC(m, tt)[r];
                                const bool inband = __builtin_fmaf(c, vun[e], vvn[e]) >= pl[tt] && !(__builtin_fmaf(c, vus[e], vvs[e]) >= ph[tt]);
                                bits_count |= inband ? (1ull << (16 * tt + r)) : 0ull;
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    bits = bits_cand | bits_count;
                }
            }
            bool any = false;
            if (!DN) {
                const f32x4* kup = reinterpret_cast<const f32x4*>(sh.kap_u[wq][m][h]);
                const f32x4* kvp = reinterpret_cast<const f32x4*>(sh.kap_v[wq][m][h]);
                float mx[NT];
#pragma unroll
                for (int tt = 0; tt < NT; tt++) mx[tt] = -3.0e38f;
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const f32x4 vu = kup[r4], vv = kvp[r4];
                    // two pairs per instruction: v_pk_fma_f32 on the accumulator's register pairs, v_max3_f32
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2 u2{vu[e], vu[e + 1]}, v2{vv[e], vv[e + 1]};
#pragma unroll
                        for (int tt = 0; tt < NT; tt++) {
                            const f32x2 c2{ACC(m, tt)[4 * r4 + e], ACC(m, tt)[4 * r4 + e + 1]};
                            const f32x2 t2v = __builtin_elementwise_fma(c2, u2, v2);
                            mx[tt] = fmaxf(fmaxf(mx[tt], t2v.x), t2v.y);
                        }
                    }
                }
#pragma unroll
                for (int tt = 0; tt < NT; tt++) any = any || (active[tt] && mx[tt] >= pbf[tt] - 0.01f);
            }
            // which accumulator registers hold a passing pair (bit 16 tt + r: row tile tt)
            if (DN ? __ballot(bits != 0) != 0 : __ballot(any) != 0) {
                if (!DN) {
                    const f32x4* kap = reinterpret_cast<const f32x4*>(sh.kap_a[wq][m][h]);
                    const f32x4* kbp = reinterpret_cast<const f32x4*>(sh.kap_b[wq][m][h]);
#pragma unroll
                    for (int r4 = 0; r4 < 4; r4++) {
                        const f32x4 va = kap[r4], vb = kbp[r4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int r = 4 * r4 + e;
#pragma unroll
                            for (int tt = 0; tt < NT; tt++)
                                bits |= (active[tt] && ACC(m, tt)[r] >= __builtin_fmaf(vb[e], pbf[tt], va[e]))
                                            ? (1ull << (16 * tt + r))
                                            : 0ull;
                        }
                    }
                }
                // OR over the wavefront, 32 bits at a time
                uint32_t o[2] = {static_cast<uint32_t>(bits), static_cast<uint32_t>(bits >> 32)};
#pragma unroll
                for (int half = 0; half < (NT > 2 ? 2 : 1); half++) {
                    uint32_t v = o[half];
                    v |= dpp<0xB1>(v);
                    v |= dpp<0x4E>(v);
                    v |= dpp<0x141>(v);
                    v |= dpp<0x140>(v);
                    v |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), 16, 64));
                    v |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), 32, 64));
                    o[half] = __builtin_amdgcn_readfirstlane(v);
                }
                rmask = o[0] | (NT > 2 ? static_cast<u64>(o[1]) << 32 : 0ull);
            }
            // rare: stage the pairs that passed
            while (rmask) {
                const int bit = __builtin_ctzll(rmask);
                rmask &= rmask - 1;
                const int r = bit & 15;
                const int tsel = bit >> 4;
                float cf = pick16(ACC(m, 0), r);
                uint32_t pbs = pb[0];
                u64 rws = rowi[0];
#pragma unroll
                for (int tt = 1; tt < NT; tt++) {
                    if (tsel == tt) {
                        cf = pick16(ACC(m, tt), r);
                        pbs = pb[tt];
                        rws = rowi[tt];
                    }
                }
                const bool pass = (bits >> bit) & 1ull;
                const u64 mp = __ballot(pass);
                if (pass) {
                    const uint32_t slotp = staged + lane_rank(mp);
                    stg_row[slotp] = static_cast<uint32_t>(rws);
                    stg_cb[slotp] = (static_cast<uint32_t>(cf) << 16) + pbs;
                    stg_q[slotp] = static_cast<uint32_t>(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) | // query of the wave
                                   (DN ? (static_cast<uint32_t>((bits_cand >> bit) & 1ull) << 30) | (static_cast<uint32_t>((bits_count >> bit) & 1ull) << 31) : 0u);
                }
                staged += static_cast<uint32_t>(__popcll(mp));
                if (staged > kMStage - 64) {
                    MF_T(td0);
                    drain_stage();
                    MF_T(td1);
                    MF_ACC(8, td0, td1);
                }
            }
        }
        MF_T(tk2);
        MF_ACC(1, tk1, tk2);
    };
    // the polled thresholds (landed some blocks ago) and, when its histogram has landed, the refreshed one
    // Rows of up to 512 bits with more than 64 queries, 1024-bit rows with more than 128 (two query tiles per wave) and the
    // dense-cutoff variant keep round 4's upkeep: EVERY wave re-derives one of its queries'
    // thresholds every fourth block from the table-wide histogram, loaded straight into registers (the wait for those loads
    // drains the row DMA -- harmless where a block is long; the staged refresh's rate, one wave per two blocks, cost these
    // batches 8-35 %).
    const bool direct_refresh = (WORDS <= 16 && nq > 64) || (WORDS == 32 && nq > 128) || DN;
    const int rshift = nblocks >= 64ull * gridDim.x ? 2 : 0;
    auto thresholds = [&]() __attribute__((always_inline)) {
        MF_T(tr0);
        // the polled thresholds (landed some blocks ago) and, when its histogram has landed, the refreshed one
        uint32_t gt = (lane < QW && ((it & 3u) == 0 || direct_refresh)) ? sh.tau_poll[wq][lane] : 0u;
        if constexpr (WORDS <= 32 || DN) {
            const int qref = q0t + static_cast<int>(((turn >> rshift) + rgroup * (QW / ngroups)) & (QW - 1));
            if (direct_refresh && qref < nq && (turn & ((1u << rshift) - 1u)) == 0) {
                uint32_t hh[PER];
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(qstate[qref].ghist, 0, kBBins * 4, 0x00020000);
                uint32_t sum = 0;
#pragma unroll
                for (int v = 0; v < PER / 4; v++) {
                    const u32x4 v4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * PER * 4 + v * 16, 0, /*sc1*/ 16);
                    hh[4 * v + 0] = v4.x;
                    hh[4 * v + 1] = v4.y;
                    hh[4 * v + 2] = v4.z;
                    hh[4 * v + 3] = v4.w;
                    sum += v4.x + v4.y + v4.z + v4.w;
                }
                uint32_t bin_k, cnt;
                threshold_from_counts<PER>(hh, sum, a.k, lane, bin_k, cnt);
                bin_k = __builtin_amdgcn_readfirstlane(bin_k);
                cnt = __builtin_amdgcn_readfirstlane(cnt);
                if (cnt >= a.k) {
                    if (lane == 0)
                        __hip_atomic_fetch_max((g_u32p) &qstate[qref].gtau, bin_k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (lane == (qref & (QW - 1)) && bin_k > gt) gt = bin_k;
                }
            }
        }
        if (pend_q >= 0 && it >= pend_at) {
            uint32_t hh[PER];
            const u32x4* hp = reinterpret_cast<const u32x4*>(&sh.hist[pend_buf][lane * PER]);
            uint32_t sum = 0;
#pragma unroll
            for (int v = 0; v < PER / 4; v++) {
                const u32x4 v4 = hp[v];
                hh[4 * v + 0] = v4.x;
                hh[4 * v + 1] = v4.y;
                hh[4 * v + 2] = v4.z;
                hh[4 * v + 3] = v4.w;
                sum += v4.x + v4.y + v4.z + v4.w;
            }
            uint32_t bin_k, cnt;
            threshold_from_counts<PER>(hh, sum, a.k, lane, bin_k, cnt);
            bin_k = __builtin_amdgcn_readfirstlane(bin_k);
            cnt = __builtin_amdgcn_readfirstlane(cnt);
            if (cnt >= a.k) {
                if (lane == 0)
                    __hip_atomic_fetch_max((g_u32p) &qstate[pend_q].gtau, bin_k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == (pend_q & (QW - 1)) && bin_k > gt) gt = bin_k;
            }
            pend_q = -1;
        }
        if (lane < QW && gt > sh.tau[wq][lane]) set_query_constants(gt);
        MF_T(tr9);
        MF_ACC(3, tr0, tr9);
    };

    for (; blk < nblocks; blk += gridDim.x, turn++, it++) {
        MF_T(tb0);
        // ---- this iteration's requests ----
        issue_poll();
        issue_block(blk + static_cast<u64>(AHEAD) * gridDim.x, (slot + AHEAD) % RING, (slot + AHEAD) % RING);
        const uint32_t ev = turn >> RS;
        const bool on_duty = wave_has_queries && !direct_refresh && (turn & ((1u << RS) - 1u)) == 0 && static_cast<int>(ev % WV) == wq && pend_q < 0;
        if (on_duty) {
            const int qref = q0t + static_cast<int>(((ev / WV) + rgroup * (QW / ngroups)) & (QW - 1));
            if (qref < nq) {
                pend_q = qref;
                pend_buf = static_cast<int>(ev & static_cast<uint32_t>(Ring::HB - 1));
                pend_at = it + LAT;
                const u32x4* hsrc = reinterpret_cast<const u32x4*>(qstate[qref].ghist);
#pragma unroll
                for (int v = 0; v < kBBins / 256; v++)
                    __builtin_amdgcn_global_load_lds((g_cvoidp) (hsrc + v * 64 + lane), (l_voidp) (&sh.hist[pend_buf][v * 256]), 16, 0, /*sc1*/ 16);
            }
        }
        MF_T(tb2);
        MF_ACC(7, tb0, tb2);
        if (wave_has_queries) {
#pragma unroll 1
            for (int t2 = NT * rgroup; t2 < NTB; t2 += NT * ngroups) {
                contract(sh.rows[slot], t2);
                tile_epilogue(blk, t2, pslot);
            }
            thresholds();
        }
        MF_T(tr1);
        wait_for_next_block();
        MF_T(tw);
        MF_ACC(4, tr1, tw);
        lds_barrier();
        slot = slot + 1 == RING ? 0 : slot + 1;
        pslot = pslot + 1 == Ring::RP ? 0 : pslot + 1;
        MF_T(tb1);
        MF_ACC(5, tw, tb1);
        MF_ACC(2, tb0, tb1);
    }
#undef ACC
    wait_vm_outstanding<0>(); // (requests past the table's end are still landing in LDS)
#if GSIM_MF_TIMING
    if (wq == 0 && lane == 0)
        for (int d = 0; d < 9; d++) atomicAdd(&rr.flags[2 + d], static_cast<uint32_t>(tacc[d] >> 6));
#endif
    if (staged) drain_stage();
    if (DN) { // this lane's counts -> the wave's per-query counts (slot r of lane half h is query (r & 3) + 8 (r >> 2) + 4 h of the tile)
#pragma unroll
        for (int r = 0; r < 16; r++)
            if (kcnt[r]) atomicAdd(&sh.kept[wq][(r & 3) + 8 * (r >> 2) + 4 * h], kcnt[r]);
        __builtin_amdgcn_wave_barrier();
        if (wq == 0 && lane == 0) atomicOr(rr.flags, 16u); // the dense route ran: the host does not re-run the batch on the VALU pass
    }
    if (has_cutoff && lane < QW && q0t + lane < nq && sh.kept[wq][lane])
        atomicAdd(&qstate[q0t + lane].kept, static_cast<u64>(sh.kept[wq][lane]));
    if (lane == 0) {
        rr.seg_count[w] = cursor < rr.seg_cap ? cursor : rr.seg_cap;
        if (cursor > rr.seg_cap) { // segment overflow: the host grows the segments to what was needed (flags[15]) and runs the batch again
            atomicOr(rr.flags, 1u);
            atomicMax(&rr.flags[15], cursor);
        }
    }
}

// ---- sample pass on the matrix cores ------------------------------------------------------
// Starting thresholds for all a.nq queries from a strided sample of row blocks: the contraction
// of the main kernel (one query tile per wave), then EVERY pair of the sampled tiles is scored
// exactly and counted in a coarse (128-bin) per-query histogram in LDS; the workgroups add their
// histograms into the table-wide ones (at the lower edge of each coarse bin, so the threshold
// derived from them is conservative) and the last workgroup turns them into gtau and clears
// them.  Replaces ceil(nq / 32) launches of the VALU sample kernel (3.3 ms for 256 queries on
// 2048-bit rows) with one launch of a few hundred microseconds.
template <int WORDS>
__global__ __launch_bounds__(kMBlock) void batch_mfma_sample_kernel(BatchArgs a, uint32_t nsb, u64 stride_blocks)
{
    constexpr int KG = WORDS / 8, CPR = WORDS / 4, RPLN = 16 / CPR, RB = kMChunks / CPR, NTB = RB / 32;
    constexpr int kCoarseWords = 64; // 128 coarse bins, two 16-bit counters per word: 64 KB for 256 queries
    constexpr int kFinePerCoarse = kBBins / (2 * kCoarseWords);
    static_assert(kMfmaQueries * kCoarseWords * 4 <= kMChunks * 16, "histograms live in the second row buffer");
    __shared__ MfmaShared sh;        // rows[0]: the sampled block, rows[1]: the histograms
    __shared__ uint32_t s_last;
    uint32_t* hist = reinterpret_cast<uint32_t*>(sh.rows[1]);
    const int lane = threadIdx.x & 63;
    const int wq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int nq = static_cast<int>(a.nq);
    const int ntiles = (nq + 31) / 32;
    const int p2 = ntiles <= 1 ? 1 : (ntiles <= 2 ? 2 : (ntiles <= 4 ? 4 : 8));
    const int tile = wq % p2, rgroup = wq / p2, ngroups = kMWaves / p2;
    const int q0t = tile * 32;
    const bool wave_has_queries = q0t < nq && rgroup * 2 < NTB;
    const BatchRare rr = *a.rare;
    BatchQueryState* qstate = rr.qstate + a.q0;
    const u32x4* __restrict__ db = reinterpret_cast<const u32x4*>(a.rows);
    for (int x = threadIdx.x; x < kMfmaQueries * kCoarseWords; x += kMBlock) hist[x] = 0;
    if (lane < 32) sh.qpop[wq][lane] = q0t + lane < nq ? a.qpop[a.q0 + q0t + lane] : 0u;

    ClassMasks km{0x11111111u, 0x22222222u, 0x44444444u};
    asm volatile("" : "+v"(km.m1), "+v"(km.m2), "+v"(km.m4));
    v4i aexp[KG][4];
    {
        const int ql = q0t + i;
        const u32x4* qp = reinterpret_cast<const u32x4*>(a.queries + static_cast<size_t>(a.q0 + ql) * WORDS);
#pragma unroll
        for (int g = 0; g < KG; g++) {
            const u32x4 x = ql < nq ? qp[2 * g + h] : u32x4{0, 0, 0, 0};
            aexp[g][0] = fp4_class<0>(x, km);
            aexp[g][1] = fp4_class<1>(x, km);
            aexp[g][2] = fp4_class<2>(x, km);
            aexp[g][3] = fp4_class<3>(x, km);
        }
    }
    // row block staging exactly as in batch_mfma_kernel (one buffer)
    constexpr int RPJ = kMBlock / CPR;
    const int line0 = wq * 4 + (lane >> 4);
    const uint32_t rowl0 = static_cast<uint32_t>(line0 * RPLN + (lane & 15) / CPR);
    const uint32_t chunk0 = static_cast<uint32_t>(((lane & 15) % CPR) ^ (line0 % CPR));
    const u32x4* dbc = db + chunk0;
    const uint32_t last_row = static_cast<uint32_t>(a.nrows - 1);
    __syncthreads();

    for (uint32_t sb = blockIdx.x; sb < nsb; sb += gridDim.x) {
        const u64 blk = static_cast<u64>(sb) * stride_blocks;
        const uint32_t first = static_cast<uint32_t>(blk) * RB + rowl0;
#pragma unroll
        for (int j = 0; j < kMChunks / kMBlock; j++) {
            uint32_t grow = first + j * RPJ;
            grow = grow < last_row ? grow : last_row;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (dbc + static_cast<u64>(grow) * CPR),
                                             (__attribute__((address_space(3))) void*) (&sh.rows[0][(j * kMWaves + wq) * 64]), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (wave_has_queries) {
#pragma unroll 1
            for (int t2 = 2 * rgroup; t2 < NTB; t2 += 2 * ngroups) {
                v16f acc[2] = {v16f{}, v16f{}};
                uint32_t pb[2] = {0, 0};
#pragma unroll
                for (int g = 0; g < KG; g++) {
                    u32x4 b[2];
#pragma unroll
                    for (int tt = 0; tt < 2; tt++) {
                        const int row = (t2 + tt) * 32 + i;
                        const int line = row / RPLN;
                        b[tt] = sh.rows[0][line * 16 + (row % RPLN) * CPR + ((2 * g + h) ^ (line % CPR))];
                        pb[tt] = bcnt_acc(b[tt].x, pb[tt]);
                        pb[tt] = bcnt_acc(b[tt].y, pb[tt]);
                        pb[tt] = bcnt_acc(b[tt].z, pb[tt]);
                        pb[tt] = bcnt_acc(b[tt].w, pb[tt]);
                    }
#pragma unroll
                    for (int tt = 0; tt < 2; tt++) {
                        acc[tt] = mfma_class<0>(aexp[g][0], fp4_class<0>(b[tt], km), acc[tt]);
                        acc[tt] = mfma_class<1>(aexp[g][1], fp4_class<1>(b[tt], km), acc[tt]);
                        acc[tt] = mfma_class<2>(aexp[g][2], fp4_class<2>(b[tt], km), acc[tt]);
                        acc[tt] = mfma_class<3>(aexp[g][3], fp4_class<3>(b[tt], km), acc[tt]);
                    }
                }
#pragma unroll
                for (int tt = 0; tt < 2; tt++) {
                    pb[tt] += static_cast<uint32_t>(__shfl_xor(static_cast<int>(pb[tt]), 32, 64));
                    const bool active = blk * RB + (t2 + tt) * 32 + i < a.nrows;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int qi = (r & 3) + 8 * (r >> 2) + 4 * h;
                        float sc = score_of(a.metric, a.alpha, a.beta, sh.qpop[wq][qi], pb[tt], static_cast<uint32_t>(acc[tt][r]));
                        sc = apply_cutoff(sc, a.cutoff);
                        const uint32_t coarse = batch_bin(sc) / kFinePerCoarse;
                        // (plain LDS atomics: aggregating equal bins across the lanes first was 4x slower)
                        if (active && q0t + qi < nq)
                            atomicAdd(&hist[(q0t + qi) * kCoarseWords + (coarse >> 1)], (coarse & 1u) ? 65536u : 1u);
                    }
                }
            }
        }
        __syncthreads(); // the next block overwrites rows[0]
    }
    // this workgroup's counts into the table-wide histograms
    for (int x = threadIdx.x; x < nq * kCoarseWords; x += kMBlock) {
        const uint32_t v = hist[x];
        if (v == 0) continue;
        BatchQueryState* gq = &qstate[x / kCoarseWords];
        const int c0 = 2 * (x % kCoarseWords);
        if (v & 0xFFFFu) atomicAdd(&gq->ghist[c0 * kFinePerCoarse], v & 0xFFFFu);
        if (v >> 16) atomicAdd(&gq->ghist[(c0 + 1) * kFinePerCoarse], v >> 16);
    }
    // the last workgroup turns every histogram into a starting threshold (as batch_scan_kernel<.., SAMPLE>)
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(rr.ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    for (int q = wq; q < nq; q += kMWaves) {
        BatchQueryState* gq = &qstate[q];
        constexpr int PER = kBBins / 64;
        uint32_t hh[PER];
        uint32_t sum = 0;
#pragma unroll
        for (int v = 0; v < PER; v++) {
            hh[v] = __hip_atomic_load(&gq->ghist[lane * PER + v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sum += hh[v];
        }
        uint32_t bin_k, cnt;
        threshold_from_counts<PER>(hh, sum, a.k, lane, bin_k, cnt);
        if (lane == 0) gq->gtau = (a.k && cnt >= a.k) ? bin_k : 0u;
        if (a.cutoff > 0.0f) {
            // rows below the cutoff were counted with score 0 (coarse bin 0): more than 1/512 of the
            // sample above it means millions of pairs for the exact path -> flag the batch for the
            // VALU pass (bit 3)
            const uint32_t total = wave_sum(sum);
            const uint32_t below = static_cast<uint32_t>(__shfl(static_cast<int>(hh[0]), 0, 64));
            if (lane == 0 && static_cast<u64>(total - below) * 512ull > total) atomicOr(rr.flags, 8u);
        }
#pragma unroll
        for (int v = 0; v < PER; v++) gq->ghist[lane * PER + v] = 0;
    }
    if (threadIdx.x == 0) *rr.ticket = 0;
}



// popc(row) of every row as a 16-bit side array (2 bytes per row, computed once per table): the matrix-core pass
// stages it with the rows instead of counting every row's bits again in each of its eight waves (one v_bcnt per
// row word and wave: 1 of its ~9 vector instructions per MFMA).
template <int CPR> __global__ __launch_bounds__(256) void row_popcount_kernel(const u32x4* __restrict__ rows, u64 nchunks, uint16_t* out)
{
    constexpr int UN = 4;
    const u64 c0 = static_cast<u64>(blockIdx.x) * (256 * UN) + threadIdx.x;
    u32x4 x[UN];
#pragma unroll
    for (int j = 0; j < UN; j++) {
        const u64 c = c0 + j * 256;
        x[j] = c < nchunks ? __builtin_nontemporal_load(rows + c) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int j = 0; j < UN; j++) {
        uint32_t p = bcnt_acc(x[j].x, bcnt_acc(x[j].y, bcnt_acc(x[j].z, bcnt_acc(x[j].w, 0u))));
#pragma unroll
        for (int d = 1; d < CPR; d <<= 1) p += static_cast<uint32_t>(__shfl_xor(static_cast<int>(p), d, 64));
        const u64 c = c0 + j * 256;
        if (c < nchunks && c % CPR == 0) out[c / CPR] = static_cast<uint16_t>(p);
    }
}

// debug hook: the pre-filter constants as the device computes them, for qa = 0..max_qa and (no
// cutoff) every threshold bin: out[(qa * nlev + lev) * 4 + {ka, kb, u, v}], nlev = 512 or 1
__global__ __launch_bounds__(256) void prefilter_table_kernel(int tversky, float alpha, float beta, uint32_t max_qa, int has_cutoff,
                                                              float cutoff, float* out)
{
    const uint32_t nlev = has_cutoff ? 1u : static_cast<uint32_t>(kBBins);
    const u64 n = static_cast<u64>(max_qa + 1) * nlev;
    const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t qa = static_cast<uint32_t>(i / nlev), lev = static_cast<uint32_t>(i % nlev);
    const PrefilterConstants k = prefilter_constants(tversky != 0, alpha, beta, qa, prefilter_level(has_cutoff != 0, cutoff, lev), true);
    out[4 * i + 0] = k.ka;
    out[4 * i + 1] = k.kb;
    out[4 * i + 2] = k.u;
    out[4 * i + 3] = k.v;
}

} // namespace

hipError_t launch_prefilter_table(int tversky, float alpha, float beta, uint32_t max_qa, int has_cutoff, float cutoff, float* d_out,
                                  hipStream_t s)
{
    const u64 n = static_cast<u64>(max_qa + 1) * (has_cutoff ? 1u : static_cast<uint32_t>(kBBins));
    hipLaunchKernelGGL(prefilter_table_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, tversky, alpha, beta,
                       max_qa, has_cutoff, cutoff, d_out);
    return hipGetLastError();
}

void prefilter_table_host(int tversky, float alpha, float beta, uint32_t max_qa, int has_cutoff, float cutoff, float* out)
{
    const uint32_t nlev = has_cutoff ? 1u : static_cast<uint32_t>(kBBins);
    for (uint32_t qa = 0; qa <= max_qa; qa++)
        for (uint32_t lev = 0; lev < nlev; lev++) {
            const PrefilterConstants k =
                prefilter_constants(tversky != 0, alpha, beta, qa, prefilter_level(has_cutoff != 0, cutoff, lev), true);
            float* o = out + 4 * (static_cast<size_t>(qa) * nlev + lev);
            o[0] = k.ka, o[1] = k.kb, o[2] = k.u, o[3] = k.v;
        }
}

namespace
{
} // namespace

hipError_t launch_row_popcounts(const void* rows, uint64_t nrows, uint32_t W, uint16_t* d_out, hipStream_t s)
{
    if (!batch_mfma_supported(W)) return hipErrorInvalidValue;
    const u64 nchunks = nrows * (W / 4);
    if (nchunks == 0) return hipSuccess;
    const dim3 grid(static_cast<uint32_t>((nchunks + 1023) / 1024)), block(256);
    const u32x4* r = static_cast<const u32x4*>(rows);
    if (W == 64) hipLaunchKernelGGL((row_popcount_kernel<16>), grid, block, 0, s, r, nchunks, d_out);
    else if (W == 32) hipLaunchKernelGGL((row_popcount_kernel<8>), grid, block, 0, s, r, nchunks, d_out);
    else if (W == 16) hipLaunchKernelGGL((row_popcount_kernel<4>), grid, block, 0, s, r, nchunks, d_out);
    else hipLaunchKernelGGL((row_popcount_kernel<2>), grid, block, 0, s, r, nchunks, d_out);
    return hipGetLastError();
}

bool batch_mfma_supported(uint32_t W)
{
    return W == 8 || W == 16 || W == 32 || W == 64; // 256 ... 2048-bit rows (multiples of the 256-bit MFMA group)
}

uint32_t batch_mfma_waves(int num_cus)
{
    return static_cast<uint32_t>(num_cus) * kMWaves;
}

// Row blocks in the sample: about a million rows, never more than 1/16 of the table.
static uint32_t batch_mfma_sample_blocks(uint32_t W, uint64_t nrows)
{
    const uint32_t rb = kMChunks / (W / 4);
    const u64 nblocks = (nrows + rb - 1) / rb;
    const u64 nsb = std::min<u64>((1u << 20) / rb, nblocks / 16);
    return static_cast<uint32_t>(nsb ? nsb : 1);
}

// Does the matrix-core sample pass apply to this table?  (Large tables only; batches with a cutoff
// need it: it also estimates how many rows the cutoff keeps.)
bool batch_mfma_sample_applies(uint32_t W, uint64_t nrows, uint32_t nq, uint32_t k, int num_cus, bool enabled)
{
    if (!enabled || k == 0 || nq > static_cast<uint32_t>(kMfmaQueries) || !batch_mfma_supported(W)) return false;
    const uint32_t rb = kMChunks / (W / 4);
    const u64 nblocks = (nrows + rb - 1) / rb;
    if (nblocks < 64) return false; // tiny tables: the scan's own threshold upkeep is enough
    const uint32_t nsb = batch_mfma_sample_blocks(W, nrows);
    return (static_cast<u64>(nsb) + num_cus - 1) / num_cus * rb <= 60000u; // 16-bit LDS counters per workgroup
}

// Sample pass for all a.nq queries in one launch; false when the table is too small for it (the
// caller then uses the VALU sample passes, which have their own size rules).
bool launch_batch_mfma_sample(const BatchArgs& a, int num_cus, hipStream_t s, hipError_t* err)
{
    *err = hipSuccess;
    if (!batch_mfma_sample_applies(a.W, a.nrows, a.nq, a.k, num_cus, (a.opts & 1u) != 0)) return false;
    const uint32_t rb = kMChunks / (a.W / 4);
    const u64 nblocks = (a.nrows + rb - 1) / rb;
    const uint32_t nsb = batch_mfma_sample_blocks(a.W, a.nrows);
    const u64 stride = nblocks / nsb;
    if (a.W == 64)
        hipLaunchKernelGGL((batch_mfma_sample_kernel<64>), dim3(num_cus), dim3(kMBlock), 0, s, a, nsb, stride);
    else if (a.W == 32)
        hipLaunchKernelGGL((batch_mfma_sample_kernel<32>), dim3(num_cus), dim3(kMBlock), 0, s, a, nsb, stride);
    else if (a.W == 16)
        hipLaunchKernelGGL((batch_mfma_sample_kernel<16>), dim3(num_cus), dim3(kMBlock), 0, s, a, nsb, stride);
    else
        hipLaunchKernelGGL((batch_mfma_sample_kernel<8>), dim3(num_cus), dim3(kMBlock), 0, s, a, nsb, stride);
    *err = hipGetLastError();
    return true;
}

// Is there a band for the dense-cutoff variant (the same for every query of a batch up to 2048-bit rows)?
bool batch_mfma_dense_applies(int metric, float alpha, float beta, float cutoff, bool enabled)
{
    return enabled && cutoff_band(metric == GSIM_METRIC_TVERSKY, alpha, beta, 2048u, cutoff, true).on;
}

// The scan of one pass (a.nq <= kMfmaQueries queries from a.q0); with a cutoff only after
// launch_batch_mfma_sample (it flags cutoffs that keep too many rows); thresholds come
// from the sample passes launched before it, finish with launch_batch_finish.
template <int WORDS, int MT, int NT, int BC, int RING, bool DN>
static void launch_variant(const BatchArgs& a, int num_cus, hipStream_t s)
{
    constexpr int RB = BC / (WORDS / 4);
    const u64 nblocks = (a.nrows + RB - 1) / RB;
    hipLaunchKernelGGL((batch_mfma_kernel<WORDS, MT, NT, BC, RING, DN>), dim3(num_cus), dim3(kMBlock), 0, s, a, nblocks);
}

hipError_t launch_batch_mfma_scan(const BatchArgs& a, int num_cus, hipStream_t s)
{
    if (a.nq > static_cast<uint32_t>(kMfmaQueries) || a.nrows == 0) return hipErrorInvalidValue;
    // 64 KB row blocks, two buffers: the geometry that won for every batch size at 1024 and 2048 bits (round 5: smaller
    // blocks in deeper rings, class planes shared through LDS, staggered waves, four-wave workgroups -- all measured,
    // profiles/r05_batch_mfma_experiments.txt); 256-bit rows: 32 KB blocks, three buffers (their popcount slots are large).
    if (a.W == 64) {
        // (with one query tile only four of the eight row groups find a pair of row tiles; single
        // tiles for all eight waves -- NT = 1 -- were 20-50 % slower: one MFMA chain per wave)
        launch_variant<64, 1, 2, kMChunks, 2, false>(a, num_cus, s);
    } else if (a.W == 32) {
        // two query tiles per wave halve the operand work: 8 % faster at 256 queries, even at 128,
        // slower below (fewer row groups per query tile)
        if (a.nq > 128) launch_variant<32, 2, 1, kMChunks, 2, false>(a, num_cus, s);
        else launch_variant<32, 1, 2, kMChunks, 2, false>(a, num_cus, s);
    } else if (a.W == 16) {
        // narrow rows: the epilogue outweighs the MFMAs, two query tiles per wave from 65 queries on
        if (a.nq > 64) launch_variant<16, 2, 1, kMChunks, 2, false>(a, num_cus, s);
        else launch_variant<16, 1, 2, kMChunks, 2, false>(a, num_cus, s);
    } else if (a.W == 8) {
        if (a.nq > 64) launch_variant<8, 2, 1, kMChunks / 2, 3, false>(a, num_cus, s);
        else launch_variant<8, 1, 2, kMChunks / 2, 3, false>(a, num_cus, s);
    } else {
        return hipErrorInvalidValue;
    }
    // A cutoff the sample pass found to keep a sizeable part of the table (flag 8, read on the device): the plain kernel
    // above has returned at once, the dense variant -- one query tile per wave, the kept rows counted in registers --
    // runs the batch; any other batch it leaves alone.  Weights without a usable band (cutoff_band): the VALU pass,
    // re-enqueued by the host when it finds flag 8 without flag 16.
    if (a.cutoff > 0.0f && batch_mfma_dense_applies(a.metric, a.alpha, a.beta, a.cutoff, (a.opts & 2u) != 0)) {
        // (2048-bit rows: ONE row tile per wave.  With two -- the plain kernel's shape -- the sixteen per-query counters and the band constants no
        // longer fit beside the 128 registers of expanded queries: 100-104 registers spilled whichever way the epilogue is ordered, 65.8 ms
        // against 39.9, round 6.  The single MFMA chain per wave is why this variant sits at 0.74 of its issue ceiling: DESIGN.md section 3)
        if (a.W == 64) launch_variant<64, 1, 1, kMChunks, 2, true>(a, num_cus, s);
        else if (a.W == 32) launch_variant<32, 1, 2, kMChunks, 2, true>(a, num_cus, s);
        else if (a.W == 16) launch_variant<16, 1, 2, kMChunks, 2, true>(a, num_cus, s);
        else launch_variant<8, 1, 2, kMChunks / 2, 3, true>(a, num_cus, s);
    }
    return hipGetLastError();
}

} // namespace gsim
