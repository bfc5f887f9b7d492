// gsim_device_common.h -- device helpers shared by the kernels in gsim_scan.hip, gsim_fused.hip, gsim_select.hip and
// gsim_batch.hip: candidate keys, coarse bins, the score arithmetic, wave64 cross-lane
// helpers, threshold search.  Internal.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gpusim_hip.h"
#include "gsim_device.h"

namespace gsim
{
namespace
{

// 16 bytes per lane; a wave64 instruction covers 1 KiB of consecutive table bytes.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kStage = 128; // candidates staged in LDS per wave (a flush is triggered above 64)

typedef unsigned long long u64;

// ---------------------------------------------------------------------------
// keys, bins, scores
// ---------------------------------------------------------------------------

// Order-preserving u32 image of a float (any sign): larger score <=> larger key.
__device__ __forceinline__ uint32_t order_key(float s)
{
    const uint32_t b = __float_as_uint(s);
    return b ^ (static_cast<uint32_t>(static_cast<int32_t>(b) >> 31) | 0x80000000u);
}

__device__ __forceinline__ float key_score(uint32_t key)
{
    const uint32_t b = key ^ ((key & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(b);
}

// candidate key: (order_key(score) << 32) | ~row  -> descending key order is the
// canonical result order (score desc, row asc); keys are unique.
__device__ __forceinline__ u64 make_key(float s, uint32_t row)
{
    return (static_cast<u64>(order_key(s)) << 32) | static_cast<u64>(~row);
}

// Monotone (non-decreasing in score) coarse bin; the scaling is by a power of
// two, hence exact.
__device__ __forceinline__ uint32_t coarse_bin(float s)
{
    const float t = fminf(fmaxf(s, 0.0f), 1.0f) * static_cast<float>(kScanBins);
    const uint32_t b = static_cast<uint32_t>(t);
    return b < static_cast<uint32_t>(kScanBins) ? b : static_cast<uint32_t>(kScanBins - 1);
}

// Coarse bin of the multi-query passes (kBBins bins; gsim_batch.hip, gsim_batch_mfma.hip).
__device__ __forceinline__ uint32_t batch_bin(float s)
{
    const float t = fminf(fmaxf(s, 0.0f), 1.0f) * static_cast<float>(kBBins);
    const uint32_t b = static_cast<uint32_t>(t);
    return b < static_cast<uint32_t>(kBBins) ? b : static_cast<uint32_t>(kBBins - 1);
}

// The reference's arithmetic, fingerprintdb_cuda.cu:89-101:
//   score = (float)common / (float)(total - common), total = popc(q) + popc(d)
// one correctly rounded IEEE f32 divide.  Tversky (build-defined; oracle
// gso_score_one is the twin): one rounding per operation, in this order.
__device__ __forceinline__ float score_of(int metric, float alpha, float beta, uint32_t a, uint32_t b,
                                          uint32_t c)
{
    if (metric == GSIM_METRIC_TVERSKY) {
        const float t1 = __fmul_rn(alpha, static_cast<float>(static_cast<int>(a - c)));
        const float t2 = __fmul_rn(beta, static_cast<float>(static_cast<int>(b - c)));
        const float den = __fadd_rn(__fadd_rn(t1, t2), static_cast<float>(c));
        return __fdiv_rn(static_cast<float>(c), den);
    }
    const int total = static_cast<int>(a + b);
    return __fdiv_rn(static_cast<float>(static_cast<int>(c)),
                     static_cast<float>(total - static_cast<int>(c)));
}

// Denominator of score_of as its own step (the multi-query kernel pre-filters on it):
// score_of(...) == __fdiv_rn((float)c, score_den(...)) bit for bit.
__device__ __forceinline__ float score_den(int metric, float alpha, float beta, uint32_t a, uint32_t b, uint32_t c)
{
    if (metric == GSIM_METRIC_TVERSKY) {
        const float t1 = __fmul_rn(alpha, static_cast<float>(static_cast<int>(a - c)));
        const float t2 = __fmul_rn(beta, static_cast<float>(static_cast<int>(b - c)));
        return __fadd_rn(__fadd_rn(t1, t2), static_cast<float>(c));
    }
    return static_cast<float>(static_cast<int>(a + b) - static_cast<int>(c));
}

// popcount-accumulate, explicitly: acc + popc(x) in ONE v_bcnt_u32_b32 (hipcc otherwise emits
// v_bcnt x, 0 followed by v_add3 trees: +25 % VALU work in the multi-query inner loop)
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// fingerprintdb_cuda.cu:101  (NaN compares false -> 0)
__device__ __forceinline__ float apply_cutoff(float s, float cutoff)
{
    return s >= cutoff ? s : 0.0f;
}

// ---------------------------------------------------------------------------
// cross-lane helpers (wave64)
// ---------------------------------------------------------------------------

template <int CTRL> __device__ __forceinline__ uint32_t dpp(uint32_t v)
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, true));
}

// row_shr:N within each row of 16 lanes; lanes without a source read 0 (prefix sums)
template <int N> __device__ __forceinline__ uint32_t dpp_shr(uint32_t v)
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x110 + N, 0xF, 0xF, true));
}

// Sum over each aligned group of LPR consecutive lanes; every lane of the group
// ends up with the total.  DPP row operations up to 16 lanes.
template <int LPR> __device__ __forceinline__ uint32_t group_sum(uint32_t v)
{
    if (LPR >= 2) v += dpp<0xB1>(v);  // quad_perm [1,0,3,2]
    if (LPR >= 4) v += dpp<0x4E>(v);  // quad_perm [2,3,0,1]
    if (LPR >= 8) v += dpp<0x141>(v); // row_half_mirror
    if (LPR >= 16) v += dpp<0x140>(v); // row_mirror
    if (LPR >= 32) v += static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), 16, 64));
    if (LPR >= 64) v += static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), 32, 64));
    return v;
}

__device__ __forceinline__ uint32_t lane_rank(u64 mask)
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    for (int d = 32; d > 0; d >>= 1) v += static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), d, 64));
    return v;
}

// Largest bin B with sum_{b >= B} count[b] >= k, and that sum; the whole histogram
// holds fewer than k entries: B = 0 and the total.  One wavefront; lane l holds the
// counts of bins [PER l, PER l + PER) in h[] and their sum in s.  k >= 1.
template <int PER>
__device__ __forceinline__ void threshold_from_counts(const uint32_t (&h)[PER], uint32_t s, uint32_t k, int lane,
                                                      uint32_t& bin_out, uint32_t& cnt_out)
{
    uint32_t incl = s; // suffix sum over lanes >= lane
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = static_cast<uint32_t>(__shfl_down(static_cast<int>(incl), d, 64));
        if (lane + d < 64) incl += t;
    }
    const u64 m = __ballot(incl >= k);
    if (m == 0) {
        bin_out = 0;
        cnt_out = static_cast<uint32_t>(__shfl(static_cast<int>(incl), 0, 64));
        return;
    }
    const int L = 63 - __clzll(static_cast<long long>(m));
    uint32_t acc = incl - s; // entries in lanes above this one
    uint32_t bin = 0, cnt = 0;
    bool found = false;
#pragma unroll
    for (int i = PER - 1; i >= 0; i--) {
        acc += h[i];
        if (!found && acc >= k) {
            found = true;
            bin = static_cast<uint32_t>(lane * PER + i);
            cnt = acc;
        }
    }
    bin_out = static_cast<uint32_t>(__shfl(static_cast<int>(bin), L, 64));
    cnt_out = static_cast<uint32_t>(__shfl(static_cast<int>(cnt), L, 64));
}

__device__ __forceinline__ void find_threshold(const uint32_t* hist, uint32_t k, int lane, uint32_t& bin_out,
                                               uint32_t& cnt_out)
{
    constexpr int PER = kScanBins / 64;
    uint32_t h[PER];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        h[i] = hist[lane * PER + i];
        s += h[i];
    }
    threshold_from_counts<PER>(h, s, k, lane, bin_out, cnt_out);
}

// The coarse histogram as a layout: base[b] = rows in the bins above b (bin b's first position in a list ordered by bin, highest
// first), B* = the bin of the k-th best (find_threshold), and the largest population among the bins >= B*.  One wavefront
// (threads 0..63 of the workgroup); base: kScanBins words of LDS.
__device__ __forceinline__ void bin_layout(const uint32_t* hist, uint32_t k, int lane, uint32_t* base, uint32_t& bstar_out, uint32_t& cnt_out,
                                           uint32_t& maxpop_out)
{
    constexpr int PER = kScanBins / 64;
    uint32_t h[PER];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        h[i] = hist[lane * PER + i];
        s += h[i];
    }
    uint32_t incl = s; // suffix sum over lanes >= lane
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = static_cast<uint32_t>(__shfl_down(static_cast<int>(incl), d, 64));
        if (lane + d < 64) incl += t;
    }
    uint32_t acc = incl - s;
#pragma unroll
    for (int i = PER - 1; i >= 0; i--) {
        base[lane * PER + i] = acc;
        acc += h[i];
    }
    uint32_t bstar, cnt;
    threshold_from_counts<PER>(h, s, k, lane, bstar, cnt);
    uint32_t mx = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) mx = (static_cast<uint32_t>(lane * PER + i) >= bstar && h[i] > mx) ? h[i] : mx;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t t = static_cast<uint32_t>(__shfl_xor(static_cast<int>(mx), d, 64));
        mx = t > mx ? t : mx;
    }
    bstar_out = bstar;
    cnt_out = cnt;
    maxpop_out = mx;
}

} // namespace
} // namespace gsim
