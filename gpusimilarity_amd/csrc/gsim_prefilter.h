// gsim_prefilter.h -- the division-free pre-filter of the matrix-core multi-query pass
// (gsim_batch_mfma.hip), as plain host + device functions.
//
// The contraction leaves c = popc(q & row) for 32 x 32 pairs in registers; scoring every pair with
// the reference's f32 divide (fingerprintdb_cuda.cu:89-101) would cost more than the contraction.
// "score >= T" is linear in the counts,
//
//     c / (al (a - c) + be (b - c) + c) >= T   <=>   c >= f al a + f be b,   f = T / (1 - T (1 - al - be))
//
// (a = popc(query), b = popc(row); Tanimoto is al = be = 1), so a pair can only pass the exact test
// if it passes   c >= ka + kb b   with ka = f al a, kb = f be.  The functions below make that bound
// CONSERVATIVE under f32 rounding (f is scaled down by 2^-12, the comparisons carry 0.05 / 0.01 of
// absolute slack) -- a pair the filter rejects is never scored, so an over-eager filter would lose
// hits silently.  tests/cpp/prefilter_check.cpp includes THIS header and checks, over every
// (a, b, c) of 256...2048-bit fingerprints, every threshold bin and a grid of cutoffs and weights,
// that the exact test never accepts a pair the filter rejects; tests/test_gpu_parity.py compares the
// constants computed on the device with the host's.
//
// Plain arithmetic only: IEEE +, *, / and fmaf, no contraction (both compilers run with
// -ffp-contract=off), so host and device agree bit for bit.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GSIM_HD __host__ __device__ __forceinline__
#else
#define GSIM_HD inline
#endif

namespace gsim
{

constexpr int kPrefilterBins = 512; // == kBBins (gsim_device.h)

struct PrefilterConstants {
    float ka, kb; // pair test:  c >= fma(kb, b, ka)
    float u, v;   // tile test:  fma(c, u, v) >= b - 0.01   (the same bound solved for b)
};

// Level of the filter for a query: with a cutoff every pair at or above it has to reach the exact
// path (it is counted: `approx`), so the level is the cutoff, whatever the top-k threshold; without
// one it is the lower edge of the query's threshold bin.
GSIM_HD float prefilter_level(bool has_cutoff, float cutoff, uint32_t tau_bin)
{
    return has_cutoff ? cutoff * (1.0f - 4.76837158203125e-7f) // cutoff (1 - 2^-21)
                      : static_cast<float>(tau_bin) * (1.0f / kPrefilterBins);
}

// T <= 0, exotic weights or an ill-conditioned bound switch the filter off (everything passes);
// `valid` = false is a padding query of the last tile: nothing passes.
GSIM_HD PrefilterConstants prefilter_constants(bool tversky, float alpha, float beta, uint32_t qa, float T, bool valid)
{
    PrefilterConstants k;
    k.ka = 0.0f;
    k.kb = 0.0f;
    k.u = 0.0f;
    k.v = 3.0e38f; // tile test always true
    if (!valid) {
        k.ka = 3.0e38f;
        k.v = -3.0e38f;
        return k;
    }
    const float al = tversky ? alpha : 1.0f;
    const float be = tversky ? beta : 1.0f;
    const float D = 1.0f - T * (1.0f - al - be);
    if (!(T > 0.0f) || !(al >= 0.0f) || !(be >= 0.0f)) return k;
    if (T > 1.0f) { // scores never exceed 1 with non-negative weights
        k.ka = 3.0e38f;
        k.v = -3.0e38f;
        return k;
    }
    if (!(D > 0.05f)) return k;
    const float f = T / D * (1.0f - 0.000244140625f); // (1 - 2^-12)
    k.ka = f * al * static_cast<float>(qa);
    k.kb = f * be;
    // c >= ka + kb b - 0.05  <=>  c u + v >= b  with u = 1 / kb, v = (0.05 - ka) u; a vanishing kb
    // (be = 0) or a huge ka (beyond: the rounding of c u + v could exceed the slack) leave the tile
    // test open and the decision to the pair test
    if (k.kb > 1.0e-6f && k.ka < 1.0e4f) {
        k.u = 1.0f / k.kb;
        k.v = (0.05f - k.ka) * k.u;
    }
    return k;
}

// c, b: the exact integer counts as floats (< 2^24)
GSIM_HD bool prefilter_tile_term_passes(const PrefilterConstants& k, float c, float b)
{
    return fmaf(c, k.u, k.v) >= b - 0.01f;
}

GSIM_HD bool prefilter_pair_passes(const PrefilterConstants& k, float c, float b)
{
    return c >= fmaf(k.kb, b, k.ka);
}

// ---- dense cutoffs on the matrix cores ---------------------------------------------------------
// `approx` counts the rows with score >= cutoff: when the cutoff keeps a sizeable part of the table those rows
// cannot all go through the exact path.  The count is linear in the counts as well, so the kernel decides almost
// every pair with two fused multiply-adds and counts in registers:
//     surely kept       fmaf(c, us, vs) >= b + 0.01      =>  RN(c / den) >= cutoff (and c != 0)
//     surely not kept   fmaf(c, un, vn) <  b - 0.01      =>  not kept
// with (us, vs) / (un, vn) the bound solved for b at f (1 + 2^-18) / f (1 - 2^-18).  Budget, in units of b: the
// rounding of f, of its products and of the fma stays below 0.007 for a + b <= 4096 while D > 0.05 (the division
// by D amplifies one rounding of D at most twenty times), the rounding of the reference's own denominator and
// quotient is worth 0.0013; the margins give 0.0156 + 0.01 on either side.  Only the pairs in between -- the
// bound passes within ~0.005 counts of an integer -- go to the exact path.  prefilter_check.cpp checks both
// implications over every (a, b, c), a grid of cutoffs and every pair's own score (and its neighbours) as cutoff.
struct CutoffBand {
    float us, vs, un, vn;
    bool on; // false: no usable band for these weights / this cutoff (the VALU pass takes dense cutoffs then)
};

GSIM_HD CutoffBand cutoff_band(bool tversky, float alpha, float beta, uint32_t qa, float cutoff, bool valid)
{
    CutoffBand k;
    k.us = 0.0f;
    k.vs = -3.0e38f; // surely kept: never
    k.un = 0.0f;
    k.vn = valid ? 3.0e38f : -3.0e38f; // surely not kept: never (padding query: always)
    k.on = false;
    const float al = tversky ? alpha : 1.0f;
    const float be = tversky ? beta : 1.0f;
    const float T = cutoff;
    const float D = 1.0f - T * (1.0f - al - be);
    if (!(T > 0.0f) || !(T <= 1.0f) || !(al >= 0.0f) || !(be > 1.0e-3f) || !(al <= 16.0f) || !(be <= 16.0f) || !(D > 0.05f)) return k;
    const float f = T / D;
    const float fs = f * (1.0f + 3.814697265625e-6f), fn = f * (1.0f - 3.814697265625e-6f); // (1 +- 2^-18)
    const float kas = fs * al * static_cast<float>(qa), kbs = fs * be;
    const float kan = fn * al * static_cast<float>(qa), kbn = fn * be;
    if (!(kbs > 1.0e-4f) || !(kbn > 1.0e-4f) || !(kas < 1.0e4f)) return k;
    k.on = true;
    if (!valid) return k;
    k.us = 1.0f / kbs;
    k.vs = -kas * k.us;
    k.un = 1.0f / kbn;
    k.vn = -kan * k.un;
    return k;
}

GSIM_HD bool band_surely_kept(const CutoffBand& k, float c, float b) { return fmaf(c, k.us, k.vs) >= b + 0.01f; }
GSIM_HD bool band_surely_not_kept(const CutoffBand& k, float c, float b) { return fmaf(c, k.un, k.vn) < b - 0.01f; }

// ---- VALU multi-query pass (gsim_batch.hip) ---------------------------------------------------
// There the denominator of the score is at hand (den = score_den(...), the f32 value the exact
// divide uses), so the tests are on c against a multiple of den:
//   rejects:   c < RN(T- den), T- = tau/512 (1 - 2^-21)   =>  RN(c / den) < tau/512, i.e. bin < tau
//   with a cutoff the kept-count needs "RN(c / den) >= cutoff" for EVERY pair; it is decided without
//   the divide unless c / den is within 2^-21 of the cutoff:
//   surely kept:      den > 0, c != 0, c >= RN(cutoff (1 + 2^-21) den)   =>  RN(c / den) >= cutoff
//   surely not kept:  c == 0 or c < RN(cutoff (1 - 2^-21) den)           =>  not (RN(c / den) >= cutoff and != 0)
GSIM_HD float valu_filter_level(uint32_t tau_bin)
{
    return (static_cast<float>(tau_bin) * (1.0f / kPrefilterBins)) * (1.0f - 4.76837158203125e-7f);
}

GSIM_HD bool valu_filter_rejects(float tm, float c, float den)
{
    return c < tm * den;
}

GSIM_HD float valu_cutoff_hi(float cutoff) { return cutoff * (1.0f + 4.76837158203125e-7f); }
GSIM_HD float valu_cutoff_lo(float cutoff) { return cutoff * (1.0f - 4.76837158203125e-7f); }

GSIM_HD bool valu_surely_kept(float cut_hi, float c, float den, uint32_t ci)
{
    return den > 0.0f && c >= cut_hi * den && ci != 0;
}

GSIM_HD bool valu_surely_not_kept(float cut_lo, float c, float den, uint32_t ci)
{
    return c < cut_lo * den || ci == 0;
}

} // namespace gsim
