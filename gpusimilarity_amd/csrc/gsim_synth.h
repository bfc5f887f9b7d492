// gsim_synth.h -- the counter-based synthetic fingerprint generators, as plain host + device
// functions: generate_kernel / generate_morgan_kernel (gsim_select.hip) fill a table in HBM with
// them, gsim_synth_row (capi_lifecycle.cpp) hands single rows to host callers (benchmark queries are
// rows of the table).  Integer arithmetic only, so host and device agree bit for bit; the test
// oracle has its own restatement (oracle/gsim_oracle.c gso_synth_*) and the parity tests compare
// the three.
//
// GSIM_SYNTH_SPARSE / GSIM_SYNTH_DENSE: word (row i, word j) = four (one)
// SplitMix64 streams ANDed, bit density 1/16 (1/2): i.i.d. bits, no structure.
//
// GSIM_SYNTH_MORGAN: the shape of what the reference's numbers are quoted on -- 1024-bit Morgan
// r=2 fingerprints (python/gpusim_utils.py:21,55-66).  Its fixture test/small.fsim has popcounts
// 20..53 (mean 34.5), a dozen bits set in more than half of the rows and a long tail of rare ones,
// mean pairwise Tanimoto 0.155; real libraries add series of analogs and exact duplicates.  A row is
//   16 "common" bits drawn per scaffold with the fixture's frequencies
// + 10..26 scaffold bits (up to two dropped per member) + 2..13 member bits,
// rows belong to a contiguous series of 16/64/256/1024 analogs (half of them), to one of 65 536
// table-wide scaffolds (a quarter) or to nobody (a quarter), and members are drawn from a space
// small enough that ~3 % of the rows are exact duplicates of another row.  Scores against such a
// table are coarse (ratios of small integers): a top-1000 holds 40..100 distinct values and the
// k-th one is shared by tens to hundreds of rows.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define GSIM_SHD __host__ __device__ __forceinline__
#else
#define GSIM_SHD inline
#endif

namespace gsim
{

GSIM_SHD uint64_t synth_splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// n-th output of the SplitMix64 stream seeded with `seed`
GSIM_SHD uint64_t synth_stream_at(uint64_t seed, uint64_t n)
{
    return synth_splitmix64(seed + n * 0x9E3779B97F4A7C15ull);
}

// kinds SPARSE (dense = false) and DENSE: one word
GSIM_SHD uint32_t synth_word_iid(uint64_t seed, bool dense, uint64_t ctr)
{
    if (dense) return static_cast<uint32_t>(synth_stream_at(seed, ctr));
    const uint64_t h0 = synth_stream_at(seed, 2 * ctr), h1 = synth_stream_at(seed, 2 * ctr + 1);
    return static_cast<uint32_t>(h0) & static_cast<uint32_t>(h0 >> 32) & static_cast<uint32_t>(h1) &
           static_cast<uint32_t>(h1 >> 32);
}

GSIM_SHD uint64_t synth_hash2(uint64_t seed, uint64_t tag, uint64_t a, uint64_t b)
{
    return synth_splitmix64(synth_splitmix64(seed + tag * 0xD1B54A32D192ED03ull + a * 0x9E3779B97F4A7C15ull) +
                            b * 0x9E3779B97F4A7C15ull);
}

// bit position of a scaffold / member feature: the product of two uniforms (density ~ -ln u: a
// long tail of rare bits), then the (word, bit) transposition spreads the frequent ones over the words
GSIM_SHD uint32_t synth_morgan_pos(uint64_t x, uint32_t W)
{
    const uint32_t u = static_cast<uint32_t>(((x & 0xFFFFu) * ((x >> 16) & 0xFFFFu)) >> 16);
    const uint32_t pos = static_cast<uint32_t>((static_cast<uint64_t>(u) * (W * 32u)) >> 16);
    return (pos & 31u) * W + (pos >> 5);
}

// frequency (x / 256) of the sixteen common bits: test/small.fsim's most frequent bits
GSIM_SHD uint32_t synth_morgan_common(uint32_t i)
{
    // two packed tables (no indexable constant array on the device side)
    const uint64_t lo = 0xA1A6AEC8DCECFDFDull; // 253 253 236 220 200 174 166 161
    const uint64_t hi = 0x4A4A545A83858DA1ull; // 161 141 133 131  90  84  74  74
    return static_cast<uint32_t>(((i < 8 ? lo : hi) >> (8 * (i & 7u))) & 0xFFu);
}

// One row of the GSIM_SYNTH_MORGAN table into out[0..W) (any writable memory: LDS on the device).
GSIM_SHD void synth_row_morgan(uint32_t* out, uint64_t seed, uint64_t row, uint32_t W)
{
    const uint32_t nbits = W * 32u;
    for (uint32_t j = 0; j < W; j++) out[j] = 0;
    const uint64_t rh = synth_hash2(seed, 1, row, 0);
    const uint64_t sh = synth_hash2(seed, 2, row >> 10, 0);
    uint64_t sid, mspace;
    const uint32_t cls = static_cast<uint32_t>(rh & 3u);
    if (cls == 0) { // table-wide scaffold
        sid = (1ull << 62) | ((rh >> 8) & 0xFFFFu);
        mspace = 1u << 14;
    } else if (cls == 1) { // singleton
        sid = (2ull << 62) | row;
        mspace = 1;
    } else { // a contiguous series of S = 16, 64, 256 or 1024 rows
        const uint32_t lg = 4u + 2u * static_cast<uint32_t>(sh & 3u);
        sid = (row >> lg) | (static_cast<uint64_t>(lg) << 56);
        mspace = 4ull << lg;
    }
    const uint64_t m = (rh >> 24) % mspace;
    const uint64_t kh = synth_hash2(seed, 3, sid, 0);
    const uint64_t mh = synth_hash2(seed, 6, sid, m);
    for (uint32_t i = 0; i < 16; i++) {
        const uint64_t c = synth_hash2(seed, 4, sid, i >> 3);
        if (((c >> (8 * (i & 7u))) & 0xFFu) < synth_morgan_common(i)) {
            const uint32_t p = (i * 67u + 5u) % nbits;
            out[p >> 5] |= 1u << (p & 31u);
        }
    }
    const uint32_t ps = 10u + static_cast<uint32_t>(kh % 17u);
    const uint32_t ndrop = static_cast<uint32_t>(mh % 3u);
    const uint32_t d0 = static_cast<uint32_t>((mh >> 8) & 0xFFu) % ps, d1 = static_cast<uint32_t>((mh >> 16) & 0xFFu) % ps;
    for (uint32_t j = 0; j < ps; j++) {
        if ((ndrop >= 1 && j == d0) || (ndrop >= 2 && j == d1)) continue;
        const uint32_t p = synth_morgan_pos(synth_hash2(seed, 5, sid, j), W);
        out[p >> 5] |= 1u << (p & 31u);
    }
    const uint32_t na = 2u + static_cast<uint32_t>((mh >> 32) % 12u);
    for (uint32_t j = 0; j < na; j++) {
        const uint32_t p = synth_morgan_pos(synth_splitmix64(mh + (j + 1) * 0x9E3779B97F4A7C15ull), W);
        out[p >> 5] |= 1u << (p & 31u);
    }
}

} // namespace gsim
