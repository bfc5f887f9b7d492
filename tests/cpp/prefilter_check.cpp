// prefilter_check.cpp -- exhaustive host-side proof test of the matrix-core pass's pre-filter.
//
// Includes the PRODUCT's arithmetic (gpusimilarity_amd/csrc/gsim_prefilter.h, the very functions the
// kernel calls) and checks it against the exact test of the kernel's drain_stage, restated here with
// the oracle's score (gso_score_one, pinned bit for bit to the device's score_of by
// test_score_arithmetic_bit_exact):
//
//   without a cutoff: a pair reaches the candidate list iff batch_bin(score) >= tau; the filter ran
//       with the constants of level tau / 512.  For every (a, b, c) and every tau <= batch_bin(score)
//       both filter stages must pass.
//   with a cutoff:    a pair is counted and considered iff score >= cutoff; the filter ran with the
//       constants of level cutoff (1 - 2^-21).  Checked for a grid of cutoffs and, for every triple,
//       for cutoff == its own score (the tightest case).
//
// a = popc(query), b = popc(row) in 0..fp_bits, c = popc(q & row) in 0..min(a, b).
// usage: prefilter_check <fp_bits> <metric 0|1> <alpha> <beta> <a_step> <full_tau_every> <threads>
// prints: checked <n> violations <v> [first: ...]
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "../../gpusimilarity_amd/csrc/gsim_prefilter.h"
extern "C" {
#include "../../oracle/gsim_oracle.h"
}

namespace
{
// gsim_device_common.h batch_bin
uint32_t batch_bin(float s)
{
    const float t = fminf(fmaxf(s, 0.0f), 1.0f) * 512.0f;
    const uint32_t b = static_cast<uint32_t>(t);
    return b < 512u ? b : 511u;
}

struct First {
    std::mutex m;
    bool set = false;
    char text[256];
};

float g_sabotage = 1.0f;

bool passes(gsim::PrefilterConstants k, float c, float b)
{
    if (g_sabotage != 1.0f && k.ka < 1.0e38f) { // negative control: pretend the bound were tighter
        k.ka *= g_sabotage;
        k.kb *= g_sabotage;
        if (k.u != 0.0f) {
            k.u = 1.0f / k.kb;
            k.v = (0.05f - k.ka) * k.u;
        }
    }
    return gsim::prefilter_tile_term_passes(k, c, b) && gsim::prefilter_pair_passes(k, c, b);
}
} // namespace

int main(int argc, char** argv)
{
    if (argc < 8) {
        std::fprintf(stderr, "usage: %s fp_bits metric alpha beta a_step full_tau_every threads\n", argv[0]);
        return 2;
    }
    const uint32_t bits = static_cast<uint32_t>(std::atoi(argv[1]));
    const int metric = std::atoi(argv[2]);
    const float alpha = static_cast<float>(std::atof(argv[3])), beta = static_cast<float>(std::atof(argv[4]));
    const uint32_t a_step = static_cast<uint32_t>(std::max(1, std::atoi(argv[5])));
    const uint32_t full_every = static_cast<uint32_t>(std::max(1, std::atoi(argv[6])));
    const int nthreads = std::max(1, std::atoi(argv[7]));
    const float sabotage = argc > 8 ? static_cast<float>(std::atof(argv[8])) : 1.0f; // > 1: an over-eager filter (the test of the test)
    const bool tv = metric == 1;
    g_sabotage = sabotage;
    std::vector<float> cutoffs;
    for (int i = 1; i <= 40; i++) cutoffs.push_back(static_cast<float>(i) * 0.025f);
    for (float x : {1e-6f, 0.001f, 0.3333333f, 0.6666667f, 0.9999999f, 1.0f, 1.0000001f}) cutoffs.push_back(x);

    std::atomic<unsigned long long> checked{0}, violations{0};
    First first;
    std::vector<uint32_t> as;
    for (uint32_t a = 0; a <= bits; a += a_step) as.push_back(a);
    if (as.back() != bits) as.push_back(bits);
    std::atomic<size_t> next{0};
    auto report = [&](const char* mode, uint32_t a, uint32_t b, uint32_t c, float s, float level, const gsim::PrefilterConstants& k) {
        violations++;
        std::lock_guard<std::mutex> g(first.m);
        if (first.set) return;
        first.set = true;
        std::snprintf(first.text, sizeof(first.text), "%s a=%u b=%u c=%u score=%.9g level=%.9g ka=%.9g kb=%.9g u=%.9g v=%.9g", mode, a, b,
                      c, s, level, k.ka, k.kb, k.u, k.v);
    };
    auto work = [&] {
        std::vector<gsim::PrefilterConstants> ktau(512), kcut(cutoffs.size());
        unsigned long long n = 0;
        for (;;) {
            const size_t ai = next++;
            if (ai >= as.size()) break;
            const uint32_t a = as[ai];
            for (uint32_t t = 0; t < 512; t++)
                ktau[t] = gsim::prefilter_constants(tv, alpha, beta, a, gsim::prefilter_level(false, 0.f, t), true);
            for (size_t i = 0; i < cutoffs.size(); i++)
                kcut[i] = gsim::prefilter_constants(tv, alpha, beta, a, gsim::prefilter_level(true, cutoffs[i], 0), true);
            const bool full = (ai % full_every) == 0;
            for (uint32_t b = 0; b <= bits; b++) {
                const float bf = static_cast<float>(b);
                const uint32_t cmax = a < b ? a : b;
                for (uint32_t c = 0; c <= cmax; c++) {
                    const float cf = static_cast<float>(c);
                    float s = gso_score_one(metric, alpha, beta, a, b, c);
                    // (a) no cutoff (cutoff <= 0): score = score >= cutoff ? score : 0 (NaN -> 0)
                    const float s0 = s >= 0.0f ? s : 0.0f;
                    const uint32_t bin = batch_bin(s0);
                    if (full) {
                        for (uint32_t t = 0; t <= bin; t++) {
                            n++;
                            if (!passes(ktau[t], cf, bf)) report("tau", a, b, c, s0, gsim::prefilter_level(false, 0.f, t), ktau[t]);
                        }
                    } else {
                        const uint32_t ts[4] = {bin, bin ? bin - 1 : 0, bin / 2, bin ? 1u : 0u};
                        for (uint32_t t : ts) {
                            n++;
                            if (!passes(ktau[t], cf, bf)) report("tau", a, b, c, s0, gsim::prefilter_level(false, 0.f, t), ktau[t]);
                        }
                    }
                    // (b) cutoffs of the grid
                    for (size_t i = 0; i < cutoffs.size(); i++) {
                        if (!(s >= cutoffs[i])) continue;
                        n++;
                        if (!passes(kcut[i], cf, bf)) report("cutoff", a, b, c, s, cutoffs[i], kcut[i]);
                    }
                    // (d) the VALU pass's tests on c against multiples of the denominator
                    {
                        const float den = gso_score_den(metric, alpha, beta, a, b, c);
                        const bool same = (cf / den == s) || (s != s && cf / den != cf / den);
                        if (!same) report("den", a, b, c, s, den, ktau[0]);
                        const uint32_t tlo = full ? 0u : (bin > 2 ? bin - 2 : 0u);
                        for (uint32_t t = tlo; t <= bin; t++) { // the exact test accepts bin >= t: the filter must not reject
                            n++;
                            if (gsim::valu_filter_rejects(gsim::valu_filter_level(t), cf, den)) report("valu-tau", a, b, c, s0, static_cast<float>(t), ktau[t]);
                        }
                        auto check_cut = [&](float cutoff) {
                            const bool kept = s >= cutoff && s != 0.0f; // apply_cutoff(s) != 0
                            n++;
                            if (gsim::valu_surely_kept(gsim::valu_cutoff_hi(cutoff), cf, den, c) && !kept) report("valu-surely", a, b, c, s, cutoff, ktau[0]);
                            if (gsim::valu_surely_not_kept(gsim::valu_cutoff_lo(cutoff), cf, den, c) && kept) report("valu-surely-not", a, b, c, s, cutoff, ktau[0]);
                        };
                        for (float cutoff : cutoffs) check_cut(cutoff);
                        if (s > 0.0f) {
                            check_cut(s);
                            check_cut(nextafterf(s, 2.0f));
                            check_cut(nextafterf(s, 0.0f));
                        }
                    }
                    // (e) the band of the dense-cutoff route: "surely kept" / "surely not kept" must agree with the exact test
                    {
                        auto check_band = [&](float cutoff) {
                            const gsim::CutoffBand kb = gsim::cutoff_band(tv, alpha, beta, a, cutoff, true);
                            if (!kb.on) return;
                            const bool kept = s >= cutoff && s != 0.0f;
                            n++;
                            gsim::CutoffBand kk = kb;
                            if (g_sabotage != 1.0f) { // negative control: a band that is too narrow by that factor
                                kk.us *= g_sabotage, kk.vs *= g_sabotage;
                            }
                            if (gsim::band_surely_kept(kk, cf, bf) && !kept) report("band-surely", a, b, c, s, cutoff, ktau[0]);
                            if (gsim::band_surely_not_kept(kb, cf, bf) && kept) report("band-surely-not", a, b, c, s, cutoff, ktau[0]);
                        };
                        for (float cutoff : cutoffs) check_band(cutoff);
                        if (s > 0.0f) {
                            check_band(s);
                            check_band(nextafterf(s, 2.0f));
                            check_band(nextafterf(s, 0.0f));
                        }
                    }
                    // (c) cutoff == this pair's own score
                    if (s > 0.0f) {
                        const gsim::PrefilterConstants k = gsim::prefilter_constants(tv, alpha, beta, a, gsim::prefilter_level(true, s, 0), true);
                        n++;
                        if (!passes(k, cf, bf)) report("cutoff=score", a, b, c, s, s, k);
                    }
                }
            }
        }
        checked += n;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; t++) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    std::printf("checked %llu violations %llu%s%s\n", checked.load(), violations.load(), first.set ? " first: " : "",
                first.set ? first.text : "");
    return violations.load() ? 1 : 0;
}
