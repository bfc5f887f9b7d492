// scan_ablation_probe.hip -- what each stage of the streaming scan costs, measured OUTSIDE the product: the production
// streaming loop (csrc/gsim_scan_inl.h: scan_rows / reduce_chunk, unchanged) driven by probe-side filter policies.
// (Until round 4 these variants lived as `#if GSIM_ABLATE` blocks inside the production kernels.)
//   stage 1  loads only                       -> scripts/hbm_read_probe.hip (same access pattern, one XOR per dword)
//   stage 3  + popcounts, DPP reduction, the reference's f32 divide     (SinkFilter: the score is consumed, nothing else)
//   stage 4  + the filter's fast path: cutoff, order key, compare with a fixed threshold, ballot   (BallotFilter)
//   stage 5  + emission into an LDS store at a fixed threshold (no threshold exchange)             (StoreFilter)
// The full kernels (threshold exchange, publish, select) are timed by bench.py.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Igpusimilarity_amd/csrc scripts/scan_ablation_probe.hip -o /tmp/abl && /tmp/abl
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "gsim_scan_inl.h"

using namespace gsim;

struct SinkFilter {
    static constexpr bool kFused = false;
    template <int LPR> __device__ __forceinline__ void offer_counts(bool active, uint32_t row, uint32_t val, const ScanArgs& a, int lane) { offer_scored(*this, active, row, val, a, lane); }
    __device__ __forceinline__ uint32_t load_gtau() const { return 0u; }
    __device__ __forceinline__ void refresh(uint32_t, int) {}
    __device__ __forceinline__ void checkpoint(uint32_t, int) {}
    __device__ __forceinline__ void offer(bool active, uint32_t, float s, uint32_t, int) { asm volatile("" ::"v"(s), "v"(active)); }
};

struct BallotFilter {
    static constexpr bool kFused = false;
    template <int LPR> __device__ __forceinline__ void offer_counts(bool active, uint32_t row, uint32_t val, const ScanArgs& a, int lane) { offer_scored(*this, active, row, val, a, lane); }
    uint32_t tau, hits;
    __device__ __forceinline__ uint32_t load_gtau() const { return 0u; }
    __device__ __forceinline__ void refresh(uint32_t, int) {}
    __device__ __forceinline__ void checkpoint(uint32_t, int) {}
    __device__ __forceinline__ void offer(bool active, uint32_t, float raw, uint32_t, int)
    {
        const float s = apply_cutoff(raw, 0.0f);
        const u64 m = __ballot(active && order_key(s) >= tau);
        hits += static_cast<uint32_t>(__popcll(m));
    }
};

struct StoreFilter {
    static constexpr bool kFused = false;
    template <int LPR> __device__ __forceinline__ void offer_counts(bool active, uint32_t row, uint32_t val, const ScanArgs& a, int lane) { offer_scored(*this, active, row, val, a, lane); }
    u64* skey;
    uint32_t* scb;
    uint32_t tau, staged;
    __device__ __forceinline__ uint32_t load_gtau() const { return 0u; }
    __device__ __forceinline__ void refresh(uint32_t, int) {}
    __device__ __forceinline__ void checkpoint(uint32_t, int) {}
    __device__ __forceinline__ void offer(bool active, uint32_t row, float raw, uint32_t cb, int)
    {
        const float s = apply_cutoff(raw, 0.0f);
        const uint32_t okey = order_key(s);
        const bool cand = active && okey >= tau;
        const u64 m = __ballot(cand);
        if (m == 0) return;
        if (cand) {
            const uint32_t slot = (staged + lane_rank(m)) & 2047u;
            skey[slot] = (static_cast<u64>(okey) << 32) | static_cast<u64>(~row);
            scb[slot] = cb;
        }
        staged += static_cast<uint32_t>(__popcll(m));
    }
};

template <int STAGE> __global__ __launch_bounds__(256) void probe(ScanArgs a, ScanGeometry g, uint32_t tau, uint32_t* out)
{
    __shared__ u64 s_key[4][2048];
    __shared__ uint32_t s_cb[4][2048];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
    const u32x4 q = reinterpret_cast<const u32x4*>(a.query)[lane % 8];
    if (STAGE == 3) {
        SinkFilter f;
        scan_rows<8, 8>(a, g, f, q, w, lane);
    } else if (STAGE == 4) {
        BallotFilter f{tau, 0u};
        scan_rows<8, 8>(a, g, f, q, w, lane);
        if (f.hits == 0xFFFFFFFFu) out[0] = 1;
    } else {
        StoreFilter f{s_key[wv], s_cb[wv], tau, 0u};
        scan_rows<8, 8>(a, g, f, q, w, lane);
        if (lane == 0) atomicAdd(&out[1], f.staged);
    }
}

template <int STAGE> double run(const ScanArgs& a, const ScanGeometry& g, uint32_t tau, uint32_t* out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double best = 1e30;
    for (int it = 0; it < 12; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<STAGE>), dim3(g.nwaves / 4), dim3(256), 0, 0, a, g, tau, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2 && ms < best) best = ms;
    }
    return a.nrows * 128.0 / (best * 1e-3) / 1e9;
}

int main()
{
    const uint64_t nrows = 100000000ull;
    void* db;
    uint32_t *q, *out;
    if (hipMalloc(&db, nrows * 128) != hipSuccess) return 1;
    hipMalloc(&q, 128);
    hipMalloc(&out, 64);
    hipMemset(db, 0x11, nrows * 128); // two bits per byte: popcount 256, every row scores the same
    hipMemset(q, 0x33, 128);
    hipMemset(out, 0, 64);
    ScanArgs a{};
    a.rows = db, a.nrows = nrows, a.W = 32, a.query = q, a.qpop = 512, a.k = 1000, a.metric = GSIM_METRIC_TANIMOTO;
    ScanGeometry g{};
    g.lanes_per_row = 8, g.unroll = 8, g.chunk_rows = 64, g.nwaves = 1024, g.nchunks = (nrows + 63) / 64;
    const uint32_t never = 0xFFFFFFFFu;
    printf("stage 3 (+popcounts, DPP, divide)        %7.1f GB/s\n", run<3>(a, g, never, out));
    printf("stage 4 (+filter fast path, no emission) %7.1f GB/s\n", run<4>(a, g, never, out));
    printf("stage 5 (+LDS emission, none passes)     %7.1f GB/s\n", run<5>(a, g, never, out));
    printf("stage 5 (+LDS emission, every row passes) %6.1f GB/s\n", run<5>(a, g, 0u, out));
    return 0;
}
