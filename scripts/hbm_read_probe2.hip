// hbm_read_probe2.hip -- read ceiling vs the chunk->wave mapping and the data (random bits vs constant).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

// MAP 0: chunk = round*nwaves + w (scan_kernel)   1: XCD-contiguous inside a round
// MAP 2: every wave streams its own contiguous region   3: rounds of 2 consecutive chunks per wave
template <int MAP> __global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ db, u64 nchunks, unsigned nwaves, unsigned* out)
{
    constexpr int U = 8;
    const int lane = threadIdx.x & 63;
    const unsigned wib = threadIdx.x >> 6, b = blockIdx.x;
    unsigned w = b * 4 + wib;
    if (MAP == 1) w = (b % 8) * (nwaves / 8) + (b / 8) * 4 + wib;
    w = __builtin_amdgcn_readfirstlane(w);
    const u64 rounds = nchunks / nwaves;
    u32x4 acc = {0, 0, 0, 0};
    auto chunk_of = [&](u64 r) -> u64 {
        if (MAP == 2) return (u64) w * rounds + r;
        if (MAP == 3) return (r / 2) * (2ull * nwaves) + 2ull * w + (r & 1);
        return r * nwaves + w;
    };
    u32x4 nxt[U];
    const u32x4* p0 = db + chunk_of(0) * (U * 64) + lane;
#pragma unroll
    for (int j = 0; j < U; j++) nxt[j] = __builtin_nontemporal_load(p0 + j * 64);
    for (u64 r = 0; r < rounds; r++) {
        u32x4 d[U];
#pragma unroll
        for (int j = 0; j < U; j++) d[j] = nxt[j];
        const u64 rn = r + 1 < rounds ? r + 1 : r;
        const u32x4* p = db + chunk_of(rn) * (U * 64) + lane;
#pragma unroll
        for (int j = 0; j < U; j++) nxt[j] = __builtin_nontemporal_load(p + j * 64);
#pragma unroll
        for (int j = 0; j < U; j++) acc ^= d[j];
    }
    const unsigned x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x12345678u) out[0] = x;
}
__global__ void fill(unsigned* p, u64 n, int random)
{
    for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
        u64 z = i * 0x9E3779B97F4A7C15ull + 12345; z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
        const unsigned v = (unsigned) z & (unsigned) (z >> 32);
        p[i] = random ? (v & (unsigned) (z >> 16)) : 0x5a5a5a5au;
    }
}
template <int MAP> double run(const u32x4* db, size_t bytes, unsigned* out)
{
    const unsigned nwaves = 1024;
    const u64 nchunks = bytes / 8192 / nwaves * nwaves;
    hipEvent_t a, b; (void) hipEventCreate(&a); (void) hipEventCreate(&b);
    double best = 1e30;
    for (int it = 0; it < 10; it++) {
        (void) hipEventRecord(a);
        hipLaunchKernelGGL((probe<MAP>), dim3(256), dim3(256), 0, 0, db, nchunks, nwaves, out);
        (void) hipEventRecord(b); (void) hipEventSynchronize(b);
        float ms; (void) hipEventElapsedTime(&ms, a, b);
        if (it >= 2 && ms < best) best = ms;
    }
    return nchunks * 8192.0 / (best * 1e-3) / 1e9;
}
int main()
{
    const size_t bytes = 12800000000ull;
    void* db; unsigned* out;
    if (hipMalloc(&db, bytes) != hipSuccess) return 1;
    (void) hipMalloc(&out, 64);
    for (int random = 0; random < 2; random++) {
        hipLaunchKernelGGL(fill, dim3(65536), dim3(256), 0, 0, (unsigned*) db, (u64) (bytes / 4), random);
        (void) hipDeviceSynchronize();
        const u32x4* p = (const u32x4*) db;
        printf("%s data: round-robin %7.1f | XCD-contiguous %7.1f | private streams %7.1f | 2-chunk runs %7.1f GB/s\n",
               random ? "sparse random" : "constant     ", run<0>(p, bytes, out), run<1>(p, bytes, out), run<2>(p, bytes, out), run<3>(p, bytes, out));
    }
    return 0;
}
