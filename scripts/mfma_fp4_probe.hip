// mfma_fp4_probe.hip -- binary inner products on the MX-FP4 matrix cores of gfx950.
// popc(q & d) over K bits is a 0/1 contraction; v_mfma_scale_f32_32x32x64_f8f6f4 with both
// operands in FP4 (E2M1) does 32x32x64 of them per instruction.  A packed word x is turned into
// FP4 operands with ONE v_and per dword: x & 0x11111111 -> nibbles {0, 0.5}, & 0x22222222 ->
// {0, 1.0}, & 0x44444444 -> {0, 2.0}; the E8M0 block scales (2^1, 2^0, 2^-1) bring each class
// back to {0, 1}.  The 0x88888888 class is the FP4 sign bit (-0), so it is shifted down first.
// Which bit lands in which k slot does not matter as long as A and B use the same map.
// Part 1 checks the lane->row/col assumptions against the host; part 2 measures issue rate
// with and without VALU work in the MFMA shadow.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v8i cls(u32x4 x, unsigned m)
{
    return v8i{(int) (x.x & m), (int) (x.y & m), (int) (x.z & m), (int) (x.w & m), 0, 0, 0, 0};
}
__device__ __forceinline__ v16f dot256(u32x4 a, u32x4 b, v16f acc)
{
    const int s1 = 0x80808080, s0 = 0x7F7F7F7F, sm = 0x7E7E7E7E;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cls(a, 0x11111111u), cls(b, 0x11111111u), acc, 4, 4, 0, s1, 0, s1);
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cls(a, 0x22222222u), cls(b, 0x22222222u), acc, 4, 4, 0, s0, 0, s0);
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cls(a, 0x44444444u), cls(b, 0x44444444u), acc, 4, 4, 0, sm, 0, sm);
    const u32x4 a3 = a >> 3, b3 = b >> 3;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cls(a3, 0x11111111u), cls(b3, 0x11111111u), acc, 4, 4, 0, s1, 0, s1);
    return acc;
}
// A: 32 rows x 8 words, B: 32 rows x 8 words; C[32][32] = popc(A[i] & B[j])
__global__ void check_kernel(const u32x4* A, const u32x4* B, float* C)
{
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    v16f acc = {};
    acc = dot256(A[i * 2 + h], B[i * 2 + h], acc);
    for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}
// issue-rate: NACC independent accumulators, VALU_PER independent v_and_b32 (vector operands,
// 2-cycle issue) per MFMA in its shadow
template <int NACC, int VALU_PER> __global__ __launch_bounds__(256) void rate_kernel(const u32x4* A, float* out, int iters)
{
    u32x4 a = A[threadIdx.x & 63], b = A[(threadIdx.x + 7) & 63];
    v16f acc[NACC];
    for (int n = 0; n < NACC; n++) acc[n] = v16f{};
    unsigned junk[8];
    for (int j = 0; j < 8; j++) junk[j] = threadIdx.x + j;
    unsigned y = threadIdx.x * 2654435761u;
    const int s0 = 0x7F7F7F7F;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int n = 0; n < NACC; n++) {
            acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cls(a, 0x22222222u), cls(b, 0x22222222u), acc[n], 4, 4, 0, s0, 0, s0);
#pragma unroll
            for (int v = 0; v < VALU_PER; v++) asm volatile("v_and_b32 %0, %1, %0" : "+v"(junk[v & 7]) : "v"(y));
        }
    }
    float t = 0;
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 16; r++) t += acc[n][r];
    unsigned ju = 0;
    for (int j = 0; j < 8; j++) ju ^= junk[j];
    out[blockIdx.x * 256 + threadIdx.x] = t + ju;
}
// burst pattern: BURST MFMAs back to back on ONE accumulator, then BURST * VALU_PER v_and_b32; the two
// accumulators alternate.  (An issue slot between dependent MFMAs costs far more than one between
// independent ones, so VALU work is better batched between bursts than interleaved.)
template <int BURST, int VALU_PER> __global__ __launch_bounds__(256) void burst_kernel(const u32x4* A, float* out, int iters)
{
    u32x4 a = A[threadIdx.x & 63], b = A[(threadIdx.x + 7) & 63];
    v16f acc[2];
    acc[0] = v16f{}; acc[1] = v16f{};
    unsigned junk[8];
    for (int j = 0; j < 8; j++) junk[j] = threadIdx.x + j;
    unsigned y = threadIdx.x * 2654435761u;
    const int s0 = 0x7F7F7F7F;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int n = 0; n < 2; n++) {
#pragma unroll
            for (int m = 0; m < BURST; m++)
                acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cls(a, 0x22222222u), cls(b, 0x22222222u), acc[n], 4, 4, 0, s0, 0, s0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < VALU_PER * BURST; v++) asm volatile("v_and_b32 %0, %1, %0" : "+v"(junk[v & 7]) : "v"(y));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float t = 0;
    for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) t += acc[n][r];
    unsigned ju = 0;
    for (int j = 0; j < 8; j++) ju ^= junk[j];
    out[blockIdx.x * 256 + threadIdx.x] = t + ju;
}
template <int BURST, int VALU_PER> void burst(const u32x4* A, float* out, int wpc)
{
    const int iters = 1000, blocks = 256 * wpc / 4;
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    float best = 1e30f;
    for (int t = 0; t < 4; t++) {
        (void) hipEventRecord(e0);
        hipLaunchKernelGGL((burst_kernel<BURST, VALU_PER>), dim3(blocks), dim3(256), 0, 0, A, out, iters);
        (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
        float ms; (void) hipEventElapsedTime(&ms, e0, e1);
        if (t && ms < best) best = ms;
    }
    const double mf = double(blocks) * 4 * iters * 2 * BURST;
    printf("burst %d v_and/mfma %2d waves/SIMD %d: %.1f cycles per MFMA per SIMD at 2.4 GHz\n", BURST, VALU_PER, wpc / 4,
           best * 1e-3 * 2.4e9 / (mf / 1024.0));
}
template <int NACC, int VALU_PER> void rate(const u32x4* A, float* out, int wpc)
{
    const int iters = 2000, blocks = 256 * wpc / 4;
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    float best = 1e30f;
    for (int t = 0; t < 4; t++) {
        (void) hipEventRecord(e0);
        hipLaunchKernelGGL((rate_kernel<NACC, VALU_PER>), dim3(blocks), dim3(256), 0, 0, A, out, iters);
        (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
        float ms; (void) hipEventElapsedTime(&ms, e0, e1);
        if (t && ms < best) best = ms;
    }
    const double mf = double(blocks) * 4 * iters * NACC; // wave-MFMAs
    printf("nacc %d v_and/mfma %2d waves/SIMD %d: %.1f cycles per MFMA per SIMD at 2.4 GHz (%.0f TFLOP/s-equivalent)\n", NACC, VALU_PER, wpc / 4,
           best * 1e-3 * 2.4e9 / (mf / 1024.0), mf * 2.0 * 32 * 32 * 64 / (best * 1e-3) / 1e12);
}
int main()
{
    std::vector<unsigned> hA(32 * 8), hB(32 * 8);
    srand(7);
    for (auto& w : hA) w = (unsigned) rand() * 2654435761u ^ (unsigned) rand();
    for (auto& w : hB) w = (unsigned) rand() * 40503u ^ ((unsigned) rand() << 7);
    u32x4 *dA, *dB; float *dC, *out;
    (void) hipMalloc(&dA, 1024); (void) hipMalloc(&dB, 1024); (void) hipMalloc(&dC, 4096); (void) hipMalloc(&out, 256 * 16 * 256 * 4);
    (void) hipMemcpy(dA, hA.data(), 1024, hipMemcpyHostToDevice); (void) hipMemcpy(dB, hB.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<float> hC(1024);
    (void) hipMemcpy(hC.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0, badT = 0;
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
        int c = 0;
        for (int w = 0; w < 8; w++) c += __builtin_popcount(hA[i * 8 + w] & hB[j * 8 + w]);
        if (hC[i * 32 + j] != (float) c) bad++;
        if (hC[j * 32 + i] != (float) c) badT++;
    }
    printf("check: C[a_row][b_row] mismatches %d (transposed reading: %d) of 1024; sample %.1f\n", bad, badT, hC[33]);
    for (int wpc : {4, 8}) {
        rate<2, 0>(dA, out, wpc); rate<2, 4>(dA, out, wpc); rate<2, 8>(dA, out, wpc); rate<2, 12>(dA, out, wpc);
        rate<1, 8>(dA, out, wpc); rate<4, 8>(dA, out, wpc);
        burst<4, 4>(dA, out, wpc); burst<4, 6>(dA, out, wpc); burst<4, 8>(dA, out, wpc); burst<4, 12>(dA, out, wpc);
        burst<8, 6>(dA, out, wpc); burst<8, 8>(dA, out, wpc); burst<2, 8>(dA, out, wpc); burst<16, 8>(dA, out, wpc);
    }
    return 0;
}
