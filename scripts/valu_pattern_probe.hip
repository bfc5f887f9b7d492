// valu_pattern_probe.hip -- cycles per instruction of the multi-query inner pattern
// (v_and_b32 v,s,v ; v_bcnt_u32_b32 acc) for different interleavings, wave64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc) { unsigned r; asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc)); return r; }

// MODE 0: 4 accumulators, SGPR query operand   1: 8 accumulators, SGPR   2: 4 acc, VGPR operand
// MODE 3: 16 accumulators, SGPR
template <int MODE> __global__ __launch_bounds__(256) void k(unsigned* out, const unsigned* __restrict__ qsrc, int iters)
{
    unsigned r[16];
    for (int j = 0; j < 16; j++) r[j] = threadIdx.x * 2654435761u + j * 40503u;
    unsigned acc[16];
    for (int j = 0; j < 16; j++) acc[j] = 0;
    typedef const __attribute__((address_space(4))) unsigned* cp;
    cp qs = (cp) qsrc;
    unsigned qs0 = qs[0], qs1 = qs[1], qs2 = qs[2], qs3 = qs[3];
    unsigned qv = qsrc[threadIdx.x & 3];
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 16; j += 4) {
                    acc[0] = bcnt_acc(r[j] & qs0, acc[0]); acc[1] = bcnt_acc(r[j + 1] & qs1, acc[1]);
                    acc[2] = bcnt_acc(r[j + 2] & qs2, acc[2]); acc[3] = bcnt_acc(r[j + 3] & qs3, acc[3]);
                }
            } else if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 16; j += 8) {
#pragma unroll
                    for (int u = 0; u < 8; u++) acc[u] = bcnt_acc(r[j + u] & ((u & 1) ? qs1 : qs0), acc[u]);
                }
            } else if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 16; j += 4) {
                    acc[0] = bcnt_acc(r[j] & qv, acc[0]); acc[1] = bcnt_acc(r[j + 1] & qv, acc[1]);
                    acc[2] = bcnt_acc(r[j + 2] & qv, acc[2]); acc[3] = bcnt_acc(r[j + 3] & qv, acc[3]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 16; u++) acc[u] = bcnt_acc(r[u] & ((u & 1) ? qs1 : qs0), acc[u]);
            }
            asm volatile("" : "+s"(qs0), "+s"(qs1), "+s"(qs2), "+s"(qs3));
        }
    }
    unsigned t = 0;
    for (int j = 0; j < 16; j++) t ^= acc[j];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int MODE> void run(unsigned* out, unsigned* q, int wpc)
{
    const int iters = 2000, blocks = 256 * wpc / 4;
    hipEvent_t a, b; (void) hipEventCreate(&a); (void) hipEventCreate(&b);
    float best = 1e30f;
    for (int it = 0; it < 4; it++) {
        (void) hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, q, iters);
        (void) hipEventRecord(b); (void) hipEventSynchronize(b);
        float ms; (void) hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    const double winstr = double(blocks) * 4 * iters * 8 * 32; // wave-instructions (16 and + 16 bcnt per rep)
    const double per_simd = winstr / 1024.0;
    printf("mode %d wpc %2d: %.2f ms, %.2f cycles/instr at 2.38 GHz, %.1f T lane-ops/s\n", MODE, wpc, best,
           best * 1e-3 * 2.38e9 / per_simd, winstr * 64 / (best * 1e-3) / 1e12);
}
int main()
{
    unsigned *out, *q;
    (void) hipMalloc(&out, 256 * 16 * 256 * 4); (void) hipMalloc(&q, 64); (void) hipMemset(q, 0x5a, 64);
    for (int wpc : {4, 8, 16}) { run<0>(out, q, wpc); run<1>(out, q, wpc); run<2>(out, q, wpc); run<3>(out, q, wpc); }
    return 0;
}
