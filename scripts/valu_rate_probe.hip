// valu_rate_probe.hip -- issue rate of the instructions the multi-query kernel lives on:
// v_bcnt_u32_b32 (popcount-accumulate) and v_and_b32, against v_add_u32, wave64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP> __global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed, int iters)
{
    unsigned a0 = threadIdx.x ^ seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned x = seed * 2654435761u + threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (OP == 0) { // 8 independent v_bcnt_u32_b32 chains
                a0 = __popc(x) + a0; a1 = __popc(x + 1) + a1; a2 = __popc(x + 2) + a2; a3 = __popc(x + 3) + a3;
                a4 = __popc(x + 4) + a4; a5 = __popc(x + 5) + a5; a6 = __popc(x + 6) + a6; a7 = __popc(x + 7) + a7;
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 1) { // v_and
                a0 &= x | 1; a1 &= x | 2; a2 &= x | 4; a3 &= x | 8; a4 &= x | 16; a5 &= x | 32; a6 &= x | 64; a7 &= x | 128;
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else { // v_add_u32
                a0 += x; a1 += x; a2 += x; a3 += x; a4 += x; a5 += x; a6 += x; a7 += x;
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP> double run(unsigned* out, int blocks)
{
    const int iters = 4000;
    hipEvent_t a, b;
    (void) hipEventCreate(&a);
    (void) hipEventCreate(&b);
    float best = 1e30f;
    for (int it = 0; it < 5; it++) {
        (void) hipEventRecord(a);
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, 12345u + it, iters);
        (void) hipEventRecord(b);
        (void) hipEventSynchronize(b);
        float ms;
        (void) hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    const double ops = double(blocks) * 256 * iters * 16 * 8; // lane-ops of the measured kind (x also costs adds for OP 0)
    return ops / (best * 1e-3) / 1e12;
}

int main()
{
    unsigned* out;
    (void) hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int wpc : {4, 8, 16}) {
        const int blocks = 256 * wpc / 4;
        printf("waves/CU=%2d  v_bcnt(+v_add for the operand) %6.2f T lane-ops/s | v_and(+v_or) %6.2f | v_add %6.2f\n", wpc,
               run<0>(out, blocks), run<1>(out, blocks), run<2>(out, blocks));
    }
    return 0;
}
