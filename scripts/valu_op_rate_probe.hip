// valu_op_rate_probe.hip -- issue cost (cycles per wave64 instruction per SIMD) of the VALU
// operations the scans use, 8 independent chains per wave, 1..3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAIN8(OP)                                                                                            \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                               \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) \
                 : "v"(y), "s"(sv));
#define OP_AND_V(i) "v_and_b32 %" #i ", %8, %" #i "\n"
#define OP_AND_S(i) "v_and_b32 %" #i ", %9, %" #i "\n"
#define OP_BCNT(i) "v_bcnt_u32_b32 %" #i ", %8, %" #i "\n"
#define OP_ADD(i) "v_add_u32 %" #i ", %8, %" #i "\n"
#define OP_LSHR(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define OP_FMA(i) "v_fma_f32 %" #i ", %8, %" #i ", %8\n"
#define OP_XOR_S(i) "v_xor_b32 %" #i ", %9, %" #i "\n"
template <int MODE> __global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned sv0)
{
    unsigned x[8];
    for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
    unsigned y = threadIdx.x * 77u + 1u;
    unsigned sv = __builtin_amdgcn_readfirstlane(sv0);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
            if (MODE == 0) CHAIN8(OP_AND_V)
            if (MODE == 1) CHAIN8(OP_AND_S)
            if (MODE == 2) CHAIN8(OP_BCNT)
            if (MODE == 3) CHAIN8(OP_ADD)
            if (MODE == 4) CHAIN8(OP_LSHR)
            if (MODE == 5) CHAIN8(OP_FMA)
            if (MODE == 6) CHAIN8(OP_XOR_S)
        }
    }
    unsigned t = 0;
    for (int j = 0; j < 8; j++) t ^= x[j];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int MODE> void run(unsigned* out, int wpc, const char* name)
{
    const int iters = 4000, blocks = 256 * wpc / 4;
    hipEvent_t a, b; (void) hipEventCreate(&a); (void) hipEventCreate(&b);
    float best = 1e30f;
    for (int it = 0; it < 4; it++) {
        (void) hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 0x5a5a5a5au);
        (void) hipEventRecord(b); (void) hipEventSynchronize(b);
        float ms; (void) hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    const double per_simd = double(blocks) * 4 * iters * 64 / 1024.0;
    printf("%-22s waves/SIMD %d: %.2f cycles/instr at 2.4 GHz\n", name, wpc / 4, best * 1e-3 * 2.4e9 / per_simd);
}
int main()
{
    unsigned* out; (void) hipMalloc(&out, 256 * 16 * 256 * 4);
    for (int wpc : {4, 8, 12}) {
        run<0>(out, wpc, "v_and_b32 v,v"); run<1>(out, wpc, "v_and_b32 s,v"); run<2>(out, wpc, "v_bcnt_u32_b32");
        run<3>(out, wpc, "v_add_u32"); run<4>(out, wpc, "v_lshrrev_b32"); run<5>(out, wpc, "v_fma_f32"); run<6>(out, wpc, "v_xor_b32 s,v");
    }
    return 0;
}
