// qbcast_probe.hip -- how should wave-uniform query words reach the multi-query inner loop?
// One block of 16 query words meets RPL register-resident rows (16 words each):
//   MODE 0: SGPR operand of v_and_b32 (what batch_scan_kernel does)
//   MODE 1: ds_read_b128 broadcast (uniform address) into VGPRs, then VGPR-operand v_and
//   MODE 2: v_mov_b32 v, s once per word, then VGPR-operand v_and
// Prints cycles per (word,row) pair per SIMD; 2 VALU instructions per pair is the floor.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc) { unsigned r; asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc)); return r; }

template <int MODE, int RPL> __global__ __launch_bounds__(256) void k(unsigned* out, const unsigned* __restrict__ qsrc, int iters)
{
    __shared__ u32x4 lq[512]; // 8 KB of query words
    for (int i = threadIdx.x; i < 512; i += 256) lq[i] = u32x4{qsrc[i & 15], qsrc[(i + 1) & 15], qsrc[(i + 2) & 15], qsrc[(i + 3) & 15]};
    __syncthreads();
    unsigned r[RPL][16];
    for (int p = 0; p < RPL; p++)
        for (int j = 0; j < 16; j++) r[p][j] = threadIdx.x * 2654435761u + j * 40503u + p * 977u;
    unsigned acc[RPL][4];
    for (int p = 0; p < RPL; p++) for (int j = 0; j < 4; j++) acc[p][j] = 0;
    typedef const __attribute__((address_space(4))) unsigned* cp;
    cp qs = (cp) qsrc;
    unsigned s[16];
    for (int j = 0; j < 16; j++) s[j] = qs[j];
    u32x4 cur[4], nxt[4];
    if (MODE == 1) for (int j = 0; j < 4; j++) cur[j] = lq[j];
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
            unsigned qv[16];
            if (MODE == 1) {
                const int b = ((i * 4 + rep + 1) & 127) * 4;
#pragma unroll
                for (int j = 0; j < 4; j++) nxt[j] = lq[b + j];
#pragma unroll
                for (int j = 0; j < 4; j++) { qv[4 * j] = cur[j].x; qv[4 * j + 1] = cur[j].y; qv[4 * j + 2] = cur[j].z; qv[4 * j + 3] = cur[j].w; }
            } else if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 16; j++) asm volatile("v_mov_b32 %0, %1" : "=v"(qv[j]) : "s"(s[j]));
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
#pragma unroll
                for (int p = 0; p < RPL; p++) {
                    if (MODE == 0) acc[p][j & 3] = bcnt_acc(r[p][j] & s[j], acc[p][j & 3]);
                    else acc[p][j & 3] = bcnt_acc(r[p][j] & qv[j], acc[p][j & 3]);
                }
            }
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) cur[j] = nxt[j];
            }
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("" : "+s"(s[j]));
        }
    }
    unsigned t = 0;
    for (int p = 0; p < RPL; p++) for (int j = 0; j < 4; j++) t ^= acc[p][j];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int MODE, int RPL> void run(unsigned* out, unsigned* q, int wpc)
{
    const int iters = 1000, blocks = 256 * wpc / 4;
    hipEvent_t a, b; (void) hipEventCreate(&a); (void) hipEventCreate(&b);
    float best = 1e30f;
    for (int it = 0; it < 4; it++) {
        (void) hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE, RPL>), dim3(blocks), dim3(256), 0, 0, out, q, iters);
        (void) hipEventRecord(b); (void) hipEventSynchronize(b);
        float ms; (void) hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    const double pairs = double(blocks) * 4 * iters * 4 * 16 * RPL; // (word,row-register) wave-pairs
    printf("mode %d rpl %d wpc %2d: %.3f ms, %.2f cycles/pair/SIMD at 2.4 GHz\n", MODE, RPL, wpc, best,
           best * 1e-3 * 2.4e9 / (pairs / 1024.0));
}
int main()
{
    unsigned *out, *q;
    (void) hipMalloc(&out, 256 * 16 * 256 * 4); (void) hipMalloc(&q, 64); (void) hipMemset(q, 0x5a, 64);
    for (int wpc : {4, 8, 12}) {
        run<0, 1>(out, q, wpc); run<0, 2>(out, q, wpc); run<0, 4>(out, q, wpc);
        run<1, 1>(out, q, wpc); run<1, 2>(out, q, wpc); run<1, 3>(out, q, wpc); run<1, 4>(out, q, wpc);
        run<2, 1>(out, q, wpc); run<2, 2>(out, q, wpc); run<2, 3>(out, q, wpc); run<2, 4>(out, q, wpc);
    }
    return 0;
}
