// gsim_litmus.hip -- litmus kernels for the three hardware behaviours the single launch (gsim_fused.hip) rests on, with the
// SAME instructions it uses (raw buffer loads / stores of 16 bytes, cache-policy bits sc1 / sc0 sc1):
//
//   1  an aligned 16-byte store with sc1 (write-through to device scope) is seen by a 16-byte sc1 load of another
//      workgroup -- on another XCD -- as ONE piece: never a mix of two stores' words          (region entries, region headers)
//   2  an aligned 16-byte store with sc0 sc1 (system scope) into pinned host memory lands as one piece for the host that
//      polls ONE of its words and then reads the others                                         (the result block's header)
//   3  two sc1 stores of one lane to different lines, entry first, header second, may become visible in the other order:
//      a reader that has seen the header may still read the OLD entry -- how often, and does a re-read always get it?
//                                                                                               (header = arrival, tagged entries)
//
// Nothing of the product calls these kernels; gsim_debug_litmus (capi_debug.cpp) launches them for the -m gpu tests, so that a
// ROCm / firmware change in any of the three shows up as a named failure instead of a flaky soak.
#include "gsim_device_common.h"

namespace gsim
{
namespace
{

__device__ __forceinline__ unsigned long long litmus_clock() { return wall_clock64(); } // (100 MHz)

// the four words of the store number `it` of slot `slot`: any two of them identify (it, slot), none is 0 for it >= 1
__device__ __forceinline__ u32x4 litmus_value(uint32_t it, uint32_t slot)
{
    return u32x4{it, it * 0x9E3779B1u + slot, (it ^ 0xA5A5A5A5u) + slot * 0x85EBCA6Bu, ~it};
}

__device__ __forceinline__ bool litmus_consistent(const u32x4& v, uint32_t slot)
{
    const u32x4 w = litmus_value(v.x, slot);
    return v.y == w.y && v.z == w.z && v.w == w.w;
}

// Test 1 and 3.  Workgroups 2 p (writer) and 2 p + 1 (reader) share 64 slots (one per lane).  Consecutive block indices land
// on different XCDs (MI355X_MICROARCH.md): the pair talks through memory, not through one L2.
//   writer, it = 1 .. iters:   entry[slot] <- value(it)   (sc1)        then, test 3 only:   header[slot] <- value(it)   (sc1)
//   reader, until it has seen the last store or the clock runs out:
//      test 1: v <- entry[slot] (sc1); torn += !consistent(v); loads++
//      test 3: h <- header[slot] (sc1); torn += !consistent(h); if h.x != last: e <- entry[slot];
//              torn += !consistent(e); if e.x < h.x: stale++, re-read until e.x >= h.x (never_landed++ when the clock runs out)
// stats[0] loads, [1] torn values, [2] headers seen (test 3), [3] entries that were behind their header at the first read,
// [4] entries that never caught up, [5] re-reads, [6] readers that ran out of time, [7] stores.
// with_header = 2: the product's shape -- the entries are stored by ANOTHER wave of the writer workgroup, a workgroup barrier (no wait for
// the stores' acknowledgements), then wave 0 stores the headers: two waves' stores race through the memory pipeline.
__global__ __launch_bounds__(128) void litmus_pair_kernel(u32x4* entries, u32x4* headers, unsigned long long* stats, uint32_t iters, int with_header,
                                                         unsigned long long budget_ticks)
{
    const uint32_t pair = blockIdx.x >> 1, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t slot = pair * 64u + lane;
    // (entries and headers of a pair are 1 KiB apart per wave-wide store, in different arrays: different cache lines, as in the product)
    const __amdgpu_buffer_rsrc_t ers = __builtin_amdgcn_make_buffer_rsrc(entries, 0, gridDim.x * 32u * 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(headers, 0, gridDim.x * 32u * 16u, 0x00020000);
    if ((blockIdx.x & 1u) == 0 && with_header == 2) {
        for (uint32_t it = 1; it <= iters; it++) {
            const u32x4 v = litmus_value(it, slot);
            if (wv == 1) __builtin_amdgcn_raw_buffer_store_b128(v, ers, slot * 16u, 0, /*sc1*/ 16);
            __builtin_amdgcn_s_barrier(); // (execution only: nobody waits for the entry stores)
            if (wv == 0) __builtin_amdgcn_raw_buffer_store_b128(v, hrs, slot * 16u, 0, /*sc1*/ 16);
            if ((it & 15u) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (threadIdx.x == 0) atomicAdd(&stats[7], static_cast<unsigned long long>(iters) * 64ull);
        return;
    }
    if (wv != 0) return; // (one wave: the same-lane writer, every reader)
    if ((blockIdx.x & 1u) == 0) {
        for (uint32_t it = 1; it <= iters; it++) {
            const u32x4 v = litmus_value(it, slot);
            __builtin_amdgcn_raw_buffer_store_b128(v, ers, slot * 16u, 0, /*sc1*/ 16);
            if (with_header) __builtin_amdgcn_raw_buffer_store_b128(v, hrs, slot * 16u, 0, /*sc1*/ 16);
            if ((it & 15u) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (bounded queue of stores in flight, as a publishing wave has)
        }
        if (lane == 0) atomicAdd(&stats[7], static_cast<unsigned long long>(iters) * 64ull);
        return;
    }
    unsigned long long loads = 0, torn = 0, seen = 0, stale = 0, never = 0, rereads = 0;
    uint32_t last = 0;
    const unsigned long long t0 = litmus_clock();
    bool timed_out = false;
    while (true) {
        asm volatile("" ::: "memory"); // (every trip reads memory again)
        if (!with_header) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ers, slot * 16u, 0, /*sc1*/ 16);
            loads++;
            if (v.x != 0 && !litmus_consistent(v, slot)) torn++;
            last = v.x;
        } else {
            const u32x4 h = __builtin_amdgcn_raw_buffer_load_b128(hrs, slot * 16u, 0, /*sc1*/ 16);
            loads++;
            if (h.x != 0 && !litmus_consistent(h, slot)) torn++;
            if (h.x != last && h.x != 0) {
                seen++;
                u32x4 e = __builtin_amdgcn_raw_buffer_load_b128(ers, slot * 16u, 0, /*sc1*/ 16);
                loads++;
                if (e.x != 0 && !litmus_consistent(e, slot)) torn++;
                if (e.x < h.x) { // the header overtook its entry: read again, as a selector does
                    stale++;
                    const unsigned long long tw = litmus_clock();
                    do {
                        asm volatile("" ::: "memory");
                        e = __builtin_amdgcn_raw_buffer_load_b128(ers, slot * 16u, 0, /*sc1*/ 16);
                        rereads++;
                        if (e.x != 0 && !litmus_consistent(e, slot)) torn++;
                        if (litmus_clock() - tw > budget_ticks / 8) {
                            never++;
                            break;
                        }
                    } while (e.x < h.x);
                }
                last = h.x;
            }
        }
        const bool done = last >= iters;
        if (__ballot(!done) == 0) break;
        if (litmus_clock() - t0 > budget_ticks) {
            timed_out = true;
            break;
        }
    }
    atomicAdd(&stats[0], loads);
    if (torn) atomicAdd(&stats[1], torn);
    if (seen) atomicAdd(&stats[2], seen);
    if (stale) atomicAdd(&stats[3], stale);
    if (never) atomicAdd(&stats[4], never);
    if (rereads) atomicAdd(&stats[5], rereads);
    if (timed_out && lane == 0) atomicAdd(&stats[6], 1ull);
}

// Test 2.  One slot per workgroup in pinned host memory.  it = 1 .. iters: slot <- value(it) with sc0 sc1 (one 16-byte store, what the
// closing workgroup's header store is), then wait until the host has written `it` into ack[b] (system-scope loads) -- the host reads
// word 1 first, the other three after an acquire fence, exactly as finish_query_sync reads a header.
__global__ __launch_bounds__(64) void litmus_host_kernel(u32x4* slots, const uint32_t* ack, unsigned long long* stats, uint32_t iters, unsigned long long budget_ticks)
{
    if (threadIdx.x != 0) return;
    const uint32_t b = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slots, 0, gridDim.x * 16u, 0x00020000);
    const unsigned long long t0 = litmus_clock();
    for (uint32_t it = 1; it <= iters; it++) {
        const u32x4 v = litmus_value(it, b);
        // (word 1 is the one the host polls: the product's header carries flags | epoch << 8 there)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.y, v.x, v.z, v.w}, rs, b * 16u, 0, /*sc0 sc1*/ 17);
        while (__hip_atomic_load(&ack[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != it) {
            if (litmus_clock() - t0 > budget_ticks) {
                atomicAdd(&stats[6], 1ull);
                return;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    atomicAdd(&stats[7], static_cast<unsigned long long>(iters));
}

} // namespace

hipError_t launch_litmus_pair(void* entries, void* headers, unsigned long long* stats, uint32_t nblocks, uint32_t iters, int with_header,
                              unsigned long long budget_ticks, hipStream_t s)
{
    hipLaunchKernelGGL(litmus_pair_kernel, dim3(nblocks), dim3(128), 0, s, static_cast<u32x4*>(entries), static_cast<u32x4*>(headers), stats, iters, with_header,
                       budget_ticks);
    return hipGetLastError();
}

hipError_t launch_litmus_host(void* slots, const uint32_t* ack, unsigned long long* stats, uint32_t nblocks, uint32_t iters, unsigned long long budget_ticks,
                              hipStream_t s)
{
    hipLaunchKernelGGL(litmus_host_kernel, dim3(nblocks), dim3(64), 0, s, static_cast<u32x4*>(slots), ack, stats, iters, budget_ticks);
    return hipGetLastError();
}

} // namespace gsim
