// launch_gap_probe.hip -- how long is the gap between two back-to-back kernels of one stream that each need every CU
// (256 workgroups x 384 threads, 157 KB of LDS: one workgroup per CU, as fused_kernel), with the default launch and
// with hipExtAnyOrderLaunch (no barrier bit between the packets)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_gap_probe scripts/launch_gap_probe.hip && /tmp/launch_gap_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ __launch_bounds__(384) void busy(unsigned long long ticks, unsigned int* sink)
{
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = wall_clock64();
    unsigned int x = 0;
    while (wall_clock64() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        x++;
    }
    if (threadIdx.x == 0) lds[0] = static_cast<unsigned char>(x);
    if (x == 0xFFFFFFFFu) sink[0] = lds[0];
}

int main()
{
    unsigned int* sink;
    (void) hipMalloc(&sink, 64);
    hipStream_t s;
    (void) hipStreamCreate(&s);
    const size_t shmem = 157 * 1024;
    (void) hipFuncSetAttribute(reinterpret_cast<const void*>(busy), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shmem));
    for (unsigned long long us : {40ull, 10ull}) {
        const unsigned long long ticks = us * 100ull; // wall_clock64: 100 MHz
        for (int mode = 0; mode < 2; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                const int n = 2000;
                (void) hipStreamSynchronize(s);
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < n; i++) {
                    if (mode == 0) hipLaunchKernelGGL(busy, dim3(256), dim3(384), shmem, s, ticks, sink);
                    else hipExtLaunchKernelGGL(busy, dim3(256), dim3(384), shmem, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink);
                }
                (void) hipStreamSynchronize(s);
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (rep) std::printf("kernel body %llu us, %s: %.2f us per launch (gap %.2f us)\n", us, mode ? "hipExtAnyOrderLaunch" : "default launch", 1e6 * el / n, 1e6 * el / n - us);
            }
        }
    }
    std::printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
