// hbm_stream.hip -- streaming read / write / copy bandwidth of the MI355X on buffers far larger than the 256 MB Infinity
// Cache (scratch tool): the ceilings the stitch kernels' traffic mix (36 % reads, 64 % writes) is measured against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_write(uint4 *__restrict__ dst, size_t n16)
{
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = v;
}
__global__ void __launch_bounds__(256) k_read(const uint4 *__restrict__ src, size_t n16, uint32_t *__restrict__ sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = src[i]; acc ^= v.x + v.y + v.z + v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
// RW: reads rd16 chunks per wr16 written (rd:wr traffic ratio like the stitch: 36:64 -> 9 reads per 16 writes)
__global__ void __launch_bounds__(256) k_mix(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16, int rd_per_16, uint32_t *__restrict__ sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        uint4 v = make_uint4(i, 1, 2, 3);
        if ((int)(i & 15) < rd_per_16) { v = src[i]; acc ^= v.x; }
        dst[i] = v;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F> static float timeit(F launch)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}

int main()
{
    const size_t bytes = 3ull << 30, n16 = bytes / 16;
    uint4 *a, *b; uint32_t *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const int blocks = 256 * 8;
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n16, sink); });
    printf("stream read  3 GB            %7.3f ms  %7.1f GB/s\n", ms, bytes / ms * 1e-6);
    ms = timeit([&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, b, n16); });
    printf("stream write 3 GB            %7.3f ms  %7.1f GB/s\n", ms, bytes / ms * 1e-6);
    ms = timeit([&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, a, b, n16, 16, sink); });
    printf("copy (3 GB read + 3 GB write) %7.3f ms  %7.1f GB/s total\n", ms, 2.0 * bytes / ms * 1e-6);
    ms = timeit([&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, a, b, n16, 9, sink); });
    printf("mix 36 %% read : 64 %% write   %7.3f ms  %7.1f GB/s total\n", ms, (bytes * (1.0 + 9.0 / 16)) / ms * 1e-6);
    ms = timeit([&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, a, b, n16, 4, sink); });
    printf("mix 20 %% read : 80 %% write   %7.3f ms  %7.1f GB/s total\n", ms, (bytes * (1.0 + 4.0 / 16)) / ms * 1e-6);
    return 0;
}
