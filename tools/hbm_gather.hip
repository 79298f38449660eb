// hbm_gather.hip -- HBM read efficiency of sector-granular gathers on MI355X (scratch tool, not part of the product).
// Every wave instruction reads 1 KB (16 B per lane); the 1 KB is made of contiguous chunks of CHUNK bytes placed at
// pseudo-random (or image-like strided) addresses inside a buffer much larger than the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// PAT 0: random chunks | 1: image-like: consecutive chunks of a wave step by ROW bytes (rows of a frame), wave base random
template <int CHUNK, int PAT, int UNROLL>
__global__ void __launch_bounds__(256) k_gather(const uint8_t *__restrict__ buf, uint64_t nbytes, int iters, uint32_t *__restrict__ sink)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
    constexpr int LPC = CHUNK / 16;            // lanes per chunk
    const uint32_t nchunks_total = (uint32_t)(nbytes / CHUNK);
    uint32_t acc = 0;
    for (int it = 0; it < iters; it += UNROLL) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t seq = (wave_id * (uint32_t)iters + it + u);
            uint64_t addr;
            if (PAT == 2) {          // the whole wave reads ONE random 64-byte line (4 lanes cover it, the rest repeat it)
                const uint32_t c = hash32(seq) % (uint32_t)(nbytes / 64);
                addr = (uint64_t)c * 64 + (lane % 4) * 16;
            } else if (PAT == 3) {   // one random missing line + 15 lines that stay hot (a fixed 1 KB window per wave)
                const uint32_t c = hash32(seq) % (uint32_t)(nbytes / 64);
                addr = lane < 4 ? (uint64_t)c * 64 + lane * 16 : (uint64_t)(wave_id % 4096) * 1024 + lane * 16;
            } else if (PAT == 0) {
                const uint32_t c = hash32(seq * (64 / LPC) + lane / LPC) % nchunks_total;
                addr = (uint64_t)c * CHUNK + (lane % LPC) * 16;
            } else {
                const uint64_t base = ((uint64_t)(hash32(seq) % (uint32_t)(nbytes / 4096 - 64))) * 4096;
                addr = base + (uint64_t)(lane / LPC) * 3840 + (lane % LPC) * 16;
            }
            v[u] = *reinterpret_cast<const uint4 *>(buf + addr);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F>
static void timeit(const char *name, F launch, double bytes)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-52s %8.3f ms  %8.1f GB/s (useful 16 B/lane)\n", name, ms, bytes / ms * 1e-6);
}

int main()
{
    const uint64_t nbytes = 3ull << 30;
    uint8_t *buf; uint32_t *sink;
    CK(hipMalloc(&buf, nbytes)); CK(hipMemset(buf, 1, nbytes)); CK(hipMalloc(&sink, 64));
    const int blocks = 256 * 16, iters = 64;
    const double bytes = (double)blocks * 4 * iters * 1024;
#define RUN(C, P, U, N) timeit(N, [&] { hipLaunchKernelGGL((k_gather<C, P, U>), dim3(blocks), dim3(256), 0, 0, buf, nbytes, iters, sink); }, bytes)
    RUN(16, 0, 4, "random 16 B pieces (64 lines / instr)");
    RUN(64, 0, 4, "random 64 B lines");
    RUN(128, 0, 4, "random 128 B");
    RUN(256, 0, 4, "random 256 B");
    RUN(512, 0, 4, "random 512 B");
    RUN(1024, 0, 4, "random 1 KB (one instr = one chunk)");
    RUN(64, 1, 4, "image rows: 16 rows x 64 B per instr");
    RUN(128, 1, 4, "image rows: 8 rows x 128 B per instr");
    RUN(256, 1, 4, "image rows: 4 rows x 256 B per instr");
    RUN(64, 0, 8, "random 64 B lines, 8 loads in flight / wave");
    RUN(64, 0, 1, "random 64 B lines, 1 load in flight / wave");
    RUN(128, 0, 8, "random 128 B, 8 loads in flight / wave");
    RUN(256, 0, 8, "random 256 B, 8 loads in flight / wave");
    RUN(64, 2, 4, "ONE random 64 B line per instr, 4 in flight");
    RUN(64, 2, 8, "ONE random 64 B line per instr, 8 in flight");
    RUN(64, 2, 16, "ONE random 64 B line per instr, 16 in flight");
    RUN(64, 3, 8, "1 random line + 15 hot lines per instr, 8 in flight");
    RUN(64, 3, 16, "1 random line + 15 hot lines per instr, 16 in flight");
    {
        // same one-line-per-instruction pattern, but over buffers that fit in L2 (2 MB x 8 XCDs) or in the Infinity Cache
        const uint64_t sizes[] = {1ull << 20, 16ull << 20, 128ull << 20};
        for (uint64_t sz : sizes) {
            char nm[96];
            snprintf(nm, sizeof nm, "ONE line per instr, 8 in flight, buffer %llu MB", (unsigned long long)(sz >> 20));
            timeit(nm, [&] { hipLaunchKernelGGL((k_gather<64, 2, 8>), dim3(blocks), dim3(256), 0, 0, buf, sz, iters, sink); }, bytes);
            snprintf(nm, sizeof nm, "16 random lines per instr, 4 in flight, buffer %llu MB", (unsigned long long)(sz >> 20));
            timeit(nm, [&] { hipLaunchKernelGGL((k_gather<64, 0, 4>), dim3(blocks), dim3(256), 0, 0, buf, sz, iters, sink); }, bytes);
        }
    }
    printf("(for PAT 2/3 the useful HBM traffic is 64 B per instruction: GB/s column / 16)\n");
    return 0;
}
