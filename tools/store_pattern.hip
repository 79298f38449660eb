// store_pattern.hip -- HBM write throughput of the stitch kernels' STORE pattern alone (scratch tool): a batch of 256 BEV
// images (1080 x 1080 x 3 B = 896 MB) written (a) as a linear stream, (b) exactly as the tile kernels write it: one wave
// = one 32 x 8 pixel tile, a lane = 4 pixels = 12 bytes (96-byte row segments), 4 x-neighbouring tiles per workgroup,
// 8 frames per workgroup, workgroup id % 8 (= XCD) owns whole batch chunks, (c..) variations of tile shape / chunking.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int BW = 1080, BH = 1080, BATCH = 256;

__global__ void __launch_bounds__(256) k_linear(uint4 *__restrict__ dst, size_t n16)
{
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = v;
}

// LX lanes along x (4 px each), 64 / LX rows; nb frames per block; xcd: 1 = chunk-per-XCD block map, 0 = chunk-major
template <int LX>
__global__ void __launch_bounds__(256) k_tiles(uint8_t *__restrict__ out, int tiles_x, int ntiles, int nb, int nchunks, int xcd, int nmath = 0)
{
    constexpr int LY = 64 / LX;
    const uint32_t ng = (uint32_t)((ntiles + 3) / 4), id = blockIdx.x;
    uint32_t chunk, group;
    if (xcd) { const uint32_t x = id & 7u, k = id >> 3; chunk = x + 8u * (k / ng); group = k % ng; }
    else { chunk = id / ng; group = id % ng; }
    if ((int)chunk >= nchunks) return;
    const int lane = threadIdx.x & 63, tile = (int)group * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int x0 = (tx * LX + lane % LX) * 4, y = ty * LY + lane / LX;
    if (x0 >= BW || y >= BH) return;
    const size_t img = (size_t)BW * BH * 3;
    uint8_t *ob = out + (size_t)chunk * nb * img + ((size_t)y * BW + x0) * 3;
    uint32_t v0 = lane, v1 = tile, v2 = chunk;
    for (int b = 0; b < nb; ++b, ob += img) {
        // nmath rounds of 3 dependent dot4 (VALU work between the stores, like the interpolation of a frame)
        for (int i = 0; i < nmath; ++i) {
            v0 = __builtin_amdgcn_udot4(v0, 0x01020304u, v1, false);
            v1 = __builtin_amdgcn_udot4(v1, 0x04030201u, v2, false);
            v2 = __builtin_amdgcn_udot4(v2, 0x01010101u, v0, false);
        }
        uint32_t *op = reinterpret_cast<uint32_t *>(ob);
        op[0] = v0 + b; op[1] = v1; op[2] = v2;
    }
}

template <typename F> static float timeit(F launch)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5;
}

template <int LX> static void run_tiles(uint8_t *out, int nb, int xcd, const char *name, int nmath = 0, bool store = true)
{
    const int tiles_x = (BW + 4 * LX - 1) / (4 * LX), tiles_y = (BH + 64 / LX - 1) / (64 / LX), ntiles = tiles_x * tiles_y;
    const int nchunks = BATCH / nb, ng = (ntiles + 3) / 4;
    const unsigned grid = xcd ? ng * ((nchunks + 7) / 8 * 8) : ng * nchunks;
    const double bytes = (double)BATCH * BW * BH * 3;
    const float ms = timeit([&] { hipLaunchKernelGGL((k_tiles<LX>), dim3(grid), dim3(256), 0, 0, out, tiles_x, ntiles, nb, nchunks, xcd, nmath); });
    printf("%-44s %7.3f ms  %7.1f GB/s\n", name, ms, bytes / ms * 1e-6);
}

int main()
{
    const size_t bytes = (size_t)BATCH * BW * BH * 3;
    uint8_t *out;
    CK(hipMalloc(&out, bytes));
    CK(hipMemset(out, 0, bytes));
    float ms = timeit([&] { hipLaunchKernelGGL(k_linear, dim3(2048), dim3(256), 0, 0, (uint4 *)out, bytes / 16); });
    printf("%-44s %7.3f ms  %7.1f GB/s\n", "linear stream, 16 B per lane", ms, bytes / ms * 1e-6);
    run_tiles<8>(out, 8, 1, "32x8 tiles, 8 frames/block, XCD chunks [product]");
    run_tiles<8>(out, 8, 0, "32x8 tiles, 8 frames/block, chunk-major");
    run_tiles<8>(out, 1, 0, "32x8 tiles, 1 frame/block");
    run_tiles<8>(out, 32, 1, "32x8 tiles, 32 frames/block, XCD chunks");
    run_tiles<8>(out, 8, 1, "  + 42 VALU per frame between the stores", 14);
    run_tiles<8>(out, 8, 1, "  + 126 VALU per frame", 42);
    run_tiles<8>(out, 8, 1, "  + 252 VALU per frame", 84);
    run_tiles<8>(out, 8, 1, "  + 504 VALU per frame", 168);
    run_tiles<16>(out, 8, 1, "64x4 tiles, 8 frames/block, XCD chunks");
    run_tiles<4>(out, 8, 1, "16x16 tiles, 8 frames/block, XCD chunks");
    return 0;
}
