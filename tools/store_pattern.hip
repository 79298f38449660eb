// store_pattern.hip -- HBM write throughput of the stitch kernels' STORE pattern alone (scratch tool): a batch of 256 BEV
// images (1080 x 1080 x 3 B = 896 MB) written (a) as a linear stream, (b) exactly as the tile kernels write it: one wave
// = one 32 x 8 pixel tile, a lane = 4 pixels = 12 bytes (96-byte row segments), 4 x-neighbouring tiles per workgroup,
// 8 frames per workgroup, workgroup id % 8 (= XCD) owns whole batch chunks, (c..) variations of tile shape / chunking.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
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

// round 3: the UNIT store pattern -- a block of 4 waves owns a rectangle of (4 << LQ) x (NQ * 4 * (64 >> LQ)) pixels; a wave-store covers
// 64 >> LQ rows of (1 << LQ) quads (LQ = 6: ONE row run of 768 bytes); rows are dealt to the waves round-robin (rr = 1) or in bands (rr = 0);
// pitch = pixels per image row (1080 dense, 1088 whole sectors)
// LW: log2 of the lanes per row inside ONE wave-store (LW == LQ: a wave-store is a band of whole unit rows; LW < LQ: the unit's width is cut into
// 1 << (LQ - LW) column strips, a wave-store covers 64 >> LW rows of one strip)
template <int LQ, int NQ, int LW = LQ>
__global__ void __launch_bounds__(256) k_units(uint8_t *__restrict__ out, int pitch, int units_x, int nunits, int nb, int nchunks, int rr, int bw_eff)
{
    const uint32_t ng = (uint32_t)nunits, id = blockIdx.x;
    const uint32_t x8 = id & 7u, k = id >> 3, chunk = x8 + 8u * (k / ng), unit = k % ng;
    if ((int)chunk >= nchunks) return;
    constexpr int W = 4 << LQ, RPS = 64 >> LW, COLS = 1 << (LQ - LW), H = NQ * 4 * RPS / COLS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ux = (int)(unit % units_x) * W, uy = (int)(unit / units_x) * H;
    const size_t img = (size_t)pitch * BH * 3;
    uint32_t off[NQ];
    bool ok[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int sidx = rr ? j * 4 + wave : wave * NQ + j;
        const int x = ux + 4 * ((sidx % COLS) * (1 << LW) + (lane & ((1 << LW) - 1))), y = uy + (sidx / COLS) * RPS + (lane >> LW);
        ok[j] = x < bw_eff && y < BH;
        off[j] = ((uint32_t)y * pitch + x) * 3;
    }
    uint8_t *ob = out + (size_t)chunk * nb * img;
    uint32_t v0 = lane, v1 = unit, v2 = chunk;
    for (int b = 0; b < nb; ++b, ob += img) {
#pragma unroll
        for (int j = 0; j < NQ; ++j)
            if (ok[j]) { uint32_t *op = reinterpret_cast<uint32_t *>(ob + off[j]); op[0] = v0 + b; op[1] = v1; op[2] = v2; }
    }
}
// the 64 x 4 tile pattern written with k_units' code: strips of 64 x 4 pixels in raster order, 4 consecutive strips per block
// (mode 0: 17 strips per row as k_tiles; mode 1: 4 strips per block but blocks never wrap to the next row: 5 blocks per row, the last one thin)
__global__ void __launch_bounds__(256) k_strips(uint8_t *__restrict__ out, int pitch, int nb, int nchunks, int mode, int ng)
{
    const uint32_t id = blockIdx.x;
    const uint32_t x8 = id & 7u, k = id >> 3, chunk = x8 + 8u * (k / ng), grp = k % ng;
    if ((int)chunk >= nchunks) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int sx, sy;
    if (mode == 0) { const int strip = (int)grp * 4 + wave; sx = strip % 17; sy = strip / 17; }
    else if (mode == 1) { sx = (int)(grp % 5) * 4 + wave; sy = (int)(grp / 5); }
    else if (mode == 2) { const int c = (int)(grp % 5), first = c < 2 ? 4 * c : 8 + 3 * (c - 2), n = c < 2 ? 4 : 3; sx = wave < n ? first + wave : 99; sy = (int)(grp / 5); }   // 4,4,3,3,3 strips
    else if (mode == 3) { sx = (int)(grp % 4) * 4 + wave; sy = (int)(grp / 4); }          // columns 0..1023 only, 4 blocks per row
    else { const int strip = (int)grp * 4 + wave; sx = strip % 16; sy = strip / 16; if (sy >= 270) sx = 99; }   // mode 4: = mode 3 (16 strips per row in raster order)
    const int x = sx * 64 + 4 * (lane & 15), y = sy * 4 + (lane >> 4);
    const bool ok = x < BW && y < BH;
    const size_t img = (size_t)pitch * BH * 3;
    const uint32_t off = ((uint32_t)y * pitch + x) * 3;
    uint8_t *ob = out + (size_t)chunk * nb * img;
    uint32_t v0 = lane, v1 = grp, v2 = chunk;
    for (int b = 0; b < nb; ++b, ob += img)
        if (ok) { uint32_t *op = reinterpret_cast<uint32_t *>(ob + off); op[0] = v0 + b; op[1] = v1; op[2] = v2; }
}
static void run_strips(uint8_t *out, int pitch, int nb, int mode, const char *name)
{
    const int ng = mode == 0 ? (17 * 270 + 3) / 4 : mode >= 3 ? 4 * 270 : 5 * 270, nchunks = BATCH / nb;
    const unsigned grid = ng * ((nchunks + 7) / 8 * 8);
    const double bytes = (double)BATCH * BW * BH * 3;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&] { hipLaunchKernelGGL(k_strips, dim3(grid), dim3(256), 0, 0, out, pitch, nb, nchunks, mode, ng); };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-60s %7.3f ms  %7.1f GB/s\n", name, ms, bytes / ms * 1e-6);
}

template <int LQ, int NQ, int LW = LQ> static void run_units(uint8_t *out, int pitch, int nb, int rr, const char *name, int bw_eff = BW)
{
    constexpr int W = 4 << LQ, H = NQ * 4 * (64 >> LW) / (1 << (LQ - LW));
    const int units_x = (BW + W - 1) / W, units_y = (BH + H - 1) / H, nunits = units_x * units_y, nchunks = BATCH / nb;
    const unsigned grid = nunits * ((nchunks + 7) / 8 * 8);
    const double bytes = (double)BATCH * BW * BH * 3;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&] { hipLaunchKernelGGL((k_units<LQ, NQ, LW>), dim3(grid), dim3(256), 0, 0, out, pitch, units_x, nunits, nb, nchunks, rr, bw_eff); };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-60s %7.3f ms  %7.1f GB/s\n", name, ms, bytes / ms * 1e-6);
}

template <typename F> static float timeit(F launch)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5;
}

// round 6: the FORMAT of a unit's wave-store, separated from its shape.  Same rectangles, rows dealt round-robin, buffer stores with the
// product's cache policy (AUX: 0 default, 3 = nt | sc0 streaming):
//   FMT 0  one buffer_store_dwordx3 per lane (12 B: lanes straddle every 64-byte sector boundary, 64 / 12 = 5.33)          [product, rounds 3 - 5]
//   FMT 1  the 4 x 12 B of a lane quad repacked with quad_perm DPP moves into 3 x 16 B; buffer_store_dwordx4 from 48 lanes: the same row run,
//          every 64-byte sector written by exactly 4 lanes; lane quads with a masked lane fall back to dwordx3 (second, wave-uniformly skipped store)
//   FMT 2  the same repack through LDS (3 ds_write_b32 at stride 12 B, one ds_read_b128)
//   FMT 3  global_store_dwordx3 (what rounds 3 - 5 measured in this tool)
// Every dword written = its own dword index inside the image + the frame: main() checks the images byte for byte after each variant.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
template <int LQ, int NQ, int FMT, int AUX>
__global__ void __launch_bounds__(256) k_units_fmt(uint8_t *__restrict__ out, int pitch, int units_x, int nunits, int nb, int nchunks, int bw_eff)
{
    __shared__ __attribute__((aligned(16))) uint32_t xch[4][64 * 3];
    const uint32_t ng = (uint32_t)nunits, id = blockIdx.x;
    const uint32_t x8 = id & 7u, k = id >> 3, chunk = x8 + 8u * (k / ng), unit = k % ng;
    if ((int)chunk >= nchunks) return;
    constexpr int W = 4 << LQ, RPS = 64 >> LQ, H = NQ * 4 * RPS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & 3;
    const int ux = (int)(unit % units_x) * W, uy = (int)(unit / units_x) * H;
    const uint32_t img = (uint32_t)pitch * BH * 3;
    constexpr uint32_t NONE = 0x80000000u;
    uint32_t off12[NQ], off16[NQ], base[NQ];
    bool anyp[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int sidx = j * 4 + wave;
        const int x = ux + 4 * (lane & ((1 << LQ) - 1)), y = uy + sidx * RPS + (lane >> LQ);
        const bool ok = x < bw_eff && y < BH;
        const uint32_t off = ((uint32_t)y * pitch + x) * 3;
        base[j] = off / 4;
        const uint64_t m = __builtin_amdgcn_ballot_w64(ok);
        const bool full = ((m >> (lane & ~3)) & 0xfull) == 0xfull;
        if (FMT == 1 || FMT == 2) {
            off16[j] = (full && q < 3) ? off - 12u * q + 16u * q : NONE;
            off12[j] = (!full && ok) ? off : NONE;
            anyp[j] = __builtin_amdgcn_ballot_w64(!full && ok) != 0;
        } else {
            off12[j] = ok ? off : NONE;
        }
    }
    for (int b = 0; b < nb; ++b) {
        uint8_t *ob = out + (size_t)(chunk * nb + b) * img;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ob, 0, img, 0x00020000u);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const uint32_t v0 = base[j] + b, v1 = base[j] + 1 + b, v2 = base[j] + 2 + b;
            if (FMT == 0) {
                __builtin_amdgcn_raw_buffer_store_b96(u32x3{v0, v1, v2}, ro, (int)off12[j], 0, AUX);
            } else if (FMT == 3) {
                if (off12[j] != NONE) { uint32_t *op = reinterpret_cast<uint32_t *>(ob + off12[j]); op[0] = v0; op[1] = v1; op[2] = v2; }
            } else {
                u32x4 o;
                if (FMT == 1) {
                    // the next lane's three dwords (quad_perm [1, 2, 3, 3]), then the funnel S[q .. q + 3] of S = {v0, v1, v2, n0, n1, n2}
                    const uint32_t n0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v0, 0xF9, 0xf, 0xf, false);
                    const uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v1, 0xF9, 0xf, 0xf, false);
                    const uint32_t n2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v2, 0xF9, 0xf, 0xf, false);
                    o.x = q == 0 ? v0 : (q == 1 ? v1 : v2);
                    o.y = q == 0 ? v1 : (q == 1 ? v2 : n0);
                    o.z = q == 0 ? v2 : (q == 1 ? n0 : n1);
                    o.w = q == 0 ? n0 : (q == 1 ? n1 : n2);
                } else {
                    uint32_t *xw = xch[wave];
                    xw[lane * 3] = v0; xw[lane * 3 + 1] = v1; xw[lane * 3 + 2] = v2;
                    __builtin_amdgcn_wave_barrier();
                    o = *reinterpret_cast<const u32x4 *>(xw + (lane & ~3) * 3 + q * 4 - (q == 3 ? 4 : 0));
                    __builtin_amdgcn_wave_barrier();
                }
                __builtin_amdgcn_raw_buffer_store_b128(o, ro, (int)off16[j], 0, AUX);
                if (anyp[j]) __builtin_amdgcn_raw_buffer_store_b96(u32x3{v0, v1, v2}, ro, (int)off12[j], 0, AUX);
            }
        }
    }
}
static bool check_units_image(const uint8_t *d_out, int pitch, int bw_eff, int nb)
{
    // frames 0 and nb + 1 (chunk 1, frame 1) of the batch: dword i of frame f holds i + (f % nb)
    const size_t img = (size_t)pitch * BH * 3;
    static uint32_t *h = nullptr;
    if (!h) h = (uint32_t *)malloc((size_t)1088 * BH * 3);
    for (int f : {0, nb + 1}) {
        CK(hipMemcpy(h, d_out + (size_t)f * img, img, hipMemcpyDeviceToHost));
        for (int y = 0; y < BH; ++y)
            for (int i = 0; i < bw_eff * 3 / 4; ++i) {
                const size_t d = ((size_t)y * pitch * 3) / 4 + i;
                if (h[d] != (uint32_t)d + (uint32_t)(f % nb)) { printf("    MISMATCH frame %d row %d dword %d: %08x != %08x\n", f, y, i, h[d], (uint32_t)d + f % nb); return false; }
            }
    }
    return true;
}
template <int LQ, int NQ, int FMT, int AUX> static void run_fmt(uint8_t *out, int pitch, int nb, const char *name, int bw_eff = BW)
{
    constexpr int W = 4 << LQ, H = NQ * 4 * (64 >> LQ);
    const int units_x = (std::max(BW, bw_eff) + W - 1) / W, units_y = (BH + H - 1) / H, nunits = units_x * units_y, nchunks = BATCH / nb;
    const unsigned grid = nunits * ((nchunks + 7) / 8 * 8);
    const double bytes = (double)BATCH * BW * BH * 3;
    CK(hipMemset(out, 0xee, (size_t)BATCH * 1088 * BH * 3));
    auto launch = [&] { hipLaunchKernelGGL((k_units_fmt<LQ, NQ, FMT, AUX>), dim3(grid), dim3(256), 0, 0, out, pitch, units_x, nunits, nb, nchunks, bw_eff); };
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 3; ++rep) { const float ms = timeit(launch); best = std::min(best, ms); sum += ms; }
    const bool ok = check_units_image(out, pitch, bw_eff, nb);
    static const char *fn[] = {"buffer dwordx3 x 64 lanes", "DPP repack, dwordx4 x 48", "LDS repack, dwordx4 x 48", "global dwordx3 x 64"};
    printf("%-44s %-26s aux %d  %7.3f ms (mean %7.3f)  %7.1f GB/s  %s\n", name, fn[FMT], AUX, best, sum / 3, bytes / best * 1e-6, ok ? "bytes ok" : "WRONG BYTES");
}
template <int LQ, int NQ> static void run_fmt_all(uint8_t *out, int pitch, int nb, const char *name, int bw_eff = BW)
{
    run_fmt<LQ, NQ, 0, 3>(out, pitch, nb, name, bw_eff);
    run_fmt<LQ, NQ, 1, 3>(out, pitch, nb, name, bw_eff);
    run_fmt<LQ, NQ, 2, 3>(out, pitch, nb, name, bw_eff);
    run_fmt<LQ, NQ, 0, 0>(out, pitch, nb, name, bw_eff);
    run_fmt<LQ, NQ, 1, 0>(out, pitch, nb, name, bw_eff);
    run_fmt<LQ, NQ, 3, 0>(out, pitch, nb, name, bw_eff);
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

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "format")) {   // round 6: store format A/B on the unit shapes (profiles/r06/store_format.log)
        const size_t alloc = (size_t)BATCH * 1088 * BH * 3;
        uint8_t *out;
        CK(hipMalloc(&out, alloc));
        for (int pass = 0; pass < 2; ++pass) {
            run_fmt_all<6, 4>(out, 1088, 16, "units 256x16, 16 fr/blk, pitch 1088, pad written", 1088);
            run_fmt_all<6, 4>(out, 1080, 16, "units 256x16, 16 fr/blk, dense");
            run_fmt_all<5, 4>(out, 1088, 16, "units 128x32, 16 fr/blk, pitch 1088, pad written", 1088);
            run_fmt_all<4, 4>(out, 1088, 16, "units 64x64, 16 fr/blk, pitch 1088, pad written", 1088);
            run_fmt_all<6, 1>(out, 1088, 16, "units 256x4, 16 fr/blk, pitch 1088, pad written", 1088);
            run_fmt_all<3, 4>(out, 1088, 16, "units 32x128, 16 fr/blk, pitch 1088, pad written", 1088);
        }
        float ms = timeit([&] { hipLaunchKernelGGL(k_linear, dim3(2048), dim3(256), 0, 0, (uint4 *)out, (size_t)BATCH * BW * BH * 3 / 16); });
        printf("%-44s %7.3f ms  %7.1f GB/s\n", "linear stream, 16 B per lane", ms, (double)BATCH * BW * BH * 3 / ms * 1e-6);
        return 0;
    }
    if (argc > 1) {   // "strips": only the strip assignments, dense, 8 frames per block (counter runs: 6 dispatches per mode, modes 0 1 3 4 in this order)
        const size_t alloc = (size_t)BATCH * 1088 * BH * 3;
        uint8_t *out;
        CK(hipMalloc(&out, alloc));
        CK(hipMemset(out, 0, alloc));
        run_strips(out, 1080, 8, 0, "strips 64x4, raster order (17 per row), dense");
        run_strips(out, 1080, 8, 1, "strips 64x4, 5 blocks per row, dense");
        run_strips(out, 1080, 8, 3, "strips 64x4, columns 0..1023, 4 blocks per row, dense");
        run_strips(out, 1080, 8, 4, "strips 64x4, columns 0..1023, 16 strips per row in raster order, dense");
        return 0;
    }
    const size_t bytes = (size_t)BATCH * BW * BH * 3, alloc = (size_t)BATCH * 1088 * BH * 3;
    uint8_t *out;
    CK(hipMalloc(&out, alloc));
    CK(hipMemset(out, 0, alloc));
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
    // round 3: unit-shaped stores (profiles/r03/store_pattern_units.log)
    run_units<6, 4>(out, 1080, 16, 1, "units 256x16, 16 frames/block, rows round-robin, dense");
    run_units<6, 4>(out, 1088, 16, 1, "units 256x16, 16 frames/block, rows round-robin, pitch 1088");
    run_units<6, 4>(out, 1088, 16, 0, "units 256x16, 16 frames/block, row bands, pitch 1088");
    run_units<6, 4>(out, 1088, 8, 1, "units 256x16, 8 frames/block, pitch 1088");
    run_units<6, 2>(out, 1088, 16, 1, "units 256x8, 16 frames/block, pitch 1088");
    run_units<6, 1>(out, 1088, 16, 1, "units 256x4, 16 frames/block, pitch 1088");
    run_units<5, 4>(out, 1088, 16, 1, "units 128x32, 16 frames/block, pitch 1088");
    run_units<4, 4>(out, 1088, 16, 1, "units 64x64, 16 frames/block, pitch 1088");
    run_units<4, 1>(out, 1088, 16, 1, "units 64x16, 16 frames/block, pitch 1088");
    run_units<3, 4>(out, 1088, 16, 1, "units 32x128, 16 frames/block, pitch 1088");
    run_units<6, 4, 5>(out, 1088, 16, 1, "units 256x16, wave-store = 2 rows x 384 B, pitch 1088");
    run_units<6, 4, 4>(out, 1088, 16, 1, "units 256x16, wave-store = 4 rows x 192 B, pitch 1088");
    run_units<6, 4, 3>(out, 1088, 16, 1, "units 256x16, wave-store = 8 rows x 96 B, pitch 1088");
    run_units<6, 4, 4>(out, 1080, 16, 1, "units 256x16, wave-store = 4 rows x 192 B, dense");
    run_units<6, 4, 3>(out, 1080, 16, 1, "units 256x16, wave-store = 8 rows x 96 B, dense");
    run_units<6, 4, 4>(out, 1088, 16, 0, "units 256x16, 4 rows x 192 B, a wave owns a strip (rr 0), pitch 1088");
    run_units<6, 4, 4>(out, 1088, 8, 1, "units 256x16, 4 rows x 192 B, 8 frames/block, pitch 1088");
    run_units<5, 4, 4>(out, 1088, 16, 1, "units 128x32, 4 rows x 192 B, pitch 1088");
    run_units<6, 2, 4>(out, 1088, 16, 1, "units 256x8, 4 rows x 192 B, pitch 1088");
    run_units<6, 1, 4>(out, 1088, 16, 1, "units 256x4, 4 rows x 192 B, pitch 1088");
    run_units<6, 1, 4>(out, 1080, 8, 1, "units 256x4, 4 rows x 192 B, 8 frames/block, dense");
    run_tiles<16>(out, 8, 1, "64x4 tiles, 8 frames/block, XCD chunks (again)");
    run_tiles<16>(out, 16, 1, "64x4 tiles, 16 frames/block, XCD chunks");
    run_tiles<8>(out, 8, 1, "32x8 tiles, 8 frames/block, XCD chunks (again)");
    run_units<6, 4>(out, 1088, 16, 1, "units 256x16, 16 frames/block, pitch 1088 (again)");
    run_units<6, 4>(out, 1088, 16, 1, "units 256x16, 16 frames/block, pitch 1088, padding columns written too", 1088);
    run_units<6, 4, 4>(out, 1088, 16, 1, "units 256x16, 4 rows x 192 B, pitch 1088, padding written too", 1088);
    run_units<6, 1, 4>(out, 1088, 8, 1, "units 256x4, 4 rows x 192 B, 8 frames/block, pitch 1088, padding written", 1088);
    run_units<6, 4>(out, 1088, 8, 1, "units 256x16, 8 frames/block, pitch 1088, padding written", 1088);
    run_units<6, 1, 4>(out, 1080, 8, 1, "units 256x4, 4 rows x 192 B, 8 frames/block, dense, columns 0..1023 only", 1024);
    run_units<6, 1, 4>(out, 1080, 8, 1, "units 256x4, 4 rows x 192 B, 8 frames/block, dense (again)");
    run_units<4, 1>(out, 1080, 8, 1, "units 64x16 (4 waves stacked), 8 frames/block, dense");
    run_units<4, 1>(out, 1088, 8, 1, "units 64x16 (4 waves stacked), 8 frames/block, pitch 1088, padding written", 1088);
    run_units<4, 4>(out, 1088, 8, 1, "units 64x64, 8 frames/block, pitch 1088, padding written", 1088);
    run_units<4, 4>(out, 1088, 16, 1, "units 64x64, 16 frames/block, pitch 1088, padding written", 1088);
    run_units<5, 4>(out, 1088, 16, 1, "units 128x32, 16 frames/block, pitch 1088, padding written", 1088);
    run_tiles<16>(out, 8, 1, "64x4 tiles, 8 frames/block, XCD chunks (third time)");
    run_strips(out, 1080, 8, 0, "strips 64x4, 4 per block in raster order (= k_tiles), dense");
    run_strips(out, 1080, 8, 1, "strips 64x4, 4 per block, 5 blocks per row (= k_units 256x4), dense");
    run_strips(out, 1088, 8, 0, "strips 64x4, raster order, pitch 1088");
    run_strips(out, 1088, 8, 1, "strips 64x4, 5 blocks per row, pitch 1088");
    run_strips(out, 1088, 16, 0, "strips 64x4, raster order, pitch 1088, 16 frames/block");
    run_strips(out, 1080, 8, 2, "strips 64x4, 5 blocks per row of 4,4,3,3,3 strips, dense");
    run_strips(out, 1080, 8, 3, "strips 64x4, columns 0..1023 only (4 blocks per row), dense  [5 % fewer bytes]");
    run_strips(out, 1080, 8, 0, "strips 64x4, raster order, dense (again)");
    run_strips(out, 1080, 8, 1, "strips 64x4, 5 blocks per row, dense (again)");
    return 0;
}
