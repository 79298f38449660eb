// microbench.hip -- gather-instruction cost model on gfx950 (scratch tool, not part of the product).
// Measures cycles per wave-level memory instruction for the access shapes the stitch kernel can use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct __attribute__((packed, aligned(1))) PU2 { uint32_t x, y; };
struct __attribute__((packed, aligned(4))) AU2 { uint32_t x, y; };
struct __attribute__((packed, aligned(4))) AU3 { uint32_t x, y, z; };
struct __attribute__((packed, aligned(4))) AU4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) PU3 { uint32_t x, y, z; };

constexpr int ROW = 3840;
constexpr int NL = 8;   // loads per iteration per thread

// All loads are issued through inline asm so that the instruction and its alignment are exactly what is named.
// MODE: 0 dwordx2 unaligned | 1 dwordx2 @4B | 2 dwordx2 @8B | 3 dword @4B | 4 dwordx3 @4B | 5 dwordx4 @4B | 6 dwordx4 @16B
//       7 ubyte | 8 dwordx3 unaligned | 9 dword unaligned
// PAT : 0 tile gather (16 rows x 4 lanes, 9-byte lane pitch) | 1 fully coalesced | 2 64x4 tile (4 rows x 16 lanes)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int PAT>
__global__ void __launch_bounds__(256) k_global(const uint8_t *__restrict__ base, int iters, uint32_t *__restrict__ sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t *p = base + (size_t)(blockIdx.x % 64) * 65536 * 4 + wave * 48;  // distinct windows per block
    uint32_t offs[NL];
    for (int j = 0; j < NL; ++j) {
        uint32_t o;
        if (PAT == 0) o = (lane / 4 + (j >> 2)) * ROW + (lane % 4) * 9 + (j & 3) * 3 + 1;
        else if (PAT == 2) o = (lane / 16 + (j >> 2)) * ROW + (lane % 16) * 9 + (j & 3) * 3 + 1;
        else o = lane * 16 + j * 1024 + 1;
        if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5) o &= ~3u;
        if (MODE == 2) o &= ~7u;
        if (MODE == 6) o &= ~15u;
        offs[j] = o;
    }
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint8_t *q = p + (it & 3) * 16;
        u32x4 v[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const uint8_t *a = q + offs[j];
            v[j] = u32x4{0, 0, 0, 0};
            if (MODE == 0 || MODE == 1 || MODE == 2) { u32x2 t; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(t) : "v"(a)); v[j].x = t.x; v[j].y = t.y; }
            else if (MODE == 3 || MODE == 9) { uint32_t t; asm volatile("global_load_dword %0, %1, off" : "=v"(t) : "v"(a)); v[j].x = t; }
            else if (MODE == 4 || MODE == 8) { u32x3 t; asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(t) : "v"(a)); v[j].x = t.x; v[j].y = t.y; v[j].z = t.z; }
            else if (MODE == 5 || MODE == 6) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(a)); }
            else if (MODE == 7) { uint32_t t; asm volatile("global_load_ubyte %0, %1, off" : "=v"(t) : "v"(a)); v[j].x = t; }
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
        for (int j = 0; j < NL; ++j) acc ^= v[j].x + v[j].y + v[j].z + v[j].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// LDS modes: 0 ds_read_b64 @8B | 1 ds_read2_b32 @4B | 2 ds_read_b32 | 3 ds_read_b64 @4B (misaligned by 4) | 4 ds_read_b128 @16B
//            5 ds_read_u8 | 6 ds_read_b64 @1B (byte-misaligned) | 7 ds_read_b96 @4B
template <int MODE>
__global__ void __launch_bounds__(256) k_lds(int iters, uint32_t *__restrict__ sink)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[32768 + 64];
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) ((uint32_t *)lds)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds;
    uint32_t offs[NL];
    for (int j = 0; j < NL; ++j) {
        // 4 B per texel staging: rows of 80 texels (320 B), lane (lx, ly) reads texel ~ (3*lx + j, ly + j/4)
        uint32_t o = wave * 8192 + ((lane / 4 + (j >> 2)) * 80 + (lane % 4) * 3 + (j & 3)) * 4;
        if (MODE == 0) o &= ~7u;
        if (MODE == 3) o = (o & ~7u) + 4;
        if (MODE == 4) o &= ~15u;
        if (MODE == 5 || MODE == 6) o += 1;
        offs[j] = lbase + o;
    }
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t q = (it & 3) * 16;
        u32x4 v[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const uint32_t a = q + offs[j];
            v[j] = u32x4{0, 0, 0, 0};
            if (MODE == 0 || MODE == 3 || MODE == 6) { u32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(a)); v[j].x = t.x; v[j].y = t.y; }
            else if (MODE == 1) { u32x2 t; asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(t) : "v"(a)); v[j].x = t.x; v[j].y = t.y; }
            else if (MODE == 2) { uint32_t t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(a)); v[j].x = t; }
            else if (MODE == 4) { asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(a)); }
            else if (MODE == 5) { uint32_t t; asm volatile("ds_read_u8 %0, %1" : "=v"(t) : "v"(a)); v[j].x = t; }
            else if (MODE == 7) { u32x3 t; asm volatile("ds_read_b96 %0, %1" : "=v"(t) : "v"(a)); v[j].x = t.x; v[j].y = t.y; v[j].z = t.z; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
        for (int j = 0; j < NL; ++j) acc ^= v[j].x + v[j].y + v[j].z + v[j].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}


// ---- VALU op throughput: 16 independent ops per iteration, 8 waves/SIMD resident -------------------------------
// OP: 0 v_dot4_u32_u8 | 1 v_dot2_u32_u16 | 2 v_alignbyte_b32 | 3 v_perm_b32 | 4 v_lshl_or_b32 | 5 v_mad_u32_u24 | 6 v_add_u32
//     7 v_pk_mad_u16 | 8 v_cvt_f32_ubyte0 | 9 v_mul_f32
template <int OP>
__global__ void __launch_bounds__(256) k_valu(int iters, uint32_t *__restrict__ sink)
{
    uint32_t r[16];
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 17 + i;
    const uint32_t w = threadIdx.x | 0x01020304u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(w));
            else if (OP == 1) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(w));
            else if (OP == 2) asm volatile("v_alignbyte_b32 %0, %1, %0, 1" : "+v"(r[i]) : "v"(w));
            else if (OP == 3) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(w));
            else if (OP == 4) asm volatile("v_lshl_or_b32 %0, %1, 16, %0" : "+v"(r[i]) : "v"(w));
            else if (OP == 5) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(w));
            else if (OP == 6) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[i]) : "v"(w));
            else if (OP == 7) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(w));
            else if (OP == 8) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(r[i]) : "v"(w));
            else if (OP == 9) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i]) : "v"(w));
        }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; ++i) acc ^= r[i];
    if (acc == 0x12345678u) sink[0] = acc;
}

// ---- store patterns: one store per wave per "frame", streaming over frames like the stitch kernel --------------
// PAT 0: 16x16 px tile (16 rows x 48 B), dwordx3 | 1: 64x4 px tile (4 rows x 192 B), dwordx3 | 2: 32x8 tile, dwordx3
//     3: fully contiguous 768 B per wave (dwordx3) | 4: contiguous 1 KB per wave, dwordx4
template <int PAT>
__global__ void __launch_bounds__(256) k_store(uint8_t *__restrict__ out, int tiles_x, int nframes, size_t img_bytes, int bw)
{
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int LX = PAT == 0 ? 4 : PAT == 1 ? 16 : 8, LY = 64 / LX;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    size_t ooff;
    if (PAT <= 2) ooff = ((size_t)(ty * LY + lane / LX) * bw + (tx * LX + lane % LX) * 4) * 3;
    else if (PAT == 3) ooff = (size_t)tile * 768 + lane * 12;
    else ooff = (size_t)tile * 1024 + lane * 16;
    if (ooff + 16 > img_bytes) return;
    uint32_t v = tile * 64 + lane;
    for (int b = 0; b < nframes; ++b) {
        uint32_t *op = reinterpret_cast<uint32_t *>(out + (size_t)b * img_bytes + ooff);
        if (PAT == 4) { op[0] = v; op[1] = v + 1; op[2] = v + 2; op[3] = v + 3; }
        else { op[0] = v; op[1] = v + 1; op[2] = v + 2; }
        v += 7;
    }
}

template <typename F>
static void timeit(const char *name, F launch, int iters, int blocks)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per CU: blocks * 4 waves * iters * NL / 256 CUs
    const double winst_per_cu = (double)blocks * 4 * iters * NL / 256.0;
    printf("%-44s %8.3f ms  %7.1f ns per wave-instr per CU  (~%5.1f clk @2.1GHz)\n", name, ms, ms * 1e6 / winst_per_cu,
           ms * 1e6 / winst_per_cu * 2.1);
}

int main()
{
    uint8_t *buf; uint32_t *sink;
    CK(hipMalloc(&buf, 64 * 65536 * 4 + (1 << 20)));
    CK(hipMemset(buf, 0x5a, 64 * 65536 * 4 + (1 << 20)));
    CK(hipMalloc(&sink, 64));
    const int iters = 2000, blocks = 256 * 8;
#define G(M, P, N) timeit(N, [&] { hipLaunchKernelGGL((k_global<M, P>), dim3(blocks), dim3(256), 0, 0, buf, iters, sink); }, iters, blocks)
    G(0, 0, "global unaligned dwordx2, tile16x16 gather");
    G(1, 0, "global dwordx2 @4B, tile16x16 gather");
    G(2, 0, "global dwordx2 @8B, tile16x16 gather");
    G(3, 0, "global dword @4B, tile16x16 gather");
    G(4, 0, "global dwordx3 @4B, tile16x16 gather");
    G(5, 0, "global dwordx4 @4B, tile16x16 gather");
    G(6, 0, "global dwordx4 @16B, tile16x16 gather");
    G(7, 0, "global ubyte, tile16x16 gather");
    G(8, 0, "global unaligned dwordx3, tile16x16 gather");
    G(9, 0, "global unaligned dword, tile16x16 gather");
    G(0, 2, "global unaligned dwordx2, tile64x4 gather");
    G(1, 2, "global dwordx2 @4B, tile64x4 gather");
    G(3, 2, "global dword @4B, tile64x4 gather");
    G(5, 2, "global dwordx4 @4B, tile64x4 gather");
    G(0, 1, "global unaligned dwordx2, coalesced 16B pitch");
    G(2, 1, "global dwordx2 @8B, coalesced 16B pitch");
    G(6, 1, "global dwordx4 @16B, coalesced");
    G(3, 1, "global dword, coalesced 16B pitch");
#define S(M, N) timeit(N, [&] { hipLaunchKernelGGL((k_lds<M>), dim3(blocks), dim3(256), 0, 0, iters, sink); }, iters, blocks)
    S(0, "lds b64 @8B gather");
    S(1, "lds 2xb32 @4B gather");
    S(2, "lds b32 gather");
    S(3, "lds b64 @4B (misaligned) gather");
    S(4, "lds b128 @16B gather");
    S(5, "lds u8 gather");
    S(6, "lds b64 @1B (byte-misaligned) gather");
    S(7, "lds b96 @4B gather");
#define V(M, N) timeit(N, [&] { hipLaunchKernelGGL((k_valu<M>), dim3(blocks), dim3(256), 0, 0, iters, sink); }, iters * 2, blocks)
    printf("-- VALU (ns column = per wave-instr per CU; x4 for per-SIMD)\n");
    V(0, "v_dot4_u32_u8"); V(1, "v_dot2_u32_u16"); V(2, "v_alignbyte_b32"); V(3, "v_perm_b32"); V(4, "v_lshl_or_b32");
    V(5, "v_mad_u32_u24"); V(6, "v_add_u32"); V(7, "v_pk_mad_u16"); V(8, "v_cvt_f32_ubyte0"); V(9, "v_mul_f32");
    {
        const int bw = 1088, bh = 1088, nframes = 64;
        const size_t img = (size_t)bw * bh * 3;
        uint8_t *out; CK(hipMalloc(&out, img * nframes));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto st = [&](const char *name, auto launch, double bytes) {
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms * 1e-6);
        };
        const double bytes = (double)img * nframes;
        st("store dwordx3 16x16 tiles (16 x 48 B)", [&] { hipLaunchKernelGGL((k_store<0>), dim3(68 * 68 / 4), dim3(256), 0, 0, out, 68, nframes, img, bw); }, bytes);
        st("store dwordx3 64x4 tiles (4 x 192 B)", [&] { hipLaunchKernelGGL((k_store<1>), dim3(17 * 272 / 4), dim3(256), 0, 0, out, 17, nframes, img, bw); }, bytes);
        st("store dwordx3 32x8 tiles (8 x 96 B)", [&] { hipLaunchKernelGGL((k_store<2>), dim3(34 * 136 / 4), dim3(256), 0, 0, out, 34, nframes, img, bw); }, bytes);
        st("store dwordx3 contiguous 768 B / wave", [&] { hipLaunchKernelGGL((k_store<3>), dim3((unsigned)(img / 768 / 4)), dim3(256), 0, 0, out, 1, nframes, img, bw); }, bytes);
        st("store dwordx4 contiguous 1 KB / wave", [&] { hipLaunchKernelGGL((k_store<4>), dim3((unsigned)(img / 1024 / 4)), dim3(256), 0, 0, out, 1, nframes, img, bw); }, bytes);
    }
    return 0;
}
