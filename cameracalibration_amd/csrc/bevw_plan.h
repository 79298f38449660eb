// bevw_plan.h -- the tile-plan schedule of BevGenerator.__call__ (BEVW_SCHED_TILE_PLAN).
//
// Idea.  Every table the per-pixel schedule reads per frame (4 LUTs = 24 B/px, 4 masks = 4 B/px) is static for a
// calibration, and after masking a BEV pixel has at most two contributing cameras (one inside a trapezoid, two on a
// seam / in a blend overlap, none under the car).  bevw_build compiles LUT + masks into a CONTRIBUTOR PLAN: per BEV
// pixel up to two 8-byte entries {byte offset of the 2x2 footprint inside the 4-camera frame set, 5+5 bit
// fractions, u8 mask/weight, camera}.  One wave64 owns a (4*LX) x (64/LX) pixel tile (4 horizontally adjacent
// pixels per lane = one 12-byte store), loads its slice of the plan ONCE into registers, and then loops over the
// frames of its batch chunk: per frame and pixel the only memory traffic is two unaligned 8-byte gathers (top and
// bottom texel pair, base address in SGPRs + per-lane 32-bit offset) and the coalesced 12-byte store.
// Address, weight and mask arithmetic is hoisted out of the batch loop; the fixed-point bilinear runs on
// v_dot4_u32_u8 / v_dot2_u32_u16.
//
// Layout.  plan[tile][slot 0..7][lane 0..63] (8 B each, so every plan load is a fully coalesced 512 B wave access);
// slots 0..3 = first contributor of the lane's 4 pixels, 4..7 = second contributor (only read when the tile header
// says some lane has one).  Blocks are 4 waves = 4 consecutive tiles (shared L1 lines on the same CU); the block
// index is mapped so that all tiles of one batch chunk run on the same XCD (block id % 8), which keeps a frame's
// source rows in one L2 while neighbouring tiles consume them.
#pragma once
#include "bevw_kernels.h"

namespace bevw {

constexpr uint32_t kMetaValid = 1u << 20;
constexpr uint32_t kMetaSlow = 1u << 21;   // footprint touches the frame border (or the 8-byte read would overrun)
constexpr uint32_t kHdrSecond = 1u;        // some lane of the tile has a second contributor
constexpr uint32_t kHdrSlow = 2u;          // some entry of the tile needs the per-tap border path
constexpr uint32_t kHdrEmpty = 4u;         // no contributor at all (car rectangle): tile is zero + car
constexpr int kPlanLX = 4;                 // lanes along x -> 16 x 16 pixel tiles

struct Plan {
    void *entries = nullptr;     // uint2[ntiles][8][64]
    void *hdr = nullptr;         // uint32[ntiles]
    void *psums = nullptr;       // uint32[batch][ntiles][3]  (balance: per-tile channel sums)
    size_t psums_cap = 0;
    int *d_max = nullptr;
    int fw = 0, fh = 0, bw = 0, bh = 0;
    int tiles_x = 0, tiles_y = 0, ntiles = 0;
    int max_contrib = 0;
    bool usable = false;
};

struct __attribute__((packed, aligned(1))) PackedU2 { uint32_t x, y; };
__device__ __forceinline__ uint2 load_u2_unaligned(const uint8_t *p)
{
    const PackedU2 v = *reinterpret_cast<const PackedU2 *>(p);
    return make_uint2(v.x, v.y);
}

// ---------------------------------------------------------------------------------------------------------------
// plan compiler: one wave per tile
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_plan_build(StitchTables T, int fw, int fh, int bw, int bh, int tiles_x, int ntiles,
                             uint2 *__restrict__ plan, uint32_t *__restrict__ hdr, int *__restrict__ max_contrib)
{
    constexpr int LX = kPlanLX, LY = 64 / LX;
    const int tile = blockIdx.x, lane = threadIdx.x;
    if (tile >= ntiles) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int x0 = (tx * LX + lane % LX) * 4, y = ty * LY + lane / LX;
    const uint32_t frame_bytes = (uint32_t)fw * fh * 3;
    uint32_t flags = 0;
    int worst = 0;
    for (int j = 0; j < 4; ++j) {
        uint2 e[2] = {make_uint2(0, 0), make_uint2(0, 0)};
        int count = 0;
        const int x = x0 + j;
        if (x < bw && y < bh) {
            const size_t o = (size_t)y * bw + x;
            for (int c = 0; c < 4; ++c) {
                const uint32_t m = T.mask[c][o];
                if (m == 0) continue;
                const int sx = T.lut1[c][o * 2], sy = T.lut1[c][o * 2 + 1];
                const uint32_t code = T.lut2[c][o] & (kQTab2 - 1);
                if (sx >= fw || sx + 1 < 0 || sy >= fh || sy + 1 < 0) continue;  // whole footprint outside: adds 0
                uint32_t meta = code | (m << 10) | ((uint32_t)c << 18) | kMetaValid, off;
                const bool interior = (unsigned)sx < (unsigned)(fw > 1 ? fw - 1 : 0) && (unsigned)sy < (unsigned)(fh > 1 ? fh - 1 : 0);
                const uint32_t toff = ((uint32_t)sy * fw + sx) * 3;
                if (interior && toff + (uint32_t)fw * 3 + 8 <= frame_bytes) {
                    off = (uint32_t)c * frame_bytes + toff;
                } else {
                    meta |= kMetaSlow;
                    off = ((uint32_t)sx & 0xffffu) | ((uint32_t)sy << 16);
                    flags |= kHdrSlow;
                }
                if (count < 2) e[count] = make_uint2(off, meta);
                ++count;
            }
        }
        if (count > 1) flags |= kHdrSecond;
        if (count > 0) flags |= 8u;
        worst = max(worst, count);
        plan[((size_t)tile * 8 + j) * 64 + lane] = e[0];
        plan[((size_t)tile * 8 + 4 + j) * 64 + lane] = e[1];
    }
    // wave-wide OR / max
    for (int off = 32; off > 0; off >>= 1) {
        flags |= __shfl_xor(flags, off, 64);
        worst = max(worst, __shfl_xor(worst, off, 64));
    }
    if (lane == 0) {
        uint32_t hflags = flags & (kHdrSecond | kHdrSlow);
        if (!(flags & 8u)) hflags |= kHdrEmpty;
        hdr[tile] = hflags;
        atomicMax(max_contrib, worst);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// per-entry evaluation
// ---------------------------------------------------------------------------------------------------------------
struct EntryRegs {
    uint32_t off;   // byte offset of the footprint in the 4-camera set (or packed sx|sy for slow entries)
    uint32_t meta;
    uint32_t wx;    // (32-fx) | fx << 24   (bytes 0 and 3 of a 4-byte window = the two x taps of one channel)
    uint32_t wy;    // (32-fy) | fy << 16
    float wf;       // blend weight float32(mask / 255.0)
};

__device__ __forceinline__ EntryRegs decode_entry(uint2 e, bool blend)
{
    EntryRegs r;
    r.off = e.x;
    r.meta = e.y;
    const uint32_t fx = e.y & 31, fy = (e.y >> 5) & 31;
    const bool fast = (e.y & kMetaValid) && !(e.y & kMetaSlow);
    r.wx = fast ? ((32 - fx) | (fx << 24)) : 0u;   // zero weights make an absent entry contribute exactly 0
    r.wy = (32 - fy) | (fy << 16);
    if (!fast) r.off = (e.y & kMetaSlow) ? e.x : 0u;
    r.wf = blend ? blend_weight_f32((int)((e.y >> 10) & 255)) : 1.f;
    return r;
}

// Fixed-point bilinear of one interior footprint from its two 8-byte rows (bytes: B0 G0 R0 B1 G1 R1 x x).
// out_c = ((p00*ax + p01*fx) * ay + (p10*ax + p11*fx) * fy + 512) >> 10  -- the separable form of
// (sum p * (wx*wy) + 512) >> 10, exact in integers.
__device__ __forceinline__ void bilinear_rows(uint2 r0, uint2 r1, uint32_t wx, uint32_t wy, int v[3])
{
    const uint32_t g0 = __builtin_amdgcn_alignbyte(r0.y, r0.x, 1), q0 = __builtin_amdgcn_alignbyte(r0.y, r0.x, 2);
    const uint32_t g1 = __builtin_amdgcn_alignbyte(r1.y, r1.x, 1), q1 = __builtin_amdgcn_alignbyte(r1.y, r1.x, 2);
    const uint32_t hb0 = __builtin_amdgcn_udot4(r0.x, wx, 0u, false), hb1 = __builtin_amdgcn_udot4(r1.x, wx, 0u, false);
    const uint32_t hg0 = __builtin_amdgcn_udot4(g0, wx, 0u, false), hg1 = __builtin_amdgcn_udot4(g1, wx, 0u, false);
    const uint32_t hr0 = __builtin_amdgcn_udot4(q0, wx, 0u, false), hr1 = __builtin_amdgcn_udot4(q1, wx, 0u, false);
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    union { uint32_t u; us2 v; } pb, pg, pr, w;
    pb.u = hb0 | (hb1 << 16); pg.u = hg0 | (hg1 << 16); pr.u = hr0 | (hr1 << 16); w.u = wy;
    v[0] = (int)(__builtin_amdgcn_udot2(pb.v, w.v, 512u, false) >> 10);
    v[1] = (int)(__builtin_amdgcn_udot2(pg.v, w.v, 512u, false) >> 10);
    v[2] = (int)(__builtin_amdgcn_udot2(pr.v, w.v, 512u, false) >> 10);
}

template <bool BLEND, bool BAL>
__device__ __forceinline__ void eval_entry(const uint8_t *__restrict__ fb, const EntryRegs &e, uint32_t row_bytes, int fw,
                                           int fh, uint32_t frame_bytes, bool tile_slow, const int *__restrict__ fdeltas,
                                           const int *sdiv, const int *hdiv, int v[3])
{
    const int cam = (e.meta >> 18) & 3;
    if (tile_slow && (e.meta & kMetaSlow)) {
        const int sx = (int)(int16_t)(e.off & 0xffffu), sy = (int)(int16_t)(e.off >> 16);
        remap_u8c3_px<BAL>(fb + (size_t)cam * frame_bytes, fw, fh, sx, sy, e.meta & 1023u, v, BAL ? fdeltas[cam] : 0, sdiv, hdiv);
    } else if (!BAL) {
        const uint2 r0 = load_u2_unaligned(fb + e.off), r1 = load_u2_unaligned(fb + e.off + row_bytes);
        bilinear_rows(r0, r1, e.wx, e.wy, v);
    } else {
        const uint2 r0 = load_u2_unaligned(fb + e.off), r1 = load_u2_unaligned(fb + e.off + row_bytes);
        int t[4][3] = {{(int)(r0.x & 255), (int)((r0.x >> 8) & 255), (int)((r0.x >> 16) & 255)},
                       {(int)(r0.x >> 24), (int)(r0.y & 255), (int)((r0.y >> 8) & 255)},
                       {(int)(r1.x & 255), (int)((r1.x >> 8) & 255), (int)((r1.x >> 16) & 255)},
                       {(int)(r1.x >> 24), (int)(r1.y & 255), (int)((r1.y >> 8) & 255)}};
        const int delta = fdeltas[cam];
#pragma unroll
        for (int q = 0; q < 4; ++q) luminance_shift_px(t[q][0], t[q][1], t[q][2], delta, sdiv, hdiv);
        const int ax = e.wx & 255, fx = e.wx >> 24, ay = e.wy & 65535, fy = e.wy >> 16;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            v[k] = ((t[0][k] * ax + t[1][k] * fx) * ay + (t[2][k] * ax + t[3][k] * fx) * fy + 512) >> 10;
    }
    if (BLEND) { v[0] = blend_mul(v[0], e.wf); v[1] = blend_mul(v[1], e.wf); v[2] = blend_mul(v[2], e.wf); }
}

struct PlanArgs {
    const uint8_t *frames;
    const uint2 *plan;
    const uint32_t *hdr;
    const int *deltas;
    const HsvTables *tab;
    const uint8_t *car;
    uint32_t *psums;
    uint8_t *out;
    int fw, fh, bw, bh;
    int tiles_x, ntiles, ngroups;
    int batch, nb, nchunks, xcd_affine;
};

// grid = ngroups * (nchunks rounded up to a multiple of 8 when xcd_affine), block = 256 (4 tiles)
template <bool BLEND, bool BAL>
__global__ void __launch_bounds__(256) k_stitch_plan(PlanArgs a)
{
    constexpr int LX = kPlanLX, LY = 64 / LX;
    __shared__ int sdiv[BAL ? 256 : 1], hdiv[BAL ? 256 : 1];
    if (BAL) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) { sdiv[i] = a.tab->sdiv[i]; hdiv[i] = a.tab->hdiv[i]; }
        __syncthreads();
    }
    const uint32_t id = blockIdx.x;
    uint32_t chunk, group;
    if (a.xcd_affine) {
        const uint32_t xcd = id & 7u, k = id >> 3;
        chunk = xcd + 8u * (k / (uint32_t)a.ngroups);
        group = k % (uint32_t)a.ngroups;
    } else {
        chunk = id / (uint32_t)a.ngroups;
        group = id % (uint32_t)a.ngroups;
    }
    if ((int)chunk >= a.nchunks) return;
    const int lane = threadIdx.x & 63;
    const int tile = (int)group * 4 + (threadIdx.x >> 6);
    if (tile >= a.ntiles) return;

    const uint32_t hdr = __builtin_amdgcn_readfirstlane(a.hdr[tile]);
    const bool second = hdr & kHdrSecond, tile_slow = hdr & kHdrSlow;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int x0 = (tx * LX + lane % LX) * 4, y = ty * LY + lane / LX;
    const bool inimg = x0 < a.bw && y < a.bh;
    const uint32_t frame_bytes = (uint32_t)a.fw * a.fh * 3, row_bytes = (uint32_t)a.fw * 3;
    const size_t set_bytes = (size_t)frame_bytes * 4, img_bytes = (size_t)a.bw * a.bh * 3;
    const uint32_t ooff = ((uint32_t)y * a.bw + x0) * 3;

    EntryRegs e0[4], e1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        e0[j] = decode_entry(a.plan[((size_t)tile * 8 + j) * 64 + lane], BLEND);
        e1[j] = decode_entry(second ? a.plan[((size_t)tile * 8 + 4 + j) * 64 + lane] : make_uint2(0, 0), BLEND);
    }
    uint32_t car0 = 0, car1 = 0, car2 = 0;
    if (!BAL && a.car != nullptr && inimg) {
        const uint32_t *cp = reinterpret_cast<const uint32_t *>(a.car + ooff);
        car0 = cp[0]; car1 = cp[1]; car2 = cp[2];
    }
    const bool car_any = __builtin_amdgcn_ballot_w64((car0 | car1 | car2) != 0) != 0;

    const int b_begin = (int)chunk * a.nb, b_end = min(a.batch, b_begin + a.nb);
#pragma unroll 2
    for (int b = b_begin; b < b_end; ++b) {
        const uint8_t *fb = a.frames + (size_t)b * set_bytes;
        const int *fdeltas = BAL ? a.deltas + b * 4 : nullptr;
        int px[4][3];
        if (hdr & kHdrEmpty) {
#pragma unroll
            for (int j = 0; j < 4; ++j) px[j][0] = px[j][1] = px[j][2] = 0;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                eval_entry<BLEND, BAL>(fb, e0[j], row_bytes, a.fw, a.fh, frame_bytes, tile_slow, fdeltas, sdiv, hdiv, px[j]);
            }
            if (second) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int w[3];
                    eval_entry<BLEND, BAL>(fb, e1[j], row_bytes, a.fw, a.fh, frame_bytes, tile_slow, fdeltas, sdiv, hdiv, w);
                    px[j][0] = min(255, px[j][0] + w[0]); px[j][1] = min(255, px[j][1] + w[1]); px[j][2] = min(255, px[j][2] + w[2]);
                }
            }
        }
        if (BAL) {
            // per-tile channel sums of the pre-gain BEV (color_balance means, surroundBEV.py:44-47)
            unsigned s0 = 0, s1 = 0, s2 = 0;
            if (inimg) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { s0 += px[j][0]; s1 += px[j][1]; s2 += px[j][2]; }
            }
            s0 = wave_sum_u32(s0); s1 = wave_sum_u32(s1); s2 = wave_sum_u32(s2);
            if (lane == 0) {
                uint32_t *ps = a.psums + ((size_t)b * a.ntiles + tile) * 3;
                ps[0] = s0; ps[1] = s1; ps[2] = s2;
            }
        } else if (car_any) {
            const uint32_t cw[3] = {car0, car1, car2};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int bi = j * 3 + k;
                    px[j][k] = min(255, px[j][k] + (int)((cw[bi >> 2] >> ((bi & 3) * 8)) & 255));
                }
        }
        if (inimg) {
            const uint32_t d0 = (uint32_t)px[0][0] | ((uint32_t)px[0][1] << 8) | ((uint32_t)px[0][2] << 16) | ((uint32_t)px[1][0] << 24);
            const uint32_t d1 = (uint32_t)px[1][1] | ((uint32_t)px[1][2] << 8) | ((uint32_t)px[2][0] << 16) | ((uint32_t)px[2][1] << 24);
            const uint32_t d2 = (uint32_t)px[2][2] | ((uint32_t)px[3][0] << 8) | ((uint32_t)px[3][1] << 16) | ((uint32_t)px[3][2] << 24);
            uint32_t *op = reinterpret_cast<uint32_t *>(a.out + (size_t)b * img_bytes + ooff);
            op[0] = d0; op[1] = d1; op[2] = d2;
        }
    }
}

// psums[b][tile][3] -> chsums[b][3] ; grid = batch, block = 256
__global__ void k_reduce_psums(const uint32_t *__restrict__ psums, int ntiles, unsigned long long *__restrict__ chsums)
{
    const uint32_t *p = psums + (size_t)blockIdx.x * ntiles * 3;
    unsigned long long acc[3] = {0, 0, 0};
    for (int t = threadIdx.x; t < ntiles; t += blockDim.x) { acc[0] += p[t * 3]; acc[1] += p[t * 3 + 1]; acc[2] += p[t * 3 + 2]; }
    __shared__ unsigned long long part[3][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        unsigned long long s = wave_sum_u64(acc[k]);
        if (lane == 0) part[k][wv] = s;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long t = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += part[threadIdx.x][i];
        chsums[blockIdx.x * 3 + threadIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static inline void plan_release(Plan &p)
{
    if (p.entries) (void)hipFree(p.entries);
    if (p.hdr) (void)hipFree(p.hdr);
    if (p.psums) (void)hipFree(p.psums);
    if (p.d_max) (void)hipFree(p.d_max);
    p = Plan();
}

// returns 0 or a hipError_t cast to a negative-free int; the caller turns it into a bevw_status
static inline hipError_t plan_build_impl(Plan &p, hipStream_t st, const StitchTables &T, int fw, int fh, int bw, int bh)
{
    plan_release(p);
    p.fw = fw; p.fh = fh; p.bw = bw; p.bh = bh;
    p.tiles_x = (bw + 4 * kPlanLX - 1) / (4 * kPlanLX);
    p.tiles_y = (bh + (64 / kPlanLX) - 1) / (64 / kPlanLX);
    p.ntiles = p.tiles_x * p.tiles_y;
    hipError_t e;
    if ((e = hipMalloc(&p.entries, (size_t)p.ntiles * 8 * 64 * sizeof(uint2))) != hipSuccess) return e;
    if ((e = hipMalloc(&p.hdr, (size_t)p.ntiles * sizeof(uint32_t))) != hipSuccess) return e;
    if ((e = hipMalloc((void **)&p.d_max, sizeof(int))) != hipSuccess) return e;
    if ((e = hipMemsetAsync(p.d_max, 0, sizeof(int), st)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_plan_build, dim3(p.ntiles), dim3(64), 0, st, T, fw, fh, bw, bh, p.tiles_x, p.ntiles,
                       static_cast<uint2 *>(p.entries), static_cast<uint32_t *>(p.hdr), p.d_max);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(&p.max_contrib, p.d_max, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    // 12-byte stores need 4-byte aligned pixel quads: bw % 4 == 0 makes every row and every image start aligned
    p.usable = p.max_contrib <= 2 && (bw % 4 == 0);
    return hipSuccess;
}

static inline hipError_t plan_stitch_impl(Plan &p, hipStream_t st, const uint8_t *d_frames, int batch, bool blend, bool balance,
                                          const int *d_deltas, const HsvTables *d_tab, const uint8_t *d_car,
                                          unsigned long long *d_chsums, uint8_t *d_out, int nb_override)
{
    hipError_t e;
    PlanArgs a;
    a.frames = d_frames; a.plan = static_cast<const uint2 *>(p.entries); a.hdr = static_cast<const uint32_t *>(p.hdr);
    a.deltas = d_deltas; a.tab = d_tab; a.car = d_car; a.out = d_out;
    a.fw = p.fw; a.fh = p.fh; a.bw = p.bw; a.bh = p.bh;
    a.tiles_x = p.tiles_x; a.ntiles = p.ntiles; a.ngroups = (p.ntiles + 3) / 4;
    a.batch = batch;
    // frames per block: enough chunks to give each of the 8 XCDs whole chunks, otherwise one frame per chunk
    int nb = nb_override > 0 ? nb_override : 16;
    if (batch < 8 * nb) nb = batch >= 8 ? batch / 8 : 1;
    a.nb = nb;
    a.nchunks = (batch + nb - 1) / nb;
    a.xcd_affine = a.nchunks >= 8 ? 1 : 0;
    const int chunks_padded = a.xcd_affine ? ((a.nchunks + 7) / 8) * 8 : a.nchunks;
    if (balance) {
        const size_t need = (size_t)batch * p.ntiles * 3 * sizeof(uint32_t);
        if (need > p.psums_cap) {
            if (p.psums) (void)hipFree(p.psums);
            p.psums = nullptr; p.psums_cap = 0;
            if ((e = hipMalloc(&p.psums, need)) != hipSuccess) return e;
            p.psums_cap = need;
        }
    }
    a.psums = static_cast<uint32_t *>(p.psums);
    const dim3 grid((unsigned)(a.ngroups * chunks_padded)), block(256);
    if (blend && balance) hipLaunchKernelGGL((k_stitch_plan<true, true>), grid, block, 0, st, a);
    else if (blend) hipLaunchKernelGGL((k_stitch_plan<true, false>), grid, block, 0, st, a);
    else if (balance) hipLaunchKernelGGL((k_stitch_plan<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_stitch_plan<false, false>), grid, block, 0, st, a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (balance) {
        hipLaunchKernelGGL(k_reduce_psums, dim3(batch), dim3(256), 0, st, a.psums, p.ntiles, d_chsums);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace bevw
