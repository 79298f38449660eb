// bevw_plan.h -- the tile-plan schedule of BevGenerator.__call__ (BEVW_SCHED_TILE_PLAN).
//
// Idea.  Every table the per-pixel schedule reads per frame (4 LUTs = 24 B/px, 4 masks = 4 B/px) is static for a
// calibration, and after masking a BEV pixel has at most two contributing cameras (one inside a trapezoid, two on a
// seam / in a blend overlap, none under the car).  bevw_build compiles LUT + masks into a PLAN, and the per-frame kernels
// do no table look-up and no projection arithmetic at all:
//   * the UNIT schedule (bevw_unit.h): the BEV is cut into rectangles whose source footprint fits one 32 KB LDS patch; a block
//     of 4 waves stages the texels of a unit once per frame (bevw_pair.h) and interpolates every pixel from the patch.  Every
//     pixel whose footprints lie inside the frames is a unit pixel: dense far field, sparse near field, seams, blend overlaps,
//     the empty car rectangle.  ONE launch per step (k_plan_units);
//   * the per-tap tile kernel (k_stitch_plan, this file): one wave per 32 x 8 pixel base tile, every 2x2 footprint read straight
//     from the frames.  It serves the base tiles the units do not take (footprints on the frame border) and every geometry the
//     units cannot serve (frame widths that are not a multiple of 4 pixels, frame sets that are not 4-byte aligned), and it is
//     the kernel with the luminance round trip per tap (BEVW_BAL_MODE=0).
// Rounds 2 - 3 carried three more schedules (per-wave pair classes, 64 x 32 block tiles, seam tiles) beside the units; round 4
// retired them (profiles/r04/retire.md has the before / after).
//
// Base tiles: 32 x 8 pixels, lane l of a wave owns the pixel quad (l % 8, l / 8) -- 4 horizontally adjacent pixels, one 12-byte
// store.  plan[tile][slot 0..7][lane] (8 B each): slots 0..3 = first contributor of the lane's 4 pixels, 4..7 = second.
// Blocks are dealt so that all tiles / units of one batch chunk run on the same XCD (block id % 8): a frame's source rows stay
// in one L2 while neighbouring units consume them.
#pragma once
#include <algorithm>
#include <vector>

#include "bevw_kernels.h"

namespace bevw {

constexpr uint32_t kMetaValid = 1u << 20;
constexpr uint32_t kMetaSlow = 1u << 21;   // footprint touches the frame border (or the 8-byte read would overrun)
constexpr uint32_t kHdrSecond = 1u;        // some lane of the tile has a second contributor
constexpr uint32_t kHdrSlow = 2u;          // some entry of the tile needs the per-tap border path
constexpr uint32_t kHdrEmpty = 4u;         // no contributor at all (car rectangle): tile is zero + car
constexpr uint32_t kHdrBlock = 1024u;      // base tile is owned by a unit (bevw_unit.h): not on the per-tap kernel's list
constexpr int kPlanLX = 8;                 // lanes along x -> 32 x 8 pixel base tiles
constexpr int kPlanLY = 64 / kPlanLX;

struct Plan {
    void *entries = nullptr;     // uint2[ntiles][8][64]
    void *hdr = nullptr;         // uint32[ntiles]
    void *psums = nullptr;       // uint32[batch][ntiles][3]  (balance: per-tile channel sums)
    size_t psums_cap = 0;
    int psums_layout = -1;       // entries per frame the zeros of psums were laid out for (plan_stitch_impl)
    // destination widths that are not a multiple of 4 pixels: the kernels' 12-byte stores need dword-aligned pixel quads,
    // so they write rows of `pitch` = bw rounded up to 4 pixels into pad_out and k_plan_unpad compacts them (one more
    // pass over the output instead of the per-pixel schedule)
    int pitch = 0;
    bool out_pitched = false;    // the caller's output images have rows of `pitch` pixels themselves (bevw_set_output_pitch): no scratch, no compaction
    void *pad_out = nullptr, *pad_car = nullptr;
    size_t pad_cap = 0;
    int *d_max = nullptr;
    int fw = 0, fh = 0, bw = 0, bh = 0;
    int tiles_x = 0, tiles_y = 0, ntiles = 0;
    int ncams = 4;
    void *groups = nullptr;      // uint32[n_groups]: byte offsets (inside the frame set) of the sampled 4-texel groups
    int n_groups = 0;
    bool band_ok = false;        // the sampled-group list exists (balance schedule 1)
    int max_contrib = 0;
    bool usable = false;
    void *list_slow = nullptr;   // base tiles no unit owns (frame-border footprints; everything when there are no units)
    int n_slow = 0;
    // unit schedule (bevw_unit.h): k-d partition compiled on the host
    void *un_desc = nullptr, *un_entries = nullptr, *un_gsrc = nullptr;
    void *un_gsrc_compact = nullptr;         // the units' group lists for the compact scratch of the balance schedule (unit_gsrc_compact); nullptr: not usable
    size_t compact_stride = 0;               // bytes between the compact scratch copies of consecutive frame sets
    void *list_un_all = nullptr;             // every unit in partition order, class in bits 28..31
    int n_un_all = 0;
    int n_un[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // units per class (diagnostics)
    int n_unit_tiles = 0;                    // base tiles the units own
    size_t un_lines = 0, un_sectors = 0;     // request arithmetic of the partition (per frame)
    int un_skew = 0;
};

struct __attribute__((packed, aligned(1))) PackedU2 { uint32_t x, y; };
__device__ __forceinline__ uint2 load_u2_unaligned(const uint8_t *p)
{
    const PackedU2 v = *reinterpret_cast<const PackedU2 *>(p);
    return make_uint2(v.x, v.y);
}

// 12 bytes at a 4-byte aligned address: one global_load_dwordx3 / global_store_dwordx3
struct __attribute__((packed, aligned(4))) AlignedU3 { uint32_t x, y, z; };

// 4 pixel dwords (B | G << 8 | R << 16) -> the 12 output bytes
__host__ __device__ __forceinline__ void pack_pixels(const uint32_t P[4], uint32_t &d0, uint32_t &d1, uint32_t &d2)
{
    d0 = px_perm(P[1], P[0], 0x04020100u);
    d1 = px_perm(P[2], P[1], 0x05040201u);
    d2 = px_perm(P[3], P[2], 0x06050402u);
}

// saturating add of the car sprite (12 bytes c0 c1 c2 at the lane's store position) onto 4 pixel dwords
__host__ __device__ __forceinline__ void add_car(uint32_t P[4], uint32_t c0, uint32_t c1, uint32_t c2)
{
    const uint32_t C[4] = {c0 & 0xffffffu, px_alignbyte(c1, c0, 3) & 0xffffffu, px_alignbyte(c2, c1, 2) & 0xffffffu, c2 >> 8};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t sb = (P[j] & 255u) + (C[j] & 255u), sg = ((P[j] >> 8) & 255u) + ((C[j] >> 8) & 255u);
        const uint32_t sr = ((P[j] >> 16) & 255u) + ((C[j] >> 16) & 255u);
        const uint32_t b = sb < 255u ? sb : 255u, g = sg < 255u ? sg : 255u, r = sr < 255u ? sr : 255u;
        P[j] = b | (g << 8) | (r << 16);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// plan compiler: one wave per base tile
// ---------------------------------------------------------------------------------------------------------------
static __global__ void k_plan_build(StitchTables T, int fw, int fh, int bw, int bh, int tiles_x, int ntiles, int ncams,
                                    uint2 *__restrict__ plan, uint32_t *__restrict__ hdr, int *__restrict__ max_contrib)
{
    const int tile = blockIdx.x, lane = threadIdx.x;
    if (tile >= ntiles) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int x0 = (tx * kPlanLX + lane % kPlanLX) * 4, y = ty * kPlanLY + lane / kPlanLX;
    const uint32_t frame_bytes = (uint32_t)fw * fh * 3;
    uint32_t flags = 0;
    int worst = 0;
    for (int j = 0; j < 4; ++j) {
        uint2 e[2] = {make_uint2(0, 0), make_uint2(0, 0)};
        int count = 0;
        const int x = x0 + j;
        if (x < bw && y < bh) {
            const size_t o = (size_t)y * bw + x;
            for (int c = 0; c < ncams; ++c) {
                const uint32_t m = T.mask[c][o];
                if (m == 0) continue;
                const int sx = T.lut1[c][o * 2], sy = T.lut1[c][o * 2 + 1];
                const uint32_t code = T.lut2[c][o] & (kQTab2 - 1);
                if (sx >= fw || sx + 1 < 0 || sy >= fh || sy + 1 < 0) continue;  // whole footprint outside: adds 0
                uint32_t meta = code | (m << 10) | ((uint32_t)c << 18) | kMetaValid, off;
                const bool interior = (unsigned)sx < (unsigned)(fw > 1 ? fw - 1 : 0) && (unsigned)sy < (unsigned)(fh > 1 ? fh - 1 : 0);
                const uint32_t toff = ((uint32_t)sy * fw + sx) * 3;
                if (interior && (toff & ~3u) + (uint32_t)fw * 3 + 12 <= frame_bytes) {  // aligned 12-byte row reads stay inside
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

// Bitmap of the 4-texel groups (12 bytes, 4-byte aligned because fw % 4 == 0) that the plan samples, over the 4-camera
// frame set: bit index = (cam * fh + y) * (fw / 4) + x / 4.  The balance schedule converts exactly these groups of every
// raw frame (luminance round trip) instead of whole frames or bounding boxes (the LUT quirk makes the right camera's
// bounding box start at texel (0,0)).
static __global__ void k_plan_touch(const uint2 *__restrict__ plan, int ntiles, int fw, int fh, uint32_t *__restrict__ bitmap)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ntiles * 8 * 64) return;
    const uint2 e = plan[i];
    if (!(e.y & kMetaValid)) return;
    const int cam = (e.y >> 18) & 3;
    int sx, sy;
    if (e.y & kMetaSlow) { sx = (int)(int16_t)(e.x & 0xffffu); sy = (int)(int16_t)(e.x >> 16); }
    else {
        const uint32_t t = e.x - (uint32_t)cam * (uint32_t)fw * fh * 3;
        sy = (int)(t / ((uint32_t)fw * 3)); sx = (int)((t % ((uint32_t)fw * 3)) / 3);
    }
    const int gw = fw / 4;
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
            const int x = sx + dx, y = sy + dy;
            if ((unsigned)x >= (unsigned)fw || (unsigned)y >= (unsigned)fh) continue;
            const uint32_t bit = ((uint32_t)cam * fh + y) * gw + x / 4;
            atomicOr(&bitmap[bit >> 5], 1u << (bit & 31));
        }
}

// luminance_balance (surroundBEV.py:57-79) applied to the sampled texel groups of every raw frame, into the COMPACT scratch (bevw_unit.h:
// unit_gsrc_compact): slot i of a frame set's scratch = HSV2BGR(sat(V + delta)) of group groups[i].  One lane = one group = 4 texels
// (12 bytes, one dwordx3 each way); groups[] holds byte offsets inside the frame set in ascending order, so neighbouring lanes read
// neighbouring memory, and they WRITE consecutive 12-byte slots: whole sectors, 3 KB per block and trip.  (Rounds 2 - 4 wrote the groups back
// at their own offsets into a scratch frame set of full size: 12-byte pieces with holes, every run of groups ending in a partially written
// sector, and the units then re-read that sparse layout.)  A block takes kLumTrips x 256 consecutive groups of one frame set (the 4 KB of
// tables in LDS are paid once per block); the offsets and the texels of all its trips are requested before the first one is converted.
// grid = xcd_frame_grid(ceil(ngroups / (256 kLumTrips)), batch); block = 256.
constexpr int kLumTrips = 8;
static __global__ void __launch_bounds__(256) k_lum_groups(const uint8_t *__restrict__ frames, uint8_t *__restrict__ scratch, size_t set_bytes,
                                                            size_t scratch_stride, uint32_t frame_bytes, const uint32_t *__restrict__ groups, int ngroups,
                                                            const int *__restrict__ deltas, const HsvTables *__restrict__ tab,
                                                            uint32_t blocks_per_frame, uint32_t nframes)
{
    __shared__ HsvTables hsv;
    __shared__ int cam_delta[4];
    uint32_t frame, blk;
    if (!xcd_frame_map(blockIdx.x, blocks_per_frame, nframes, frame, blk)) return;   // grid: xcd_frame_grid()
    hsv_tables_to_lds(hsv, tab);
    if (threadIdx.x < 4) cam_delta[threadIdx.x] = deltas[frame * 4 + threadIdx.x];   // k_lum_delta's (a kernel of its own: bevwarp.hip luminance_stats)
    const uint8_t *fin = frames + (size_t)frame * set_bytes;
    uint8_t *fout = scratch + (size_t)frame * scratch_stride;
    const int g0 = (int)blk * (kLumTrips * 256) + (int)threadIdx.x;
    uint32_t goff[kLumTrips];
    AlignedU3 v[kLumTrips];
#pragma unroll
    for (int t = 0; t < kLumTrips; ++t) goff[t] = g0 + t * 256 < ngroups ? groups[g0 + t * 256] : 0u;
#pragma unroll
    for (int t = 0; t < kLumTrips; ++t) v[t] = *reinterpret_cast<const AlignedU3 *>(fin + goff[t]);   // (past the list: group 0 once more, not stored)
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kLumTrips; ++t) {
        const int gi = g0 + t * 256;
        if (gi >= ngroups) break;
        const uint32_t w[3] = {v[t].x, v[t].y, v[t].z};
        const int cam = (int)(goff[t] >= frame_bytes) + (int)(goff[t] >= 2u * frame_bytes) + (int)(goff[t] >= 3u * frame_bytes);
        const int delta = cam_delta[cam];
        // the 4 texels of the group as dwords (byte 3 is ignored), shifted, and packed back into the 12 bytes
        uint32_t P[4] = {w[0], __builtin_amdgcn_alignbyte(w[1], w[0], 3), __builtin_amdgcn_alignbyte(w[2], w[1], 2), w[2] >> 8};
#pragma unroll
        for (int k = 0; k < 4; ++k) P[k] = luminance_shift_bgr(P[k], delta, hsv);
        uint32_t o[3];
        pack_pixels(P, o[0], o[1], o[2]);
        AlignedU3 ov; ov.x = o[0]; ov.y = o[1]; ov.z = o[2];
        *reinterpret_cast<AlignedU3 *>(fout + (size_t)gi * 12) = ov;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// per-entry evaluation (the per-tap tile kernel)
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
                                           const HsvTables &hsv, int v[3])
{
    const int cam = (e.meta >> 18) & 3;
    if (tile_slow && (e.meta & kMetaSlow)) {
        const int sx = (int)(int16_t)(e.off & 0xffffu), sy = (int)(int16_t)(e.off >> 16);
        remap_u8c3_px<BAL>(fb + (size_t)cam * frame_bytes, fw, fh, sx, sy, e.meta & 1023u, v, BAL ? fdeltas[cam] : 0, &hsv);
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
        for (int q = 0; q < 4; ++q) luminance_shift_px(t[q][0], t[q][1], t[q][2], delta, hsv);
        const int ax = e.wx & 255, fx = e.wx >> 24, ay = e.wy & 65535, fy = e.wy >> 16;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            v[k] = ((t[0][k] * ax + t[1][k] * fx) * ay + (t[2][k] * ax + t[3][k] * fx) * fy + 512) >> 10;
    }
    if (BLEND) { v[0] = blend_mul(v[0], e.wf); v[1] = blend_mul(v[1], e.wf); v[2] = blend_mul(v[2], e.wf); }
}

struct UnitDesc;   // bevw_unit.h

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
    int pitch;                   // pixels per row of `out` / `car`: bw, or bw rounded up to 4 (padded scratch, see plan_stitch_impl)
    int tiles_x, ntiles, ngroups;
    int ncams;                   // images per frame set: 4 for BevGenerator, 1 for a plain cv2.remap
    int batch, nb, nchunks, xcd_affine;
    const uint32_t *tile_list;   // per-tap kernel: base-tile indices (nullptr: every tile), ngroups = ceil(nlist / 4); units: unit ids, ngroups = nlist
    int nlist;
    // units (bevw_unit.h)
    const UnitDesc *un_desc;
    const uint4 *un_entries;     // one uint4 per lane and quad slot: the plan entries of the lane's 4 pixels
    const uint32_t *un_gsrc;
    int un_skew;                 // unit_skew constant of the plan
    uint32_t set_stride;         // units: bytes between consecutive frame sets (0: fw * fh * 3 * ncams; else the compact scratch, bevw_unit.h)
    // channel sums (balance): psums[frame][nsum][3], ONE writer per entry and frame -- unit u writes entry u, the per-tap kernel entry
    // sum_base + its position in the tile list (sum_base = number of units; 0 when it serves every tile)
    int nsum, sum_base;
};

// Block index -> (batch chunk, tile group).  Blocks are dealt to the 8 XCDs round-robin (block id % 8), and each XCD has
// its own L2, so the map decides what an L2 sees:
//   xcd_affine 1: an XCD owns whole batch chunks (all tiles of frames b0..b0+nb), neighbouring tiles share its L2
//   xcd_affine 2: as 1, the chunks of an XCD interleaved unit by unit (BEVW_PLAN_XCDMAP=2)
//   xcd_affine 0: plain chunk-major order (few chunks)
__device__ __forceinline__ bool plan_block_map(const PlanArgs &a, uint32_t id, uint32_t &chunk, uint32_t &group)
{
    const uint32_t ng = (uint32_t)a.ngroups;
    if (a.xcd_affine == 1) {
        const uint32_t xcd = id & 7u, k = id >> 3;
        chunk = xcd + 8u * (k / ng);
        group = k % ng;
    } else if (a.xcd_affine == 2) {
        // the chunks of an XCD interleaved: unit g of every chunk the XCD owns runs back to back, so the unit's plan slice (entries +
        // group list) is fetched from memory once per XCD and served from its L2 to the other chunks
        const uint32_t xcd = id & 7u, k = id >> 3, cpx = ((uint32_t)a.nchunks + 7u) >> 3;
        chunk = xcd + 8u * (k % cpx);
        group = k / cpx;
    } else {
        chunk = id / ng;
        group = id % ng;
    }
    return (int)chunk < a.nchunks;
}

// The per-tap tile kernel.  grid: blocks of 4 waves = 4 base tiles of the list; one tile per wave.
// LUM: luminance round trip per fetched texel (raw frames); SUMS: emit per-tile channel sums and leave the car to the gain pass
// (two waves per SIMD: the 128-VGPR budget of rounds 1 - 3 spilled 0.5 - 1.2 KB per lane to scratch -- tools/kernel_resources.sh)
template <bool BLEND, bool LUM, bool SUMS = LUM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) k_stitch_plan(PlanArgs a)
{
    constexpr bool BAL = LUM;
    __shared__ __attribute__((aligned(16))) uint32_t hsv_words[BAL ? sizeof(HsvTables) / 4 : 1];   // (no LDS for the variants without the luminance round trip)
    const HsvTables &hsv = *reinterpret_cast<const HsvTables *>(hsv_words);
    if (BAL) {
        hsv_tables_to_lds(*reinterpret_cast<HsvTables *>(hsv_words), a.tab);
        __syncthreads();
    }
    uint32_t chunk, group;
    if (!plan_block_map(a, blockIdx.x, chunk, group)) return;
    const int lane = threadIdx.x & 63;
    const int slot = (int)group * (int)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (slot >= a.nlist) return;
    const int tile = a.tile_list ? (int)__builtin_amdgcn_readfirstlane(a.tile_list[slot]) : slot;

    const uint32_t hdr = __builtin_amdgcn_readfirstlane(a.hdr[tile]);
    const bool second = hdr & kHdrSecond, tile_slow = hdr & kHdrSlow;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int x0 = (tx * kPlanLX + lane % kPlanLX) * 4, y = ty * kPlanLY + lane / kPlanLX;
    const bool inimg = x0 < a.bw && y < a.bh;
    const uint32_t frame_bytes = (uint32_t)a.fw * a.fh * 3, row_bytes = (uint32_t)a.fw * 3;
    const size_t set_bytes = (size_t)frame_bytes * a.ncams, img_bytes = (size_t)a.pitch * a.bh * 3;
    const uint32_t ooff = ((uint32_t)y * a.pitch + x0) * 3;

    EntryRegs e0[4], e1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        e0[j] = decode_entry(a.plan[((size_t)tile * 8 + j) * 64 + lane], BLEND);
        e1[j] = decode_entry(second ? a.plan[((size_t)tile * 8 + 4 + j) * 64 + lane] : make_uint2(0, 0), BLEND);
    }
    uint32_t car0 = 0, car1 = 0, car2 = 0;
    if (!SUMS && a.car != nullptr && inimg) {
        const uint32_t *cp = reinterpret_cast<const uint32_t *>(a.car + ooff);
        car0 = cp[0]; car1 = cp[1]; car2 = cp[2];
    }
    const bool car_any = __builtin_amdgcn_ballot_w64((car0 | car1 | car2) != 0) != 0;

    const int b_begin = (int)chunk * a.nb, b_end = min(a.batch, b_begin + a.nb);
#pragma unroll 1
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
                eval_entry<BLEND, BAL>(fb, e0[j], row_bytes, a.fw, a.fh, frame_bytes, tile_slow, fdeltas, hsv, px[j]);
            }
            if (second) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int w[3];
                    eval_entry<BLEND, BAL>(fb, e1[j], row_bytes, a.fw, a.fh, frame_bytes, tile_slow, fdeltas, hsv, w);
                    px[j][0] = min(255, px[j][0] + w[0]); px[j][1] = min(255, px[j][1] + w[1]); px[j][2] = min(255, px[j][2] + w[2]);
                }
            }
        }
        if (SUMS) {
            // per-tile channel sums of the pre-gain BEV (color_balance means, surroundBEV.py:44-47); pixels outside
            // the image have no plan entry and contribute 0.  One entry per listed tile (PlanArgs::sum_base).
            unsigned s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { s0 += px[j][0]; s1 += px[j][1]; s2 += px[j][2]; }
            s0 = wave_sum_u32(s0); s1 = wave_sum_u32(s1); s2 = wave_sum_u32(s2);
            if (lane == 0) {
                uint32_t *ps = a.psums + ((size_t)b * a.nsum + a.sum_base + slot) * 3;
                ps[0] = s0; ps[1] = s1; ps[2] = s2;
            }
        }
        uint32_t P[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) P[j] = (uint32_t)px[j][0] | ((uint32_t)px[j][1] << 8) | ((uint32_t)px[j][2] << 16);
        if (!SUMS && car_any) add_car(P, car0, car1, car2);
        if (inimg) {
            uint32_t d0, d1, d2;
            pack_pixels(P, d0, d1, d2);
            uint32_t *op = reinterpret_cast<uint32_t *>(a.out + (size_t)b * img_bytes + ooff);
            op[0] = d0; op[1] = d1; op[2] = d2;
        }
    }
}

}  // namespace bevw

#include "bevw_pair.h"
#include "bevw_unit.h"

namespace bevw {

// psums[b][tile][3] -> chsums[b][3] ; grid = batch, block = 256
static __global__ void k_reduce_psums(const uint32_t *__restrict__ psums, int ntiles, unsigned long long *__restrict__ chsums)
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

// rows of `pitch` pixels <-> rows of `bw` pixels (destination widths that are not a multiple of 4, see Plan::pitch)
static __global__ void k_plan_pad(const uint8_t *__restrict__ src, int bw, int pitch, int rows, uint8_t *__restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row_bytes = (size_t)pitch * 3;
    if (i >= row_bytes * rows) return;
    const size_t y = i / row_bytes, k = i % row_bytes;
    dst[i] = k < (size_t)bw * 3 ? src[y * bw * 3 + k] : (uint8_t)0;
}
static __global__ void k_plan_unpad(const uint8_t *__restrict__ src, int bw, int pitch, size_t rows, uint8_t *__restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row_bytes = (size_t)bw * 3;
    if (i >= row_bytes * rows) return;
    const size_t y = i / row_bytes, k = i % row_bytes;
    dst[i] = src[y * pitch * 3 + k];
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static inline void plan_release(Plan &p)
{
    void *ptrs[] = {p.un_desc, p.un_entries, p.un_gsrc, p.un_gsrc_compact, p.list_un_all, p.entries, p.hdr, p.groups, p.psums, p.pad_out, p.pad_car, p.d_max, p.list_slow};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    p = Plan();
}

static inline hipError_t plan_upload_list(const std::vector<uint32_t> &v, void **dptr)
{
    *dptr = nullptr;
    if (v.empty()) return hipSuccess;
    hipError_t e = hipMalloc(dptr, v.size() * sizeof(uint32_t));
    if (e != hipSuccess) return e;
    return hipMemcpy(*dptr, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
}

// the compiled units of a plan -> device memory
static inline hipError_t plan_upload_units(Plan &p, const UnitPlanHost &up)
{
    hipError_t e;
    if ((e = hipMalloc(&p.un_desc, up.desc.size() * sizeof(UnitDesc))) != hipSuccess) return e;
    if ((e = hipMemcpy(p.un_desc, up.desc.data(), up.desc.size() * sizeof(UnitDesc), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMalloc(&p.un_entries, up.entries.size() * sizeof(uint32_t))) != hipSuccess) return e;
    if ((e = hipMemcpy(p.un_entries, up.entries.data(), up.entries.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = plan_upload_list(up.gsrc, &p.un_gsrc)) != hipSuccess) return e;
    if ((e = plan_upload_list(up.all, &p.list_un_all)) != hipSuccess) return e;
    p.n_un_all = (int)up.all.size();
    for (int c = 0; c < kUnitClasses; ++c) p.n_un[c] = (int)up.list[c].size();
    p.n_unit_tiles = (int)up.claimed_tiles;
    p.un_lines = up.lines; p.un_sectors = up.sectors; p.un_skew = (int)up.skew;
    return hipSuccess;
}

// pixels per output row the plan kernels write: the caller's pitch (bevw_set_output_pitch), else bw rounded up to 4 (12-byte stores)
static inline int plan_pitch(int bw, int out_pitch) { return out_pitch > 0 ? out_pitch : (bw + 3) & ~3; }

static inline hipError_t plan_build_impl(Plan &p, hipStream_t st, const StitchTables &T, int fw, int fh, int bw, int bh, int ncams = 4,
                                         bool units = true, const UnitTuning &unit_tune = UnitTuning(), int out_pitch = 0)
{
    plan_release(p);
    p.ncams = ncams;
    p.fw = fw; p.fh = fh; p.bw = bw; p.bh = bh;
    p.tiles_x = (bw + 4 * kPlanLX - 1) / (4 * kPlanLX);
    p.tiles_y = (bh + kPlanLY - 1) / kPlanLY;
    p.ntiles = p.tiles_x * p.tiles_y;
    hipError_t e;
    if ((e = hipMalloc(&p.entries, (size_t)p.ntiles * 8 * 64 * sizeof(uint2))) != hipSuccess) return e;
    if ((e = hipMalloc(&p.hdr, (size_t)p.ntiles * sizeof(uint32_t))) != hipSuccess) return e;
    if ((e = hipMalloc((void **)&p.d_max, sizeof(int))) != hipSuccess) return e;
    if ((e = hipMemsetAsync(p.d_max, 0, sizeof(int), st)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_plan_build, dim3(p.ntiles), dim3(64), 0, st, T, fw, fh, bw, bh, p.tiles_x, p.ntiles, ncams,
                       static_cast<uint2 *>(p.entries), static_cast<uint32_t *>(p.hdr), p.d_max);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(&p.max_contrib, p.d_max, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    p.band_ok = false;
    std::vector<uint32_t> groups_host;   // the sampled 4-texel groups, ascending (what Plan::groups holds on the device)
    if (fw % 4 == 0) {
        const size_t nbits = (size_t)ncams * fh * (fw / 4), nwords = (nbits + 31) / 32;
        uint32_t *d_bits = nullptr;
        if ((e = hipMalloc((void **)&d_bits, nwords * 4)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(d_bits, 0, nwords * 4, st)) != hipSuccess) return e;
        const size_t nent = (size_t)p.ntiles * 8 * 64;
        hipLaunchKernelGGL(k_plan_touch, dim3((unsigned)((nent + 255) / 256)), dim3(256), 0, st, static_cast<const uint2 *>(p.entries),
                           p.ntiles, fw, fh, d_bits);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        std::vector<uint32_t> bits(nwords), list;
        if ((e = hipMemcpyAsync(bits.data(), d_bits, nwords * 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
        (void)hipFree(d_bits);
        const uint32_t gw = (uint32_t)fw / 4, row_bytes = (uint32_t)fw * 3;
        for (size_t wi = 0; wi < nwords; ++wi) {
            uint32_t m = bits[wi];
            while (m) {
                const uint32_t bit = (uint32_t)wi * 32 + (uint32_t)__builtin_ctz(m);
                m &= m - 1;
                list.push_back((bit / gw) * row_bytes + (bit % gw) * 12);   // (cam * fh + y) rows of the set, 12 B per group
            }
        }
        p.n_groups = (int)list.size();
        if ((e = plan_upload_list(list, &p.groups)) != hipSuccess) return e;
        p.band_ok = true;
        groups_host.swap(list);
    }
    std::vector<uint32_t> hdr((size_t)p.ntiles);
    if ((e = hipMemcpyAsync(hdr.data(), p.hdr, hdr.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    p.usable = p.max_contrib <= 2 && (((size_t)fw * fh * 3) % 4 == 0) && (size_t)fw * fh * 3 * ncams < (1ull << 31);
    // units (bevw_unit.h): compiled on the host from the LUTs; they need rows of whole 4-texel groups (fw % 4 == 0) and 31-bit offsets
    if (units && p.usable && fw % 4 == 0) {
        const size_t bpx = (size_t)bw * bh;
        std::vector<int16_t> h1[4];
        std::vector<uint16_t> h2[4];
        std::vector<uint8_t> hm[4];
        for (int c = 0; c < ncams; ++c) {
            h1[c].resize(bpx * 2); h2[c].resize(bpx); hm[c].resize(bpx);
            if ((e = hipMemcpy(h1[c].data(), T.lut1[c], bpx * 4, hipMemcpyDeviceToHost)) != hipSuccess) return e;
            if ((e = hipMemcpy(h2[c].data(), T.lut2[c], bpx * 2, hipMemcpyDeviceToHost)) != hipSuccess) return e;
            if ((e = hipMemcpy(hm[c].data(), T.mask[c], bpx, hipMemcpyDeviceToHost)) != hipSuccess) return e;
        }
        UnitPlanHost up;
        std::vector<uint32_t> hdr_un = hdr;
        unit_compile(h1, h2, hm, ncams, fw, fh, bw, bh, plan_pitch(bw, out_pitch), p.tiles_x, p.tiles_y, hdr_un, up, unit_tune);
        if (!up.desc.empty()) {
            hdr.swap(hdr_un);
            if ((e = plan_upload_units(p, up)) != hipSuccess) return e;
            std::vector<uint32_t> gc;
            if (p.band_ok && unit_gsrc_compact(up.gsrc, groups_host, gc) && unit_compact_stride(groups_host.size()) < (1ull << 31)) {
                if ((e = plan_upload_list(gc, &p.un_gsrc_compact)) != hipSuccess) return e;
                p.compact_stride = unit_compact_stride(groups_host.size());
            }
        }
    }
    // what no unit owns stays on the per-tap kernel
    std::vector<uint32_t> left;
    for (int t = 0; t < p.ntiles; ++t)
        if (!(hdr[(size_t)t] & kHdrBlock)) left.push_back((uint32_t)t);
    p.n_slow = (int)left.size();
    if ((e = plan_upload_list(left, &p.list_slow)) != hipSuccess) return e;
    // 12-byte stores need 4-byte aligned pixel quads: rows of `pitch` pixels (bw % 4 != 0: padded scratch + k_plan_unpad)
    p.pitch = plan_pitch(bw, out_pitch);
    p.out_pitched = out_pitch > 0 && out_pitch != bw;
    return hipSuccess;
}

// nb: frames per block (0 = default); xcd_map: 1 = an XCD owns whole batch chunks; units: 0 = the per-tap kernel over every tile (debug)
struct PlanTuning { int nb = 0; int xcd_map = 1; int units = 1; };

// One step of the tile plan on `st`.
//   balance   = luminance round trip per tap on RAW frames (per-tap kernel over every tile) + per-tile channel sums;
//   d_scratch = the compact scratch plan_lum_band filled from d_frames (balance schedule 1): the units read IT (p.compact_stride bytes per
//               frame set, group lists p.un_gsrc_compact), the per-tap kernel serves what no unit owns from the RAW frames with the luminance
//               round trip per tap (d_deltas, d_tab); everything on the per-tap kernel when the units cannot run;
//   sums      = per-unit / per-tile channel sums, the car left to the gain pass (with d_scratch; or: d_frames are luminance-shifted already).
// balance and sums end with k_reduce_psums into d_chsums.  psums_frames / psums_first: the psums buffer is sized for psums_frames frame sets
// and this call's frames start at slot psums_first of it (two half-batches of one balance step run concurrently on two streams: run_device).
static inline hipError_t plan_stitch_impl(Plan &p, hipStream_t st, const uint8_t *d_frames, int batch, bool blend, bool balance,
                                          const int *d_deltas, const HsvTables *d_tab, const uint8_t *d_car,
                                          unsigned long long *d_chsums, uint8_t *d_out, const PlanTuning &tune, bool sums = false,
                                          int psums_frames = 0, int psums_first = 0, const uint8_t *d_scratch = nullptr)
{
    hipError_t e;
    PlanArgs a = {};
    a.frames = d_frames; a.plan = static_cast<const uint2 *>(p.entries); a.hdr = static_cast<const uint32_t *>(p.hdr);
    a.deltas = d_deltas; a.tab = d_tab; a.car = d_car; a.out = d_out;
    a.fw = p.fw; a.fh = p.fh; a.bw = p.bw; a.bh = p.bh;
    a.pitch = p.pitch;
    const bool padded = p.pitch != p.bw, scratch = padded && !p.out_pitched;
    if (padded) {
        const size_t img = (size_t)p.pitch * p.bh * 3, need = scratch ? img * (size_t)batch : 0;
        if (need > p.pad_cap) {
            if (p.pad_out) (void)hipFree(p.pad_out);
            p.pad_out = nullptr; p.pad_cap = 0;
            if ((e = hipMalloc(&p.pad_out, need)) != hipSuccess) return e;
            p.pad_cap = need;
        }
        if (d_car && !p.pad_car && (e = hipMalloc(&p.pad_car, img)) != hipSuccess) return e;
        if (d_car) {
            hipLaunchKernelGGL(k_plan_pad, dim3((unsigned)((img + 255) / 256)), dim3(256), 0, st, d_car, p.bw, p.pitch, p.bh,
                               static_cast<uint8_t *>(p.pad_car));
            a.car = static_cast<const uint8_t *>(p.pad_car);
        }
        if (scratch) a.out = static_cast<uint8_t *>(p.pad_out);
    }
    a.tiles_x = p.tiles_x; a.ntiles = p.ntiles;
    a.ncams = p.ncams;
    a.un_desc = static_cast<const UnitDesc *>(p.un_desc);
    a.un_entries = static_cast<const uint4 *>(p.un_entries);
    a.un_gsrc = static_cast<const uint32_t *>(p.un_gsrc);
    a.un_skew = p.un_skew;
    // the units need 4-byte aligned frame sets (dword-addressed group loads) and are not combined with the per-tap luminance kernel
    const bool compact = d_scratch != nullptr;
    const bool use_units = !balance && tune.units && p.n_un_all > 0 && (((uintptr_t)d_frames) & 3u) == 0 &&
                           (!compact || (p.un_gsrc_compact != nullptr && (((uintptr_t)d_scratch) & 3u) == 0));
    a.batch = batch;
    // frames per block: enough chunks to give each of the 8 XCDs whole chunks, otherwise one frame per chunk.  A block reads its plan
    // slice (4 bytes per pixel + the group list) once per chunk: 16 frames per block instead of 8 halve that traffic -- 1.8 M of the
    // 13.6 M read requests of a config-3 step -- for -5 % (direct), -7 % (blend) (profiles/r03/sweeps.log; 32 frames: the tail of 8 long
    // chunks costs more than it saves).
    // Batches of 32 .. 127 frame sets: 8 frames per block even when that leaves fewer than 8 chunks (the 4K rig at batch 32: 4 chunks in
    // plain chunk-major order, -9 % against 8 chunks of 4 frames).  An explicit BEVW_PLAN_NB is taken as it is.
    int nb = tune.nb > 0 ? tune.nb : (batch >= 128 ? 16 : (batch >= 32 ? 8 : (batch >= 8 ? batch / 8 : 1)));
    if (nb > batch) nb = batch;
    a.nb = nb;
    a.nchunks = (batch + nb - 1) / nb;
    a.xcd_affine = (a.nchunks >= 8 && tune.xcd_map) ? tune.xcd_map : 0;
    const bool with_sums = balance || sums;
    // channel-sum entries per frame: one per unit + one per base tile left to the per-tap kernel (or one per tile without units).  Every
    // entry has exactly one writer per frame (no atomics, round 5: 2.4 M atomic adds per config-4 step cost 58 us of the 600), and every writer
    // writes every frame of the call (units without a contributor write zeros): no entry depends on what the buffer held before
    a.nsum = use_units ? p.n_un_all + p.n_slow : p.ntiles;
    a.sum_base = use_units ? p.n_un_all : 0;
    if (with_sums) {
        if (psums_frames < batch) { psums_frames = batch; psums_first = 0; }
        const size_t per_frame = (size_t)(p.n_un_all + p.n_slow > p.ntiles ? p.n_un_all + p.n_slow : p.ntiles) * 3;   // either layout fits
        const size_t need = (size_t)psums_frames * per_frame * sizeof(uint32_t);
        if (need > p.psums_cap) {
            // (never while another stream still uses the buffer: the first half-batch call of a step sizes it for the whole step)
            if (p.psums) (void)hipFree(p.psums);
            p.psums = nullptr; p.psums_cap = 0;
            if ((e = hipMalloc(&p.psums, need)) != hipSuccess) return e;
            p.psums_cap = need;
            p.psums_layout = -1;
        }
        p.psums_layout = a.nsum;   // (entries per frame of the layout in use: plan_sum_entries)
        a.psums = static_cast<uint32_t *>(p.psums) + (size_t)psums_first * a.nsum * 3;
    }
    auto grid_blocks = [&]() -> unsigned {
        if (a.xcd_affine >= 1) return (unsigned)(a.ngroups * (((a.nchunks + 7) / 8) * 8));
        return (unsigned)(a.ngroups * a.nchunks);
    };
    const dim3 block(256);
    if (use_units) {
        if (sums) a.car = nullptr;   // the car is added behind the gains
        a.tile_list = static_cast<const uint32_t *>(p.list_un_all); a.nlist = p.n_un_all; a.ngroups = p.n_un_all;
        if (compact) { a.frames = d_scratch; a.set_stride = (uint32_t)p.compact_stride; a.un_gsrc = static_cast<const uint32_t *>(p.un_gsrc_compact); }
        const dim3 grid(grid_blocks());
        if (blend && sums) hipLaunchKernelGGL((k_plan_units<true, true>), grid, block, 0, st, a);
        else if (blend) hipLaunchKernelGGL((k_plan_units<true, false>), grid, block, 0, st, a);
        else if (sums) hipLaunchKernelGGL((k_plan_units<false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_plan_units<false, false>), grid, block, 0, st, a);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    const int n_tap = use_units ? p.n_slow : p.ntiles;
    if (n_tap) {
        a.tile_list = use_units ? static_cast<const uint32_t *>(p.list_slow) : nullptr;
        a.nlist = n_tap; a.ngroups = (n_tap + 3) / 4;
        if (sums) a.car = nullptr;
        a.frames = d_frames; a.set_stride = 0;   // the per-tap kernel reads whole frames: RAW ones in the balance modes
        const dim3 grid(grid_blocks());
        if (balance || (compact && sums)) {
            if (blend) hipLaunchKernelGGL((k_stitch_plan<true, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_stitch_plan<false, true>), grid, block, 0, st, a);
        } else if (compact) {   // luminance round trip per tap, no channel sums (camera-per-GPU shards: the stitch rank balances the colours)
            if (blend) hipLaunchKernelGGL((k_stitch_plan<true, true, false>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_stitch_plan<false, true, false>), grid, block, 0, st, a);
        } else if (blend && sums) hipLaunchKernelGGL((k_stitch_plan<true, false, true>), grid, block, 0, st, a);
        else if (blend) hipLaunchKernelGGL((k_stitch_plan<true, false>), grid, block, 0, st, a);
        else if (sums) hipLaunchKernelGGL((k_stitch_plan<false, false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_stitch_plan<false, false>), grid, block, 0, st, a);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (with_sums && d_chsums != nullptr) {   // (nullptr: the caller's gain pass adds the partial sums itself: plan_sum_entries)
        hipLaunchKernelGGL(k_reduce_psums, dim3(batch), dim3(256), 0, st, a.psums, a.nsum, d_chsums);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (scratch) {
        const size_t rows = (size_t)batch * p.bh;
        for (size_t r0 = 0; r0 < rows; r0 += (size_t)1 << 20) {   // <= 2^20 rows per launch keeps the grid below 2^31 blocks
            const size_t nr = rows - r0 < ((size_t)1 << 20) ? rows - r0 : ((size_t)1 << 20);
            hipLaunchKernelGGL(k_plan_unpad, dim3((unsigned)((nr * p.bw * 3 + 255) / 256)), dim3(256), 0, st,
                               static_cast<const uint8_t *>(p.pad_out) + r0 * p.pitch * 3, p.bw, p.pitch, nr, d_out + r0 * p.bw * 3);
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

// the partial channel sums the last plan_stitch_impl call with sums left for frame `first` of the psums buffer, and their number per frame
static inline const uint32_t *plan_sum_entries(const Plan &p, int first, int &nsum)
{
    nsum = p.psums_layout;
    return static_cast<const uint32_t *>(p.psums) + (size_t)first * (size_t)(nsum > 0 ? nsum : 0) * 3;
}

// a plan that holds only a WIDE unit schedule (analytic projection mode: plan_analytic_units) -> one launch of k_plan_unit_wide
static inline hipError_t plan_unit_wide_launch(const Plan &p, hipStream_t st, const uint8_t *d_frames, int batch, bool blend, const uint8_t *d_car,
                                               uint8_t *d_out, const PlanTuning &tune)
{
    if (p.n_un_all == 0) return hipSuccess;
    PlanArgs a = {};
    a.frames = d_frames; a.car = d_car; a.out = d_out;
    a.fw = p.fw; a.fh = p.fh; a.bw = p.bw; a.bh = p.bh; a.pitch = p.pitch;
    a.tiles_x = p.tiles_x; a.ntiles = p.ntiles; a.ncams = p.ncams;
    a.un_desc = static_cast<const UnitDesc *>(p.un_desc);
    a.un_entries = static_cast<const uint4 *>(p.un_entries);
    a.un_gsrc = static_cast<const uint32_t *>(p.un_gsrc);
    a.un_skew = p.un_skew;
    a.batch = batch;
    int nb = tune.nb > 0 ? tune.nb : (batch >= 128 ? 16 : (batch >= 32 ? 8 : (batch >= 8 ? batch / 8 : 1)));   // as plan_stitch_impl
    if (nb > batch) nb = batch;
    a.nb = nb;
    a.nchunks = (batch + nb - 1) / nb;
    a.xcd_affine = (a.nchunks >= 8 && tune.xcd_map) ? tune.xcd_map : 0;
    a.tile_list = static_cast<const uint32_t *>(p.list_un_all); a.nlist = p.n_un_all; a.ngroups = p.n_un_all;
    const unsigned grid = a.xcd_affine ? (unsigned)(a.ngroups * (((a.nchunks + 7) / 8) * 8)) : (unsigned)(a.ngroups * a.nchunks);
    if (blend) hipLaunchKernelGGL((k_plan_unit_wide<true>), dim3(grid), dim3(kUnitThreads), 0, st, a);
    else hipLaunchKernelGGL((k_plan_unit_wide<false>), dim3(grid), dim3(kUnitThreads), 0, st, a);
    return hipGetLastError();
}

// luminance-shift the sampled texel groups of every raw frame of the batch into the compact scratch (p.compact_stride bytes per frame set)
static inline hipError_t plan_lum_band(const Plan &p, hipStream_t st, const uint8_t *d_frames, uint8_t *d_scratch, int batch,
                                       const int *d_deltas, const HsvTables *d_tab)
{
    if (p.n_groups == 0 || p.compact_stride == 0) return hipSuccess;
    const size_t set_bytes = (size_t)p.fw * p.fh * 3 * p.ncams;
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        const unsigned bpf = (unsigned)(p.n_groups + 256 * kLumTrips - 1) / (256 * kLumTrips);
        hipLaunchKernelGGL(k_lum_groups, dim3(xcd_frame_grid(bpf, (unsigned)nb)), dim3(256), 0, st, d_frames + (size_t)b0 * set_bytes,
                           d_scratch + (size_t)b0 * p.compact_stride, set_bytes, p.compact_stride, (uint32_t)p.fw * p.fh * 3,
                           static_cast<const uint32_t *>(p.groups), p.n_groups, d_deltas + (size_t)b0 * 4, d_tab, bpf, (uint32_t)nb);
    }
    return hipGetLastError();
}

}  // namespace bevw
