// bevw_plan.h -- the tile-plan schedule of BevGenerator.__call__ (BEVW_SCHED_TILE_PLAN).
//
// Idea.  Every table the per-pixel schedule reads per frame (4 LUTs = 24 B/px, 4 masks = 4 B/px) is static for a
// calibration, and after masking a BEV pixel has at most two contributing cameras (one inside a trapezoid, two on a
// seam / in a blend overlap, none under the car).  bevw_build compiles LUT + masks into a CONTRIBUTOR PLAN: per BEV
// pixel up to two 8-byte entries {byte offset of the 2x2 footprint inside the 4-camera frame set, 5+5 bit
// fractions, u8 mask/weight, camera}.  One wave64 owns a (4*LX) x (64/LX) pixel tile (4 horizontally adjacent
// pixels per lane = one 12-byte store), loads its slice of the plan ONCE into registers, and then loops over the
// frames of its batch chunk.  Address, weight and mask arithmetic is hoisted out of the batch loop; the fixed-point
// bilinear runs on v_dot4_u32_u8 / v_dot2_u32_u16.  Three ways to get the texels, chosen per tile when the plan is built:
//   * block-staged (bevw_block.h, round 2): 2 x 4 base tiles form a 64 x 32 block tile whose footprint is staged ONCE per
//     frame into a patch all waves of the block share (dense single-contributor regions: 71 % of the tiles of config 3);
//   * pair-staged (bevw_pair.h, round 2): the tile's source texels are fetched in row-run groups, turned ONCE into
//     dot-product-ready texel pairs in a wave-private LDS patch, and every pixel reads two 8-byte pair entries;
//   * gather (plan_gather_tile): the schedule of round 1 for what cannot be pair-staged (frame widths that are not a
//     multiple of 4 pixels, the handful of sparse two-contributor tiles): every footprint row as an aligned 12-byte window.
// (Round 1's sector-staged schedule -- 64-byte sectors through an LDS-DMA ring, footprints realigned per pixel -- was
// measured against the pair-staged one, profiles/r02/sweeps.log, and removed.)
// A step is ONE launch (k_plan_all): every tile class of the batch in one grid, longest-running classes first.
//
// Layout.  plan[tile][slot 0..7][lane 0..63] (8 B each, so every plan load is a fully coalesced 512 B wave access);
// slots 0..3 = first contributor of the lane's 4 pixels, 4..7 = second contributor (only read when the tile header
// says some lane has one).  Blocks are 4 waves = 4 consecutive tiles (shared L1 lines on the same CU); the block
// index is mapped so that all tiles of one batch chunk run on the same XCD (block id % 8), which keeps a frame's
// source rows in one L2 while neighbouring tiles consume them.
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
constexpr uint32_t kHdrTransposed = 16u;   // lanes of a quad run along BEV y (see lane_xy)
constexpr uint32_t kHdrInterleaved = 32u;  // x-major tile whose lanes COMPUTE interleaved pixels (see pixel ownership)
constexpr int kPlanLXDefault = 8;          // lanes along x -> 32 x 8 pixel tiles (best of 4 / 8 / 16 on config 3)

struct Plan {
    void *entries = nullptr;     // uint2[ntiles][8][64]
    void *hdr = nullptr;         // uint32[ntiles]
    void *psums = nullptr;       // uint32[batch][ntiles][3]  (balance: per-tile channel sums)
    size_t psums_cap = 0;
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
    int lx = kPlanLXDefault;     // lanes of a wave along x; tile = (4 * lx) x (64 / lx) pixels
    int ncams = 4;
    void *groups = nullptr;      // uint32[n_groups]: byte offsets (inside the frame set) of the sampled 4-texel groups
    int n_groups = 0;
    bool band_ok = false;        // the sampled-group list exists (balance schedule 1)
    int max_contrib = 0;
    bool usable = false;
    // tile classes (lists of tile indices, row-major order kept): each class has its own lean kernel
    void *list_single = nullptr, *list_double = nullptr, *list_slow = nullptr, *list_empty = nullptr;
    int n_single = 0, n_double = 0, n_slow = 0, n_empty = 0;
    // pair-staged variant (bevw_pair.h): tiles whose footprints fit kPairRounds x 64 groups of 4 texels; pr_* lists hold
    // them, rp_* the single / double tiles that stay on the L1-gather kernels when this schedule is selected
    void *entries_pr = nullptr;  // uint2[ntiles][8][64]: x = LDS byte address of the pair entry of row 0 | row 1 << 16, y = meta
    void *gsrc = nullptr;        // uint32[ntiles][8][64]: per-lane source offset of the group of [slice][round]
    // pair classes: single-contributor tiles by mode (whole-tile 1 / 2 / 4 rounds, sliced), then two-contributor tiles
    // (whole-tile 1 / 2 / 4 rounds)
    static constexpr int kPairClasses = 7;
    void *list_pr[kPairClasses] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int n_pr[kPairClasses] = {0, 0, 0, 0, 0, 0, 0};

    void *list_rp_single = nullptr, *list_rp_double = nullptr, *list_rp_empty = nullptr;
    int n_rp_single = 0, n_rp_double = 0, n_rp_empty = 0;
    // block-staged variant (bevw_block.h): 64 x 32 block tiles compiled on the host; their base tiles are in none of the
    // pr / rp lists
    void *bt_entries = nullptr, *bt_gsrc = nullptr, *bt_pos = nullptr;
    void *sm_entries = nullptr, *sm_gsrc = nullptr, *sm_pos = nullptr, *list_sm = nullptr;   // seam block tiles (bevw_block.h)
    int n_sm = 0;
    void *list_bt = nullptr;                 // block-tile ids
    int n_bt = 0;
    int n_bt_tiles = 0;                      // base tiles they cover
    // unit schedule (bevw_unit.h): k-d partition compiled on the host; their base tiles are in none of the pr / rp lists either
    void *un_desc = nullptr, *un_entries = nullptr, *un_gsrc = nullptr;
    static constexpr int kUnitLists = 7;
    void *list_un[kUnitLists] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int n_un[kUnitLists] = {0, 0, 0, 0, 0, 0, 0};
    void *list_un_all = nullptr;             // every unit in partition order, class in bits 28..31
    int n_un_all = 0;
    size_t un_lines = 0, un_sectors = 0;     // request arithmetic of the partition (per frame)
    int un_skew = 0;
    bool paired_ok = false;
};

struct __attribute__((packed, aligned(1))) PackedU2 { uint32_t x, y; };
__device__ __forceinline__ uint2 load_u2_unaligned(const uint8_t *p)
{
    const PackedU2 v = *reinterpret_cast<const PackedU2 *>(p);
    return make_uint2(v.x, v.y);
}

// 12 bytes from a 4-byte aligned address: one global_load_dwordx3.  On gfx950 a dword-aligned gather of up to 16 B per
// lane costs ~14 clk per wave instruction in the texture addresser, a byte-misaligned dwordx2 twice that
// (tools/microbench.hip), so footprints are fetched as the aligned 12-byte window around them and realigned with
// v_alignbyte_b32.
struct __attribute__((packed, aligned(4))) AlignedU3 { uint32_t x, y, z; };
__device__ __forceinline__ uint2 load_footprint_row(const uint8_t *p_aligned, uint32_t mis)
{
    const AlignedU3 v = *reinterpret_cast<const AlignedU3 *>(p_aligned);
    return make_uint2(__builtin_amdgcn_alignbyte(v.y, v.x, mis), __builtin_amdgcn_alignbyte(v.z, v.y, mis));
}

// Lane -> pixel-quad position inside a (4*LX) x LY tile.  The vector memory pipe works on quads of 4 consecutive lanes
// and pays one cache access per distinct line a quad touches (measured: ~1.2 accesses/clk/CU, tools/microbench.hip and
// TCP_TOTAL_CACHE_ACCESSES), so the 4 lanes of a quad should sample neighbouring texels of ONE source row.  Where the
// BEV x axis runs along source rows (front/back cameras) that is the natural x-major order; where the BEV y axis does
// (left/right cameras: the image is rotated by ~90 degrees) the lanes of a quad are stacked along y instead.  The
// plan compiler picks per tile whichever order touches fewer lines.
__device__ __forceinline__ void lane_xy(int lane, int LX, bool transposed, int &lx, int &ly)
{
    const int LY = 64 / LX;
    if (transposed) { ly = lane % LY; lx = lane / LY; }
    else { lx = lane % LX; ly = lane / LX; }
}

// Pixel ownership.  A lane STORES 4 horizontally adjacent pixels (one 12-byte piece of a BEV row).  In x-major tiles
// it COMPUTES an interleaved set instead: lane l of a quad (a 16-pixel row segment) computes pixels {l, l+4, l+8, l+12},
// so that load instruction j of the quad fetches the footprints of the ADJACENT pixels 4j..4j+3, which sit in one or
// two 64-byte lines (a quad of non-interleaved lanes spreads over 16 pixels = ~50 source bytes and pays ~1.9 lines).
// The 4x4 exchange back to store order goes through a wave-private 1 KB LDS patch (4 ds_write_b32 + 1 ds_read_b128;
// the LDS pipe is otherwise idle in this kernel).  In y-major (transposed) tiles the quad's lanes are already adjacent
// along the source row, so compute order == store order.
__device__ __forceinline__ void quad_exchange(uint32_t P[4], uint32_t *xp_wave, int lane)
{
    const int base = (lane >> 2) * 16 + (lane & 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) xp_wave[base + 4 * j] = P[j];
    __builtin_amdgcn_wave_barrier();
    const uint4 v = *reinterpret_cast<const uint4 *>(xp_wave + lane * 4);
    __builtin_amdgcn_wave_barrier();
    P[0] = v.x; P[1] = v.y; P[2] = v.z; P[3] = v.w;
}

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
// plan compiler: one wave per tile
// ---------------------------------------------------------------------------------------------------------------
// 64-byte line id of the top footprint row of the first contributor of BEV pixel (x, y); ~0 when there is none
__device__ inline uint32_t plan_line_id(const StitchTables &T, int ncams, int fw, int fh, int bw, int bh, int x, int y)
{
    if (x >= bw || y >= bh) return 0xffffffffu;
    const size_t o = (size_t)y * bw + x;
    for (int c = 0; c < ncams; ++c) {
        if (T.mask[c][o] == 0) continue;
        const int sx = T.lut1[c][o * 2], sy = T.lut1[c][o * 2 + 1];
        if ((unsigned)sx >= (unsigned)fw || (unsigned)sy >= (unsigned)fh) continue;
        return (((uint32_t)c * fh + sy) * fw + sx) * 3 >> 6;
    }
    return 0xffffffffu;
}

// number of distinct line ids over the 4 lanes of every quad, summed over the wave
__device__ inline int quad_distinct_lines(uint32_t id, int lane)
{
    const int q0 = lane & ~3;
    int first = 1;
    for (int k = 0; k < 3; ++k) {
        const uint32_t other = __shfl(id, q0 + k, 64);
        if (q0 + k < lane && other == id) first = 0;
    }
    int n = first;
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    return n;
}

__global__ void k_plan_build(StitchTables T, int fw, int fh, int bw, int bh, int tiles_x, int ntiles, int LX, int orient,
                             int interleave, int ncams, uint2 *__restrict__ plan, uint32_t *__restrict__ hdr, int *__restrict__ max_contrib)
{
    const int LY = 64 / LX;
    const int tile = blockIdx.x, lane = threadIdx.x;
    if (tile >= ntiles) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    // choose the lane order that touches fewer 64-byte lines per quad (orient: 0 auto, 1 x-major, 2 y-major)
    bool transposed = orient == 2;
    if (orient == 0) {
        int cost[2] = {0, 0};
        for (int t = 0; t < 2; ++t) {
            int lx, ly;
            lane_xy(lane, LX, t != 0, lx, ly);
            for (int j = 0; j < 4; ++j) {
                const int xq = (t == 0 && interleave) ? (tx * LX + (lx & ~3)) * 4 + 4 * j + (lx & 3) : (tx * LX + lx) * 4 + j;
                cost[t] += quad_distinct_lines(plan_line_id(T, ncams, fw, fh, bw, bh, xq, ty * LY + ly), lane);
            }
        }
        transposed = cost[1] < cost[0];
    }
    int lx_, ly_;
    lane_xy(lane, LX, transposed, lx_, ly_);
    const bool inter = interleave && !transposed;
    const int x0 = (tx * LX + lx_) * 4, y = ty * LY + ly_;
    const uint32_t frame_bytes = (uint32_t)fw * fh * 3;
    uint32_t flags = 0;
    int worst = 0;
    for (int j = 0; j < 4; ++j) {
        uint2 e[2] = {make_uint2(0, 0), make_uint2(0, 0)};
        int count = 0;
        // compute pixel of slot j: store order x0 + j, or the interleaved one inside the quad's 16-pixel segment
        const int x = inter ? (tx * LX + (lx_ & ~3)) * 4 + 4 * j + (lx_ & 3) : x0 + j;
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
        if (transposed) hflags |= kHdrTransposed;
        if (inter) hflags |= kHdrInterleaved;
        hdr[tile] = hflags;
        atomicMax(max_contrib, worst);
    }
}

// Bitmap of the 4-texel groups (12 bytes, 4-byte aligned because fw % 4 == 0) that the plan samples, over the 4-camera
// frame set: bit index = (cam * fh + y) * (fw / 4) + x / 4.  The balance schedule converts exactly these groups of every
// raw frame (luminance round trip) instead of whole frames or bounding boxes (the LUT quirk makes the right camera's
// bounding box start at texel (0,0)).
__global__ void k_plan_touch(const uint2 *__restrict__ plan, int ntiles, int fw, int fh, uint32_t *__restrict__ bitmap)
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

// luminance_balance (surroundBEV.py:57-79) applied to the sampled texel groups of every raw frame:
// scratch = HSV2BGR(sat(V + delta)).  One lane = one group = 4 texels (12 bytes, one dwordx3 each way); groups[] holds
// byte offsets inside the frame set in ascending order, so neighbouring lanes touch neighbouring memory.
// grid = (ceil(ngroups / 256), batch); block = 256.
__global__ void __launch_bounds__(256) k_lum_groups(const uint8_t *__restrict__ frames, uint8_t *__restrict__ scratch, size_t set_bytes,
                                                     uint32_t frame_bytes, const uint32_t *__restrict__ groups, int ngroups,
                                                     const int *__restrict__ deltas, const HsvTables *__restrict__ tab,
                                                     uint32_t blocks_per_frame, uint32_t nframes)
{
    __shared__ int sdiv[256], hdiv[256];
    uint32_t frame, blk;
    if (!xcd_frame_map(blockIdx.x, blocks_per_frame, nframes, frame, blk)) return;   // grid: xcd_frame_grid()
    for (int i = threadIdx.x; i < 256; i += 256) { sdiv[i] = tab->sdiv[i]; hdiv[i] = tab->hdiv[i]; }
    __syncthreads();
    const int gi = (int)blk * 256 + threadIdx.x;
    if (gi >= ngroups) return;
    const int b = (int)frame;
    const uint32_t goff = groups[gi];
    const size_t off = (size_t)b * set_bytes + goff;
    const AlignedU3 v = *reinterpret_cast<const AlignedU3 *>(frames + off);
    const uint32_t w[3] = {v.x, v.y, v.z};
    const int delta = deltas[b * 4 + (int)(goff / frame_bytes)];
    // the 4 texels of the group as dwords (byte 3 is ignored), shifted, and packed back into the 12 bytes
    uint32_t P[4] = {w[0], __builtin_amdgcn_alignbyte(w[1], w[0], 3), __builtin_amdgcn_alignbyte(w[2], w[1], 2), w[2] >> 8};
#pragma unroll
    for (int t = 0; t < 4; ++t) P[t] = luminance_shift_bgr(P[t], delta, sdiv, hdiv);
    uint32_t o[3];
    pack_pixels(P, o[0], o[1], o[2]);
    AlignedU3 ov; ov.x = o[0]; ov.y = o[1]; ov.z = o[2];
    *reinterpret_cast<AlignedU3 *>(scratch + off) = ov;
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

// Same arithmetic with the y weights pre-scaled by 64: (S * 64 + 512 * 64) >> 16 == (S + 512) >> 10, so the result
// byte sits in bits 16..23 of each accumulator and the 12 output bytes of a lane are assembled with v_perm_b32
// instead of 12 shifts.  wy64 = (32-fy)*64 | (fy*64) << 16 (<= 2048 each; H <= 8160, so the sum stays < 2^32).
__device__ __forceinline__ void bilinear_rows_b2(uint2 r0, uint2 r1, uint32_t wx, uint32_t wy64, uint32_t acc[3])
{
    const uint32_t g0 = __builtin_amdgcn_alignbyte(r0.y, r0.x, 1), q0 = __builtin_amdgcn_alignbyte(r0.y, r0.x, 2);
    const uint32_t g1 = __builtin_amdgcn_alignbyte(r1.y, r1.x, 1), q1 = __builtin_amdgcn_alignbyte(r1.y, r1.x, 2);
    const uint32_t hb0 = __builtin_amdgcn_udot4(r0.x, wx, 0u, false), hb1 = __builtin_amdgcn_udot4(r1.x, wx, 0u, false);
    const uint32_t hg0 = __builtin_amdgcn_udot4(g0, wx, 0u, false), hg1 = __builtin_amdgcn_udot4(g1, wx, 0u, false);
    const uint32_t hr0 = __builtin_amdgcn_udot4(q0, wx, 0u, false), hr1 = __builtin_amdgcn_udot4(q1, wx, 0u, false);
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    union { uint32_t u; us2 v; } pb, pg, pr, w;
    pb.u = hb0 | (hb1 << 16); pg.u = hg0 | (hg1 << 16); pr.u = hr0 | (hr1 << 16); w.u = wy64;
    acc[0] = __builtin_amdgcn_udot2(pb.v, w.v, 32768u, false);
    acc[1] = __builtin_amdgcn_udot2(pg.v, w.v, 32768u, false);
    acc[2] = __builtin_amdgcn_udot2(pr.v, w.v, 32768u, false);
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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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
    int group_major;             // xcd_affine: block order inside an XCD is (tile group, chunk) instead of (chunk, tile group)
    const uint2 *plan_pr;        // pair-staged entries (plan_pair_body)
    const uint32_t *gsrc;        // group source offsets [ntiles][kPairRounds][64]
    const uint32_t *tile_list;   // class kernels: tile indices; nlist entries, ngroups = ceil(nlist / 4)
    int nlist;
    // block-staged classes (bevw_block.h): tile_list holds block-tile ids, ngroups = nlist
    const uint2 *bt_entries;
    const uint32_t *bt_gsrc;
    const uint32_t *bt_pos;
    // seam block tiles (64 x 16, two contributors)
    const uint2 *sm_entries;
    const uint32_t *sm_gsrc;
    const uint32_t *sm_pos;
    // units (bevw_unit.h): tile_list holds unit ids, ngroups = nlist
    const UnitDesc *un_desc;
    const uint4 *un_entries;     // one uint4 per lane and quad slot: the plan entries of the lane's 4 pixels
    const uint32_t *un_gsrc;
    int un_skew;                 // unit_skew constant of the plan
};

// Block index -> (batch chunk, tile group).  Blocks are dealt to the 8 XCDs round-robin (block id % 8), and each XCD has
// its own L2, so the map decides what an L2 sees:
//   xcd_affine 1: an XCD owns whole batch chunks (all tiles of frames b0..b0+nb), neighbouring tiles share its L2
//   xcd_affine 0: plain chunk-major order (few chunks)
__device__ __forceinline__ bool plan_block_map(const PlanArgs &a, uint32_t id, uint32_t &chunk, uint32_t &group)
{
    const uint32_t ng = (uint32_t)a.ngroups;
    if (a.xcd_affine == 1) {
        const uint32_t xcd = id & 7u, k = id >> 3;
        if (a.group_major) {
            // the chunks of an XCD back to back for every tile group: the blocks that read the same plan entries run at the same time
            const uint32_t cpx = ((uint32_t)a.nchunks + 7u) >> 3;
            chunk = xcd + 8u * (k % cpx);
            group = k / cpx;
        } else {
            chunk = xcd + 8u * (k / ng);
            group = k % ng;
        }
    } else {
        chunk = id / ng;
        group = id % ng;
    }
    return (int)chunk < a.nchunks;
}

// grid: see plan_grid_blocks(); block = 64 * waves-per-block threads (one tile per wave)
// LUM: luminance round trip per fetched texel (raw frames); SUMS: emit per-tile channel sums and leave the car to k_gain
template <int LX, bool BLEND, bool LUM, bool SUMS = LUM>
__global__ void __launch_bounds__(1024) k_stitch_plan(PlanArgs a)
{
    constexpr bool BAL = LUM;
    constexpr int LY = 64 / LX;
    __shared__ int sdiv[BAL ? 256 : 1], hdiv[BAL ? 256 : 1];
    __shared__ __attribute__((aligned(16))) uint32_t xpose[16 * 256];
    if (BAL) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) { sdiv[i] = a.tab->sdiv[i]; hdiv[i] = a.tab->hdiv[i]; }
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
    int lx_, ly_;
    lane_xy(lane, LX, (hdr & kHdrTransposed) != 0, lx_, ly_);
    const int x0 = (tx * LX + lx_) * 4, y = ty * LY + ly_;
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
        if (SUMS) {
            // per-tile channel sums of the pre-gain BEV (color_balance means, surroundBEV.py:44-47); pixels outside
            // the image have no plan entry and contribute 0
            unsigned s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { s0 += px[j][0]; s1 += px[j][1]; s2 += px[j][2]; }
            s0 = wave_sum_u32(s0); s1 = wave_sum_u32(s1); s2 = wave_sum_u32(s2);
            if (lane == 0) {
                uint32_t *ps = a.psums + ((size_t)b * a.ntiles + tile) * 3;
                ps[0] = s0; ps[1] = s1; ps[2] = s2;
            }
        }
        uint32_t P[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) P[j] = (uint32_t)px[j][0] | ((uint32_t)px[j][1] << 8) | ((uint32_t)px[j][2] << 16);
        if (hdr & kHdrInterleaved) quad_exchange(P, xpose + (threadIdx.x >> 6) * 256, lane);
        if (!SUMS && car_any) add_car(P, car0, car1, car2);
        if (inimg) {
            uint32_t d0, d1, d2;
            pack_pixels(P, d0, d1, d2);
            uint32_t *op = reinterpret_cast<uint32_t *>(a.out + (size_t)b * img_bytes + ooff);
            op[0] = d0; op[1] = d1; op[2] = d2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// gather class kernels (no luminance round trip): every contributor of the tile is an interior footprint.
//   NSLOT = 1: <= 1 contributor per pixel (inside a trapezoid)                -> list_single / list_rp_single
//   NSLOT = 2: some pixel has two (direct-stitch seams, blend overlaps)       -> list_double / list_rp_double
// Register diet: per pixel and slot only {offset, misalignment, wx, wy (, wf)} live across the batch loop, so many waves
// fit a SIMD and the gathers of many tiles overlap.  Same block -> (chunk, tile) mapping as k_stitch_plan.
// SUMS: emit per-tile channel sums (balance on pre-shifted frames) and leave the car to k_gain.
// ---------------------------------------------------------------------------------------------------------------
// one wave: tile `tile`, frames [b_begin, b_end); xpose_wave: 1 KB of LDS private to the wave (quad exchange)
template <int LX, int NSLOT, bool BLEND, bool SUMS>
__device__ __forceinline__ void plan_gather_tile(const PlanArgs &a, int tile, int b_begin, int b_end, uint32_t *xpose_wave)
{
    constexpr int LY = 64 / LX;
    const int lane = threadIdx.x & 63;
    const uint32_t hdr = __builtin_amdgcn_readfirstlane(a.hdr[tile]);
    const bool interleaved = (hdr & kHdrInterleaved) != 0;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    int lx_, ly_;
    lane_xy(lane, LX, (hdr & kHdrTransposed) != 0, lx_, ly_);
    const int x0 = (tx * LX + lx_) * 4, y = ty * LY + ly_;
    const bool inimg = x0 < a.bw && y < a.bh;
    const uint32_t row_bytes = (uint32_t)a.fw * 3;
    const size_t set_bytes = (size_t)a.fw * a.fh * 3 * a.ncams, img_bytes = (size_t)a.pitch * a.bh * 3;
    const uint32_t ooff = ((uint32_t)y * a.pitch + x0) * 3;

    uint32_t off[NSLOT][4], mis[NSLOT][4], wx[NSLOT][4], wy[NSLOT][4];
    float wf[NSLOT][4];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint2 e = a.plan[((size_t)tile * 8 + s * 4 + j) * 64 + lane];
            const uint32_t fx = e.y & 31, fy = (e.y >> 5) & 31;
            const bool valid = e.y & kMetaValid;
            const uint32_t o = valid ? e.x : 0u;
            off[s][j] = o & ~3u;   // frames are 4-byte aligned (checked by the host), so this is an aligned address
            mis[s][j] = o & 3u;
            wx[s][j] = valid ? ((32 - fx) | (fx << 24)) : 0u;  // zero x-weights: an absent entry contributes exactly 0
            wy[s][j] = ((32 - fy) << 6) | (fy << 22);          // y weights x 64 (bilinear_rows_b2)
            wf[s][j] = BLEND ? blend_weight_f32((int)((e.y >> 10) & 255)) : 1.f;
        }
    uint32_t car0 = 0, car1 = 0, car2 = 0;
    if (!SUMS && a.car != nullptr && inimg) {
        const uint32_t *cp = reinterpret_cast<const uint32_t *>(a.car + ooff);
        car0 = cp[0]; car1 = cp[1]; car2 = cp[2];
    }
    const bool car_any = __builtin_amdgcn_ballot_w64((car0 | car1 | car2) != 0) != 0;

    const uint8_t *fb = a.frames + (size_t)b_begin * set_bytes;
    uint8_t *ob = a.out + (size_t)b_begin * img_bytes + ooff;
#pragma unroll 1
    for (int b = b_begin; b < b_end; ++b, fb += set_bytes, ob += img_bytes) {
        const uint8_t *fb1 = fb + row_bytes;
        uint32_t acc[4][3];
        {
            uint2 r0[4], r1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r0[j] = load_footprint_row(fb + off[0][j], mis[0][j]);
                r1[j] = load_footprint_row(fb1 + off[0][j], mis[0][j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) bilinear_rows_b2(r0[j], r1[j], wx[0][j], wy[0][j], acc[j]);
        }
        uint32_t P[4];
        if (!BLEND && NSLOT == 1) {
            // result bytes sit in bits 16..23 of the accumulators: gather them into pixel dwords B | G << 8 | R << 16
#pragma unroll
            for (int j = 0; j < 4; ++j)
                P[j] = __builtin_amdgcn_perm(acc[j][2], __builtin_amdgcn_perm(acc[j][1], acc[j][0], 0x0c0c0602u), 0x0c060100u);
        } else {
            int px[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint32_t v = (acc[j][k] >> 16) & 255u;
                    px[j][k] = BLEND ? (int)((float)v * wf[0][j]) : (int)v;
                }
            if (NSLOT == 2) {
                uint2 r0[4], r1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    r0[j] = load_footprint_row(fb + off[NSLOT - 1][j], mis[NSLOT - 1][j]);
                    r1[j] = load_footprint_row(fb1 + off[NSLOT - 1][j], mis[NSLOT - 1][j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t w[3];
                    bilinear_rows_b2(r0[j], r1[j], wx[NSLOT - 1][j], wy[NSLOT - 1][j], w);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const uint32_t v = (w[k] >> 16) & 255u;
                        px[j][k] = min(255, px[j][k] + (BLEND ? (int)((float)v * wf[NSLOT - 1][j]) : (int)v));
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) P[j] = (uint32_t)px[j][0] | ((uint32_t)px[j][1] << 8) | ((uint32_t)px[j][2] << 16);
        }
        if (SUMS) {
            // channel sums of the tile for color_balance: a lane's 4 pixels sum to <= 1020 per channel and a wave to
            // <= 65280, so B and G travel packed in one dword through the butterfly
            uint32_t sb = 0, sg = 0, sr = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sb = __builtin_amdgcn_udot4(P[j], 0x00000001u, sb, false);
                sg = __builtin_amdgcn_udot4(P[j], 0x00000100u, sg, false);
                sr = __builtin_amdgcn_udot4(P[j], 0x00010000u, sr, false);
            }
            uint32_t bg = sb | (sg << 16);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { bg += __shfl_xor(bg, o, 64); sr += __shfl_xor(sr, o, 64); }
            if (lane == 0) {
                uint32_t *ps = a.psums + ((size_t)b * a.ntiles + tile) * 3;
                ps[0] = bg & 0xffffu; ps[1] = bg >> 16; ps[2] = sr;
            }
        }
        if (interleaved) quad_exchange(P, xpose_wave, lane);
        if (car_any) add_car(P, car0, car1, car2);
        if (inimg) {
            uint32_t d0, d1, d2;
            pack_pixels(P, d0, d1, d2);
            uint32_t *op = reinterpret_cast<uint32_t *>(ob);
            op[0] = d0; op[1] = d1; op[2] = d2;
        }
    }
}

template <int LX, int NSLOT, bool BLEND, bool SUMS>
__device__ __forceinline__ void plan_gather_block(const PlanArgs &a, uint32_t block_id, uint32_t *xpose)
{
    uint32_t chunk, group;
    if (!plan_block_map(a, block_id, chunk, group)) return;
    const int slot = (int)group * 4 + (threadIdx.x >> 6);
    if (slot >= a.nlist) return;
    const int tile = (int)__builtin_amdgcn_readfirstlane(a.tile_list[slot]);
    const int b_begin = (int)chunk * a.nb;
    plan_gather_tile<LX, NSLOT, BLEND, SUMS>(a, tile, b_begin, min(a.batch, b_begin + a.nb), xpose + (threadIdx.x >> 6) * 256);
}

template <int LX, int NSLOT, bool BLEND, bool SUMS>
__global__ void __launch_bounds__(256) k_plan_lean(PlanArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t xpose[4 * 256];
    plan_gather_block<LX, NSLOT, BLEND, SUMS>(a, blockIdx.x, xpose);
}

// tiles without any contributor (under the car): out = car (or 0) for every frame of the chunk
template <int LX>
__device__ __forceinline__ void plan_empty_body(const PlanArgs &a, uint32_t block_id)
{
    constexpr int LY = 64 / LX;
    const uint32_t chunk = block_id / (uint32_t)a.ngroups, group = block_id % (uint32_t)a.ngroups;
    const int lane = threadIdx.x & 63;
    const int slot = (int)group * (int)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (slot >= a.nlist) return;
    const int tile = (int)a.tile_list[slot];
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    int lx_, ly_;
    lane_xy(lane, LX, false, lx_, ly_);
    const int x0 = (tx * LX + lx_) * 4, y = ty * LY + ly_;
    if (!(x0 < a.bw && y < a.bh)) return;
    const size_t img_bytes = (size_t)a.pitch * a.bh * 3;
    const uint32_t ooff = ((uint32_t)y * a.pitch + x0) * 3;
    uint32_t c0 = 0, c1 = 0, c2 = 0;
    if (a.car != nullptr) {
        const uint32_t *cp = reinterpret_cast<const uint32_t *>(a.car + ooff);
        c0 = cp[0]; c1 = cp[1]; c2 = cp[2];
    }
    const int b_begin = (int)chunk * a.nb, b_end = min(a.batch, b_begin + a.nb);
    for (int b = b_begin; b < b_end; ++b) {
        uint32_t *op = reinterpret_cast<uint32_t *>(a.out + (size_t)b * img_bytes + ooff);
        op[0] = c0; op[1] = c1; op[2] = c2;
    }
}

template <int LX>
__global__ void __launch_bounds__(1024) k_plan_empty(PlanArgs a) { plan_empty_body<LX>(a, blockIdx.x); }

}  // namespace bevw
#include "bevw_pair.h"
#include "bevw_block.h"
#include "bevw_unit.h"
namespace bevw {

// Every tile class of a step in ONE launch: the class kernels write disjoint tiles and never depend on each other, but
// consecutive launches on a stream are separated by a barrier (the tail of one class and the ramp of the next cost
// ~10 us each, five times per step).  Blocks are dealt to the classes in the same order as the separate launches
// (launch position i owns blocks start[i] .. start[i+1] and runs class kind[i]; every start is a multiple of 8, so a
// block's XCD is what it was in the separate launch).
constexpr int kPlanAllMax = 13;   // classes of one merged launch (launch positions in use)
struct PlanAllArgs {
    PlanArgs a;
    const uint32_t *list[kPlanAllMax];
    int nlist[kPlanAllMax];
    int ngroups[kPlanAllMax];
    uint32_t start[kPlanAllMax + 1];   // block ranges in launch order
    // launch position -> class: 2 empty, 3 gather single, 5..8 pair-staged single (whole-tile 1 / 2 / 4 rounds, sliced),
    // 9, 10, 11 pair-staged double (1 / 2 / 4 rounds), 12 block-staged (bevw_block.h, 4 waves per block tile), 13 seam block tiles,
    // 18 units (bevw_unit.h) of every class in the partition's own order
    int kind[kPlanAllMax];
    int n;                             // launch positions in use
};

// The two-contributor classes set the register budget (~140 VGPRs, 3 workgroups per CU); measured, the single-contributor
// classes lose nothing at that occupancy (profiles/r01_sweeps.log: two launches split by register budget are slower).
#ifndef BEVW_PLAN_ALL_WAVES
#define BEVW_PLAN_ALL_WAVES 3   // waves per SIMD the merged kernel is compiled for (168 VGPRs: 3 blocks of 32 KB per CU; 4 spills 2 registers)
#endif
template <int LX, bool BLEND, bool SUMS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BEVW_PLAN_ALL_WAVES, BEVW_PLAN_ALL_WAVES))) k_plan_all(PlanAllArgs q)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage_0[4 * kPairPatch];
    int pos = 0;
#pragma unroll
    for (int c = 1; c < kPlanAllMax; ++c) pos += (c < q.n && blockIdx.x >= q.start[c]) ? 1 : 0;
    PlanArgs a = q.a;
    a.tile_list = q.list[pos]; a.nlist = q.nlist[pos]; a.ngroups = q.ngroups[pos];
    const uint32_t id = blockIdx.x - q.start[pos];
    switch (q.kind[pos]) {
        case 5: plan_pair_body<LX, 1, BLEND, SUMS, 1, 1>(a, id, stage_0); break;
        case 6: plan_pair_body<LX, 1, BLEND, SUMS, 1, 2>(a, id, stage_0); break;
        case 7: plan_pair_body<LX, 1, BLEND, SUMS, 1, 4>(a, id, stage_0); break;
        case 8: plan_pair_body<LX, 1, BLEND, SUMS, 4, 2>(a, id, stage_0); break;
        case 9: plan_pair_body<LX, 2, BLEND, SUMS, 1, 1>(a, id, stage_0); break;
        case 10: plan_pair_body<LX, 2, BLEND, SUMS, 1, 2>(a, id, stage_0); break;
        case 11: plan_pair_body<LX, 2, BLEND, SUMS, 1, 4>(a, id, stage_0); break;
        case 12: plan_block_body<BLEND, SUMS, 2>(a, id, stage_0); break;
        case 13: plan_seam_body<BLEND, SUMS>(a, id, stage_0); break;
        case 18: plan_unit_any<BLEND, SUMS>(a, id, stage_0); break;
        case 2: plan_empty_body<LX>(a, id); break;
        case 3: plan_gather_block<LX, 1, BLEND, SUMS>(a, id, reinterpret_cast<uint32_t *>(stage_0)); break;
        // (the two-contributor gather class -- a handful of sparse seam tiles, 110+ VGPRs -- stays out of the merged kernel: it
        // would set the register budget of every other class; plan_launch_lx gives it its own launch)
        default: break;
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
    void *ptrs[] = {p.entries_pr, p.gsrc, p.list_pr[0], p.list_pr[1], p.list_pr[2], p.list_pr[3], p.list_pr[4], p.list_pr[5], p.list_pr[6],
                    p.list_rp_single, p.list_rp_double, p.list_rp_empty, p.bt_entries, p.bt_gsrc, p.bt_pos, p.list_bt, p.sm_entries, p.sm_gsrc, p.sm_pos, p.list_sm,
                    p.un_desc, p.un_entries, p.un_gsrc, p.list_un_all, p.list_un[0], p.list_un[1], p.list_un[2], p.list_un[3], p.list_un[4], p.list_un[5], p.list_un[6],
                    p.entries, p.hdr, p.groups, p.psums, p.pad_out, p.pad_car, p.d_max, p.list_single, p.list_double, p.list_slow, p.list_empty};
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

// pixels per output row the plan kernels write: the caller's pitch (bevw_set_output_pitch), else bw rounded up to 4 (12-byte stores)
static inline int plan_pitch(int bw, int out_pitch) { return out_pitch > 0 ? out_pitch : (bw + 3) & ~3; }

static inline hipError_t plan_build_impl(Plan &p, hipStream_t st, const StitchTables &T, int fw, int fh, int bw, int bh, int lx,
                                         int orient = 0, int interleave = 1, bool column_major_transposed = true, int super_tile = 1,
                                         int ncams = 4, bool block_tiles = true, bool seam_tiles = true, bool units = true,
                                         const UnitTuning &unit_tune = UnitTuning(), int out_pitch = 0)
{
    plan_release(p);
    if (lx != 4 && lx != 8 && lx != 16) lx = kPlanLXDefault;
    p.lx = lx;
    p.ncams = ncams;
    p.fw = fw; p.fh = fh; p.bw = bw; p.bh = bh;
    p.tiles_x = (bw + 4 * lx - 1) / (4 * lx);
    p.tiles_y = (bh + (64 / lx) - 1) / (64 / lx);
    p.ntiles = p.tiles_x * p.tiles_y;
    hipError_t e;
    if ((e = hipMalloc(&p.entries, (size_t)p.ntiles * 8 * 64 * sizeof(uint2))) != hipSuccess) return e;
    if ((e = hipMalloc(&p.hdr, (size_t)p.ntiles * sizeof(uint32_t))) != hipSuccess) return e;
    if ((e = hipMalloc((void **)&p.d_max, sizeof(int))) != hipSuccess) return e;
    if ((e = hipMemsetAsync(p.d_max, 0, sizeof(int), st)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_plan_build, dim3(p.ntiles), dim3(64), 0, st, T, fw, fh, bw, bh, p.tiles_x, p.ntiles, lx, orient, interleave, ncams,
                       static_cast<uint2 *>(p.entries), static_cast<uint32_t *>(p.hdr), p.d_max);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(&p.max_contrib, p.d_max, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    p.band_ok = false;
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
    }
    p.paired_ok = false;
    if (fw % 4 == 0 && (size_t)fw * fh * 3 * ncams < (1ull << 31)) {   // rows are whole groups of 4 texels (12 bytes)
        if ((e = hipMalloc(&p.entries_pr, (size_t)p.ntiles * 8 * 64 * sizeof(uint2))) != hipSuccess) return e;
        if ((e = hipMalloc(&p.gsrc, (size_t)p.ntiles * kPairSrcSlots * 64 * sizeof(uint32_t))) != hipSuccess) return e;
        hipLaunchKernelGGL(k_plan_pair_build, dim3(p.ntiles), dim3(64), 0, st, static_cast<const uint2 *>(p.entries),
                           static_cast<uint32_t *>(p.hdr), p.ntiles, (uint32_t)fw * 3, (uint32_t)((size_t)fw * fh * 3 * ncams),
                           static_cast<uint2 *>(p.entries_pr), static_cast<uint32_t *>(p.gsrc), lx);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        p.paired_ok = true;
    }
    std::vector<uint32_t> hdr((size_t)p.ntiles);
    if ((e = hipMemcpyAsync(hdr.data(), p.hdr, hdr.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    // block tiles (bevw_block.h): compiled on the host from the LUTs; claimed base tiles get kHdrBlock in the host copy
    if (block_tiles && p.paired_ok && lx == 8) {
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
        // units (bevw_unit.h) first: they take every single-contributor base tile without border footprints; block tiles are the
        // round-2 schedule for the same tiles and are compiled only when the units are switched off
        bool have_units = false;
        if (units) {
            UnitPlanHost up;
            std::vector<uint32_t> hdr_un = hdr;
            unit_compile(h1, h2, hm, ncams, fw, fh, bw, bh, plan_pitch(bw, out_pitch), p.tiles_x, p.tiles_y, hdr_un, up, unit_tune);
            if (!up.desc.empty()) {
                have_units = true;
                hdr.swap(hdr_un);
                if ((e = hipMalloc(&p.un_desc, up.desc.size() * sizeof(UnitDesc))) != hipSuccess) return e;
                if ((e = hipMemcpy(p.un_desc, up.desc.data(), up.desc.size() * sizeof(UnitDesc), hipMemcpyHostToDevice)) != hipSuccess) return e;
                if ((e = hipMalloc(&p.un_entries, up.entries.size() * sizeof(uint32_t))) != hipSuccess) return e;
                if ((e = hipMemcpy(p.un_entries, up.entries.data(), up.entries.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess) return e;
                if ((e = plan_upload_list(up.gsrc, &p.un_gsrc)) != hipSuccess) return e;
                for (int c = 0; c < kUnitClasses; ++c) {
                    p.n_un[c] = (int)up.list[c].size();
                    if ((e = plan_upload_list(up.list[c], &p.list_un[c])) != hipSuccess) return e;
                }
                p.n_un_all = (int)up.all.size();
                if ((e = plan_upload_list(up.all, &p.list_un_all)) != hipSuccess) return e;
                p.un_lines = up.lines; p.un_sectors = up.sectors; p.un_skew = (int)up.skew;
            }
        }
        BlockPlanHost bp;
        std::vector<uint32_t> hdr_bt = hdr;
        if (!have_units) block_compile(h1, h2, hm, ncams, fw, fh, bw, bh, p.tiles_x, p.tiles_y, hdr_bt, bp);
        SeamPlanHost sp;
        if (seam_tiles) seam_compile(h1, h2, hm, ncams, fw, fh, bw, bh, p.tiles_x, p.tiles_y, hdr_bt, sp);
        // worth two more launches only when the block tiles take a good part of the work (the 4K rig: 18 of 4166 tiles)
        size_t claimed = 0, busy = 0;
        for (size_t t = 0; t < hdr.size(); ++t) {
            if (hdr[t] & kHdrEmpty) continue;
            ++busy;
            if (hdr_bt[t] & kHdrBlock) ++claimed;
        }
        if (have_units || (!bp.pos.empty() && claimed * 4 >= busy)) {
            hdr.swap(hdr_bt);
            if (!bp.pos.empty()) {
                if ((e = hipMalloc(&p.bt_entries, bp.entries.size() * sizeof(uint2))) != hipSuccess) return e;
                if ((e = hipMemcpy(p.bt_entries, bp.entries.data(), bp.entries.size() * sizeof(uint2), hipMemcpyHostToDevice)) != hipSuccess) return e;
                if ((e = plan_upload_list(bp.gsrc, &p.bt_gsrc)) != hipSuccess) return e;
                if ((e = plan_upload_list(bp.pos, &p.bt_pos)) != hipSuccess) return e;
                p.n_bt = (int)bp.list[0].size();
                if ((e = plan_upload_list(bp.list[0], &p.list_bt)) != hipSuccess) return e;
            }
            if (!sp.pos.empty()) {
                if ((e = hipMalloc(&p.sm_entries, sp.entries.size() * sizeof(uint2))) != hipSuccess) return e;
                if ((e = hipMemcpy(p.sm_entries, sp.entries.data(), sp.entries.size() * sizeof(uint2), hipMemcpyHostToDevice)) != hipSuccess) return e;
                if ((e = plan_upload_list(sp.gsrc, &p.sm_gsrc)) != hipSuccess) return e;
                if ((e = plan_upload_list(sp.pos, &p.sm_pos)) != hipSuccess) return e;
                if ((e = plan_upload_list(sp.list, &p.list_sm)) != hipSuccess) return e;
                p.n_sm = (int)sp.list.size();
            }
        }
    }
    // classify tiles (order kept): slow > empty > double > single
    std::vector<uint32_t> ls, ld, lw, le;
    for (int t = 0; t < p.ntiles; ++t) {
        const uint32_t h = hdr[(size_t)t];
        if (h & kHdrSlow) lw.push_back((uint32_t)t);
        else if (h & kHdrEmpty) le.push_back((uint32_t)t);
        else if (h & kHdrSecond) ld.push_back((uint32_t)t);
        else ls.push_back((uint32_t)t);
    }
    // HBM serves 128-byte lines (a 64-byte sector request costs as much, tools/hbm_gather.hip), so the 4 waves of a
    // block -- which share one L1 -- should cover ONE contiguous stretch of source rows: consecutive list entries are
    // x-neighbours for x-major tiles (BEV x runs along source rows) and y-neighbours for y-major (transposed) tiles.
    auto order = [&](std::vector<uint32_t> &v) {
        if (super_tile > 1) {
            // blocks of super_tile x super_tile tiles are contiguous in the list: the waves of one workgroup (one L1)
            // then share source rows in both directions
            const uint32_t tx = (uint32_t)p.tiles_x, S = (uint32_t)super_tile;
            auto key = [tx, S](uint32_t t) {
                const uint32_t x = t % tx, y = t / tx;
                return (((uint64_t)(y / S) * 4096 + (x / S)) * S + (y % S)) * S + (x % S);
            };
            std::stable_sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
            return;
        }
        if (!column_major_transposed) return;
        std::vector<uint32_t> xm, ym;
        for (uint32_t t : v) ((hdr[t] & kHdrTransposed) ? ym : xm).push_back(t);
        const uint32_t tx = (uint32_t)p.tiles_x;
        std::stable_sort(ym.begin(), ym.end(), [tx](uint32_t a, uint32_t b) {
            return (a % tx) != (b % tx) ? (a % tx) < (b % tx) : (a / tx) < (b / tx);
        });
        v = xm;
        v.insert(v.end(), ym.begin(), ym.end());
    };
    order(ls); order(ld); order(lw);
    {
        std::vector<uint32_t> pr[Plan::kPairClasses], rs, rd, re;
        for (uint32_t t : ls) { if (hdr[t] & kHdrBlock) ++p.n_bt_tiles; else if (hdr[t] & kHdrPaired) pr[(hdr[t] >> 8) & 3u].push_back(t); else rs.push_back(t); }
        for (uint32_t t : ld) { if (hdr[t] & kHdrBlock) ++p.n_bt_tiles; else if (hdr[t] & kHdrPaired) pr[4 + ((hdr[t] >> 8) & 3u)].push_back(t); else rd.push_back(t); }
        for (uint32_t t : le) { if (hdr[t] & kHdrBlock) ++p.n_bt_tiles; else re.push_back(t); }
        p.n_rp_empty = (int)re.size();
        if ((e = plan_upload_list(re, &p.list_rp_empty)) != hipSuccess) return e;
        for (int c = 0; c < Plan::kPairClasses; ++c) {
            p.n_pr[c] = (int)pr[c].size();
            if ((e = plan_upload_list(pr[c], &p.list_pr[c])) != hipSuccess) return e;
        }
        p.n_rp_single = (int)rs.size(); p.n_rp_double = (int)rd.size();
        if ((e = plan_upload_list(rs, &p.list_rp_single)) != hipSuccess) return e;
        if ((e = plan_upload_list(rd, &p.list_rp_double)) != hipSuccess) return e;
    }
    p.n_single = (int)ls.size(); p.n_double = (int)ld.size(); p.n_slow = (int)lw.size(); p.n_empty = (int)le.size();
    if ((e = plan_upload_list(ls, &p.list_single)) != hipSuccess) return e;
    if ((e = plan_upload_list(ld, &p.list_double)) != hipSuccess) return e;
    if ((e = plan_upload_list(lw, &p.list_slow)) != hipSuccess) return e;
    if ((e = plan_upload_list(le, &p.list_empty)) != hipSuccess) return e;
    // 12-byte stores need 4-byte aligned pixel quads: rows of `pitch` pixels (bw % 4 != 0: padded scratch + k_plan_unpad)
    // and the aligned 12-byte footprint reads need every frame of a set to start on a 4-byte boundary
    p.pitch = plan_pitch(bw, out_pitch);
    p.out_pitched = out_pitch > 0 && out_pitch != bw;
    p.usable = p.max_contrib <= 2 && (((size_t)fw * fh * 3) % 4 == 0) && (size_t)fw * fh * 3 * ncams < (1ull << 31);
    return hipSuccess;
}

// rows of `pitch` pixels <-> rows of `bw` pixels (destination widths that are not a multiple of 4, see Plan::pitch)
__global__ void k_plan_pad(const uint8_t *__restrict__ src, int bw, int pitch, int rows, uint8_t *__restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row_bytes = (size_t)pitch * 3;
    if (i >= row_bytes * rows) return;
    const size_t y = i / row_bytes, k = i % row_bytes;
    dst[i] = k < (size_t)bw * 3 ? src[y * bw * 3 + k] : (uint8_t)0;
}
__global__ void k_plan_unpad(const uint8_t *__restrict__ src, int bw, int pitch, size_t rows, uint8_t *__restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row_bytes = (size_t)bw * 3;
    if (i >= row_bytes * rows) return;
    const size_t y = i / row_bytes, k = i % row_bytes;
    dst[i] = src[y * pitch * 3 + k];
}

// nb: frames per block (0 = default); lean: 0 = one generic kernel over every tile (debug); lds_pad: dynamic LDS added to the
// single-contributor gather kernel to cap it at 5 waves per SIMD (more resident gather waves thrash the L1);
// xcd_map: 1 = an XCD owns whole batch chunks; staged: 0 = gather classes only (k_plan_lean), 1 = pair-staged classes
// (bevw_pair.h) for every tile that has a pair plan; one_launch: 1 = all tile classes of a step in one kernel
// (k_plan_all), 0 = one launch per class
// bt_merged: 1 = the block tiles are a class of the merged launch (4 waves per block tile), 0 = their own 8-wave kernel first
struct PlanTuning { int nb = 0; int lean = 1; int lds_pad = 16384; int xcd_map = 1; int staged = 1; int one_launch = 1; int bt_merged = 1; int group_major = 0; int unit_spatial = 1; };

template <int LX>
static inline hipError_t plan_launch_lx(Plan &p, hipStream_t st, PlanArgs a, bool blend, bool balance, bool lean, int lds_pad,
                                        bool sums, bool staged, bool one_launch, bool bt_in_merged_launch, bool unit_spatial)
{
    hipError_t e;
    const dim3 block(256);   // 4 waves = 4 tiles per workgroup
    auto grid_blocks = [&]() -> unsigned {
        if (a.xcd_affine == 1) return (unsigned)(a.ngroups * (((a.nchunks + 7) / 8) * 8));
        return (unsigned)(a.ngroups * a.nchunks);
    };
    auto set_list = [&](void *list, int n) { a.tile_list = static_cast<const uint32_t *>(list); a.nlist = n; a.ngroups = (n + 3) / 4; };
    if (balance || !lean) {
        // generic kernel over every tile (luminance round trip per tap, per-tile channel sums)
        set_list(nullptr, p.ntiles);
        const dim3 grid(grid_blocks());
        if (blend && balance) hipLaunchKernelGGL((k_stitch_plan<LX, true, true>), grid, block, 0, st, a);
        else if (balance) hipLaunchKernelGGL((k_stitch_plan<LX, false, true>), grid, block, 0, st, a);
        else if (blend) hipLaunchKernelGGL((k_stitch_plan<LX, true, false>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_stitch_plan<LX, false, false>), grid, block, 0, st, a);
        return hipGetLastError();
    }
    // sums = balance on pre-shifted frames: per-tile channel sums, the car is added by k_gain afterwards
    if (sums) a.car = nullptr;
    // class lists: with the pair-staged schedule the single / double classes split into the pair classes (list_pr[]) and
    // the sparse rest, which stays on the gather kernel (k_plan_lean)
    void *l_single = staged ? p.list_rp_single : p.list_single;
    void *l_double = staged ? p.list_rp_double : p.list_double;
    void *l_empty = staged ? p.list_rp_empty : p.list_empty;
    const int n_single = staged ? p.n_rp_single : p.n_single;
    const int n_double = staged ? p.n_rp_double : p.n_double;
    const int n_empty = staged ? p.n_rp_empty : p.n_empty;
#define BEVW_LAUNCH_CLASS(KERNEL, NS, SHMEM, ...)                                                                  \
    do {                                                                                                            \
        const dim3 grid(grid_blocks());                                                                             \
        if (blend && sums) hipLaunchKernelGGL((KERNEL<LX, NS, true, true __VA_ARGS__>), grid, block, SHMEM, st, a);  \
        else if (blend) hipLaunchKernelGGL((KERNEL<LX, NS, true, false __VA_ARGS__>), grid, block, SHMEM, st, a);    \
        else if (sums) hipLaunchKernelGGL((KERNEL<LX, NS, false, true __VA_ARGS__>), grid, block, SHMEM, st, a);     \
        else hipLaunchKernelGGL((KERNEL<LX, NS, false, false __VA_ARGS__>), grid, block, SHMEM, st, a);              \
        if ((e = hipGetLastError()) != hipSuccess) return e;                                                        \
    } while (0)
#define BEVW_COMMA ,
    // block-staged classes first: their blocks (8 waves, one barrier per frame) run longest
    // (running the block class on a second stream next to the per-wave classes was measured: no consistent gain, profiles/r02/sweeps.log)
    const bool bt_merged = staged && one_launch && bt_in_merged_launch && p.n_bt > 0;   // block tiles as a class of k_plan_all (4 waves each)
    if (staged && p.n_bt && !bt_merged) {
        a.tile_list = static_cast<const uint32_t *>(p.list_bt); a.nlist = p.n_bt; a.ngroups = p.n_bt;
        const dim3 grid(grid_blocks()), block8(512);
        if (blend && sums) hipLaunchKernelGGL((k_plan_block<true, true>), grid, block8, 0, st, a);
        else if (blend) hipLaunchKernelGGL((k_plan_block<true, false>), grid, block8, 0, st, a);
        else if (sums) hipLaunchKernelGGL((k_plan_block<false, true>), grid, block8, 0, st, a);
        else hipLaunchKernelGGL((k_plan_block<false, false>), grid, block8, 0, st, a);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (staged && !one_launch) {             // the unit classes as launches of their own (per-class mode)
#define BEVW_LAUNCH_UNIT(C)                                                                                                                  \
    if (p.n_un[C]) {                                                                                                                         \
        a.tile_list = static_cast<const uint32_t *>(p.list_un[C]); a.nlist = p.n_un[C]; a.ngroups = p.n_un[C];                               \
        const dim3 grid(grid_blocks());                                                                                                      \
        if (blend && sums) hipLaunchKernelGGL((k_plan_unit<true, true, C>), grid, block, 0, st, a);                                          \
        else if (blend) hipLaunchKernelGGL((k_plan_unit<true, false, C>), grid, block, 0, st, a);                                            \
        else if (sums) hipLaunchKernelGGL((k_plan_unit<false, true, C>), grid, block, 0, st, a);                                             \
        else hipLaunchKernelGGL((k_plan_unit<false, false, C>), grid, block, 0, st, a);                                                      \
        if ((e = hipGetLastError()) != hipSuccess) return e;                                                                                 \
    }
        BEVW_LAUNCH_UNIT(2) BEVW_LAUNCH_UNIT(3) BEVW_LAUNCH_UNIT(4) BEVW_LAUNCH_UNIT(5) BEVW_LAUNCH_UNIT(6) BEVW_LAUNCH_UNIT(1) BEVW_LAUNCH_UNIT(0)
#undef BEVW_LAUNCH_UNIT
    }
    if (staged && p.n_sm && !one_launch) {   // the seam block tiles as a launch of their own (per-class mode)
        a.tile_list = static_cast<const uint32_t *>(p.list_sm); a.nlist = p.n_sm; a.ngroups = p.n_sm;
        const dim3 grid(grid_blocks());
        if (blend && sums) hipLaunchKernelGGL((k_plan_seam<true, true>), grid, block, 0, st, a);
        else if (blend) hipLaunchKernelGGL((k_plan_seam<true, false>), grid, block, 0, st, a);
        else if (sums) hipLaunchKernelGGL((k_plan_seam<false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_plan_seam<false, false>), grid, block, 0, st, a);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (staged && one_launch) {
        PlanAllArgs q;
        q.a = a;
        // launch order: the classes whose blocks run longest first (the sliced and 4-round pair tiles, the sparse gather
        // tiles, then the shorter pair classes, then the empty tiles), so that the short blocks fill the tail of the grid
        // (profiles/r01_sweeps.log, profiles/r02/sweeps.log)
        struct Cls { int kind; void *list; int n; };
        // launch order: the per-wave classes with the longest-running blocks first (sliced and 4-round tiles ... 1-round tiles), the block
        // tiles behind them when the per-wave side holds a good part of the work (config 3 / blend: -1 % against block tiles first; a remap
        // plan is nearly all block tiles and they go first: +2.5 % otherwise), the empty tiles last (profiles/r02/sweeps.log, run49)
        int n_wave_side = n_single + n_double;
        for (int c = 0; c < Plan::kPairClasses; ++c) n_wave_side += p.n_pr[c];
        const bool bt_last = n_wave_side * 5 >= p.n_bt_tiles;
        const Cls bt_cls = {12, p.list_bt, bt_merged ? p.n_bt : 0}, none = {12, nullptr, 0};
        // units (bevw_unit.h) in front: they hold the bulk of the step; the classes with 4 rounds of groups run longest
        (void)unit_spatial;   // (class-by-class unit lists exist only as per-class launches: BEVW_PLAN_ONELAUNCH=0)
        const Cls cls[] = {{18, p.list_un_all, p.n_un_all},
                           bt_last ? none : bt_cls, {8, p.list_pr[3], p.n_pr[3]}, {11, p.list_pr[6], p.n_pr[6]}, {7, p.list_pr[2], p.n_pr[2]}, {3, l_single, n_single},
                           {10, p.list_pr[5], p.n_pr[5]}, {13, p.list_sm, p.n_sm}, {9, p.list_pr[4], p.n_pr[4]}, {6, p.list_pr[1], p.n_pr[1]},
                           {5, p.list_pr[0], p.n_pr[0]}, bt_last ? bt_cls : none, {2, l_empty, n_empty}};
        uint32_t at = 0;
        int np = 0;
        for (const Cls &c : cls) {
            if (!c.n) continue;
            q.kind[np] = c.kind;
            q.list[np] = static_cast<const uint32_t *>(c.list); q.nlist[np] = c.n; q.ngroups[np] = c.kind >= 12 ? c.n : (c.n + 3) / 4;
            q.start[np] = at;
            a.ngroups = q.ngroups[np];
            const unsigned nblk = c.kind == 2 ? (unsigned)(a.ngroups * a.nchunks) : grid_blocks();
            at += (nblk + 7u) & ~7u;
            ++np;
        }
        for (int i = np; i < kPlanAllMax; ++i) { q.kind[i] = -1; q.list[i] = nullptr; q.nlist[i] = 0; q.ngroups[i] = 1; }
        for (int i = np; i <= kPlanAllMax; ++i) q.start[i] = at;
        q.n = np;
        if (at) {
            if (blend && sums) hipLaunchKernelGGL((k_plan_all<LX, true, true>), dim3(at), block, 0, st, q);
            else if (blend) hipLaunchKernelGGL((k_plan_all<LX, true, false>), dim3(at), block, 0, st, q);
            else if (sums) hipLaunchKernelGGL((k_plan_all<LX, false, true>), dim3(at), block, 0, st, q);
            else hipLaunchKernelGGL((k_plan_all<LX, false, false>), dim3(at), block, 0, st, q);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        // the sparse two-contributor tiles: their 142-VGPR body would cap the merged kernel's occupancy (see k_plan_all)
        if (n_double) { set_list(l_double, n_double); BEVW_LAUNCH_CLASS(k_plan_lean, 2, 0); }
    } else {
        if (staged && p.n_pr[0]) { set_list(p.list_pr[0], p.n_pr[0]); BEVW_LAUNCH_CLASS(k_plan_pair, 1, 0, BEVW_COMMA 1 BEVW_COMMA 1); }
        if (staged && p.n_pr[1]) { set_list(p.list_pr[1], p.n_pr[1]); BEVW_LAUNCH_CLASS(k_plan_pair, 1, 0, BEVW_COMMA 1 BEVW_COMMA 2); }
        if (staged && p.n_pr[2]) { set_list(p.list_pr[2], p.n_pr[2]); BEVW_LAUNCH_CLASS(k_plan_pair, 1, 0, BEVW_COMMA 1 BEVW_COMMA 4); }
        if (staged && p.n_pr[3]) { set_list(p.list_pr[3], p.n_pr[3]); BEVW_LAUNCH_CLASS(k_plan_pair, 1, 0, BEVW_COMMA 4 BEVW_COMMA 2); }
        if (staged && p.n_pr[4]) { set_list(p.list_pr[4], p.n_pr[4]); BEVW_LAUNCH_CLASS(k_plan_pair, 2, 0, BEVW_COMMA 1 BEVW_COMMA 1); }
        if (staged && p.n_pr[5]) { set_list(p.list_pr[5], p.n_pr[5]); BEVW_LAUNCH_CLASS(k_plan_pair, 2, 0, BEVW_COMMA 1 BEVW_COMMA 2); }
        if (staged && p.n_pr[6]) { set_list(p.list_pr[6], p.n_pr[6]); BEVW_LAUNCH_CLASS(k_plan_pair, 2, 0, BEVW_COMMA 1 BEVW_COMMA 4); }
        if (n_empty) {
            set_list(l_empty, n_empty);
            hipLaunchKernelGGL((k_plan_empty<LX>), dim3((unsigned)(a.ngroups * a.nchunks)), block, 0, st, a);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        if (n_single) { set_list(l_single, n_single); BEVW_LAUNCH_CLASS(k_plan_lean, 1, lds_pad); }
        if (n_double) { set_list(l_double, n_double); BEVW_LAUNCH_CLASS(k_plan_lean, 2, 0); }
    }
#undef BEVW_LAUNCH_CLASS
#undef BEVW_COMMA
    if (p.n_slow) {
        set_list(p.list_slow, p.n_slow);
        const dim3 grid(grid_blocks());
        if (blend && sums) hipLaunchKernelGGL((k_stitch_plan<LX, true, false, true>), grid, block, 0, st, a);
        else if (blend) hipLaunchKernelGGL((k_stitch_plan<LX, true, false>), grid, block, 0, st, a);
        else if (sums) hipLaunchKernelGGL((k_stitch_plan<LX, false, false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_stitch_plan<LX, false, false>), grid, block, 0, st, a);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

// balance = per-tap luminance round trip on RAW frames (generic kernel); sums = frames are already luminance-shifted
// (k_lum_band), run the lean kernels and emit per-tile channel sums.  Both end with k_reduce_psums.
static inline hipError_t plan_stitch_impl(Plan &p, hipStream_t st, const uint8_t *d_frames, int batch, bool blend, bool balance,
                                          const int *d_deltas, const HsvTables *d_tab, const uint8_t *d_car,
                                          unsigned long long *d_chsums, uint8_t *d_out, const PlanTuning &tune, bool sums = false)
{
    hipError_t e;
    PlanArgs a;
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
    a.tiles_x = p.tiles_x; a.ntiles = p.ntiles; a.ngroups = (p.ntiles + 3) / 4;
    a.ncams = p.ncams;
    a.tile_list = nullptr; a.nlist = p.ntiles;
    a.plan_pr = static_cast<const uint2 *>(p.entries_pr);
    a.gsrc = static_cast<const uint32_t *>(p.gsrc);
    a.bt_entries = static_cast<const uint2 *>(p.bt_entries);
    a.bt_gsrc = static_cast<const uint32_t *>(p.bt_gsrc);
    a.bt_pos = static_cast<const uint32_t *>(p.bt_pos);
    a.sm_entries = static_cast<const uint2 *>(p.sm_entries);
    a.sm_gsrc = static_cast<const uint32_t *>(p.sm_gsrc);
    a.sm_pos = static_cast<const uint32_t *>(p.sm_pos);
    a.un_desc = static_cast<const UnitDesc *>(p.un_desc);
    a.un_entries = static_cast<const uint4 *>(p.un_entries);
    a.un_gsrc = static_cast<const uint32_t *>(p.un_gsrc);
    a.un_skew = p.un_skew;
    // pair-staged schedule: needs 4-byte aligned frame sets (dword-addressed group loads) and is not combined with the
    // per-tap luminance kernel
    const bool use_staged = !balance && tune.lean && tune.staged && p.paired_ok && (((uintptr_t)d_frames) & 3u) == 0;
    a.batch = batch;
    // frames per block: enough chunks to give each of the 8 XCDs whole chunks, otherwise one frame per chunk.  A block reads its plan
    // slice (4 bytes per pixel + the group list) once per chunk: 16 frames per block instead of 8 halve that traffic -- 1.8 M of the
    // 13.6 M read requests of a config-3 step -- for -5 % (direct), -7 % (blend) (profiles/r03/sweeps.log; 32 frames: the tail of 8 long
    // chunks costs more than it saves; round 2's 8-byte plan and class-ordered lists measured 16 slower)
    // Batches of 32 .. 127 frame sets: 8 frames per block even when that leaves fewer than 8 chunks (the 4K rig at batch 32: 4 chunks in
    // plain chunk-major order, -9 % against 8 chunks of 4 frames).  An explicit BEVW_PLAN_NB is taken as it is.
    int nb = tune.nb > 0 ? tune.nb : (batch >= 128 ? 16 : (batch >= 32 ? 8 : (batch >= 8 ? batch / 8 : 1)));
    if (nb > batch) nb = batch;
    a.nb = nb;
    a.nchunks = (batch + nb - 1) / nb;
    a.xcd_affine = (a.nchunks >= 8 && tune.xcd_map) ? 1 : 0;
    a.group_major = tune.group_major;
    if (balance || sums) {
        const size_t need = (size_t)batch * p.ntiles * 3 * sizeof(uint32_t);
        if (need > p.psums_cap) {
            if (p.psums) (void)hipFree(p.psums);
            p.psums = nullptr; p.psums_cap = 0;
            if ((e = hipMalloc(&p.psums, need)) != hipSuccess) return e;
            p.psums_cap = need;
        }
    }
    a.psums = static_cast<uint32_t *>(p.psums);
    if (sums && (e = hipMemsetAsync(p.psums, 0, (size_t)batch * p.ntiles * 3 * sizeof(uint32_t), st)) != hipSuccess) return e;
    switch (p.lx) {
        case 8: e = plan_launch_lx<8>(p, st, a, blend, balance, tune.lean != 0, tune.lds_pad, sums, use_staged, tune.one_launch != 0, tune.bt_merged != 0, tune.unit_spatial != 0); break;
        case 16: e = plan_launch_lx<16>(p, st, a, blend, balance, tune.lean != 0, tune.lds_pad, sums, use_staged, tune.one_launch != 0, tune.bt_merged != 0, tune.unit_spatial != 0); break;
        default: e = plan_launch_lx<4>(p, st, a, blend, balance, tune.lean != 0, tune.lds_pad, sums, use_staged, tune.one_launch != 0, tune.bt_merged != 0, tune.unit_spatial != 0); break;
    }
    if (e != hipSuccess) return e;
    if (balance || sums) {
        hipLaunchKernelGGL(k_reduce_psums, dim3(batch), dim3(256), 0, st, a.psums, p.ntiles, d_chsums);
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

// a plan that holds only a WIDE unit schedule (analytic projection mode: bevwarp.hip analytic_units_build) -> one launch of k_plan_unit_wide
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
    a.xcd_affine = (a.nchunks >= 8 && tune.xcd_map) ? 1 : 0;
    a.group_major = 0;
    a.tile_list = static_cast<const uint32_t *>(p.list_un_all); a.nlist = p.n_un_all; a.ngroups = p.n_un_all;
    const unsigned grid = a.xcd_affine ? (unsigned)(a.ngroups * (((a.nchunks + 7) / 8) * 8)) : (unsigned)(a.ngroups * a.nchunks);
    if (blend) hipLaunchKernelGGL((k_plan_unit_wide<true>), dim3(grid), dim3(kUnitThreads), 0, st, a);
    else hipLaunchKernelGGL((k_plan_unit_wide<false>), dim3(grid), dim3(kUnitThreads), 0, st, a);
    return hipGetLastError();
}

// luminance-shift the sampled texel groups of every raw frame of the batch into `scratch` (same layout as `frames`)
static inline hipError_t plan_lum_band(const Plan &p, hipStream_t st, const uint8_t *d_frames, uint8_t *d_scratch, int batch,
                                       const int *d_deltas, const HsvTables *d_tab)
{
    if (p.n_groups == 0) return hipSuccess;
    const size_t set_bytes = (size_t)p.fw * p.fh * 3 * p.ncams;
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        const unsigned bpf = (unsigned)(p.n_groups + 255) / 256;
        hipLaunchKernelGGL(k_lum_groups, dim3(xcd_frame_grid(bpf, (unsigned)nb)), dim3(256), 0, st, d_frames + (size_t)b0 * set_bytes,
                           d_scratch + (size_t)b0 * set_bytes, set_bytes, (uint32_t)p.fw * p.fh * 3,
                           static_cast<const uint32_t *>(p.groups), p.n_groups, d_deltas + (size_t)b0 * 4, d_tab, bpf, (uint32_t)nb);
    }
    return hipGetLastError();
}

}  // namespace bevw
