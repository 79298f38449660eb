// bevw_block.h -- the BLOCK-STAGED schedule of the tile plan (included by bevw_plan.h after bevw_pair.h; round 2).
//
// Why.  The stitch is bound by the NUMBER of vector-L1 -> L2 requests (profiles/r02/sweeps.log: the step follows
// reads + writes at ~80 G requests/s whatever their size; TA 97 % busy, the L1 stalled on L2 data 80 % of the time).  With
// one wave per 32 x 8 tile every wave fetches the 128-byte lines its own footprint touches -- 17.9 lines per 256 output
// pixels on BASELINE config 3, most of them shared with the neighbouring tiles (tools/analyze_requests.py) -- and writes
// rows of 96 bytes (19.1 sectors per 256 pixels, 14 of them partial).  Here the 8 waves of a block share ONE staged
// footprint:
//
//   * a block owns a 64 x 32 pixel block tile (= 2 x 4 base tiles); the distinct texel groups of the whole block tile are
//     fetched once (8.5 lines per 256 pixels when every block tile fits, -52 %), each wave loading 64 consecutive groups
//     of the ascending list per round, converted to pair entries exactly as in bevw_pair.h and stored to a patch the
//     block shares;
//   * wave w interpolates rows 4w .. 4w+3 of the block tile, 16 lanes x 4 pixels along x: its stores are rows of 192
//     contiguous bytes (15.5 sectors per 256 pixels, 7 partial);
//   * the patch is double-buffered per frame: the groups of frame b+1 are converted into the other half while frame b is
//     interpolated, ONE s_barrier per frame (lgkmcnt only -- the register prefetch of frame b+2 and the stores stay in
//     flight across it).
//
// Block tiles are compiled on the HOST at plan-build time (block_compile: the LUTs come back once, ~30 MB) for the block
// tiles whose pixels have at most one contributor each, no border footprint, and at most 512 distinct groups (one round
// = one group per lane of the block, 2 x 16 KB of LDS); their base tiles carry kHdrBlock and leave the per-wave classes.
// (A second class with two rounds -- <= 1024 groups, 64 KB, 2 blocks per CU -- and full 8-wave block tiles with two contributors
// per pixel were built and measured no faster than the per-wave classes they replaced: profiles/r02/sweeps.log.)
// Seams and blend overlaps get SEAM block tiles instead: 64 x 16 pixels, 4 waves, two contributors per pixel sharing one 512-group
// patch (seam_compile / plan_seam_body below; only as a class of the merged launch's 256-thread blocks, and not for balance handles).
// Sparse block tiles (near the car every pixel samples its own texels) stay on the per-wave pair classes.
#pragma once
#include <algorithm>
#include <vector>

namespace bevw {

constexpr uint32_t kHdrBlock = 1024u;          // base tile belongs to a block tile (bevw_block.h): not in the per-wave lists
constexpr int kBlockW = 64, kBlockH = 32;      // block tile in pixels = 2 x 4 base tiles of 32 x 8
constexpr int kBlockWaves = 8;
constexpr int kBlockRoundGroups = kBlockWaves * 64;              // groups per round: one per lane of the block
constexpr int kBlockMaxRounds = 1;   // rounds of 512 groups per block tile (a 2-round class, 64 KB of LDS, measured no faster: sweeps.log)
constexpr int kBlockRoundBytes = kBlockWaves * kPairRoundBytes;  // 16 KB of pair entries per round

struct BlockPlanHost {
    std::vector<uint2> entries;      // [nbt][8 waves][4 pixel slots][64 lanes]
    std::vector<uint32_t> gsrc;      // [nbt][rounds][8 waves][64 lanes]
    std::vector<uint32_t> pos;       // [nbt]  block-tile x | y << 16
    std::vector<uint32_t> list[kBlockMaxRounds];   // block tiles by rounds
};

// Host-side plan compiler of the block tiles.  tables: host copies of the LUTs of every camera.  hdr: base-tile headers
// (32 x 8 tiles, tiles_x per row); claimed base tiles get kHdrBlock.  The contributor rule is k_plan_build's.
static inline void block_compile(const std::vector<int16_t> lut1[4], const std::vector<uint16_t> lut2[4], const std::vector<uint8_t> mask[4],
                                 int ncams, int fw, int fh, int bw, int bh, int tiles_x, int tiles_y, std::vector<uint32_t> &hdr,
                                 BlockPlanHost &out)
{
    const int nbx = (bw + kBlockW - 1) / kBlockW, nby = (bh + kBlockH - 1) / kBlockH;
    const uint32_t frame_bytes = (uint32_t)fw * fh * 3, gpr = (uint32_t)fw / 4;
    const size_t set_bytes = (size_t)frame_bytes * ncams;
    auto lds_addr = [](uint32_t slot, uint32_t k) {
        return (slot >> 6) * (uint32_t)kPairRoundBytes + (k >> 1) * 1024u + (slot & 63u) * 16u + (k & 1u) * 8u;
    };
    std::vector<uint32_t> keys;
    std::vector<uint2> base((size_t)kBlockW * kBlockH);   // per pixel: source byte offset, meta (0 = no contributor)
    for (int by = 0; by < nby; ++by)
        for (int bx = 0; bx < nbx; ++bx) {
            // the 8 base tiles: all present ones single-contributor and border-free, at least one not empty
            bool ok = true, any = false;
            for (int k = 0; k < 8 && ok; ++k) {
                const int tx = 2 * bx + (k & 1), ty = 4 * by + (k >> 1);
                if (tx >= tiles_x || ty >= tiles_y) continue;
                const uint32_t h = hdr[(size_t)ty * tiles_x + tx];
                if (h & (kHdrSlow | kHdrSecond)) ok = false;
                if (!(h & kHdrEmpty)) any = true;
            }
            if (!ok || !any) continue;
            keys.clear();
            for (int py = 0; py < kBlockH && ok; ++py)
                for (int px = 0; px < kBlockW; ++px) {
                    uint2 e = make_uint2(0u, 0u);
                    const int x = bx * kBlockW + px, y = by * kBlockH + py;
                    if (x < bw && y < bh) {
                        const size_t o = (size_t)y * bw + x;
                        for (int c = 0; c < ncams; ++c) {
                            const uint32_t m = mask[c][o];
                            if (m == 0) continue;
                            const int sx = lut1[c][o * 2], sy = lut1[c][o * 2 + 1];
                            if (sx >= fw || sx + 1 < 0 || sy >= fh || sy + 1 < 0) continue;   // whole footprint outside: adds 0
                            const uint32_t code = lut2[c][o] & (kQTab2 - 1);
                            const uint32_t off = (uint32_t)c * frame_bytes + ((uint32_t)sy * fw + sx) * 3;
                            const uint32_t key = off / 12u;
                            if ((size_t)(key + gpr) * 12u + 16u > set_bytes) ok = false;      // the last group's 16-byte window
                            e = make_uint2(off, code | (m << 10) | ((uint32_t)c << 18) | kMetaValid);
                            keys.push_back(key);
                            keys.push_back(key + gpr);
                            break;   // single-contributor tiles: the first contributor is the only one
                        }
                    }
                    base[(size_t)py * kBlockW + px] = e;
                }
            if (!ok) continue;
            std::sort(keys.begin(), keys.end());
            keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
            const int count = (int)keys.size();
            if (count == 0 || count > kBlockMaxRounds * kBlockRoundGroups) continue;
            const int rounds = (count + kBlockRoundGroups - 1) / kBlockRoundGroups;
            const uint32_t id = (uint32_t)out.pos.size();
            out.pos.push_back((uint32_t)bx | ((uint32_t)by << 16));
            out.list[rounds - 1].push_back(id);
            auto slot_of = [&](uint32_t key) { return (uint32_t)(std::lower_bound(keys.begin(), keys.end(), key) - keys.begin()); };
            out.entries.resize((size_t)(id + 1) * kBlockWaves * 4 * 64);
            uint2 *ent = out.entries.data() + (size_t)id * kBlockWaves * 4 * 64;
            for (int w = 0; w < kBlockWaves; ++w)
                for (int j = 0; j < 4; ++j)
                    for (int lane = 0; lane < 64; ++lane) {
                        const uint2 e = base[(size_t)(w * 4 + (lane >> 4)) * kBlockW + (lane & 15) * 4 + j];
                        uint2 o = make_uint2(0u, 0u);
                        if (e.y & kMetaValid) {
                            const uint32_t key = e.x / 12u, pk = (e.x - key * 12u) / 3u;
                            o = make_uint2(lds_addr(slot_of(key), pk) | (lds_addr(slot_of(key + gpr), pk) << 16), e.y);
                        }
                        ent[((size_t)w * 4 + j) * 64 + lane] = o;
                    }
            out.gsrc.resize((size_t)(id + 1) * kBlockMaxRounds * kBlockRoundGroups);
            uint32_t *gs = out.gsrc.data() + (size_t)id * kBlockMaxRounds * kBlockRoundGroups;
            for (int s = 0; s < kBlockMaxRounds * kBlockRoundGroups; ++s) gs[s] = s < count ? keys[(size_t)s] * 12u : kPairNoGroup;
            for (int k = 0; k < 8; ++k) {
                const int tx = 2 * bx + (k & 1), ty = 4 * by + (k >> 1);
                if (tx < tiles_x && ty < tiles_y) hdr[(size_t)ty * tiles_x + tx] |= kHdrBlock;
            }
        }
}

// Seam block tiles: 64 x 16 pixels (2 x 2 base tiles), 4 waves, at most TWO contributors per pixel (seams, blend overlaps).  Both
// contributors' groups share the 512-group patch (two loads per lane); only tiles left unclaimed by block_compile and holding at least
// one two-contributor base tile are taken.  entries: [nbt][4 sub-tiles][2 contributors][4 pixel slots][64 lanes].
struct SeamPlanHost {
    std::vector<uint2> entries;
    std::vector<uint32_t> gsrc;      // [nbt][8 slot waves][64 lanes]
    std::vector<uint32_t> pos;       // [nbt]  block-tile x | (first pixel row / 16) << 16
    std::vector<uint32_t> list;
};
constexpr int kSeamH = 16;
static inline void seam_compile(const std::vector<int16_t> lut1[4], const std::vector<uint16_t> lut2[4], const std::vector<uint8_t> mask[4],
                                int ncams, int fw, int fh, int bw, int bh, int tiles_x, int tiles_y, std::vector<uint32_t> &hdr, SeamPlanHost &out)
{
    const int nbx = (bw + kBlockW - 1) / kBlockW, nby = (bh + kSeamH - 1) / kSeamH;
    const uint32_t frame_bytes = (uint32_t)fw * fh * 3, gpr = (uint32_t)fw / 4;
    const size_t set_bytes = (size_t)frame_bytes * ncams;
    constexpr size_t kEntPerTile = (size_t)4 * 2 * 4 * 64;
    auto lds_addr = [](uint32_t slot, uint32_t k) {
        return (slot >> 6) * (uint32_t)kPairRoundBytes + (k >> 1) * 1024u + (slot & 63u) * 16u + (k & 1u) * 8u;
    };
    std::vector<uint32_t> keys;
    std::vector<uint2> base((size_t)2 * kBlockW * kSeamH);   // [contributor][pixel]
    for (int by = 0; by < nby; ++by)
        for (int bx = 0; bx < nbx; ++bx) {
            bool ok = true, seam = false;
            for (int k = 0; k < 4 && ok; ++k) {
                const int tx = 2 * bx + (k & 1), ty = 2 * by + (k >> 1);
                if (tx >= tiles_x || ty >= tiles_y) continue;
                const uint32_t h = hdr[(size_t)ty * tiles_x + tx];
                if (h & (kHdrSlow | kHdrBlock)) ok = false;
                if (h & kHdrSecond) seam = true;
            }
            if (!ok || !seam) continue;
            keys.clear();
            for (int py = 0; py < kSeamH && ok; ++py)
                for (int px = 0; px < kBlockW; ++px) {
                    uint2 e[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
                    int count = 0;
                    const int x = bx * kBlockW + px, y = by * kSeamH + py;
                    if (x < bw && y < bh) {
                        const size_t o = (size_t)y * bw + x;
                        for (int c = 0; c < ncams; ++c) {
                            const uint32_t m = mask[c][o];
                            if (m == 0) continue;
                            const int sx = lut1[c][o * 2], sy = lut1[c][o * 2 + 1];
                            if (sx >= fw || sx + 1 < 0 || sy >= fh || sy + 1 < 0) continue;   // whole footprint outside: adds 0
                            const uint32_t code = lut2[c][o] & (kQTab2 - 1);
                            const uint32_t off = (uint32_t)c * frame_bytes + ((uint32_t)sy * fw + sx) * 3;
                            const uint32_t key = off / 12u;
                            if ((size_t)(key + gpr) * 12u + 16u > set_bytes) ok = false;
                            if (count < 2) {
                                e[count] = make_uint2(off, code | (m << 10) | ((uint32_t)c << 18) | kMetaValid);
                                keys.push_back(key);
                                keys.push_back(key + gpr);
                            }
                            ++count;
                        }
                    }
                    if (count > 2) ok = false;
                    base[(size_t)py * kBlockW + px] = e[0];
                    base[(size_t)kBlockW * kSeamH + (size_t)py * kBlockW + px] = e[1];
                }
            if (!ok) continue;
            std::sort(keys.begin(), keys.end());
            keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
            const int count = (int)keys.size();
            if (count == 0 || count > kBlockRoundGroups) continue;
            const uint32_t id = (uint32_t)out.pos.size();
            out.pos.push_back((uint32_t)bx | ((uint32_t)by << 16));
            out.list.push_back(id);
            auto slot_of = [&](uint32_t key) { return (uint32_t)(std::lower_bound(keys.begin(), keys.end(), key) - keys.begin()); };
            out.entries.resize((size_t)(id + 1) * kEntPerTile);
            uint2 *ent = out.entries.data() + (size_t)id * kEntPerTile;
            for (int w = 0; w < 4; ++w)
                for (int sc = 0; sc < 2; ++sc)
                    for (int j = 0; j < 4; ++j)
                        for (int lane = 0; lane < 64; ++lane) {
                            const uint2 e = base[(size_t)sc * kBlockW * kSeamH + (size_t)(w * 4 + (lane >> 4)) * kBlockW + (lane & 15) * 4 + j];
                            uint2 o = make_uint2(0u, 0u);
                            if (e.y & kMetaValid) {
                                const uint32_t key = e.x / 12u, pk = (e.x - key * 12u) / 3u;
                                o = make_uint2(lds_addr(slot_of(key), pk) | (lds_addr(slot_of(key + gpr), pk) << 16), e.y);
                            }
                            ent[(((size_t)w * 2 + sc) * 4 + j) * 64 + lane] = o;
                        }
            out.gsrc.resize((size_t)(id + 1) * kBlockRoundGroups);
            uint32_t *gs = out.gsrc.data() + (size_t)id * kBlockRoundGroups;
            for (int sidx = 0; sidx < kBlockRoundGroups; ++sidx) gs[sidx] = sidx < count ? keys[(size_t)sidx] * 12u : kPairNoGroup;
            for (int k = 0; k < 4; ++k) {
                const int tx = 2 * bx + (k & 1), ty = 2 * by + (k >> 1);
                if (tx < tiles_x && ty < tiles_y) hdr[(size_t)ty * tiles_x + tx] |= kHdrBlock;
            }
        }
}

// s_barrier that waits for this wave's LDS traffic only: the register prefetch (vmcnt) and the stores stay in flight
__device__ __forceinline__ void block_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one block: block tile from the class list, frames of the chunk.  lds: 2 x 16 KB.
// NSUB 1: 8 waves, wave w owns rows 4w .. 4w+3 (k_plan_block, 512 threads).  NSUB 2: 4 waves, wave w owns rows 4w .. 4w+3 and
// 16+4w .. 16+4w+3 -- two groups, eight pixels and two stores per lane and frame -- so that the block tile fits the 256-thread blocks
// of the merged launch (k_plan_all).
template <bool BLEND, bool SUMS, int NSUB>
__device__ __forceinline__ void plan_block_body(const PlanArgs &a, uint32_t block_id, uint8_t *lds)
{
    static_assert(NSUB == 1 || NSUB == 2, "8 or 4 waves per block tile");
    constexpr int kWaves = kBlockWaves / NSUB;
    uint32_t chunk, group;
    if (!plan_block_map(a, block_id, chunk, group)) return;   // uniform over the block
    if ((int)group >= a.nlist) return;
    const uint32_t bt = __builtin_amdgcn_readfirstlane(a.tile_list[group]);
    const uint32_t pos = __builtin_amdgcn_readfirstlane(a.bt_pos[bt]);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bx = (int)(pos & 0xffffu), by = (int)(pos >> 16);
    const int x0 = bx * kBlockW + (lane & 15) * 4;
    const size_t set_bytes = (size_t)a.fw * a.fh * 3 * a.ncams, img_bytes = (size_t)a.pitch * a.bh * 3;
    constexpr int kHalf = kBlockRoundBytes;   // one frame's patch

    uint32_t i0[NSUB][4], i1[NSUB][4], wxa[NSUB][4], wy[NSUB][4], gs[NSUB], ooff[NSUB], ooff_masked[NSUB];
    uint32_t car[NSUB][3];
    float wf[NSUB][4];
    int sum_tile[NSUB];
    bool sum_plain;
    uint32_t car_or = 0;
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        const int vw = wave + kWaves * s;   // the wave of the 8-wave layout this sub-tile belongs to
        const int y = by * kBlockH + vw * 4 + (lane >> 4);
        const bool inimg = x0 < a.bw && y < a.bh;
        ooff[s] = ((uint32_t)y * a.pitch + x0) * 3;
        ooff_masked[s] = inimg ? ooff[s] : kPairNoGroup;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint2 e = a.bt_entries[(((size_t)bt * kBlockWaves + vw) * 4 + j) * 64 + lane];
            const uint32_t fx = e.y & 31, fy = (e.y >> 5) & 31;
            const bool valid = e.y & kMetaValid;
            i0[s][j] = (e.x & 0xffffu) >> 3; i1[s][j] = e.x >> 19;
            wxa[s][j] = valid ? ((32 - fx) | (fx << 8)) : 0u;   // zero x weights: an absent entry contributes exactly 0
            wy[s][j] = ((32 - fy) << 6) | (fy << 22);
            wf[s][j] = BLEND ? blend_weight_f32((int)((e.y >> 10) & 255)) : 1.f;
        }
        gs[s] = a.bt_gsrc[((size_t)bt * kBlockWaves + vw) * 64 + lane];
        car[s][0] = car[s][1] = car[s][2] = 0;
        if (!SUMS && a.car != nullptr && inimg) {
            const uint32_t *cp = reinterpret_cast<const uint32_t *>(a.car + ooff[s]);
            car[s][0] = cp[0]; car[s][1] = cp[1]; car[s][2] = cp[2];
        }
        car_or |= car[s][0] | car[s][1] | car[s][2];
        // balance: sub-tile vw owns the channel-sum slot of base tile vw of the block tile (only the per-frame total is used); where
        // that base tile does not exist (right / bottom edge) it adds into the first one (plan_stitch_impl zeroes psums)
        const int sum_tx = 2 * bx + (vw & 1), sum_ty = 4 * by + (vw >> 1);
        const bool sum_own = sum_tx < a.tiles_x && sum_ty * a.tiles_x + sum_tx < a.ntiles;
        sum_tile[s] = sum_own ? sum_ty * a.tiles_x + sum_tx : 4 * by * a.tiles_x + 2 * bx;
    }
    sum_plain = 2 * bx + 1 < a.tiles_x && (4 * by + 3) * a.tiles_x + 2 * bx + 1 < a.ntiles;   // all 8 base tiles exist
    const bool car_any = __builtin_amdgcn_ballot_w64(car_or != 0) != 0;

    const int b_begin = (int)chunk * a.nb, b_end = min(a.batch, b_begin + a.nb);
    constexpr int D = 2;
    pair_u32x4 pf[D][NSUB];
    auto issue = [&](int b, int ring) {
        const uint8_t *src = a.frames + (size_t)min(b, b_end - 1) * set_bytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, (uint32_t)set_bytes, kBufferWord3);
#pragma unroll
        for (int s = 0; s < NSUB; ++s) pf[ring][s] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)gs[s], 0, kPairLoadAux);
    };
    auto land = [&](int ring) {   // frame parity == ring == patch half
#pragma unroll
        for (int s = 0; s < NSUB; ++s)
            pair_convert_store(pf[ring][s], lds + ring * kHalf + (wave + kWaves * s) * kPairRoundBytes, lane);
    };
    constexpr bool kFast = !BLEND && !SUMS;
    auto acc_to_px = [](const uint32_t acc[3]) {
        return __builtin_amdgcn_perm(acc[2], __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u), 0x0c060100u);
    };
    auto frame = [&](int b, int ring) {
        const uint2 *const pw = reinterpret_cast<const uint2 *>(lds + ring * kHalf);
        issue(b + D, ring);   // the ring slot of frame b was converted one step ago
        uint32_t d[NSUB][3];
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            uint32_t acc[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) bilinear_pairs(pw[i0[s][j]], pw[i1[s][j]], wxa[s][j], wxa[s][j] << 16, wy[s][j], acc[j]);
            if (kFast && !car_any) {
                pack_accs(acc, d[s][0], d[s][1], d[s][2]);
            } else {
                uint32_t P[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (BLEND) {
                        uint32_t px = 0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) px |= (uint32_t)(int)((float)((acc[j][k] >> 16) & 255u) * wf[s][j]) << (8 * k);
                        P[j] = px;
                    } else {
                        P[j] = acc_to_px(acc[j]);
                    }
                }
                if (SUMS) {
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
                    if (lane == 0 && b < b_end) {
                        uint32_t *ps = a.psums + ((size_t)b * a.ntiles + sum_tile[s]) * 3;
                        if (sum_plain) { ps[0] = bg & 0xffffu; ps[1] = bg >> 16; ps[2] = sr; }
                        else { atomicAdd(ps + 0, bg & 0xffffu); atomicAdd(ps + 1, bg >> 16); atomicAdd(ps + 2, sr); }
                    }
                }
                if (car_any) add_car(P, car[s][0], car[s][1], car[s][2]);
                pack_pixels(P, d[s][0], d[s][1], d[s][2]);
            }
        }
        land(ring ^ 1);       // frame b+1 into the other half: nobody reads it before the barrier
        {
            uint8_t *img = a.out + (size_t)min(b, b_end - 1) * img_bytes;   // past the chunk: re-writes the last frame with the same bytes
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(img, 0, (uint32_t)img_bytes, kBufferWord3);
#pragma unroll
            for (int s = 0; s < NSUB; ++s)
                __builtin_amdgcn_raw_buffer_store_b96(pair_u32x3{d[s][0], d[s][1], d[s][2]}, ro, (int)ooff_masked[s], 0, kPairStoreAux);
        }
        block_lds_barrier();  // half[ring ^ 1] complete for everybody; half[ring] free for frame b+2
    };
#pragma unroll
    for (int u = 0; u < D; ++u) issue(b_begin + u, u);
    land(0);
    block_lds_barrier();
#pragma unroll 1
    for (int b = b_begin; b < b_end; b += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) frame(b + u, u);
    }
}

// the block-staged class as a kernel of its own: 8 waves per block tile (per-class launches: BEVW_PLAN_ONELAUNCH=0)
template <bool BLEND, bool SUMS>
__global__ void __launch_bounds__(512) k_plan_block(PlanArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t patch[2 * kBlockRoundBytes];
    plan_block_body<BLEND, SUMS, 1>(a, blockIdx.x, patch);
}

// one block of 4 waves: seam block tile (64 x 16, two contributors per pixel) from the class list, frames of the chunk.  Wave w owns rows
// 4w .. 4w+3; every lane loads two groups per frame (slot waves w and w + 4); the second contributor is added with saturation (cv2.add,
// surroundBEV.py:318-320).  lds: 2 x 16 KB.
template <bool BLEND, bool SUMS>
__device__ __forceinline__ void plan_seam_body(const PlanArgs &a, uint32_t block_id, uint8_t *lds)
{
    uint32_t chunk, group;
    if (!plan_block_map(a, block_id, chunk, group)) return;   // uniform over the block
    if ((int)group >= a.nlist) return;
    const uint32_t bt = __builtin_amdgcn_readfirstlane(a.tile_list[group]);
    const uint32_t pos = __builtin_amdgcn_readfirstlane(a.sm_pos[bt]);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bx = (int)(pos & 0xffffu), y0 = (int)(pos >> 16) * kSeamH;
    const int x0 = bx * kBlockW + (lane & 15) * 4, y = y0 + wave * 4 + (lane >> 4);
    const bool inimg = x0 < a.bw && y < a.bh;
    const size_t set_bytes = (size_t)a.fw * a.fh * 3 * a.ncams, img_bytes = (size_t)a.pitch * a.bh * 3;
    const uint32_t ooff = ((uint32_t)y * a.pitch + x0) * 3;
    const uint32_t ooff_masked = inimg ? ooff : kPairNoGroup;
    constexpr int kHalf = kBlockRoundBytes;

    uint32_t i0[2][4], i1[2][4], wxa[2][4], wy[2][4], gs[2];
    float wf[2][4];
#pragma unroll
    for (int sc = 0; sc < 2; ++sc)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint2 e = a.sm_entries[((((size_t)bt * 4 + wave) * 2 + sc) * 4 + j) * 64 + lane];
            const uint32_t fx = e.y & 31, fy = (e.y >> 5) & 31;
            const bool valid = e.y & kMetaValid;
            i0[sc][j] = (e.x & 0xffffu) >> 3; i1[sc][j] = e.x >> 19;
            wxa[sc][j] = valid ? ((32 - fx) | (fx << 8)) : 0u;   // zero x weights: an absent entry contributes exactly 0
            wy[sc][j] = ((32 - fy) << 6) | (fy << 22);
            wf[sc][j] = BLEND ? blend_weight_f32((int)((e.y >> 10) & 255)) : 1.f;
        }
#pragma unroll
    for (int r = 0; r < 2; ++r) gs[r] = a.sm_gsrc[((size_t)bt * kBlockWaves + wave + 4 * r) * 64 + lane];
    uint32_t car0 = 0, car1 = 0, car2 = 0;
    if (!SUMS && a.car != nullptr && inimg) {
        const uint32_t *cp = reinterpret_cast<const uint32_t *>(a.car + ooff);
        car0 = cp[0]; car1 = cp[1]; car2 = cp[2];
    }
    const bool car_any = __builtin_amdgcn_ballot_w64((car0 | car1 | car2) != 0) != 0;
    const int sum_tx = 2 * bx + (wave & 1), sum_ty = y0 / 8 + (wave >> 1);
    const bool sum_own = sum_tx < a.tiles_x && sum_ty * a.tiles_x + sum_tx < a.ntiles;
    const bool sum_plain = 2 * bx + 1 < a.tiles_x && (y0 / 8 + 1) * a.tiles_x + 2 * bx + 1 < a.ntiles;   // all 4 base tiles exist
    const int sum_tile = sum_own ? sum_ty * a.tiles_x + sum_tx : (y0 / 8) * a.tiles_x + 2 * bx;

    const int b_begin = (int)chunk * a.nb, b_end = min(a.batch, b_begin + a.nb);
    constexpr int D = 2;
    pair_u32x4 pf[D][2];
    auto issue = [&](int b, int ring) {
        const uint8_t *src = a.frames + (size_t)min(b, b_end - 1) * set_bytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, (uint32_t)set_bytes, kBufferWord3);
#pragma unroll
        for (int r = 0; r < 2; ++r) pf[ring][r] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)gs[r], 0, kPairLoadAux);
    };
    auto land = [&](int ring) {
#pragma unroll
        for (int r = 0; r < 2; ++r) pair_convert_store(pf[ring][r], lds + ring * kHalf + (wave + 4 * r) * kPairRoundBytes, lane);
    };
    auto frame = [&](int b, int ring) {
        const uint2 *const pw = reinterpret_cast<const uint2 *>(lds + ring * kHalf);
        issue(b + D, ring);
        uint32_t P[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int px[3];
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) {
                uint32_t acc[3];
                bilinear_pairs(pw[i0[sc][j]], pw[i1[sc][j]], wxa[sc][j], wxa[sc][j] << 16, wy[sc][j], acc);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint32_t v = (acc[k] >> 16) & 255u;
                    const int c = BLEND ? (int)((float)v * wf[sc][j]) : (int)v;
                    px[k] = sc == 0 ? c : min(255, px[k] + c);
                }
            }
            P[j] = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
        }
        if (SUMS) {
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
            if (lane == 0 && b < b_end) {
                uint32_t *ps = a.psums + ((size_t)b * a.ntiles + sum_tile) * 3;
                if (sum_plain) { ps[0] = bg & 0xffffu; ps[1] = bg >> 16; ps[2] = sr; }
                else { atomicAdd(ps + 0, bg & 0xffffu); atomicAdd(ps + 1, bg >> 16); atomicAdd(ps + 2, sr); }
            }
        }
        if (car_any) add_car(P, car0, car1, car2);
        uint32_t d0, d1, d2;
        pack_pixels(P, d0, d1, d2);
        land(ring ^ 1);
        {
            uint8_t *img = a.out + (size_t)min(b, b_end - 1) * img_bytes;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(img, 0, (uint32_t)img_bytes, kBufferWord3);
            __builtin_amdgcn_raw_buffer_store_b96(pair_u32x3{d0, d1, d2}, ro, (int)ooff_masked, 0, kPairStoreAux);
        }
        block_lds_barrier();
    };
#pragma unroll
    for (int u = 0; u < D; ++u) issue(b_begin + u, u);
    land(0);
    block_lds_barrier();
#pragma unroll 1
    for (int b = b_begin; b < b_end; b += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) frame(b + u, u);
    }
}

template <bool BLEND, bool SUMS>
__global__ void __launch_bounds__(256) k_plan_seam(PlanArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t patch[2 * kBlockRoundBytes];
    plan_seam_body<BLEND, SUMS>(a, blockIdx.x, patch);
}

}  // namespace bevw
