// bevw_pair.h -- the PAIR-STAGED schedule of the tile plan (included by bevw_plan.h; round 2).
//
// Why.  The sector-staged body (plan_staged_body) reads every 2x2 footprint back from LDS as raw interleaved BGR bytes
// at an arbitrary byte offset: two 16-byte windows per pixel (ds_read2_b64, 8 LDS cycles each) and 14 select / realign
// instructions in front of the dot products -- 124 integer VALU instructions (4.4 clk each on gfx950) and ~115 LDS cycles
// per tile-frame, issued from 2.8 waves per SIMD (142 VGPRs).  Here the realignment is done ONCE PER SOURCE TEXEL while
// the texels are staged instead of once per BEV pixel and tap:
//
//   * source texels are fetched in GROUPS: 16 bytes from a 4-byte aligned address 12 * g (g = group index inside the
//     4-camera frame set; rows are whole numbers of groups because fw % 4 == 0), i.e. texels 4g .. 4g+4 of one source
//     row, one global_load_dwordx4 per lane and round, consecutive lanes = consecutive groups of a row (row-run
//     requests: whole 128-byte lines instead of the single 64-byte sectors of the per-pixel gathers);
//   * each lane turns its 16 bytes into the four horizontal texel PAIRS (4g+k, 4g+k+1), k = 0..3, laid out for the dot
//     products: 8 bytes { b0 b1 g0 g1 | r0 r1 0 0 } (8 v_perm_b32), and stores them with two conflict-free
//     ds_write_b128 (pairs 0,1 of all lanes in the first KB of the round's patch, pairs 2,3 in the second);
//   * a BEV pixel reads one 8-byte pair entry per footprint row (ds_read_b64, 2 LDS cycles, 8-byte aligned by
//     construction) and needs 6 v_dot4 + 3 v_lshl_or + 3 v_dot2 + 2 v_perm: 14 VALU per pixel instead of 31.
//
// Frame b+1's groups are in flight in registers (one dwordx4 per round) while frame b is interpolated; the LDS patch is
// single-buffered and wave-private (program order of one wave orders the reads of frame b before the writes of b+1).
// Tiles whose footprints need more than kPairRounds * 64 groups stay on the L1-gather class.
// The arithmetic is the one of bilinear_rows_b2: (sum p * w + 512) >> 10 in the separable form, exact in integers.
#pragma once

namespace bevw {

constexpr uint32_t kHdrPaired = 128u;        // tile has a pair-staging plan; rounds - 1 in header bits 8..9
constexpr int kPairRounds = 4;               // max rounds (64 groups each) per tile-frame
constexpr int kPairRoundBytes = 2048;        // LDS per round: 64 groups x 4 pairs x 8 B
constexpr int kPairPatch = kPairRounds * kPairRoundBytes;   // LDS per wave

struct __attribute__((packed, aligned(4))) AlignedU4 { uint32_t x, y, z, w; };

// plan compiler: one wave per tile.  Reads the base entries (byte offset of the footprint, meta), collects the distinct
// groups in ascending address order (lane = slot % 64, round = slot / 64), and rewrites every entry to the LDS byte
// addresses of its two pair entries.  Interleaved tiles are rewritten to store order exactly as k_plan_stage_build does.
__global__ void __launch_bounds__(64) k_plan_pair_build(const uint2 *__restrict__ plan, uint32_t *__restrict__ hdr, int ntiles,
                                                         uint32_t row_bytes, uint32_t set_bytes, uint2 *__restrict__ plan_pr,
                                                         uint32_t *__restrict__ gsrc)
{
    constexpr int kMaxGroups = kPairRounds * 64;
    __shared__ uint32_t cand[64 * 16];
    __shared__ uint32_t list[kMaxGroups + 1];
    __shared__ int s_count;
    const int tile = blockIdx.x, lane = threadIdx.x;
    if (tile >= ntiles) return;
    const uint32_t h = hdr[tile];
    if (h & (kHdrSlow | kHdrEmpty)) return;
    const uint32_t gpr = row_bytes / 12u;   // groups per source row
    uint2 e[8];
    bool overrun = false;
    for (int k = 0; k < 8; ++k) {
        e[k] = plan[((size_t)tile * 8 + k) * 64 + lane];
        uint32_t c0 = 0xffffffffu, c1 = 0xffffffffu;
        if (e[k].y & kMetaValid) {
            c0 = e[k].x / 12u;       // group of texel pair (sx, sx+1) in row sy: offset = (row * fw + sx) * 3 = 12 * (row * gpr) + 3 * sx
            c1 = c0 + gpr;           // same columns, row sy + 1
            if ((size_t)c1 * 12u + 16u > (size_t)set_bytes) overrun = true;   // the 16-byte window of the last group would overrun
        }
        cand[lane * 16 + 2 * k] = c0;
        cand[lane * 16 + 2 * k + 1] = c1;
    }
    if (lane == 0) s_count = 0;
    __syncthreads();
    uint32_t last = 0;
    bool first = true, fits = !__any(overrun);
    for (int it = 0; it <= kMaxGroups && fits; ++it) {
        uint32_t m = 0xffffffffu;
        for (int i = 0; i < 16; ++i) {
            const uint32_t v = cand[lane * 16 + i];
            if (v != 0xffffffffu && (first || v > last)) m = min(m, v);
        }
        for (int off = 32; off > 0; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off, 64));
        if (m == 0xffffffffu) break;
        if (it == kMaxGroups) { fits = false; break; }
        if (lane == 0) { list[it] = m; s_count = it + 1; }
        last = m; first = false;
    }
    __syncthreads();
    const int count = s_count;
    if (!fits || count == 0) return;
    auto slot_of = [&](uint32_t key) {
        int lo = 0, hi = count;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (list[mid] < key) lo = mid + 1; else hi = mid; }
        return (uint32_t)lo;
    };
    auto lds_addr = [](uint32_t slot, uint32_t k) {
        return (slot >> 6) * (uint32_t)kPairRoundBytes + (k >> 1) * 1024u + (slot & 63u) * 16u + (k & 1u) * 8u;
    };
    const bool inter = (h & kHdrInterleaved) != 0;
    for (int k = 0; k < 8; ++k) {
        uint2 o = make_uint2(0u, e[k].y & ~kMetaValid);
        if (e[k].y & kMetaValid) {
            const uint32_t key = e[k].x / 12u, pk = (e[k].x - key * 12u) / 3u;
            o = make_uint2(lds_addr(slot_of(key), pk) | (lds_addr(slot_of(key + gpr), pk) << 16), e[k].y);
        }
        const int j = k & 3, sl = k & 4;
        const int dst_lane = inter ? (lane & ~3) + j : lane, dst_slot = inter ? sl + (lane & 3) : k;
        plan_pr[((size_t)tile * 8 + dst_slot) * 64 + dst_lane] = o;
    }
    for (int r = 0; r < kPairRounds; ++r) {
        const int slot = r * 64 + lane;
        gsrc[((size_t)tile * kPairRounds + r) * 64 + lane] = (slot < count ? list[slot] : list[0]) * 12u;
    }
    if (lane == 0) hdr[tile] = h | kHdrPaired | ((uint32_t)((count + 63) / 64 - 1) << 8);
}

// 16 source bytes (texels 4g .. 4g+4 of one row) -> the four pair entries, stored for lane `lane` of round patch `rp`
typedef uint32_t pair_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pair_u32x4 pair_load_group(const uint8_t *p)
{
    const AlignedU4 v = *reinterpret_cast<const AlignedU4 *>(p);   // one global_load_dwordx4 from a 4-byte aligned address
    return pair_u32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void pair_convert_store(const pair_u32x4 &d, uint8_t *rp, int lane)
{
    uint4 A, B;
    A.x = __builtin_amdgcn_perm(d.y, d.x, 0x04010300u);   // pair 0: b0 b1 g0 g1   (source bytes 0 3 1 4)
    A.y = __builtin_amdgcn_perm(d.y, d.x, 0x0c0c0502u);   //         r0 r1 0 0     (2 5)
    A.z = __builtin_amdgcn_perm(d.y, d.x, 0x07040603u);   // pair 1: bytes 3 6 4 7
    A.w = __builtin_amdgcn_perm(d.z, d.y, 0x0c0c0401u);   //         5 8
    B.x = __builtin_amdgcn_perm(d.z, d.y, 0x06030502u);   // pair 2: bytes 6 9 7 10
    B.y = __builtin_amdgcn_perm(d.z, d.z, 0x0c0c0300u);   //         8 11
    B.z = __builtin_amdgcn_perm(d.w, d.z, 0x05020401u);   // pair 3: bytes 9 12 10 13
    B.w = __builtin_amdgcn_perm(d.w, d.z, 0x0c0c0603u);   //         11 14
    reinterpret_cast<uint4 *>(rp)[lane] = A;
    reinterpret_cast<uint4 *>(rp + 1024)[lane] = B;
}

// one pixel from its two pair entries: accumulators with the result byte in bits 16..23 (as bilinear_rows_b2)
__device__ __forceinline__ void bilinear_pairs(uint2 q0, uint2 q1, uint32_t wxa, uint32_t wxb, uint32_t wy64, uint32_t acc[3])
{
    const uint32_t hb0 = __builtin_amdgcn_udot4(q0.x, wxa, 0u, false), hb1 = __builtin_amdgcn_udot4(q1.x, wxa, 0u, false);
    const uint32_t hg0 = __builtin_amdgcn_udot4(q0.x, wxb, 0u, false), hg1 = __builtin_amdgcn_udot4(q1.x, wxb, 0u, false);
    const uint32_t hr0 = __builtin_amdgcn_udot4(q0.y, wxa, 0u, false), hr1 = __builtin_amdgcn_udot4(q1.y, wxa, 0u, false);
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    union { uint32_t u; us2 v; } pb, pg, pr, w;
    pb.u = hb0 | (hb1 << 16); pg.u = hg0 | (hg1 << 16); pr.u = hr0 | (hr1 << 16); w.u = wy64;
    acc[0] = __builtin_amdgcn_udot2(pb.v, w.v, 32768u, false);
    acc[1] = __builtin_amdgcn_udot2(pg.v, w.v, 32768u, false);
    acc[2] = __builtin_amdgcn_udot2(pr.v, w.v, 32768u, false);
}

// 12 accumulators (result byte in bits 16..23) of a lane's 4 pixels -> the 12 output bytes B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
__device__ __forceinline__ void pack_accs(const uint32_t acc[4][3], uint32_t &d0, uint32_t &d1, uint32_t &d2)
{
    // perm(hi, lo, sel): byte 2 of lo = index 2, byte 2 of hi = index 6
    // the two halves of each output dword have zeros where the other half has data: combine with v_or_b32 (a 2-clk VOP2)
    d0 = __builtin_amdgcn_perm(acc[1][0], acc[0][2], 0x06020c0cu) | __builtin_amdgcn_perm(acc[0][1], acc[0][0], 0x0c0c0602u);
    d1 = __builtin_amdgcn_perm(acc[2][1], acc[2][0], 0x06020c0cu) | __builtin_amdgcn_perm(acc[1][2], acc[1][1], 0x0c0c0602u);
    d2 = __builtin_amdgcn_perm(acc[3][2], acc[3][1], 0x06020c0cu) | __builtin_amdgcn_perm(acc[3][0], acc[2][2], 0x0c0c0602u);
}

// one wave: tile from the class list, frames of the chunk.  lds: 4 * kPairPatch bytes (one patch per wave)
template <int LX, int NSLOT, bool BLEND, bool SUMS>
__device__ __forceinline__ void plan_pair_body(const PlanArgs &a, uint32_t block_id, uint8_t *lds)
{
    constexpr int LY = 64 / LX;
    uint32_t chunk, group;
    if (!plan_block_map(a, block_id, chunk, group)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = (int)group * 4 + wave;
    if (slot >= a.nlist) return;
    const int tile = (int)__builtin_amdgcn_readfirstlane(a.tile_list[slot]);
    const uint32_t hdr = __builtin_amdgcn_readfirstlane(a.hdr[tile]);
    const int nr = (int)((hdr >> 8) & 3u) + 1;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    int lx_, ly_;
    lane_xy(lane, LX, (hdr & kHdrTransposed) != 0, lx_, ly_);
    const int x0 = (tx * LX + lx_) * 4, y = ty * LY + ly_;
    const bool inimg = x0 < a.bw && y < a.bh;
    const size_t set_bytes = (size_t)a.fw * a.fh * 3 * a.ncams, img_bytes = (size_t)a.bw * a.bh * 3;
    const uint32_t ooff = ((uint32_t)y * a.bw + x0) * 3;
    uint8_t *const patch = lds + wave * kPairPatch;
    const uint2 *const pw = reinterpret_cast<const uint2 *>(patch);

    // per entry (store order): qword index of the two pair entries inside the wave's patch, x / y weights
    uint32_t i0[NSLOT][4], i1[NSLOT][4], wxa[NSLOT][4], wxb[NSLOT][4], wy[NSLOT][4];
    float wf[NSLOT][4];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint2 e = a.plan_pr[((size_t)tile * 8 + s * 4 + j) * 64 + lane];
            const uint32_t fx = e.y & 31, fy = (e.y >> 5) & 31;
            const bool valid = e.y & kMetaValid;
            i0[s][j] = (e.x & 0xffffu) >> 3; i1[s][j] = e.x >> 19;
            wxa[s][j] = valid ? ((32 - fx) | (fx << 8)) : 0u;   // zero x weights: an absent entry contributes exactly 0
            wxb[s][j] = wxa[s][j] << 16;
            wy[s][j] = ((32 - fy) << 6) | (fy << 22);           // y weights x 64
            wf[s][j] = BLEND ? blend_weight_f32((int)((e.y >> 10) & 255)) : 1.f;
        }
    uint32_t gs[kPairRounds];
#pragma unroll
    for (int r = 0; r < kPairRounds; ++r) gs[r] = r < nr ? a.gsrc[((size_t)tile * kPairRounds + r) * 64 + lane] : 0u;
    uint32_t car0 = 0, car1 = 0, car2 = 0;
    if (!SUMS && a.car != nullptr && inimg) {
        const uint32_t *cp = reinterpret_cast<const uint32_t *>(a.car + ooff);
        car0 = cp[0]; car1 = cp[1]; car2 = cp[2];
    }
    const bool car_any = __builtin_amdgcn_ballot_w64((car0 | car1 | car2) != 0) != 0;

    const int b_begin = (int)chunk * a.nb, b_end = min(a.batch, b_begin + a.nb);

    pair_u32x4 pf[kPairRounds];
    // wave-uniform base + 32-bit lane offset: the loads and the store use the SGPR-base addressing form
    auto issue = [&](int b) {
        const uint8_t *src = a.frames + (size_t)b * set_bytes;
#pragma unroll
        for (int r = 0; r < kPairRounds; ++r)
            if (r < nr) pf[r] = pair_load_group(src + gs[r]);
    };
    auto land = [&]() {
#pragma unroll
        for (int r = 0; r < kPairRounds; ++r)
            if (r < nr) pair_convert_store(pf[r], patch + r * kPairRoundBytes, lane);
    };
    issue(b_begin);
    land();
#pragma unroll 1
    for (int b = b_begin; b < b_end; ++b) {
        // frame b+1's groups travel while frame b is interpolated (past the end of the chunk: the last frame again, so that
        // the loop body is one straight line and the only vector-memory wait in it is the one in front of land())
        issue(min(b + 1, b_end - 1));
        uint32_t d0, d1, d2;
        if (!BLEND && NSLOT == 1 && !SUMS) {
            uint32_t acc[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) bilinear_pairs(pw[i0[0][j]], pw[i1[0][j]], wxa[0][j], wxb[0][j], wy[0][j], acc[j]);
            if (car_any) {
                uint32_t P[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    P[j] = __builtin_amdgcn_perm(acc[j][2], __builtin_amdgcn_perm(acc[j][1], acc[j][0], 0x0c0c0602u), 0x0c060100u);
                add_car(P, car0, car1, car2);
                pack_pixels(P, d0, d1, d2);
            } else {
                pack_accs(acc, d0, d1, d2);
            }
        } else {
            uint32_t P[4];
            int px[4][3];
#pragma unroll
            for (int s = 0; s < NSLOT; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t acc[3];
                    bilinear_pairs(pw[i0[s][j]], pw[i1[s][j]], wxa[s][j], wxb[s][j], wy[s][j], acc);
                    if (!BLEND && NSLOT == 1) {
                        P[j] = __builtin_amdgcn_perm(acc[2], __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u), 0x0c060100u);
                    } else {
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const uint32_t v = (acc[k] >> 16) & 255u;
                            const int c = BLEND ? (int)((float)v * wf[s][j]) : (int)v;
                            px[j][k] = s == 0 ? c : min(255, px[j][k] + c);
                        }
                    }
                }
            if (BLEND || NSLOT == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) P[j] = (uint32_t)px[j][0] | ((uint32_t)px[j][1] << 8) | ((uint32_t)px[j][2] << 16);
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
                if (lane == 0) {
                    uint32_t *ps = a.psums + ((size_t)b * a.ntiles + tile) * 3;
                    ps[0] = bg & 0xffffu; ps[1] = bg >> 16; ps[2] = sr;
                }
            }
            if (car_any) add_car(P, car0, car1, car2);
            pack_pixels(P, d0, d1, d2);
        }
        land();   // every LDS read of frame b precedes these writes in program order; waits for the loads only (the store
                  // of frame b is issued after it, the store of frame b-1 is older than the loads)
        if (inimg) {
            uint32_t *op = reinterpret_cast<uint32_t *>(a.out + (size_t)b * img_bytes + ooff);
            op[0] = d0; op[1] = d1; op[2] = d2;
        }
    }
}

// the pair-staged class as a kernel of its own (per-class launches: BEVW_PLAN_ONELAUNCH=0, and the unit profiles are taken on)
template <int LX, int NSLOT, bool BLEND, bool SUMS>
__global__ void __launch_bounds__(256) k_plan_pair(PlanArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t patch[4 * kPairPatch];
    plan_pair_body<LX, NSLOT, BLEND, SUMS>(a, blockIdx.x, patch);
}

}  // namespace bevw
