// bevw_pair.h -- the PAIR-STAGED schedule of the tile plan (included by bevw_plan.h; round 2).
//
// Why.  Round 1's sector-staged body (removed; profiles/r01/, profiles/r02/sweeps.log) read every 2x2 footprint back from
// LDS as raw interleaved BGR bytes at an arbitrary byte offset: two 16-byte windows per pixel (ds_read2_b64, 8 LDS cycles
// each) and 14 select / realign instructions in front of the dot products -- 124 integer VALU instructions (4.7 clk each
// on gfx950, profiles/r02/valu_rates.log) and ~115 LDS cycles per tile-frame, issued from 2.8 waves per SIMD (142 VGPRs).
// Here the realignment is done ONCE PER SOURCE TEXEL while the texels are staged instead of once per BEV pixel and tap:
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
// Modes (per tile, chosen by the plan compiler; header bits 8..9):
//   whole-tile staging, 1 / 2 / 4 rounds: every group of the tile is staged, then the lane's 4 pixels are interpolated
//                                        (neighbouring pixels share groups; <= 256 groups per tile)
//   sliced staging, 2 rounds per slice : sparse tiles (near the car every pixel samples its own texels: up to 2 groups per
//                                        pixel and row pair, 512 per tile).  Pixel slot j of all lanes (64 pixels) is a
//                                        slice with its own group list of <= 128 groups; the four slices of a frame are
//                                        staged and interpolated one after the other through the same patch.  No tile is
//                                        left for the per-pixel L1 gathers (8 gather instructions + 28 VALU per pixel).
// Pipeline: the groups of the next TWO steps (step = frame, or slice of a frame) are in flight in registers (one dwordx4
// per round and step) while the current step is interpolated; the LDS patch (<= 8 KB per wave) is single-buffered and
// wave-private (program order of one wave orders the reads of step t before the writes of step t+1).  The loop body is
// straight-line code (frame indices past the end of the chunk are clamped), so the compiler's vector-memory waits are
// exact: the wait in front of a conversion covers the loads of that step only, never the younger loads or the store.
// The arithmetic is the one of bilinear_rows_b2: (sum p * w + 512) >> 10 in the separable form, exact in integers.
#pragma once

namespace bevw {

constexpr uint32_t kHdrPaired = 128u;        // tile has a pair-staging plan; mode in header bits 8..9
constexpr int kPairRoundBytes = 2048;        // LDS per round: 64 groups x 4 pairs x 8 B
constexpr int kPairMaxRounds = 4;            // rounds per step of whole-tile staging
constexpr int kPairSliceRounds = 2;          // rounds per step of sliced staging
constexpr int kPairPatch = kPairMaxRounds * kPairRoundBytes;   // LDS per wave
constexpr int kPairSrcSlots = 8;             // gsrc entries per tile and lane: [round] (whole tile) or [slice][2 rounds]
// mode -> (slices, rounds): 0 = (1, 1), 1 = (1, 2), 2 = (1, 4), 3 = (4, 2)
constexpr int kPairModes = 4;
constexpr uint32_t kPairNoGroup = 0x80000000u;   // source offset of a lane without a group (frame sets are < 2 GB)
constexpr uint32_t kBufferWord3 = 0x00020000u;   // raw buffer descriptor, dword 3 (gfx9 family: DATA_FORMAT 32)
// cache-policy bits of the group loads / tile stores (gfx94x/95x: 1 = sc0, 2 = nt, 16 = sc1).  Measured: see
// profiles/r02/sweeps.log ("cache policy"); the defaults are what ships.
#ifndef BEVW_LOAD_AUX
#define BEVW_LOAD_AUX 0
#endif
#ifndef BEVW_STORE_AUX
#define BEVW_STORE_AUX 0
#endif
constexpr int kPairLoadAux = BEVW_LOAD_AUX, kPairStoreAux = BEVW_STORE_AUX;

struct __attribute__((packed, aligned(4))) AlignedU4 { uint32_t x, y, z, w; };

// distinct values of cand[lane * 16 + i] (i in the bit mask `use`) in ascending order -> list[], at most maxn;
// returns the count, or maxn + 1 when there are more.  One wave; cand / list in LDS.
__device__ inline int pair_distinct(const uint32_t *cand, uint32_t use, uint32_t *list, int maxn, int lane)
{
    uint32_t last = 0;
    bool first = true;
    int n = 0;
    for (;;) {
        uint32_t m = 0xffffffffu;
        for (int i = 0; i < 16; ++i) {
            if (!((use >> i) & 1u)) continue;
            const uint32_t v = cand[lane * 16 + i];
            if (v != 0xffffffffu && (first || v > last)) m = min(m, v);
        }
        for (int off = 32; off > 0; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off, 64));
        if (m == 0xffffffffu) break;
        if (n == maxn) return maxn + 1;
        if (lane == 0) list[n] = m;
        ++n;
        last = m; first = false;
    }
    return n;
}

// plan compiler: one wave per tile.  Reads the base entries (byte offset of the footprint, meta), brings them to store
// order (tiles compiled lane-interleaved are rewritten to pixel slot j of lane l = pixel 4 l + j), picks the mode, assigns every distinct group
// a slot (ascending address order: lane = slot % 64, round = slot / 64) and rewrites every entry to the LDS byte addresses
// of its two pair entries.
__global__ void __launch_bounds__(64) k_plan_pair_build(const uint2 *__restrict__ plan, uint32_t *__restrict__ hdr, int ntiles,
                                                         uint32_t row_bytes, uint32_t set_bytes, uint2 *__restrict__ plan_pr,
                                                         uint32_t *__restrict__ gsrc, int LX)
{
    constexpr int kMax = kPairMaxRounds * 64;   // groups per step, whole-tile staging
    constexpr int kMaxSlice = kPairSliceRounds * 64;
    __shared__ uint2 ent[8][64];
    __shared__ uint32_t cand[64 * 16];
    __shared__ uint32_t list[kMax + 1];
    const int tile = blockIdx.x, lane = threadIdx.x;
    if (tile >= ntiles) return;
    const uint32_t h = hdr[tile];
    if (h & (kHdrSlow | kHdrEmpty)) return;
    const uint32_t gpr = row_bytes / 12u;   // groups per source row
    const bool inter = (h & kHdrInterleaved) != 0;
    for (int k = 0; k < 8; ++k) {
        const int j = k & 3, sl = k & 4;
        const int dst_lane = inter ? (lane & ~3) + j : lane, dst_slot = inter ? sl + (lane & 3) : k;
        ent[dst_slot][dst_lane] = plan[((size_t)tile * 8 + k) * 64 + lane];
    }
    __syncthreads();
    uint2 e[8];
    bool overrun = false;
    for (int k = 0; k < 8; ++k) {
        e[k] = ent[k][lane];
        uint32_t c0 = 0xffffffffu, c1 = 0xffffffffu;
        if (e[k].y & kMetaValid) {
            c0 = e[k].x / 12u;       // group of texel pair (sx, sx+1) in row sy: offset = (row * fw + sx) * 3 = 12 * (row * gpr) + 3 * sx
            c1 = c0 + gpr;           // same columns, row sy + 1
            if ((size_t)c1 * 12u + 16u > (size_t)set_bytes) overrun = true;   // the 16-byte window of the last group would overrun
        }
        cand[lane * 16 + 2 * k] = c0;
        cand[lane * 16 + 2 * k + 1] = c1;
    }
    __syncthreads();
    if (__any(overrun)) return;
    auto slot_of = [&](uint32_t key, int count) {
        int lo = 0, hi = count;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (list[mid] < key) lo = mid + 1; else hi = mid; }
        return (uint32_t)lo;
    };
    auto lds_addr = [](uint32_t slot, uint32_t k) {
        return (slot >> 6) * (uint32_t)kPairRoundBytes + (k >> 1) * 1024u + (slot & 63u) * 16u + (k & 1u) * 8u;
    };
    auto rewrite = [&](int k, int count) {
        uint2 o = make_uint2(0u, e[k].y & ~kMetaValid);
        if (e[k].y & kMetaValid) {
            const uint32_t key = e[k].x / 12u, pk = (e[k].x - key * 12u) / 3u;
            o = make_uint2(lds_addr(slot_of(key, count), pk) | (lds_addr(slot_of(key + gpr, count), pk) << 16), e[k].y);
        }
        plan_pr[((size_t)tile * 8 + k) * 64 + lane] = o;
    };
    auto write_src = [&](int base, int count, int rounds) {
        for (int r = 0; r < rounds; ++r) {
            const int slot = r * 64 + lane;
            // lanes without a group carry an out-of-range offset: the buffer load returns zeros without a memory access
            gsrc[((size_t)tile * kPairSrcSlots + base + r) * 64 + lane] = slot < count ? list[slot] * 12u : kPairNoGroup;
        }
    };
    int mode;
    const int count = pair_distinct(cand, 0xffffu, list, kMax, lane);
    __syncthreads();
    if (count == 0) return;
    // two-contributor tiles (seams, blend overlaps) have pair classes for <= 256 groups (whole-tile staging in 1 / 2 / 4
    // rounds); the few sparser ones stay on the gather class
    if ((h & kHdrSecond) && count > kMax) return;
    if (count <= kMax) {
        mode = count <= 64 ? 0 : (count <= 128 ? 1 : 2);
        write_src(0, count, kPairMaxRounds);
        for (int k = 0; k < 8; ++k) rewrite(k, count);
    } else {
        // sliced: four slices of 64 pixels, each with its own group list (single-contributor tiles: <= 2 groups per pixel).
        // A slice is a COMPACT piece of the tile -- pixels 64 s .. 64 s + 63 in row-major order (two rows of a 32 x 8 tile),
        // lane l computing pixel 64 s + l -- so that the slices touch disjoint source rows; with "pixel slot j of every lane"
        // every slice re-requested most of the tile's lines (160 against 83 distinct lines per tile-frame on config 3).
        // The kernel brings the four pixels of a lane back to store order through LDS (plan_pair_body, SLICES == 4).
        {
            const int W = 4 * LX, LY = 64 / LX;
            const bool transposed = (h & kHdrTransposed) != 0;
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int pix = 64 * sidx + lane, x = pix % W, y = pix / W;
                const int src_lane = transposed ? (x >> 2) * LY + y : y * LX + (x >> 2);
                e[sidx] = ent[x & 3][src_lane];
                e[4 + sidx] = make_uint2(0u, 0u);
            }
            __syncthreads();
            for (int k = 0; k < 8; ++k) {
                uint32_t c0 = 0xffffffffu, c1 = 0xffffffffu;
                if (e[k].y & kMetaValid) { c0 = e[k].x / 12u; c1 = c0 + gpr; }
                cand[lane * 16 + 2 * k] = c0;
                cand[lane * 16 + 2 * k + 1] = c1;
            }
            __syncthreads();
        }
        int worst = 0;
        for (int j = 0; j < 4; ++j) {
            __syncthreads();
            const int cj = pair_distinct(cand, (3u << (2 * j)) | (3u << (2 * (4 + j))), list, kMaxSlice, lane);
            if (cj > kMaxSlice) return;   // cannot happen for single-contributor tiles (2 groups per pixel)
            worst = max(worst, cj);
        }
        for (int j = 0; j < 4; ++j) {
            __syncthreads();
            const int cj = pair_distinct(cand, (3u << (2 * j)) | (3u << (2 * (4 + j))), list, kMaxSlice, lane);
            __syncthreads();
            write_src(j * kPairSliceRounds, cj, kPairSliceRounds);
            rewrite(j, cj);
            rewrite(4 + j, cj);
        }
        (void)worst;
        mode = 3;
    }
    if (lane == 0) hdr[tile] = h | kHdrPaired | ((uint32_t)mode << 8);
}

typedef uint32_t pair_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t pair_u32x3 __attribute__((ext_vector_type(3)));
// 16 source bytes (texels 4g .. 4g+4 of one row) -> the four pair entries: A = pairs 0, 1, B = pairs 2, 3
__host__ __device__ __forceinline__ void pair_convert(uint32_t dx, uint32_t dy, uint32_t dz, uint32_t dw, uint4 &A, uint4 &B)
{
    A.x = px_perm(dy, dx, 0x04010300u);   // pair 0: b0 b1 g0 g1   (source bytes 0 3 1 4)
    A.y = px_perm(dy, dx, 0x0c0c0502u);   //         r0 r1 0 0     (2 5)
    A.z = px_perm(dy, dx, 0x07040603u);   // pair 1: bytes 3 6 4 7
    A.w = px_perm(dz, dy, 0x0c0c0401u);   //         5 8
    B.x = px_perm(dz, dy, 0x06030502u);   // pair 2: bytes 6 9 7 10
    B.y = px_perm(dz, dz, 0x0c0c0300u);   //         8 11
    B.z = px_perm(dw, dz, 0x05020401u);   // pair 3: bytes 9 12 10 13
    B.w = px_perm(dw, dz, 0x0c0c0603u);   //         11 14
}
// ... stored for lane `lane` of round patch `rp`
__device__ __forceinline__ void pair_convert_store(const pair_u32x4 &d, uint8_t *rp, int lane)
{
    uint4 A, B;
    pair_convert(d.x, d.y, d.z, d.w, A, B);
    reinterpret_cast<uint4 *>(rp)[lane] = A;
    reinterpret_cast<uint4 *>(rp + 1024)[lane] = B;
}

// one pixel from its two pair entries: accumulators with the result byte in bits 16..23 (as bilinear_rows_b2)
__host__ __device__ __forceinline__ void bilinear_pairs(uint2 q0, uint2 q1, uint32_t wxa, uint32_t wxb, uint32_t wy64, uint32_t acc[3])
{
    const uint32_t hb0 = px_dot4(q0.x, wxa, 0u), hb1 = px_dot4(q1.x, wxa, 0u);
    const uint32_t hg0 = px_dot4(q0.x, wxb, 0u), hg1 = px_dot4(q1.x, wxb, 0u);
    const uint32_t hr0 = px_dot4(q0.y, wxa, 0u), hr1 = px_dot4(q1.y, wxa, 0u);
    acc[0] = px_dot2(hb0 | (hb1 << 16), wy64, 32768u);
    acc[1] = px_dot2(hg0 | (hg1 << 16), wy64, 32768u);
    acc[2] = px_dot2(hr0 | (hr1 << 16), wy64, 32768u);
}

// 12 accumulators (result byte in bits 16..23) of a lane's 4 pixels -> the 12 output bytes B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
__host__ __device__ __forceinline__ void pack_accs(const uint32_t acc[4][3], uint32_t &d0, uint32_t &d1, uint32_t &d2)
{
    // perm(hi, lo, sel): byte 2 of lo = index 2, byte 2 of hi = index 6; the two halves of each output dword have zeros
    // where the other half has data: combine with v_or_b32 (a 2.5-clk VOP2)
    d0 = px_perm(acc[1][0], acc[0][2], 0x06020c0cu) | px_perm(acc[0][1], acc[0][0], 0x0c0c0602u);
    d1 = px_perm(acc[2][1], acc[2][0], 0x06020c0cu) | px_perm(acc[1][2], acc[1][1], 0x0c0c0602u);
    d2 = px_perm(acc[3][2], acc[3][1], 0x06020c0cu) | px_perm(acc[3][0], acc[2][2], 0x0c0c0602u);
}

// one wave: tile from the class list, frames of the chunk.  lds: the block's patches, 4 x (rounds per step x 2 KB).
// SLICES 1: whole-tile staging, 4: one slice per pixel slot (NSLOT == 1 only); ROUNDS: group rounds per step.
template <int LX, int NSLOT, bool BLEND, bool SUMS, int SLICES, int ROUNDS>
__device__ __forceinline__ void plan_pair_body(const PlanArgs &a, uint32_t block_id, uint8_t *lds)
{
    uint32_t chunk, group;
    if (!plan_block_map(a, block_id, chunk, group)) return;
    const int slot = (int)group * 4 + (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (slot >= a.nlist) return;
    const int tile = (int)__builtin_amdgcn_readfirstlane(a.tile_list[slot]);
    const uint32_t hdr = __builtin_amdgcn_readfirstlane(a.hdr[tile]);
    static_assert(SLICES == 1 || (SLICES == 4 && NSLOT == 1), "sliced staging is built for single-contributor tiles");
    static_assert(ROUNDS >= 1 && ROUNDS <= (SLICES == 1 ? kPairMaxRounds : kPairSliceRounds), "rounds per step");
    constexpr int LY = 64 / LX;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    int lx_, ly_;
    lane_xy(lane, LX, (hdr & kHdrTransposed) != 0, lx_, ly_);
    const int x0 = (tx * LX + lx_) * 4, y = ty * LY + ly_;
    const bool inimg = x0 < a.bw && y < a.bh;
    const size_t set_bytes = (size_t)a.fw * a.fh * 3 * a.ncams, img_bytes = (size_t)a.pitch * a.bh * 3;
    const uint32_t ooff = ((uint32_t)y * a.pitch + x0) * 3;
    const uint32_t ooff_masked = inimg ? ooff : kPairNoGroup;   // out of range of the image's buffer descriptor: not written
    // LDS this class needs per wave: the staged rounds, and 1 KB for the slice -> store order exchange of the sliced class
    constexpr int kWavePatch = SLICES == 1 ? ROUNDS * kPairRoundBytes : kPairSliceRounds * kPairRoundBytes + 1024;
    uint8_t *const patch = lds + wave * kWavePatch;
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
    uint32_t gs[SLICES][ROUNDS];
#pragma unroll
    for (int s = 0; s < SLICES; ++s)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) gs[s][r] = a.gsrc[((size_t)tile * kPairSrcSlots + s * kPairSliceRounds + r) * 64 + lane];
    uint32_t car0 = 0, car1 = 0, car2 = 0;
    if (!SUMS && a.car != nullptr && inimg) {
        const uint32_t *cp = reinterpret_cast<const uint32_t *>(a.car + ooff);
        car0 = cp[0]; car1 = cp[1]; car2 = cp[2];
    }
    const bool car_any = __builtin_amdgcn_ballot_w64((car0 | car1 | car2) != 0) != 0;

    const int b_begin = (int)chunk * a.nb, b_end = min(a.batch, b_begin + a.nb);

    // D steps (frames, or slices of a frame) have their groups in flight in registers ahead of the one being interpolated
    // (4 and 8 measured no faster: profiles/r02/sweeps.log)
    constexpr int D = 2;
    pair_u32x4 pf[D][ROUNDS];
    // Loads and stores go through raw buffer descriptors (wave-uniform base in SGPRs + 32-bit lane offset: no 64-bit address
    // arithmetic, and a lane whose offset is out of range -- kPairNoGroup, or a pixel quad right of the image -- costs no
    // memory access and no branch, so every vector-memory instruction is issued unconditionally and counted exactly).
    auto issue = [&](int b, int s, int ring) {   // s, ring: compile-time after unrolling
        const uint8_t *src = a.frames + (size_t)min(b, b_end - 1) * set_bytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, (uint32_t)set_bytes, kBufferWord3);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) pf[ring][r] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)gs[s][r], 0, kPairLoadAux);
    };
    auto land = [&](int ring) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) pair_convert_store(pf[ring][r], patch + r * kPairRoundBytes, lane);
    };
    // pixel j of contributor s from the patch -> accumulators
    auto pixel = [&](int s, int j, uint32_t acc[3]) { bilinear_pairs(pw[i0[s][j]], pw[i1[s][j]], wxa[s][j], wxb[s][j], wy[s][j], acc); };
    // generic path: per-tile channel sums (balance) and the car sprite on the 4 pixel dwords (B | G << 8 | R << 16)
    auto finish = [&](int b, uint32_t P[4]) {
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
                uint32_t *ps = a.psums + ((size_t)b * a.ntiles + tile) * 3;
                ps[0] = bg & 0xffffu; ps[1] = bg >> 16; ps[2] = sr;
            }
        }
        if (car_any) add_car(P, car0, car1, car2);
    };
    auto store = [&](int b, uint32_t d0, uint32_t d1, uint32_t d2) {
        // a frame index past the end of the chunk re-writes the last frame with the same bytes
        uint8_t *img = a.out + (size_t)min(b, b_end - 1) * img_bytes;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(img, 0, (uint32_t)img_bytes, kBufferWord3);
        __builtin_amdgcn_raw_buffer_store_b96(pair_u32x3{d0, d1, d2}, ro, (int)ooff_masked, 0, kPairStoreAux);
    };
    // contribution of entry (s, j) accumulated onto px (saturating add of the second contributor)
    auto contrib = [&](int s, int j, int px[3]) {
        uint32_t acc[3];
        pixel(s, j, acc);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t v = (acc[k] >> 16) & 255u;
            const int c = BLEND ? (int)((float)v * wf[s][j]) : (int)v;
            px[k] = s == 0 ? c : min(255, px[k] + c);
        }
    };
    constexpr bool kFast = !BLEND && NSLOT == 1 && !SUMS;   // accumulators go straight to the output bytes
    auto acc_to_px = [](const uint32_t acc[3]) {
        return __builtin_amdgcn_perm(acc[2], __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u), 0x0c060100u);
    };

    if (SLICES == 1) {
        // step = frame; two frames per loop trip (ring slot = frame parity)
        auto frame = [&](int b, int ring) {
            issue(b + D, 0, ring);            // ring slot of frame b is free: its groups were converted one step ago
            uint32_t d0, d1, d2;
            if (kFast) {
                uint32_t acc[4][3];
#pragma unroll
                for (int j = 0; j < 4; ++j) pixel(0, j, acc[j]);
                if (car_any) {
                    uint32_t P[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) P[j] = acc_to_px(acc[j]);
                    add_car(P, car0, car1, car2);
                    pack_pixels(P, d0, d1, d2);
                } else {
                    pack_accs(acc, d0, d1, d2);
                }
            } else {
                uint32_t P[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int px[3];
#pragma unroll
                    for (int s = 0; s < NSLOT; ++s) contrib(s, j, px);
                    P[j] = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
                }
                finish(b, P);
                pack_pixels(P, d0, d1, d2);
            }
            land((ring + 1) % D);             // frame b+1 (issued D-1 steps ago); every LDS read of frame b is older
            store(b, d0, d1, d2);
        };
#pragma unroll
        for (int u = 0; u < D; ++u) issue(b_begin + u, 0, u);
        land(0);
#pragma unroll 1
        for (int b = b_begin; b < b_end; b += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) frame(b + u, u);
        }
    } else {
        // step = (frame, slice j): lane l computes pixel 64 j + l of the tile (row-major); ring slot = slice parity
        uint32_t *const xch = reinterpret_cast<uint32_t *>(patch + kPairSliceRounds * kPairRoundBytes);
        const int xrd = ly_ * (4 * LX) + 4 * lx_;   // the lane's 4 stored pixels in row-major pixel order
        issue(b_begin, 0, 0);
        issue(b_begin, 1, 1);
        land(0);
#pragma unroll 1
        for (int b = b_begin; b < b_end; ++b) {
            uint32_t P[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                issue(j < 2 ? b : b + 1, (j + 2) & 3, j & 1);
                if (kFast) {
                    uint32_t acc[3];
                    pixel(0, j, acc);
                    P[j] = acc_to_px(acc);
                } else {
                    int px[3];
                    contrib(0, j, px);
                    P[j] = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
                }
                xch[64 * j + lane] = P[j];
                land((j & 1) ^ 1);
            }
            const uint4 q = *reinterpret_cast<const uint4 *>(xch + xrd);   // same wave: LDS operations complete in order
            P[0] = q.x; P[1] = q.y; P[2] = q.z; P[3] = q.w;
            uint32_t d0, d1, d2;
            if (kFast) {
                if (car_any) add_car(P, car0, car1, car2);
            } else {
                finish(b, P);
            }
            pack_pixels(P, d0, d1, d2);
            store(b, d0, d1, d2);
        }
    }
}

// the pair-staged classes as kernels of their own (per-class launches: BEVW_PLAN_ONELAUNCH=0, and the unit profiles are
// taken on)
template <int LX, int NSLOT, bool BLEND, bool SUMS, int SLICES, int ROUNDS>
__global__ void __launch_bounds__(256) k_plan_pair(PlanArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t patch[4 * (SLICES == 1 ? ROUNDS * kPairRoundBytes : kPairSliceRounds * kPairRoundBytes + 1024)];
    plan_pair_body<LX, NSLOT, BLEND, SUMS, SLICES, ROUNDS>(a, blockIdx.x, patch);
}

}  // namespace bevw
