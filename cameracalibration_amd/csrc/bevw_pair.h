// bevw_pair.h -- texel PAIRS: the staging format of the unit schedule (bevw_unit.h).  Included by bevw_plan.h.
//
// Source texels are fetched in GROUPS: 16 bytes from the 4-byte aligned address 12 * g (g = group index inside the frame set;
// rows are whole numbers of groups because fw % 4 == 0), i.e. texels 4g .. 4g+4 of one source row, one buffer_load_dwordx4 per
// lane.  Each lane turns its 16 bytes ONCE into the four horizontal texel pairs (4g+k, 4g+k+1), k = 0..3, laid out for the dot
// products: 8 bytes { b0 b1 g0 g1 | r0 r1 0 0 } (8 v_perm_b32, two ds_write_b128).  A BEV pixel then reads one 8-byte pair
// entry per footprint row (ds_read_b64, 8-byte aligned by construction) and needs 6 v_dot4 + 3 v_lshl_or + 3 v_dot2 + 2 v_perm:
// the realignment of the interleaved BGR bytes is paid once per source texel instead of once per BEV pixel and tap.
// The arithmetic is cv2.remap's (sum p * w + 512) >> 10 in the separable form, exact in integers (surroundBEV.py:116-117).
//
// (Rounds 2 - 3 also ran per-wave and per-block schedules on this format -- "pair classes", "block tiles", "seam tiles"; round 4
// retired them: every staged pixel is a unit pixel now.  Their measurements stay in profiles/r02/, profiles/r03/.)
#pragma once

namespace bevw {

constexpr uint32_t kPairNoGroup = 0x80000000u;   // source offset of a lane without a group (frame sets are < 2 GB): out of the buffer's range
constexpr uint32_t kBufferWord3 = 0x00020000u;   // raw buffer descriptor, dword 3 (gfx9 family: DATA_FORMAT 32)
// cache-policy bits of the group loads / pixel stores (gfx94x/95x: 1 = sc0, 2 = nt, 16 = sc1).  Round 2's kernels were slower with every
// non-default policy (profiles/r02/sweeps.log, "cache policy"); the unit kernel is not: its output is written once and never read, and as
// streaming stores (nt | sc0) it leaves the L2 to the texel groups and the plan -- config 3 0.439 -> 0.410 ms (nt alone 0.419, nt | sc1
// 0.410; nt LOADS 0.519: the groups ARE re-read, by the neighbouring units) -- where the rows are whole sectors; the dense layout loses
// (0.46 -> 0.51) and keeps the default.  profiles/r04/ab_store_policy.log, run16_store_policy.log
#ifndef BEVW_LOAD_AUX
#define BEVW_LOAD_AUX 0
#endif
#ifndef BEVW_STORE_AUX
#define BEVW_STORE_AUX 0
#endif
#ifndef BEVW_STREAM_AUX
#define BEVW_STREAM_AUX 3
#endif
constexpr int kPairLoadAux = BEVW_LOAD_AUX, kPairStoreAux = BEVW_STORE_AUX, kPairStreamAux = BEVW_STREAM_AUX;   // (stream: output rows of whole sectors, bevw_unit.h unit_store_quad)

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

// one pixel from its two pair entries: accumulators with the result byte in bits 16..23.  The y weights come pre-scaled by 64:
// (S * 64 + 512 * 64) >> 16 == (S + 512) >> 10, so the 12 output bytes of a lane are assembled with v_perm_b32 instead of shifts.
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

// LDS writes of this wave done, then the block's barrier (vector-memory operations stay in flight across it)
__device__ __forceinline__ void block_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace bevw
