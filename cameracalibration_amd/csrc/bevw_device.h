// bevw_device.h -- device-side arithmetic shared by every kernel of libbevwarp (gfx950 only).
//
// Everything here restates, for one output element, what the reference obtains from cv2 at the call sites cited
// per function (paths are relative to the reference tree).  The translation unit is compiled with
// -ffp-contract=off: OpenCV's x86-64 baseline code has no fused multiply-add, so neither may we.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>

namespace bevw {

// ---- cache policy of data that passes exactly once -------------------------------------------------------------------------
// Round 4's last measurement (profiles/r04/README.md section 17): the unit kernel's output as streaming stores took 5 - 7 % off config 3
// and 20 % off the undistort -- written once, never read, it only displaced the texel groups from the L2.  The same holds for other
// once-through streams; these switches put the `nt` policy on them so that one A/B each (BEVW_CFLAGS=-DBEVW_..._NT=1,
// tools/ab_bench.py with BEVW_LIB_PATH) can tell.  With 0 the code below is exactly the plain load / store.
// Round 5's A/B (profiles/r05/ab_call1_policies_and_maps.log, config 4 at 1.923 ms): GAIN_NT 1.895, VSUM_NT 1.880, both 1.844 -> both ON;
// PLAN_NT: config 3 0.430 -> 0.433, undistort 0.079 -> 0.104 -> stays OFF (the plan slice of a unit IS re-read, by the block of the next chunk).
//   BEVW_GAIN_NT   k_gain_lut: the pre-gain BEV batch (read once) and the output (written once)      -- config 4, 0.35 of 2.0 ms
//   BEVW_VSUM_NT   k_vsum: the raw frames of the luminance statistics (3.8 GB read per config-4 step)  -- config 4, 0.65 of 2.0 ms
//   BEVW_PLAN_NT   k_plan_units: the unit's plan entries and group offsets (read once per block and 16 frames: 110 MB per config-3 step)
//   BEVW_COEF_NT   JPEG: the coefficient blocks (k_jpeg_coef writes them once, the inverse DCT kernels read them once: 2 x 472 MB per slice)
#ifndef BEVW_GAIN_NT
#define BEVW_GAIN_NT 1
#endif
#ifndef BEVW_VSUM_NT
#define BEVW_VSUM_NT 1
#endif
#ifndef BEVW_PLAN_NT
#define BEVW_PLAN_NT 0
#endif
#ifndef BEVW_COEF_NT
#define BEVW_COEF_NT 0
#endif
template <int NT, typename T> __host__ __device__ __forceinline__ T once_load(const T *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (NT) return __builtin_nontemporal_load(p);
#endif
    return *p;
}
template <int NT, typename T> __host__ __device__ __forceinline__ void once_store(T *p, T v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (NT) { __builtin_nontemporal_store(v, p); return; }
#endif
    *p = v;
}

// ---- byte-permute / dot-product instructions of the staged stitch kernels, callable from host code as well ----------
// On the device these are the gfx950 instructions themselves (v_perm_b32, v_alignbyte_b32, v_dot4_u32_u8, v_dot2_u32_u16).
// The host versions restate the instructions bit for bit; they exist ONLY so that tests/native/unit_emulate.cpp can run
// the kernels' per-lane arithmetic and the plan compiler's output on a CPU (no product path ever calls them on the host).
__host__ __device__ __forceinline__ uint32_t px_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const uint64_t v = ((uint64_t)hi << 32) | lo;   // byte s of {hi, lo}: 0..3 = lo, 4..7 = hi; 0x0c = 0x00, >= 0x0d = 0xff
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t s = (sel >> (8 * i)) & 255u;
        uint32_t b;
        if (s < 8) b = (uint32_t)(v >> (8 * s)) & 255u;
        else if (s < 12) b = ((v >> (16 * (s - 8) + 15)) & 1u) ? 255u : 0u;   // sign of bytes 1, 3, 5, 7
        else b = s == 12 ? 0u : 255u;
        r |= b << (8 * i);
    }
    return r;
#endif
}
__host__ __device__ __forceinline__ uint32_t px_alignbyte(uint32_t hi, uint32_t lo, uint32_t n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, n);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (n & 3u)));
#endif
}
__host__ __device__ __forceinline__ uint32_t px_dot4(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    for (int i = 0; i < 4; ++i) c += ((a >> (8 * i)) & 255u) * ((b >> (8 * i)) & 255u);
    return c;
#endif
}
__host__ __device__ __forceinline__ uint32_t px_dot2(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    union { uint32_t u; us2 v; } x, y;
    x.u = a; y.u = b;
    return __builtin_amdgcn_udot2(x.v, y.v, c, false);
#else
    return c + (a & 0xffffu) * (b & 0xffffu) + (a >> 16) * (b >> 16);
#endif
}

constexpr int kQBits = 5;          // INTER_BITS
constexpr int kQOne = 32;          // INTER_TAB_SIZE
constexpr int kQTab2 = 1024;       // INTER_TAB_SIZE2

// ---- rounding / saturation (cvRound = round-half-even, saturate_cast<>) -----------------------------------
__device__ __forceinline__ int rne_d(double v)
{
    // cvtsd2si semantics: out-of-range and NaN give INT_MIN
    if (!(v > -2147483648.5 && v < 2147483647.5)) return INT_MIN;
    return __double2int_rn(v);
}
__device__ __forceinline__ int rne_f(float v)
{
    if (!(v > -2147483904.0f && v < 2147483520.0f)) return INT_MIN;
    return __float2int_rn(v);
}
__device__ __forceinline__ int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int sat_s16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
__device__ __forceinline__ int sat_u16(int v) { return v < 0 ? 0 : (v > 65535 ? 65535 : v); }

// ---- cvtColor BGR<->HSV 8-bit + V shift: luminance_balance (SurroundBirdEyeView/surroundBEV.py:57-79) ------
// Tables of the round trip, built once on the host (make_hsv_tables, csrc/bevwarp.hip), copied into LDS by every kernel that shifts texels.
struct alignas(16) HsvTables {   // (16: hsv_tables_to_lds copies it with 16-byte LDS stores)
    int sdiv[256];    // cvRound((255 << 12) / (1.0 * i))                                          (BGR2HSV, 8 bit)
    int hdiv[256];    // cvRound((180 << 12) / (6.0 * i))
    // HSV2BGR's hue arithmetic, per 8-bit H value, indexed by the LOW BYTE of the signed hue quotient q = (num * hdiv[diff] + 2048) >> 12
    // (q in [-30, 150]; H = q < 0 ? q + 180 : q, so bytes 226 .. 255 hold H = 150 .. 179):
    //   .x = bits of the float32 factor g of the middle candidate  v * (1 - s * g):  g = f in odd sectors, 1.f - f in even sectors,
    //        with h = float(H) * (6.f / 180.f), sector = floor(h), f = h - sector -- the floats OpenCV's HSV2RGB_f computes;
    //   .y = v_perm_b32 selector that deals the candidates {byte 0: v, byte 1: v (1 - s), byte 2: v (1 - s g)} to (B, G, R): the sector table
    //        {{1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0}} with candidates 2 / 3 (only one of them is ever selected) as "middle".
    uint2 hue[256];
};
static_assert(sizeof(HsvTables) == 4096 && alignof(HsvTables) >= 16, "HsvTables is copied as 256 x 16 bytes");

// host side: the table entry of hue byte i (see HsvTables::hue); plain float32 arithmetic, no contraction
static inline void hsv_hue_entry(int i, uint32_t &gbits, uint32_t &sel)
{
    const int H = i < 151 ? i : (i >= 226 ? i - 256 + 180 : 0);   // bytes 151 .. 225 are never produced
    const float hscale = 6.f / 180.f;
    float h = (float)H * hscale;
    if (h >= 6.f) h -= 6.f;                                          // fmod(h, 6): never taken for H < 180
    int sector = (int)floorf(h);
    h -= (float)sector;
    if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
    const float g = (sector & 1) ? h : 1.f - h;
    memcpy(&gbits, &g, 4);
    static const uint32_t kSel[6] = {0x0c000201u, 0x0c020001u, 0x0c010002u, 0x0c010200u, 0x0c020100u, 0x0c000102u};
    sel = kSel[sector];
}

// the block's copy of the tables: one 16-byte piece per thread and trip (256 pieces), then the caller synchronises
__device__ __forceinline__ void hsv_tables_to_lds(HsvTables &lds, const HsvTables *__restrict__ tab)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(tab);
    uint4 *dst = reinterpret_cast<uint4 *>(&lds);
    for (int i = threadIdx.x; i < (int)(sizeof(HsvTables) / 16); i += blockDim.x) dst[i] = src[i];
}

// One texel through BGR2HSV -> V = sat_u8(V + delta) -> HSV2BGR.  The round trip is lossy, so it is applied even
// when delta == 0, exactly as the reference does.
// Packed form: texel in, texel out as B | G << 8 | R << 16 (byte 3 of the input is ignored, of the result 0).
// Round 5: ~45 VALU instructions per texel instead of ~85 (k_lum_groups was VALU-bound).  What went:
//   * H is never materialised: the low byte of the signed quotient indexes HsvTables::hue, which holds everything HSV2BGR derives
//     from H (float conversion, * 6/180, floor, fraction, the sector's candidate permutation);
//   * three candidates instead of four: a sector selects t2 = v (1 - s f) or t3 = v (1 - s (1 - f)), never both;
//   * the maximum channel comes back as V itself: cvRound(float(V) * (1/255.f) * 255.f) == V for every V in 0 .. 255.
// Exhaustive check against the oracle (all 2^24 colours x deltas): tests/native/hsv_exhaustive.cpp, tests/test_hsv_exhaustive.py.
__host__ __device__ __forceinline__ uint32_t luminance_shift_bgr(uint32_t texel, int delta, const HsvTables &T)
{
    const int b = (int)(texel & 255u), g = (int)((texel >> 8) & 255u), r = (int)((texel >> 16) & 255u);
    const int mx = b > g ? b : g, mn = b < g ? b : g;
    const int v = mx > r ? mx : r, vmin = mn < r ? mn : r;
    const int diff = v - vmin;
    // 24-bit multiplies (v_mad_u32_u24 / v_mad_i32_i24; a plain 32-bit product compiles to the slow v_mad_u64_u32): diff <= 255, sdiv < 2^20,
    // |num| <= 1275, hdiv < 2^17
#if defined(__HIP_DEVICE_COMPILE__)
#define BEVW_UMUL24(a, b) ((int)__umul24((unsigned)(a), (unsigned)(b)))
#define BEVW_MUL24(a, b) __mul24((a), (b))
#else
#define BEVW_UMUL24(a, b) ((a) * (b))
#define BEVW_MUL24(a, b) ((a) * (b))
#endif
    const int s = (BEVW_UMUL24(diff, T.sdiv[v]) + (1 << 11)) >> 12;   // 0 .. 255
    // hue numerator: the channel that holds the maximum picks the formula (R first, then G, as OpenCV's masks do).  Masks and bit-selects
    // (v_bfi_b32) instead of compare + v_cndmask_b32: v >= r, g, so (v - r - 1) >> 31 is all ones exactly when v == r
    const uint32_t mr = (uint32_t)((v - r - 1) >> 31), mg = (uint32_t)((v - g - 1) >> 31);
    const uint32_t c0 = (uint32_t)(g - b), c1 = (uint32_t)(b - r + 2 * diff), c2 = (uint32_t)(r - g + 4 * diff);
    const uint32_t c12 = (c1 & mg) | (c2 & ~mg);
    const int num = (int)((c0 & mr) | (c12 & ~mr));
    const uint32_t hi = ((uint32_t)((BEVW_MUL24(num, T.hdiv[diff]) + (1 << 11)) >> 12)) & 255u;
#undef BEVW_UMUL24
#undef BEVW_MUL24
    const uint2 hq = T.hue[hi];
    int v2 = v + delta;
    v2 = v2 < 0 ? 0 : (v2 > 255 ? 255 : v2);
    // HSV -> BGR, float path (cvtColor HSV2BGR on 8U: S / 255, V / 255; candidates * 255 rounded)
    const float fs = (float)s * (1.f / 255.f), fv = (float)v2 * (1.f / 255.f);
    float gf;
#if defined(__HIP_DEVICE_COMPILE__)
    gf = __uint_as_float(hq.x);
#else
    memcpy(&gf, &hq.x, 4);
#endif
    const float t1 = fv * (1.f - fs);
    const float tm = fv * (1.f - fs * gf);
    // cvRound(t * 255) for t in [0, 1]: adding 1.5 * 2^23 rounds to nearest-even at unit precision and leaves the integer
    // (0..255, no saturation possible) in the low mantissa byte
    const float kMagic = 12582912.f;
    const float r1 = t1 * 255.f + kMagic, rm = tm * 255.f + kMagic;
    uint32_t u1, um;
#if defined(__HIP_DEVICE_COMPILE__)
    u1 = __float_as_uint(r1); um = __float_as_uint(rm);
#else
    memcpy(&u1, &r1, 4); memcpy(&um, &rm, 4);
#endif
    // candidates: byte 0 = V', byte 1 = low byte of u1, byte 2 = low byte of um
    const uint32_t C = px_perm(um, px_perm(u1, (uint32_t)v2, 0x0c0c0400u), 0x0c040100u);
    return px_perm(0u, C, hq.y);
}

__host__ __device__ __forceinline__ void luminance_shift_px(int &b, int &g, int &r, int delta, const HsvTables &T)
{
    const uint32_t o = luminance_shift_bgr((uint32_t)b | ((uint32_t)g << 8) | ((uint32_t)r << 16), delta, T);
    b = (int)(o & 255u);
    g = (int)((o >> 8) & 255u);
    r = (int)((o >> 16) & 255u);
}

// Linear block id of a 1-D grid -> (frame, block inside the frame) such that XCD id % 8 owns WHOLE frames: the rows a kernel
// writes are then completed inside one L2 (tools/store_pattern.hip: 4.7 TB/s with such a map, 2.9 TB/s when the blocks of a
// frame are dealt round-robin to the XCDs).  Grid = blocks_per_frame * 8 * ceil(nframes / 8) blocks.
__device__ __forceinline__ bool xcd_frame_map(uint32_t id, uint32_t blocks_per_frame, uint32_t nframes, uint32_t &frame, uint32_t &blk)
{
    const uint32_t xcd = id & 7u, k = id >> 3;
    frame = xcd + 8u * (k / blocks_per_frame);
    blk = k % blocks_per_frame;
    return frame < nframes;
}
static inline unsigned xcd_frame_grid(unsigned blocks_per_frame, unsigned nframes) { return blocks_per_frame * 8u * ((nframes + 7u) / 8u); }

// ---- cv2.remap u8c3, INTER_LINEAR fixed point, BORDER_CONSTANT 0 -------------------------------------------
// call sites: surroundBEV.py:110-111,116-117; intrinsicCalib.py:193-195; Tools/undistort.py:66
// out = (sum p * w15 + 2^14) >> 15 with w15 = 32 * w10  ==  (sum p * w10 + 512) >> 10, w10 = 5-bit x 5-bit products.
// ties_even (BEVW_COMPAT_REMAP 1): the sum S of the four weighted taps is exact in either arithmetic; the classic fixed-point kernels
// round it as (S + 512) >> 10 = S / 1024 half UP, a float32 kernel that ends in cvRound (OpenCV >= 4.11's linear kernels, when they take the
// fixed-point maps) rounds the same exact value half to EVEN -- they differ only where S = 512 (mod 1024) and the rounded-up result is odd.
__device__ __forceinline__ int remap_round10(int S, int ties_even)
{
    int r = (S + 512) >> 10;
    if (ties_even && (S & 1023) == 512 && (r & 1)) --r;
    return r;
}
template <bool LUM>
__device__ __forceinline__ void remap_u8c3_px(const uint8_t *__restrict__ src, int sw, int sh, int sx, int sy,
                                              unsigned code, int out[3], int delta, const HsvTables *hsv, int ties_even = 0)
{
    const int fx = code & 31, fy = (code >> 5) & 31;
    const int ax = kQOne - fx, ay = kQOne - fy;
    const int w00 = ax * ay, w01 = fx * ay, w10 = ax * fy, w11 = fx * fy;
    const unsigned xlim = sw > 1 ? sw - 1 : 0, ylim = sh > 1 ? sh - 1 : 0;
    if ((unsigned)sx < xlim && (unsigned)sy < ylim) {
        const uint8_t *p0 = src + ((size_t)sy * sw + sx) * 3;
        const uint8_t *p1 = p0 + (size_t)sw * 3;
        int t[4][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { t[0][k] = p0[k]; t[1][k] = p0[3 + k]; t[2][k] = p1[k]; t[3][k] = p1[3 + k]; }
        if (LUM) {
#pragma unroll
            for (int q = 0; q < 4; ++q) luminance_shift_px(t[q][0], t[q][1], t[q][2], delta, *hsv);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            out[k] = remap_round10(t[0][k] * w00 + t[1][k] * w01 + t[2][k] * w10 + t[3][k] * w11, ties_even);
    } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
        out[0] = out[1] = out[2] = 0;
    } else {
        const bool x0 = sx >= 0 && sx < sw, x1 = sx + 1 >= 0 && sx + 1 < sw;
        const bool y0 = sy >= 0 && sy < sh, y1 = sy + 1 >= 0 && sy + 1 < sh;
        int t[4][3];
        const bool ok[4] = {x0 && y0, x1 && y0, x0 && y1, x1 && y1};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (ok[q]) {
                const uint8_t *p = src + ((size_t)(sy + (q >> 1)) * sw + (sx + (q & 1))) * 3;
                t[q][0] = p[0]; t[q][1] = p[1]; t[q][2] = p[2];
                if (LUM) luminance_shift_px(t[q][0], t[q][1], t[q][2], delta, *hsv);
            } else {
                t[q][0] = t[q][1] = t[q][2] = 0;  // border value enters after the balance step
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            out[k] = remap_round10(t[0][k] * w00 + t[1][k] * w01 + t[2][k] * w10 + t[3][k] * w11, ties_even);
    }
}

// ---- cv2.warpPerspective coordinate generation (surroundBEV.py:113-114, extrinsicCalib.py:166-169) ---------
// M is the INVERSE homography; the destination is walked in column blocks of bw0 pixels whose origin x0 enters the
// fp64 expression, so the association (X0 + M0*x1) is kept.
__device__ __forceinline__ void perspective_coord(const double *__restrict__ M, int x, int y, int bw0, int &sx,
                                                  int &sy, unsigned &code)
{
    const int x0 = (x / bw0) * bw0, x1 = x - x0;
    const double X0 = M[0] * x0 + M[1] * y + M[2];
    const double Y0 = M[3] * x0 + M[4] * y + M[5];
    const double W0 = M[6] * x0 + M[7] * y + M[8];
    double W = W0 + M[6] * x1;
    W = (W != 0.0) ? (double)kQOne / W : 0.0;
    double fX = (X0 + M[0] * x1) * W, fY = (Y0 + M[3] * x1) * W;
    const double lo = (double)INT_MIN, hi = (double)INT_MAX;
    // std::max(lo, std::min(hi, v)) with the operand order of the reference implementation (NaN -> hi)
    fX = (fX < hi) ? fX : hi; fX = (lo < fX) ? fX : lo;
    fY = (fY < hi) ? fY : hi; fY = (lo < fY) ? fY : lo;
    const int X = rne_d(fX), Y = rne_d(fY);
    sx = sat_s16(X >> kQBits);
    sy = sat_s16(Y >> kQBits);
    code = (unsigned)((Y & (kQOne - 1)) * kQOne + (X & (kQOne - 1)));
}

// ---- float-weight bilinear of 16-bit sources: the LUT quirk, Camera.get_bev_maps (surroundBEV.py:105-108) ---
template <typename T, int CN>
__device__ __forceinline__ void remap_f32_px(const T *__restrict__ src, int sw, int sh, int sx, int sy, unsigned code,
                                             int out[CN])
{
    const float fx = (float)(code & 31) * (1.f / 32), fy = (float)((code >> 5) & 31) * (1.f / 32);
    const float ax = 1.f - fx, ay = 1.f - fy;
    const float w0 = ay * ax, w1 = ay * fx, w2 = fy * ax, w3 = fy * fx;
    const unsigned xlim = sw > 1 ? sw - 1 : 0, ylim = sh > 1 ? sh - 1 : 0;
    if ((unsigned)sx < xlim && (unsigned)sy < ylim) {
        const T *S = src + ((size_t)sy * sw + sx) * CN;
        const size_t st = (size_t)sw * CN;
#pragma unroll
        for (int k = 0; k < CN; ++k)
            out[k] = rne_f((float)S[k] * w0 + (float)S[k + CN] * w1 + (float)S[k + st] * w2 + (float)S[k + st + CN] * w3);
    } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
#pragma unroll
        for (int k = 0; k < CN; ++k) out[k] = 0;
    } else {
        const bool x0 = sx >= 0 && sx < sw, x1 = sx + 1 >= 0 && sx + 1 < sw;
        const bool y0 = sy >= 0 && sy < sh, y1 = sy + 1 >= 0 && sy + 1 < sh;
#pragma unroll
        for (int k = 0; k < CN; ++k) {
            float v0 = (x0 && y0) ? (float)src[((size_t)sy * sw + sx) * CN + k] : 0.f;
            float v1 = (x1 && y0) ? (float)src[((size_t)sy * sw + sx + 1) * CN + k] : 0.f;
            float v2 = (x0 && y1) ? (float)src[((size_t)(sy + 1) * sw + sx) * CN + k] : 0.f;
            float v3 = (x1 && y1) ? (float)src[((size_t)(sy + 1) * sw + sx + 1) * CN + k] : 0.f;
            out[k] = rne_f(v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3);
        }
    }
}

// ---- OpenCV >= 4.11: float32 linear kernels of warpPerspective (8U / 16U / 32F, C1 / C3 / C4) -- a FAMILY of candidate restatements ----
// The classic kernels above (OpenCV 2.4 ... 4.10) quantise the source position to 1/32 pixel and interpolate in fixed point (8U) or with
// tabulated float weights (16S / 16U).  OpenCV 4.11 added kernels that keep the position in float32: (sx, sy) = (M x) / w evaluated in
// float, ix = floor(sx), alpha = sx - ix, out = cvRound(lerp(lerp(p00, p01, alpha), lerp(p10, p11, alpha), beta)), border taps replaced by
// the border value one by one.  Their exact operation order is not reproducible without the library at hand (and differs between its
// own SIMD body -- fused multiply-adds on AVX2 / NEON builds -- and scalar tail), so BEVW_COMPAT_WARP selects a MEMBER of a family spanned
// by the choices below; the implementation probes of tests/golden/ (stored whole) decide which member, if any, IS a given cv2
// (tests/test_cv2_goldens.py tries them all).  Engine and oracle implement every member with the same float32 operations.
constexpr int kWarpF32 = 1;          // 0: the classic fixed-point path (all other bits ignored)
constexpr int kWarpCoordFma = 2;     // coordinates: fma(M0, x, fma(M1, y, M2)) instead of (x * M0 + y * M1) + M2
constexpr int kWarpInterFma = 4;     // interpolation with fused multiply-adds
constexpr int kWarpInterTwoWeights = 8;   // (1 - t) * a + t * b instead of a + t * (b - a)
constexpr int kWarpCoordF64 = 16;    // numerators and denominator in double from the double matrix, sx = float(X / W)
constexpr int kWarpCoordRecip = 32;  // sx = X * (1 / w) instead of X / w
constexpr int kWarpModes = 64;

__host__ __device__ __forceinline__ float warp_lerp(float a, float b, float t, int flags)
{
    if (flags & kWarpInterTwoWeights) {
        const float u = 1.f - t;
        return (flags & kWarpInterFma) ? fmaf(t, b, u * a) : u * a + t * b;
    }
    return (flags & kWarpInterFma) ? fmaf(t, b - a, a) : a + t * (b - a);
}

// one destination pixel of cv2.warpPerspective(src, M_inv given, INTER_LINEAR, BORDER_CONSTANT 0) through member `flags` of the family.
// T = uint8_t / uint16_t; results are the integers cvRound leaves before the saturating store.
template <typename T, int CN>
__device__ __forceinline__ void warp_f32_px(const T *__restrict__ src, int sw, int sh, const double *__restrict__ M, int x, int y, int flags,
                                            int out[CN])
{
    float sx, sy;
    if (flags & kWarpCoordF64) {
        const double X = M[0] * x + M[1] * y + M[2], Y = M[3] * x + M[4] * y + M[5], W = M[6] * x + M[7] * y + M[8];
        sx = (float)(X / W); sy = (float)(Y / W);
    } else {
        const float m0 = (float)M[0], m1 = (float)M[1], m2 = (float)M[2], m3 = (float)M[3], m4 = (float)M[4], m5 = (float)M[5];
        const float m6 = (float)M[6], m7 = (float)M[7], m8 = (float)M[8], fx = (float)x, fy = (float)y;
        float X, Y, W;
        if (flags & kWarpCoordFma) {
            X = fmaf(m0, fx, fmaf(m1, fy, m2)); Y = fmaf(m3, fx, fmaf(m4, fy, m5)); W = fmaf(m6, fx, fmaf(m7, fy, m8));
        } else {
            X = (fx * m0 + fy * m1) + m2; Y = (fx * m3 + fy * m4) + m5; W = (fx * m6 + fy * m7) + m8;
        }
        if (flags & kWarpCoordRecip) { const float iw = 1.f / W; sx = X * iw; sy = Y * iw; }
        else { sx = X / W; sy = Y / W; }
    }
#pragma unroll
    for (int k = 0; k < CN; ++k) out[k] = 0;
    if (!(sx > -2.f && sx < (float)sw + 1.f && sy > -2.f && sy < (float)sh + 1.f)) return;   // every tap outside (also NaN / inf): border value
    const float flx = floorf(sx), fly = floorf(sy);
    const int ix = (int)flx, iy = (int)fly;
    const float alpha = sx - flx, beta = sy - fly;
    const bool x0 = ix >= 0 && ix < sw, x1 = ix + 1 >= 0 && ix + 1 < sw, y0 = iy >= 0 && iy < sh, y1 = iy + 1 >= 0 && iy + 1 < sh;
#pragma unroll
    for (int k = 0; k < CN; ++k) {
        const float p00 = (x0 && y0) ? (float)src[((size_t)iy * sw + ix) * CN + k] : 0.f;
        const float p01 = (x1 && y0) ? (float)src[((size_t)iy * sw + ix + 1) * CN + k] : 0.f;
        const float p10 = (x0 && y1) ? (float)src[((size_t)(iy + 1) * sw + ix) * CN + k] : 0.f;
        const float p11 = (x1 && y1) ? (float)src[((size_t)(iy + 1) * sw + ix + 1) * CN + k] : 0.f;
        out[k] = rne_f(warp_lerp(warp_lerp(p00, p01, alpha, flags), warp_lerp(p10, p11, alpha, flags), beta, flags));
    }
}

// ---- BlendMask weight: (img * float32(mask / 255.0)).astype(uint8)  (surroundBEV.py:187-188, 279-280) -------
__host__ __device__ __forceinline__ float blend_weight_f32(int mask_u8) { return (float)((double)mask_u8 / 255.0); }
__device__ __forceinline__ int blend_mul(int v, float w) { return (int)((float)v * w); }  // truncation
// The same truncation without floats, for the unit kernels (round 6): for every v, m in 0 .. 255
//     trunc(f32(v) * f32(m / 255.0)) == (v * m) / 255 == (v * (m * 32897)) >> 23
// (the float product is within 2^-17 relative of v m / 255 and never crosses an integer it should not: all 65,536 pairs are checked by
// tests/native/blend_exhaustive.cpp, which compiles these two functions and blend_weight_f32 for the host).  m * 32897 <= 8,388,735 fits
// 24 bits and v * that fits 32: v_mul_u32_u24 + v_lshrrev_b32.
__host__ __device__ __forceinline__ uint32_t blend_weight_q23(uint32_t mask_u8) { return mask_u8 * 32897u; }
__host__ __device__ __forceinline__ uint32_t blend_apply_q23(uint32_t v, uint32_t wq)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(v, wq) >> 23;
#else
    return (v * wq) >> 23;
#endif
}

// ---- wave64 / block reductions for the balance statistics --------------------------------------------------
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

}  // namespace bevw
