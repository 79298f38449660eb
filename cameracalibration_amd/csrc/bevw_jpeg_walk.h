// The three walkers of the JPEG decoder's entropy passes (round 4) -- what a lane (or, for the scalar walker, a wave) does with one subsequence.
// They compute exactly what decode_sub (bevw_jpeg.h: the plain, branchy statement of jdhuff.c's decode_mcu) computes; these are the forms the
// kernels run.  __host__ __device__ like everything a lane does: tests/native/jpeg_emulate.cpp runs them on the CPU against decode_sub, state
// for state and coefficient for coefficient.  Wave-level parts (ballots, the LDS list of completed blocks, readfirstlane, scalar loads, one
// inline v_mad_i32_i24) have a host form of the same meaning next to them.
#pragma once
#include "bevw_jpeg.h"
#include "bevw_device.h"

namespace bevw {
namespace jpg {

__host__ __device__ __forceinline__ uint32_t s_word(const void *base, uint32_t dword_index)   // base, index wave-uniform: an s_load_dword on the device
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) uint32_t *scalar_words_t;
    return ((scalar_words_t)(uintptr_t)base)[dword_index];
#else
    return static_cast<const uint32_t *>(base)[dword_index];
#endif
}
__host__ __device__ __forceinline__ uint32_t uni(uint32_t v)   // the value of the wave's first lane, in an SGPR
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
    return v;
#endif
}
__host__ __device__ __forceinline__ uint32_t mul24u(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}
// acc + a * b for |a| < 2^23, b 0 or 1: one v_mad_i32_i24 (written out: the compiler turns the product back into compare + select, or into a
// 64-bit multiply-add)
__host__ __device__ __forceinline__ int32_t mad24_sel(int32_t a, uint32_t b, int32_t acc)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
#else
    return acc + a * (int32_t)b;
#endif
}
__host__ __device__ __forceinline__ uint32_t umin2(uint32_t a, uint32_t b) { return a < b ? a : b; }
__host__ __device__ __forceinline__ uint32_t umax2(uint32_t a, uint32_t b) { return a > b ? a : b; }
__host__ __device__ __forceinline__ uint32_t s_bswap(uint32_t x) { return uni(__builtin_bswap32(x)); }   // (v_perm_b32 on the loaded VGPR, then to an SGPR)
static_assert(offsetof(HuffTab, ub) == 512 && offsetof(HuffTab, valoff) == 552 && offsetof(HuffTab, vals) == 624, "decode_sub_scalar reads HuffTab by dword index");

// ---- the lane walker of the synchronisation passes ---------------------------------------------------------------------------------------
// decode_sub<false> written for 64 lanes that walk 64 different subsequences: every conditional of the symbol step is taken by SOME lane in
// nearly every step (a block ends, a DC symbol, a zero run, end of block ...), so a branch only adds its exec-mask bookkeeping to a body
// that is executed anyway.  Here a step is straight-line selects; the two real branches left are the codes longer than the look-ahead
// (three dependent table reads) and the refill of the bit window (a memory load).  Same states, same sums, bit for bit.
__host__ __device__ inline SubOut decode_sub_lanes(const WordSource &src, const HuffTab *tabs_lds, const Geom &G, uint64_t entry, uint32_t end_bit)
{
    uint32_t p = (uint32_t)entry, z = (uint32_t)(entry >> 32) & 255u, k = (uint32_t)(entry >> 40) & 255u;
    int32_t cnt = 0, dc_all = 0, dc1 = 0, dc2 = 0;
    uint32_t off = p & 31u;
    // the words: this walk never leaves the subsequence's column (it stops at the first symbol at or behind end_bit: <= kColWords - 1 words
    // from the column's first, look-ahead included), so the next word is one stride further
    const uint32_t *next = src.col + (size_t)((p >> 5) - src.word0) * src.stride;
    uint32_t w0 = __builtin_bswap32(next[0]), w1 = __builtin_bswap32(next[src.stride]), nraw = next[2 * (size_t)src.stride];
    next += 3 * (size_t)src.stride;
    const uint32_t luma_last = (uint32_t)G.nY - 1u, bpm = (uint32_t)G.bpm;
    const uint16_t *const fast = reinterpret_cast<const uint16_t *>(tabs_lds);   // table t: fast[t * (sizeof(HuffTab) / 2) + i]
    constexpr uint32_t kTab16 = (uint32_t)sizeof(HuffTab) / 2u;
    while (p < end_bit) {
        const uint32_t c = umax2(z, luma_last) - luma_last;                                   // component of block z of the MCU
        const uint32_t t = 2u * c + umin2(k, 1u);
        const uint32_t window = (uint32_t)(((((uint64_t)w0) << 32) | w1) >> (32u - off));
        uint32_t e = fast[mul24u(t, kTab16) + (window >> 24)];
        if (e == 0u) {   // 9 .. 16 bits
            const HuffTab &H = *reinterpret_cast<const HuffTab *>(fast + mul24u(t, kTab16));
            const uint32_t peek = window >> 16;
            uint32_t len = 9u + (peek >= H.ub[1]) + (peek >= H.ub[2]) + (peek >= H.ub[3]) + (peek >= H.ub[4]) + (peek >= H.ub[5]) + (peek >= H.ub[6]) + (peek >= H.ub[7]);
            uint32_t sym = H.vals[((peek >> (16u - len)) + (uint32_t)H.valoff[len]) & 255u];
            if (peek >= H.ub[8]) { len = 16; sym = 0; }
            e = (len << 8) | sym;
        }
        const uint32_t len = e >> 8, sym = e & 255u;
        const bool dc = k == 0u;
        const uint32_t s = dc ? umin2(sym, 16u) : (sym & 15u);
        const uint32_t raw = ((window << len) >> 1) >> (31u - s);                           // the s extra bits (0 for s = 0)
        const uint32_t one_s = 1u << s;
        const int32_t v = raw < (one_s >> 1) ? (int32_t)(raw + 1u - one_s) : (int32_t)raw;   // HUFF_EXTEND (0 stays 0)
        const int32_t dcv = dc ? v : 0;                                                     // |dcv| < 2^16: 24-bit multiplies
        dc_all += dcv;
        dc1 = mad24_sel(dcv, c & 1u, dc1);   // (component 1)
        dc2 = mad24_sel(dcv, c >> 1, dc2);   // (component 2; component 0 = all - these two)
        const uint32_t run = sym >> 4;
        const uint32_t k_ac = s ? k + run + 1u : (run == 15u ? k + 16u : 64u);
        k = dc ? 1u : k_ac;
        const uint32_t done = k >> 6;                                                       // (k < 128) the block is complete
        cnt += (int32_t)done;
        k &= done - 1u;
        z += done;
        z = z == bpm ? 0u : z;
        const uint32_t used = len + s;
        p += used;
        off += used;
        if (off >= 32u) {
            off -= 32u;
            w0 = w1;
            w1 = __builtin_bswap32(nraw);
            nraw = *next;
            next += src.stride;
        }
    }
    SubOut R;
    R.exit = pack_state(p, z, k);
    R.cnt = cnt; R.dc0 = dc_all - dc1 - dc2; R.dc1 = dc1; R.dc2 = dc2;
    return R;
}

// ---- the lane walker of the final pass ----------------------------------------------------------------------------------------------------
// decode_sub<true> in the same straight-line form (see decode_sub_lanes): every subsequence is walked from its true entry state and the
// coefficients are stored.  A block belongs to the lane in whose range it STARTS: the owner assembles it in its LDS slot (lbuf, 64 int16,
// zero at entry), keeps decoding past end_bit until the block is complete, and the wave stores the blocks completed in a step whole --
// eight per pass, 16 bytes per lane (wlist: who completed which block).  All 64 lanes stay in the loop until the last one is done.
// Block position: cidx = index of the MCU = index of its chroma blocks, lidx = index of its first luma block; hs, vs are 1 or 2.
__host__ __device__ inline void decode_sub_store(const WordSource &src, const HuffTab *tabs_lds, const Geom &G, uint64_t entry, uint32_t end_bit,
                                        int16_t *__restrict__ coef, uint32_t blk, uint32_t blk_cap, int32_t pred0, int32_t pred1, int32_t pred2,
                                        const uint8_t *nat, int16_t *lbuf, bool alive, uint32_t *wave_list)
{
    uint32_t p = (uint32_t)entry, z = (uint32_t)(entry >> 32) & 255u, k = (uint32_t)(entry >> 40) & 255u;
    uint32_t widx = p >> 5, off = p & 31u;
    uint32_t w0 = __builtin_bswap32(src.at(widx)), w1 = __builtin_bswap32(src.at(widx + 1)), nraw = src.at(widx + 2);
    const uint32_t luma_last = (uint32_t)G.nY - 1u, bpm = (uint32_t)G.bpm, mcux = (uint32_t)G.mcux;
    const uint32_t hs = (uint32_t)G.hs, hshift = hs - 1u, wb0 = (uint32_t)G.wb[0], row_step = ((uint32_t)G.vs - 1u) * wb0;
    const uint32_t off1 = (uint32_t)G.blk_off[1], off2 = (uint32_t)G.blk_off[2];
    uint32_t cidx = blk / bpm, mx = cidx % mcux;
    uint32_t lidx = (cidx / mcux) * (uint32_t)G.vs * wb0 + mx * hs;
    bool own = k == 0u;   // the block in progress started inside this lane's range
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = (int)(threadIdx.x & 63u);
    int16_t *const wave_lbuf = lbuf - (size_t)lane * kLaneBlock;   // lbuf of lane 0 of this wave
#endif
    const uint16_t *const fast = reinterpret_cast<const uint16_t *>(tabs_lds);
    constexpr uint32_t kTab16 = (uint32_t)sizeof(HuffTab) / 2u;
    for (;;) {
        const bool go = alive && (p < end_bit || k != 0u) && blk < blk_cap;
#if defined(__HIP_DEVICE_COMPILE__)
        if (!__any(go)) break;
#else
        if (!go) break;      // (the host runs one lane at a time and stores its blocks itself)
#endif
        bool flush = false;      // this lane completed a block of its own in this step (block flush_idx of the image)
        uint32_t flush_idx = 0;
        if (go) {
            const uint32_t c = umax2(z, luma_last) - luma_last;
            const uint32_t t = 2u * c + umin2(k, 1u);
            const uint32_t window = (uint32_t)(((((uint64_t)w0) << 32) | w1) >> (32u - off));
            uint32_t e = fast[mul24u(t, kTab16) + (window >> 24)];
            if (e == 0u) {   // 9 .. 16 bits
                const HuffTab &H = *reinterpret_cast<const HuffTab *>(fast + mul24u(t, kTab16));
                const uint32_t peek = window >> 16;
                uint32_t len = 9u + (peek >= H.ub[1]) + (peek >= H.ub[2]) + (peek >= H.ub[3]) + (peek >= H.ub[4]) + (peek >= H.ub[5]) + (peek >= H.ub[6]) + (peek >= H.ub[7]);
                uint32_t sym = H.vals[((peek >> (16u - len)) + (uint32_t)H.valoff[len]) & 255u];
                if (peek >= H.ub[8]) { len = 16; sym = 0; }
                e = (len << 8) | sym;
            }
            const uint32_t len = e >> 8, sym = e & 255u;
            const bool dc = k == 0u;
            const uint32_t s = dc ? umin2(sym, 16u) : (sym & 15u);
            const uint32_t raw = ((window << len) >> 1) >> (31u - s);
            const uint32_t one_s = 1u << s;
            const int32_t v = raw < (one_s >> 1) ? (int32_t)(raw + 1u - one_s) : (int32_t)raw;   // HUFF_EXTEND
            // the DC prediction of the block's component
            const int32_t dcv = dc ? v : 0;
            pred0 += c == 0u ? dcv : 0;
            pred1 += c == 1u ? dcv : 0;
            pred2 += c == 2u ? dcv : 0;
            const int32_t pred = c == 0u ? pred0 : (c == 1u ? pred1 : pred2);
            // the coefficient goes into the owner's block: the DC value at 0, an AC value at the natural position of k + run
            const uint32_t run = sym >> 4, kpos = k + run;
            const bool store = own && (dc || (s != 0u && kpos <= 63u));
            const uint32_t at = dc ? 0u : (uint32_t)nat[kpos & 63u];
            if (store) lbuf[at] = (int16_t)(dc ? pred : v);
            const uint32_t k_ac = s ? kpos + 1u : (run == 15u ? k + 16u : 64u);
            k = dc ? 1u : k_ac;
            const uint32_t done = k >> 6;   // (k < 128) the block is complete
            flush = done != 0u && own;
            flush_idx = c == 0u ? lidx + (z >> hshift) * wb0 + (z & hshift) : (c == 1u ? off1 : off2) + cidx;
            own = own || done != 0u;
            blk += done;
            k &= done - 1u;
            z += done;
            const uint32_t wrap = z == bpm ? 1u : 0u;   // the MCU is complete
            z = wrap ? 0u : z;
            cidx += wrap;
            mx += wrap;
            lidx += wrap ? hs : 0u;
            const bool row_end = mx == mcux;             // (only right after a wrap)
            mx = row_end ? 0u : mx;
            lidx += row_end ? row_step : 0u;
            const uint32_t used = len + s;
            p += used;
            off += used;
            if (off >= 32u) {
                off -= 32u;
                w0 = w1;
                w1 = __builtin_bswap32(nraw);
                ++widx;
                nraw = src.at(widx + 2);
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned long long done_mask = __ballot(flush);
        if (done_mask) {
            const int nf = __popcll(done_mask);
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(done_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)done_mask, 0u));
            if (flush) wave_list[rank] = (flush_idx << 6) | (uint32_t)lane;
            for (int g0 = 0; g0 < nf; g0 += 8) {
                const int g = g0 + (lane >> 3);
                if (g < nf) {
                    const uint32_t o = wave_list[g];
                    uint4 *blk_l = reinterpret_cast<uint4 *>(wave_lbuf + (size_t)(o & 63u) * kLaneBlock) + (lane & 7);
                    if (BEVW_COEF_NT) {   // (bevw_device.h: the coefficient blocks pass once)
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        const uint4 v = *blk_l;
                        __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4 *>(coef + (size_t)(o >> 6) * 64) + (lane & 7));
                    } else {
                        *(reinterpret_cast<uint4 *>(coef + (size_t)(o >> 6) * 64) + (lane & 7)) = *blk_l;
                    }
                    *blk_l = make_uint4(0u, 0u, 0u, 0u);
                }
            }
        }
#else
        if (flush) {
            memcpy(coef + (size_t)flush_idx * 64, lbuf, 128);
            memset(lbuf, 0, 128);
        }
#endif
    }
}

// ---- the scalar walker -------------------------------------------------------------------------------------------------------------
// decode_sub<false> for ONE subsequence per wave, computed on the scalar unit.  The last synchronisation rounds of a photograph (flat sky,
// saturated areas: runs of identical short blocks in which a shifted decoder stays consistent for many subsequences) advance one
// subsequence per round and chain: a serial chain of walks by a lone wave, each bound by the dependent-instruction latency of the vector
// pipeline (~170 instructions of 4+ cycles per symbol).  Here every state variable is an SGPR, every branch a scalar branch, the Huffman
// tables are read with s_load_dword from the batch's table sets in memory (the scalar cache holds them) and only the words of the
// entropy-coded data come through the vector memory path (their own counter: a table look-up never waits for a word in flight).

__host__ __device__ inline SubOut decode_sub_scalar(const WordSource &src, const HuffTab *tabs_in_memory, const Geom &G, uint64_t entry, uint32_t end_bit_)
{
    const uint32_t e_hi = uni((uint32_t)(entry >> 32));
    uint32_t p = uni((uint32_t)entry), z = e_hi & 255u, k = (e_hi >> 8) & 255u;
    const uint32_t end_bit = uni(end_bit_), nY = uni((uint32_t)G.nY), bpm = uni((uint32_t)G.bpm);
    const uint32_t stride = uni(src.stride), word0 = uni(src.word0);
    int32_t cnt = 0, dc0 = 0, dc1 = 0, dc2 = 0;
    // the words: vector loads of a uniform address (lane 0's value is taken where it is consumed)
    auto word_at = [&](uint32_t w) -> uint32_t {
        const uint32_t d = w - word0;
        return (src.col && d < (uint32_t)kColWords) ? src.col[(size_t)d * stride] : src.words[w];
    };
    uint32_t widx = p >> 5, off = p & 31u;
    uint32_t w0 = s_bswap(word_at(widx)), w1 = s_bswap(word_at(widx + 1));
    uint32_t nraw = word_at(widx + 2);   // stays in its VGPR until the refill
    while (p < end_bit) {
        const uint32_t c = z < nY ? 0u : 1u + z - nY;
        const HuffTab *T = tabs_in_memory + (2u * c + ((k + 63u) >> 6));   // (k <= 63: 0 for the DC table, 1 for the AC table; a comparison would detour through a VGPR)
        const uint32_t window = (uint32_t)(((((uint64_t)w0) << 32) | w1) >> (32u - off));   // (off < 32: one s_lshr_b64)
        const uint32_t peek = window >> 16, hi8 = peek >> 8;
        const uint32_t e = (s_word(T, hi8 >> 1) >> ((hi8 & 1u) * 16u)) & 0xffffu;
        uint32_t len, sym;
        if (e) {
            len = e >> 8;
            sym = e & 255u;
        } else {
            len = 9u;
#pragma unroll
            for (int i = 1; i <= 7; ++i) len += peek >= s_word(T, 128 + i) ? 1u : 0u;
            const uint32_t vi = ((peek >> (16u - len)) + s_word(T, 138 + len)) & 255u;
            sym = (s_word(T, 156 + (vi >> 2)) >> ((vi & 3u) * 8u)) & 255u;
            if (peek >= s_word(T, 128 + 8)) { len = 16; sym = 0; }
        }
        const uint32_t s = k == 0 ? (sym > 16u ? 16u : sym) : (sym & 15u);
        int32_t v = 0;
        if (s) {
            v = (int32_t)((window << len) >> (32u - s));
            v = v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
        }
        if (k == 0) {
            if (c == 0) dc0 += v;
            else if (c == 1) dc1 += v;
            else dc2 += v;
            k = 1;
        } else if (s) {
            k += (sym >> 4) + 1u;
        } else {
            k = (sym >> 4) == 15u ? k + 16u : 64u;
        }
        if (k >= 64u) {
            k = 0;
            ++cnt;
            if (++z == bpm) z = 0;
        }
        const uint32_t used = len + s;
        p += used;
        off += used;
        if (off >= 32u) {
            off -= 32u;
            w0 = w1;
            w1 = s_bswap(nraw);
            ++widx;
            nraw = word_at(widx + 2);
        }
    }
    SubOut R;
    R.exit = pack_state(p, z, k);
    R.cnt = cnt; R.dc0 = dc0; R.dc1 = dc1; R.dc2 = dc2;
    return R;
}

}  // namespace jpg
}  // namespace bevw
