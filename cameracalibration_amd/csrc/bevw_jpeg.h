// bevw_jpeg.h -- the JPEG wire format either side of the path, gfx950 only.
//
// The reference reads its camera frames with cv2.imread (main.py:74-77, Tools/undistort.py:65) and writes the stitched
// image with cv2.imwrite (SurroundBirdEyeView/surroundBEV.py:340, Tools/undistort.py:73); for ".jpg" both are libjpeg(-turbo) with the
// library's defaults (baseline Huffman, ISLOW DCT, fancy upsampling; quality 95, 4:2:0, Annex-K tables).  This header holds
// the device side of both directions, bit-exact against that library (oracle/jpegoracle.c, pinned against Pillow's
// libjpeg-turbo), so that frames can enter and leave HBM compressed:
//
//   decode   host: marker parsing and a plain copy of the entropy-coded bytes into pinned memory (the gather the H2D transfer needs);
//            k_jpeg_find_end / k_jpeg_count_raw / k_jpeg_unstuff / k_jpeg_subs : the 0xFF 0x00 stuffing, RSTn markers and fill bytes
//                  removed on the GPU (chunk counts + prefix), the restart segments located and cut into subsequences;
//            k_jpeg_columns : the words of every subsequence laid out as a column (lanes of a wave read neighbouring addresses);
//            k_jpeg_sync0 / k_jpeg_sync : Huffman decoding is sequential by nature.  The entropy-coded segment is cut into
//                  subsequences of kSubBits bits; every lane decodes one from a GUESSED state (first bit of the
//                  subsequence, DC of the first block of an MCU), then again from the exit state of its predecessor,
//                  until no exit state changes -- JPEG's Huffman codes self-synchronise within a few dozen symbols, so
//                  this takes 2 - 3 rounds (correctness does not rest on it: the fixed point is reached by induction
//                  from the first subsequence of every restart segment, whose state is known).  A prefix sum of the blocks
//                  completed and of the DC differences per component turns the states into output positions;
//                  (three walkers run these passes -- bevw_jpeg_walk.h; decode_sub below is their plain statement);
//            k_jpeg_coef  : decodes every subsequence once more from its TRUE entry state and writes the coefficients;
//            k_jpeg_idct  : jpeg_idct_islow, 8 blocks per wave (lane = block x column, transpose through LDS);
//            k_jpeg_color : fancy (triangle) chroma upsampling + YCbCr -> BGR, written straight into the caller's frame layout
//                  (the camera case: k_jpeg_idct_color_h2v2, luma inverse DCT and colour in one pass).
//   encode   k_jenc_ycc   : BGR -> YCbCr + chroma downsampling with libjpeg's edge replication (4:2:0: k_jenc_ycc_h2v2, 8 x 2 pixels per lane);
//            k_jenc_fdct  : jpeg_fdct_islow + quantisation, dummy blocks of partial MCUs;
//            k_jenc_len / k_jenc_scan / k_jenc_bits / k_jenc_stuff : Huffman code lengths per block, a prefix sum to bit
//                  offsets, every block written at its offset (atomic OR on shared words), 0xFF byte stuffing by a
//                  second prefix sum; the file header is assembled once per batch.
//
// Everything a lane does is a __host__ __device__ function, so tests/native/jpeg_emulate.cpp runs the same arithmetic and the
// same synchronisation fixed point on a CPU against the oracle (no product path calls them on the host).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

namespace bevw {
namespace jpg {

constexpr int kSubBits = 1024;   // bits per subsequence of the entropy-coded data (a multiple of 32)
constexpr uint32_t kNoRestart = 0xffffffffu;

// jutils.c jpeg_natural_order: zigzag position -> row-major position
__host__ __device__ __forceinline__ int natural_of(int k)
{
    const uint8_t t[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                           41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                           30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    return t[k & 63];
}
// the inverse: row-major position -> zigzag position
__host__ __device__ __forceinline__ int zigzag_of(int n)
{
    const uint8_t t[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
                           41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
                           46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
    return t[n & 63];
}

// ---- geometry shared by all images of one batch -----------------------------------------------------------------------
struct Geom {
    int32_t w, h, nc, hs, vs;    // image size, components (1 or 3), luma sampling factors (chroma is 1 x 1)
    int32_t mcux, mcuy, bpm, nY; // MCUs per row / column, blocks per MCU, luma blocks per MCU
    int32_t wb[3], hb[3];        // sample planes in blocks (whole MCUs)
    int32_t blk_off[3];          // first block of each component inside one image's coefficient buffer
    int32_t nblk;                // blocks per image
    int32_t plane_off[3];        // byte offset of each sample plane inside one image's plane buffer
    int32_t plane_bytes;
    int32_t dw, dh;              // the chroma components' real size (jdmaster.c downsampled_width / _height)
};

inline Geom make_geom(int w, int h, int nc, int hs, int vs)
{
    Geom G;
    memset(&G, 0, sizeof G);
    G.w = w; G.h = h; G.nc = nc;
    if (nc == 1) hs = vs = 1;
    G.hs = hs; G.vs = vs;
    G.mcux = (w + 8 * hs - 1) / (8 * hs);
    G.mcuy = (h + 8 * vs - 1) / (8 * vs);
    G.nY = hs * vs;
    G.bpm = nc == 1 ? 1 : G.nY + 2;
    int boff = 0, poff = 0;
    for (int c = 0; c < nc; ++c) {
        G.wb[c] = G.mcux * (c == 0 ? hs : 1);
        G.hb[c] = G.mcuy * (c == 0 ? vs : 1);
        G.blk_off[c] = boff;
        boff += G.wb[c] * G.hb[c];
        G.plane_off[c] = poff;
        poff += G.wb[c] * G.hb[c] * 64;
    }
    G.nblk = boff;
    G.plane_bytes = poff;
    G.dw = (w + hs - 1) / hs;
    G.dh = (h + vs - 1) / vs;
    return G;
}

// ---- decoding tables ----------------------------------------------------------------------------------------------------
struct HuffTab {          // jdhuff.c jpeg_make_d_derived_tbl, 8-bit look-ahead (libjpeg's own HUFF_LOOKAHEAD)
    uint16_t fast[256];   // len << 8 | symbol for codes of <= 8 bits, 0 otherwise
    uint32_t ub[10];      // ub[l - 8], l = 8 .. 16: the first 16-bit window (left-justified) that is NOT a code of length <= l (+ 1 pad)
    int32_t valoff[18];   // index of the first symbol of length l minus its first code
    uint8_t vals[256];
};
static_assert(sizeof(HuffTab) % 16 == 0, "HuffTab is copied to LDS in 16-byte pieces");
struct TableSet { HuffTab t[6]; };   // [2 c] = DC table of component c, [2 c + 1] = its AC table

struct ImageDesc {
    uint32_t stream_word;   // first 32-bit word of this image's un-stuffed entropy-coded bytes in the batch stream buffer
    uint32_t stream_bytes;
    uint32_t seg_first;     // this image's first entry in the segment tables (nseg + 1 entries each)
    uint32_t nseg;          // restart segments (1 without DRI)
    uint32_t sub_first;     // first subsequence slot of this image
    uint32_t nsub;
    uint32_t tables;        // TableSet index
    uint32_t quant;         // quantiser triple index (3 x 64 uint16, row-major)
    uint32_t seg_blocks;    // blocks per restart segment (kNoRestart without DRI)
    uint32_t raw_bytes;     // entropy-coded bytes as they are in the file (0xFF 0x00 stuffing, RSTn markers, the closing marker behind them)
    uint32_t chunk_first;   // this image's first entry in the per-chunk counters of the un-stuffing kernels
    uint32_t error;         // set on the device: the restart markers in the data do not match DRI (nsub is then 0)
};
// What the host hands over per image: stream_word (slot of the raw bytes = slot of the un-stuffed bytes), raw_bytes, seg_first, nseg (from
// DRI), sub_first (from an upper bound of nsub), tables, quant, seg_blocks, chunk_first.  stream_bytes, nsub and error are written by the GPU.

// a state of the sequential decoder between two symbols: bit position | block inside the MCU << 32 | zigzag position << 40
__host__ __device__ __forceinline__ uint64_t pack_state(uint32_t p, uint32_t z, uint32_t k) { return (uint64_t)p | ((uint64_t)z << 32) | ((uint64_t)k << 40); }

struct SubOut { uint64_t exit; int32_t cnt, dc0, dc1, dc2; };

constexpr int kLaneBlock = 64;                  // int16 per lane of k_jpeg_coef's LDS block slots: with the 8-bit look-ahead tables four work-groups fit a CU
constexpr int kColWords = kSubBits / 32 + 4;   // words a lane can touch while it stays inside its own subsequence (+ look-ahead)

// Where a lane reads the 32-bit words of the entropy-coded data from.  `words` is the image's un-stuffed stream.  With 64 lanes walking 64
// different 128-byte stretches of it, every word load of a wave touches 64 cache lines, the lines of 32 waves do not fit the L1 and the
// stream was fetched from memory 14 x (profiles/r03_jpeg/pmc_decode_v3.txt).  k_jpeg_columns therefore lays the kColWords words of every
// subsequence out as a COLUMN (word w of subsequence j at col[w * stride + j]): lanes of a wave, which advance at about the same pace, then
// read neighbouring addresses.  Words beyond the column (a lane finishing a long block past its range) come from the stream itself.
struct WordSource {
    const uint32_t *words;   // the image's stream (32-bit words)
    const uint32_t *col;     // this lane's column, or nullptr
    uint32_t stride;         // subsequences of the image
    uint32_t word0;          // stream index of the column's first word
    __host__ __device__ __forceinline__ uint32_t at(uint32_t w) const
    {
        const uint32_t d = w - word0;
        return (col && d < (uint32_t)kColWords) ? col[(size_t)d * stride] : words[w];
    }
};

// jdhuff.c decode_mcu_slow over the bits [entry position, end_bit) of one image's entropy-coded data, starting between two symbols
// in state `entry`.  Returns the state in which the first symbol at or after end_bit is met, the number of blocks completed and
// the sum of the DC differences per component.  WRITE: the entry state is the true one; `blk` is the index (scan order) of the
// block in progress, pred the DC predictions; coefficients are written (row-major int16, DC already predicted) until blk_cap.
//
// This is the PLAIN statement of what a lane does with a subsequence -- one branch per case, as jdhuff.c reads.  The kernels run the walkers of
// bevw_jpeg_walk.h (straight-line for 64 divergent lanes, scalar for the serial tail, storing for the final pass); tests/native/jpeg_emulate.cpp
// runs this function and those walkers side by side on every subsequence and every entry state the fixed point goes through.
template <bool WRITE>
__host__ __device__ inline SubOut decode_sub(const WordSource &src, const HuffTab *tabs, const Geom &G, uint64_t entry,
                                             uint32_t end_bit, int16_t *__restrict__ coef, uint32_t blk, uint32_t blk_cap, int32_t pred0,
                                             int32_t pred1, int32_t pred2, const uint8_t *nat = nullptr, int16_t *lbuf = nullptr)
{
    // WRITE: a block belongs to the lane in whose range it STARTS.  The owner assembles it in lbuf (64 int16 of its own, zero at entry),
    // keeps decoding past end_bit until the block is complete, and the block is stored whole -- no read-modify-write of zeroed lines, no
    // zero fill of the coefficient buffer.  The block in progress at a lane's entry (k != 0) is its predecessor's: it is decoded for the
    // state only.
    uint32_t p = (uint32_t)entry, z = (uint32_t)(entry >> 32) & 255u, k = (uint32_t)(entry >> 40) & 255u;
    SubOut R;
    R.cnt = 0; R.dc0 = R.dc1 = R.dc2 = 0;
    // bit window: w0 | w1 = the two (big-endian) words around the read position, `off` = bits of w0 already used, nraw = the word after
    // them.  One 32-bit window holds a whole symbol (code <= 16 bits + <= 16 extra bits).
    uint32_t widx = p >> 5, off = p & 31u;
    uint32_t w0 = __builtin_bswap32(src.at(widx)), w1 = __builtin_bswap32(src.at(widx + 1)), nraw = src.at(widx + 2);
    // position of the block in progress (WRITE): MCU (mx, my)
    int mx = 0, my = 0;
    if (WRITE) {
        const uint32_t mcu = blk / (uint32_t)G.bpm;
        mx = (int)(mcu % (uint32_t)G.mcux);
        my = (int)(mcu / (uint32_t)G.mcux);
    }
    bool own = k == 0;   // WRITE: the block in progress started inside this lane's range
    while (WRITE ? ((p < end_bit || k != 0) && blk < blk_cap) : p < end_bit) {
        const int c = (int)z < G.nY ? 0 : 1 + (int)z - G.nY;
        const HuffTab &T = tabs[2 * c + (k ? 1 : 0)];
        const uint32_t window = off ? (w0 << off) | (w1 >> (32u - off)) : w0;
        const uint32_t peek = window >> 16;
        uint32_t len, sym;
        const uint32_t e = T.fast[peek >> 8];
        if (e) {
            len = e >> 8;
            sym = e & 255u;
        } else {
            // a code of 9 .. 16 bits: its length is 9 + the number of limits the window has reached
            len = 9u + (peek >= T.ub[1]) + (peek >= T.ub[2]) + (peek >= T.ub[3]) + (peek >= T.ub[4]) + (peek >= T.ub[5]) + (peek >= T.ub[6]) + (peek >= T.ub[7]);
            sym = T.vals[((peek >> (16u - len)) + (uint32_t)T.valoff[len]) & 255u];
            if (peek >= T.ub[8]) { len = 16; sym = 0; }   // not a code at all (corrupt data): libjpeg warns and yields 0
        }
        // the extra bits: a DC symbol IS their count, an AC symbol is run << 4 | count
        const uint32_t s = k == 0 ? (sym > 16u ? 16u : sym) : (sym & 15u);
        int32_t v = 0;
        if (s) {
            v = (int32_t)((window << len) >> (32u - s));
            v = v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;   // HUFF_EXTEND
        }
        if (k == 0) {
            if (c == 0) { R.dc0 += v; pred0 += v; }
            else if (c == 1) { R.dc1 += v; pred1 += v; }
            else { R.dc2 += v; pred2 += v; }
            if (WRITE && own) lbuf[0] = (int16_t)(c == 0 ? pred0 : (c == 1 ? pred1 : pred2));
            k = 1;
        } else if (s) {
            k += sym >> 4;
            if (WRITE && own && k <= 63u) lbuf[nat ? (int)nat[k] : natural_of((int)k)] = (int16_t)v;
            ++k;
        } else {
            k = (sym >> 4) == 15u ? k + 16u : 64u;   // ZRL : EOB
        }
        if (k >= 64u) {   // block complete
            if (WRITE && own) {
                int bx, by;
                if (c == 0) { bx = mx * G.hs + (int)z % G.hs; by = my * G.vs + (int)z / G.hs; }
                else { bx = mx; by = my; }
                memcpy(coef + ((size_t)G.blk_off[c] + (size_t)by * G.wb[c] + bx) * 64, lbuf, 128);
                memset(lbuf, 0, 128);
            }
            own = true;
            k = 0;
            ++R.cnt;
            ++blk;
            if (++z == (uint32_t)G.bpm) {
                z = 0;
                if (WRITE && ++mx == G.mcux) { mx = 0; ++my; }
            }
        }
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
    R.exit = pack_state(p, z, k);
    return R;
}

// ---- jidctint.c / jfdctint.c ------------------------------------------------------------------------------------------------
#define BEVW_JFIX_0_298631336 2446
#define BEVW_JFIX_0_390180644 3196
#define BEVW_JFIX_0_541196100 4433
#define BEVW_JFIX_0_765366865 6270
#define BEVW_JFIX_0_899976223 7373
#define BEVW_JFIX_1_175875602 9633
#define BEVW_JFIX_1_501321110 12299
#define BEVW_JFIX_1_847759065 15137
#define BEVW_JFIX_1_961570560 16069
#define BEVW_JFIX_2_053119869 16819
#define BEVW_JFIX_2_562915447 20995
#define BEVW_JFIX_3_072711026 25172

__host__ __device__ __forceinline__ int32_t descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }

// one 8-point pass of jpeg_idct_islow (CONST_BITS 13): in[0..7] -> out[0..7] descaled by `shift`
__host__ __device__ __forceinline__ void idct_1d(const int32_t in[8], int32_t out[8], int shift)
{
    int32_t z2 = in[2], z3 = in[6];
    int32_t z1 = (z2 + z3) * BEVW_JFIX_0_541196100;
    int32_t tmp2 = z1 + z3 * (-BEVW_JFIX_1_847759065);
    int32_t tmp3 = z1 + z2 * BEVW_JFIX_0_765366865;
    int32_t tmp0 = (int32_t)((uint32_t)(in[0] + in[4]) << 13);
    int32_t tmp1 = (int32_t)((uint32_t)(in[0] - in[4]) << 13);
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int32_t z4 = tmp1 + tmp3;
    const int32_t z5 = (z3 + z4) * BEVW_JFIX_1_175875602;
    tmp0 *= BEVW_JFIX_0_298631336; tmp1 *= BEVW_JFIX_2_053119869; tmp2 *= BEVW_JFIX_3_072711026; tmp3 *= BEVW_JFIX_1_501321110;
    z1 *= -BEVW_JFIX_0_899976223; z2 *= -BEVW_JFIX_2_562915447; z3 *= -BEVW_JFIX_1_961570560; z4 *= -BEVW_JFIX_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    out[0] = descale(tmp10 + tmp3, shift); out[7] = descale(tmp10 - tmp3, shift);
    out[1] = descale(tmp11 + tmp2, shift); out[6] = descale(tmp11 - tmp2, shift);
    out[2] = descale(tmp12 + tmp1, shift); out[5] = descale(tmp12 - tmp1, shift);
    out[3] = descale(tmp13 + tmp0, shift); out[4] = descale(tmp13 - tmp0, shift);
}
// sample_range_limit + CENTERJSAMPLE indexed with x & RANGE_MASK (jdmaster.c prepare_range_limit_table)
__host__ __device__ __forceinline__ uint32_t range_limit(int32_t x)
{
    const int32_t i = x & 1023;
    return (uint32_t)(i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896)));
}
// one 8-point pass of jpeg_fdct_islow: pass 0 = rows (results scaled up by 4), pass 1 = columns (descaled)
__host__ __device__ __forceinline__ void fdct_1d(const int32_t in[8], int32_t out[8], int pass)
{
    const int32_t tmp0 = in[0] + in[7], tmp7 = in[0] - in[7], tmp1 = in[1] + in[6], tmp6 = in[1] - in[6];
    const int32_t tmp2 = in[2] + in[5], tmp5 = in[2] - in[5], tmp3 = in[3] + in[4], tmp4 = in[3] - in[4];
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    const int sh = pass == 0 ? 11 : 15;   // CONST_BITS -/+ PASS1_BITS
    if (pass == 0) { out[0] = (tmp10 + tmp11) * 4; out[4] = (tmp10 - tmp11) * 4; }
    else { out[0] = descale(tmp10 + tmp11, 2); out[4] = descale(tmp10 - tmp11, 2); }
    int32_t z1 = (tmp12 + tmp13) * BEVW_JFIX_0_541196100;
    out[2] = descale(z1 + tmp13 * BEVW_JFIX_0_765366865, sh);
    out[6] = descale(z1 + tmp12 * (-BEVW_JFIX_1_847759065), sh);
    z1 = tmp4 + tmp7;
    int32_t z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
    const int32_t z5 = (z3 + z4) * BEVW_JFIX_1_175875602;
    const int32_t t4 = tmp4 * BEVW_JFIX_0_298631336, t5 = tmp5 * BEVW_JFIX_2_053119869, t6 = tmp6 * BEVW_JFIX_3_072711026,
                  t7 = tmp7 * BEVW_JFIX_1_501321110;
    z1 *= -BEVW_JFIX_0_899976223; z2 *= -BEVW_JFIX_2_562915447; z3 *= -BEVW_JFIX_1_961570560; z4 *= -BEVW_JFIX_0_390180644;
    z3 += z5; z4 += z5;
    out[7] = descale(t4 + z1 + z3, sh); out[5] = descale(t5 + z2 + z4, sh);
    out[3] = descale(t6 + z2 + z3, sh); out[1] = descale(t7 + z1 + z4, sh);
}
// jcdctmgr.c quantize: coef / (8 q), rounded half away from zero.  The division is a multiplication by recip = floor(2^32 / (8 q)) + 1
// and a shift -- exact here: the numerator stays below 2^16 and 8 q below 2^11, so n * (recip * 8q - 2^32) < 2^32 (libjpeg-turbo's SIMD
// quantiser does the same with 16-bit reciprocals); an integer division costs ~40 instructions per coefficient on this GPU.
__host__ __device__ __forceinline__ int32_t quantize(int32_t v, int32_t q, uint32_t recip)
{
    const uint32_t qv = (uint32_t)q << 3;
    const uint32_t t = (uint32_t)(v < 0 ? -v : v) + (qv >> 1);
    const int32_t r = (int32_t)(((uint64_t)t * recip) >> 32);
    return v < 0 ? -r : r;
}

// ---- colour -------------------------------------------------------------------------------------------------------------------
// a * b for |a|, |b| < 2^23 (samples and the 17-bit colour constants): one full-rate 24-bit multiply on the device, where a 32-bit one is quarter rate
__host__ __device__ __forceinline__ int32_t mul_small(int32_t a, int32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b);
#else
    return a * b;
#endif
}
// jdcolor.c ycc_rgb_convert (SCALEBITS 16) for one pixel, packed B | G << 8 | R << 16
__host__ __device__ __forceinline__ uint32_t ycc_to_bgr(int y, int cb, int cr)
{
    const int32_t cr_r = (mul_small(91881, cr - 128) + 32768) >> 16;
    const int32_t cb_b = (mul_small(116130, cb - 128) + 32768) >> 16;
    const int32_t g_off = (mul_small(-22554, cb - 128) + 32768 - mul_small(46802, cr - 128)) >> 16;
    int r = y + cr_r, g = y + g_off, b = y + cb_b;
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    g = g < 0 ? 0 : (g > 255 ? 255 : g);
    b = b < 0 ? 0 : (b > 255 ? 255 : b);
    return (uint32_t)b | ((uint32_t)g << 8) | ((uint32_t)r << 16);
}
// jccolor.c rgb_ycc_convert
__host__ __device__ __forceinline__ void bgr_to_ycc(int b, int g, int r, int &y, int &cb, int &cr)
{
    y = (mul_small(19595, r) + mul_small(38470, g) + mul_small(7471, b) + 32768) >> 16;
    cb = (mul_small(-11059, r) - mul_small(21709, g) + (b << 15) + (128 << 16) + 32767) >> 16;
    cr = ((r << 15) - mul_small(27439, g) - mul_small(5329, b) + (128 << 16) + 32767) >> 16;
}

// jdsample.c: the chroma sample of component plane C (pitch cp, real size dw x dh) seen by luma position (x, y)
__host__ __device__ __forceinline__ int upsample_at(const uint8_t *__restrict__ C, int cp, int dw, int dh, int hs, int vs, int x, int y)
{
    if (hs == 1) return C[(size_t)y * cp + x];
    const int cx = x >> 1;
    if (vs == 1) {   // h2v1_fancy_upsample (box filter when the component is <= 2 samples wide)
        if (dw <= 2) return C[(size_t)y * cp + cx];
        int nx = (x & 1) ? cx + 1 : cx - 1;
        nx = nx < 0 ? 0 : (nx > dw - 1 ? dw - 1 : nx);
        const int t = C[(size_t)y * cp + cx], o = C[(size_t)y * cp + nx];
        return (3 * t + o + 1 + (x & 1)) >> 2;
    }
    const int cy = y >> 1;   // h2v2_fancy_upsample
    if (dw <= 2) return C[(size_t)cy * cp + cx];
    int ny = (y & 1) ? cy + 1 : cy - 1, nx = (x & 1) ? cx + 1 : cx - 1;
    ny = ny < 0 ? 0 : (ny > dh - 1 ? dh - 1 : ny);
    nx = nx < 0 ? 0 : (nx > dw - 1 ? dw - 1 : nx);
    const int tcs = 3 * C[(size_t)cy * cp + cx] + C[(size_t)ny * cp + cx];
    const int ocs = 3 * C[(size_t)cy * cp + nx] + C[(size_t)ny * cp + nx];
    return (3 * tcs + ocs + 8 - (x & 1)) >> 4;
}

// ---- encoding tables ------------------------------------------------------------------------------------------------------
struct EncHuff { uint16_t code[256]; uint8_t len[256]; };   // jchuff.c jpeg_make_c_derived_tbl
struct EncTables {
    EncHuff dc[2], ac[2];     // luma, chroma
    uint16_t q[2][64];        // quantisers, row-major
    uint32_t recip[2][64];    // floor(2^32 / (8 q)) + 1 (quantize)
};

__host__ __device__ __forceinline__ int bit_length(int v) { return v ? 32 - __builtin_clz((unsigned)v) : 0; }

// jchuff.c encode_one_block on a block stored in zigzag order.  EMIT = false: the number of bits only.
// EMIT: bits are ORed into `out` (32-bit words, most significant bit first) starting at bit `pos`; the first and the last word a
// block touches may be shared with its neighbours (atomic), the words in between are its own.
template <bool EMIT>
// nonzero: bit k set for every k >= 1 with zz[k] != 0 (bit 0 is ignored) -- the loop visits the coefficients that exist, not all 63 positions
// (64 lanes = 64 blocks walk in step: the wave runs as long as its fullest block has coefficients)
__host__ __device__ inline uint32_t encode_block(const int16_t *__restrict__ zz, int last_dc, const EncHuff &D, const EncHuff &A, uint32_t *out,
                                                 uint32_t pos, uint64_t nonzero)
{
    uint32_t bits = 0;
    uint64_t acc = 0;   // EMIT: pending bits, right-aligned
    int nacc = 0;
    uint32_t widx = pos >> 5;
    int lead = (int)(pos & 31u);   // bits of out[widx] that belong to earlier blocks
    bool first = true;
    auto put = [&](uint32_t code, int len) {
        bits += (uint32_t)len;
        if (!EMIT || !len) return;
        acc = (acc << len) | (code & ((1u << len) - 1u));
        nacc += len;
        while (lead + nacc >= 32) {
            const int take = 32 - lead;
            const uint32_t w = (uint32_t)(acc >> (nacc - take)) & (take == 32 ? 0xffffffffu : ((1u << take) - 1u));
            nacc -= take;
#if defined(__HIP_DEVICE_COMPILE__)
            if (first) atomicOr(&out[widx], w); else out[widx] = w;
#else
            if (first) out[widx] |= w; else out[widx] = w;
#endif
            first = false;
            lead = 0;
            ++widx;
        }
    };
    int temp = zz[0] - last_dc, temp2 = temp;
    if (temp < 0) { temp = -temp; --temp2; }
    int nb = bit_length(temp);
    put(D.code[nb], D.len[nb]);
    if (nb) put((uint32_t)temp2, nb);
    int prev = 0;
    for (uint64_t m = nonzero & ~1ull; m; m &= m - 1ull) {
        const int k = __builtin_ctzll(m);
        int r = k - prev - 1;
        prev = k;
        temp = zz[k];
        while (r > 15) { put(A.code[0xF0], A.len[0xF0]); r -= 16; }
        temp2 = temp;
        if (temp < 0) { temp = -temp; --temp2; }
        nb = bit_length(temp);
        put(A.code[(r << 4) + nb], A.len[(r << 4) + nb]);
        put((uint32_t)temp2, nb);
    }
    if (prev < 63) put(A.code[0], A.len[0]);   // zeros behind the last coefficient: EOB
    if (EMIT && nacc > 0) {   // the tail shares its word with the next block
        const uint32_t w = (uint32_t)(acc & ((1ull << nacc) - 1ull)) << (32 - lead - nacc);
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(&out[widx], w);
#else
        out[widx] |= w;
#endif
    }
    return bits;
}

// The two halves of encode_block<false>: the bits of the AC coefficients (known as soon as the block is quantised: k_jenc_fdct) and of
// the DC difference (needs the previous block of the component: k_jenc_scan).  aclen / dclen = EncHuff::len of the component's tables.
__host__ __device__ inline uint64_t nonzero_mask(const int16_t *zz)   // bit k = zz[k] != 0, k >= 1; bit 0 set (see ac_code_bits_octet)
{
    uint64_t m = 1;
    for (int k = 1; k < 64; ++k) m |= zz[k] ? 1ull << k : 0ull;
    return m;
}
__host__ __device__ inline uint32_t ac_code_bits(const int16_t *zz, const uint8_t *aclen)
{
    uint32_t bits = 0;
    int r = 0;
    for (int k = 1; k < 64; ++k) {
        int t = zz[k];
        if (t == 0) { ++r; continue; }
        bits += (uint32_t)(r >> 4) * aclen[0xF0];
        r &= 15;
        if (t < 0) t = -t;
        const int nb = bit_length(t);
        bits += (uint32_t)aclen[(r << 4) + nb] + (uint32_t)nb;
        r = 0;
    }
    if (r > 0) bits += aclen[0];
    return bits;
}
// ac_code_bits split over the 8 lanes of a block: lane r prices the coefficients at zigzag positions 8 r .. 8 r + 7.  `nonzero` = bit k set
// for every k >= 1 with zz[k] != 0, and bit 0 set (the DC position stands for "the run starts behind me"): the run in front of a coefficient
// is its distance to the next lower set bit.  The sum of the eight lanes' results is ac_code_bits (the lane that holds position 63 adds the
// EOB); tests/native/jpeg_emulate.cpp checks exactly that on every block it encodes.
__host__ __device__ inline uint32_t ac_code_bits_octet(const int16_t *z8, int r, uint64_t nonzero, const uint8_t *aclen)
{
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = 8 * r + i;
        int t = z8[i];
        if (k == 0 || t == 0) continue;
        const uint64_t below = nonzero & ((1ull << k) - 1ull);   // (never 0: bit 0)
        const int prev = 63 - __builtin_clzll(below);
        const int run = k - prev - 1;
        if (t < 0) t = -t;
        const int nb = bit_length(t);
        bits += (uint32_t)(run >> 4) * aclen[0xF0] + (uint32_t)aclen[((run & 15) << 4) + nb] + (uint32_t)nb;
    }
    if (r == 7 && !(nonzero >> 63)) bits += aclen[0];   // zeros behind the last coefficient: EOB
    return bits;
}
__host__ __device__ __forceinline__ uint32_t dc_code_bits(int diff, const uint8_t *dclen)
{
    const int nb = bit_length(diff < 0 ? -diff : diff);
    return (uint32_t)dclen[nb] + (uint32_t)nb;
}

// index (scan order) of the block whose DC is the prediction of block `blk`, or -1 for the first block of a component
__host__ __device__ __forceinline__ int dc_predecessor(int blk, const Geom &G)
{
    const int mcu = blk / G.bpm, z = blk % G.bpm;
    if (z < G.nY) return z > 0 ? blk - 1 : (mcu > 0 ? blk - G.bpm + G.nY - 1 : -1);
    return mcu > 0 ? blk - G.bpm : -1;
}


// ---- per-lane bodies of the encoder's first two kernels ----------------------------------------------------------------------
// k_jenc_ycc: the lane of chroma position (xc, yo) of the padded chroma plane converts its hs x vs pixels (jccolor.c), writes their
// luma samples and the downsampled chroma pair (jcsample.c h2v2_downsample: bias 1, 2, 1, 2 ...; h2v1_downsample: 0, 1, 0, 1 ...).
// Reads are clamped to the image: libjpeg replicates the last column / row BEFORE averaging (expand_right_edge), but pads the
// chroma planes below the last real chroma row with COPIES of that row (jcprepct.c pre_process_data).
__host__ __device__ inline void enc_ycc_at(const uint8_t *__restrict__ bgr, size_t pitch, const Geom &G, int xc, int yo, uint8_t *__restrict__ Y,
                                           uint8_t *__restrict__ Cb, uint8_t *__restrict__ Cr)
{
    const int yw = G.wb[0] * 8, cw = G.wb[1] * 8;
    int scb = 0, scr = 0;
    const int ysrc = yo < G.dh ? yo : G.dh - 1;
    for (int j = 0; j < G.vs; ++j)
        for (int i = 0; i < G.hs; ++i) {
            const int X = xc * G.hs + i;
            const int sx = X < G.w ? X : G.w - 1;
            {   // luma of the position itself
                const int Yp = yo * G.vs + j;
                const int sy = Yp < G.h ? Yp : G.h - 1;
                const uint8_t *px = bgr + (size_t)sy * pitch + (size_t)sx * 3;
                int y, cb, cr;
                bgr_to_ycc(px[0], px[1], px[2], y, cb, cr);
                Y[(size_t)Yp * yw + X] = (uint8_t)y;
                if (yo == ysrc) { scb += cb; scr += cr; continue; }
            }
            const int Ys = ysrc * G.vs + j;
            const int sy = Ys < G.h ? Ys : G.h - 1;
            const uint8_t *px = bgr + (size_t)sy * pitch + (size_t)sx * 3;
            int y, cb, cr;
            bgr_to_ycc(px[0], px[1], px[2], y, cb, cr);
            scb += cb;
            scr += cr;
        }
    if (G.hs == 2 && G.vs == 2) {
        const int bias = 1 + (xc & 1);
        scb = (scb + bias) >> 2;
        scr = (scr + bias) >> 2;
    } else if (G.hs == 2) {
        const int bias = xc & 1;
        scb = (scb + bias) >> 1;
        scr = (scr + bias) >> 1;
    }
    Cb[(size_t)yo * cw + xc] = (uint8_t)scb;
    Cr[(size_t)yo * cw + xc] = (uint8_t)scr;
}

// enc_ycc_at for the camera case (4:2:0): one lane = 8 x 2 luma pixels = 4 chroma samples.  Inside the image (and with dword-aligned rows) the two
// rows are read as 6 dwords each and everything is written as dwords (Y: 2 x 8 bytes, Cb / Cr: 4 bytes each); tiles that touch the right or
// bottom edge, where jccolor / jcsample replicate pixels, go sample by sample through enc_ycc_at.  Same arithmetic, same planes
// (tests/native/jpeg_emulate.cpp builds the planes both ways and compares them).
__host__ __device__ inline void enc_ycc_h2v2_tile(const uint8_t *__restrict__ bgr, size_t pitch, const Geom &G, int t, int yo, uint8_t *__restrict__ Y,
                                                  uint8_t *__restrict__ Cb, uint8_t *__restrict__ Cr)
{
    const int yw = G.wb[0] * 8, cw = G.wb[1] * 8, X0 = 8 * t, Y0 = 2 * yo;
    if (4 * t >= cw) return;
    const bool inside = X0 + 8 <= G.w && Y0 + 2 <= G.h && yo < G.dh && pitch % 4 == 0 && ((uintptr_t)bgr & 3u) == 0;
    if (!inside) {
        for (int xc = 4 * t; xc < 4 * t + 4 && xc < cw; ++xc) enc_ycc_at(bgr, pitch, G, xc, yo, Y, Cb, Cr);
        return;
    }
    int cb[2][8], cr[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(bgr + (size_t)(Y0 + j) * pitch + (size_t)X0 * 3);
        uint32_t w[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = row[i];
        uint32_t yy[2] = {0u, 0u};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = 3 * i;
            const int b = (int)((w[n >> 2] >> (8 * (n & 3))) & 255u), g = (int)((w[(n + 1) >> 2] >> (8 * ((n + 1) & 3))) & 255u),
                      r = (int)((w[(n + 2) >> 2] >> (8 * ((n + 2) & 3))) & 255u);
            int y;
            bgr_to_ycc(b, g, r, y, cb[j][i], cr[j][i]);
            yy[i >> 2] |= (uint32_t)y << (8 * (i & 3));
        }
        uint32_t *yo32 = reinterpret_cast<uint32_t *>(Y + (size_t)(Y0 + j) * yw + X0);
        yo32[0] = yy[0];
        yo32[1] = yy[1];
    }
    uint32_t ob = 0, orr = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int bias = 1 + ((4 * t + i) & 1);   // jcsample.c h2v2_downsample: 1, 2, 1, 2, ...
        ob |= (uint32_t)((cb[0][2 * i] + cb[0][2 * i + 1] + cb[1][2 * i] + cb[1][2 * i + 1] + bias) >> 2) << (8 * i);
        orr |= (uint32_t)((cr[0][2 * i] + cr[0][2 * i + 1] + cr[1][2 * i] + cr[1][2 * i + 1] + bias) >> 2) << (8 * i);
    }
    *reinterpret_cast<uint32_t *>(Cb + (size_t)yo * cw + 4 * t) = ob;
    *reinterpret_cast<uint32_t *>(Cr + (size_t)yo * cw + 4 * t) = orr;
}

// k_jenc_fdct: which samples block g (scan order) transforms.  jccoefct.c compress_data: blocks of the last MCU column / row that
// lie beyond the component's own width_in_blocks / height_in_blocks are dummies -- AC = 0, DC = the quantised DC of the previous
// block of the MCU (the block to the left; for a dummy ROW the last block of the row above).  The chain always ends at a real
// block (rx, ry), and a block's DC depends on its samples only, so a dummy is "the root's DC, nothing else".
__host__ __device__ __forceinline__ void enc_block_root(const Geom &G, int g, int &comp, int &rx, int &ry, bool &dc_only)
{
    const int mcu = g / G.bpm, z = g % G.bpm;
    const int mx = mcu % G.mcux, my = mcu / G.mcux;
    int bx, by, rwb, rhb;
    if (z < G.nY) {
        comp = 0;
        bx = mx * G.hs + z % G.hs;
        by = my * G.vs + z / G.hs;
        rwb = (G.w + 7) >> 3;
        rhb = (G.h + 7) >> 3;
    } else {
        comp = 1 + z - G.nY;
        bx = mx;
        by = my;
        rwb = (G.dw + 7) >> 3;
        rhb = (G.dh + 7) >> 3;
    }
    dc_only = false;
    rx = bx;
    ry = by;
    if (by >= rhb) {
        dc_only = true;
        ry = by - 1;
        rx = mx * (comp == 0 ? G.hs : 1) + (comp == 0 ? G.hs : 1) - 1;
    }
    if (rx >= rwb) {
        dc_only = true;
        rx = rwb - 1;
    }
}

// ---- per-lane bodies of the un-stuffing kernels (k_jpeg_find_end, k_jpeg_count_raw, k_jpeg_unstuff) --------------------------------------
struct RawBytes {   // a lane's 16 bytes of the file's entropy-coded data with one byte of context either side
    uint32_t w[4];
    uint32_t prev, next;
    __host__ __device__ __forceinline__ uint32_t at(int i) const { return i < 0 ? prev : (i > 15 ? next : (w[i >> 2] >> (8 * (i & 3))) & 255u); }
};
__host__ __device__ __forceinline__ bool is_rst(uint32_t b) { return (b & 0xF8u) == 0xD0u; }

// position of the first 0xFF among the lane's bytes that is followed by anything but 0x00 / 0xFF / RSTn (the marker that ends the data), or ~0
__host__ __device__ __forceinline__ uint32_t raw_first_terminator(const RawBytes &R, uint32_t pos, uint32_t raw_bytes)
{
    for (int i = 0; i < 16; ++i) {
        if (pos + i >= raw_bytes) break;
        if (R.at(i) == 255u) {
            const uint32_t nx = pos + i + 1u < raw_bytes ? R.at(i + 1) : 0xD9u;
            if (nx != 0u && nx != 255u && !is_rst(nx)) return pos + (uint32_t)i;
        }
    }
    return 0xffffffffu;
}

// bit i of keep: byte i of the lane's 16 stays; bit i of rst: a restart marker starts at byte i (the next kept byte opens a segment)
__host__ __device__ __forceinline__ void raw_classify(const RawBytes &R, uint32_t pos, uint32_t end, uint32_t raw_bytes, uint32_t &keep, uint32_t &rst)
{
    keep = rst = 0;
    for (int i = 0; i < 16; ++i) {
        if (pos + i >= end) break;
        const uint32_t b = R.at(i), nx = pos + i + 1u < raw_bytes ? R.at(i + 1) : 0xD9u, pv = (pos + i) ? R.at(i - 1) : 0u;   // nx may be the closing marker's 0xFF
        bool drop = false;
        if (b == 255u) {
            if (is_rst(nx)) { drop = true; rst |= 1u << i; }
            else if (nx == 255u) drop = true;                       // fill byte
        } else if (pv == 255u && (b == 0u || is_rst(b))) drop = true;   // stuffed zero / second byte of RSTn
        if (!drop) keep |= 1u << i;
    }
}

// ---- host side: marker parsing (jdmarker.c) and the staging copy -----------------------------------------------------------
struct RawHuff { uint8_t bits[17]; uint8_t vals[256]; bool ok; };
struct Parsed {
    int w, h, nc, hs, vs, ri, orientation;
    int tq[3], td[3], ta[3];
    uint16_t q[4][64];   // row-major
    bool qok[4];
    RawHuff dc[4], ac[4];
    size_t scan_off;     // first entropy-coded byte
};
enum { kParseOk = 0, kParseFormat = -1, kParseUnsupported = -2 };

inline int exif_orientation(const uint8_t *p, size_t n)
{
    if (n < 14 || memcmp(p, "Exif\0\0", 6) != 0) return 0;
    const uint8_t *t = p + 6;
    const size_t tn = n - 6;
    const bool le = t[0] == 'I' && t[1] == 'I';
    if (!le && !(t[0] == 'M' && t[1] == 'M')) return 0;
    auto rd16 = [&](size_t o) -> uint32_t { return le ? (uint32_t)t[o] | ((uint32_t)t[o + 1] << 8) : ((uint32_t)t[o] << 8) | t[o + 1]; };
    auto rd32 = [&](size_t o) -> uint32_t { return le ? rd16(o) | (rd16(o + 2) << 16) : (rd16(o) << 16) | rd16(o + 2); };
    const size_t off = rd32(4);
    if (off + 2 > tn) return 0;
    const uint32_t cnt = rd16(off);
    for (uint32_t i = 0; i < cnt; ++i) {
        const size_t e = off + 2 + 12 * (size_t)i;
        if (e + 12 > tn) return 0;
        if (rd16(e) == 0x0112) return (int)rd16(e + 8);
    }
    return 0;
}

// What cv2.imread accepts and this engine decodes: baseline / extended-sequential Huffman, 8 bit, one interleaved scan, grey or
// YCbCr with luma 1x1 / 2x1 / 2x2.  Everything else is refused by name (why): the caller falls back to its own decoder knowingly.
inline int parse_header(const uint8_t *d, size_t n, Parsed &P, std::string &why)
{
    memset(&P, 0, sizeof P);
    if (!d || n < 4 || d[0] != 0xFF || d[1] != 0xD8) { why = "not a JPEG file (no SOI)"; return kParseFormat; }
    size_t i = 2;
    bool sof = false;
    int id[3] = {0, 0, 0};
    for (;;) {
        if (i + 4 > n || d[i] != 0xFF) { why = "truncated or corrupt marker stream"; return kParseFormat; }
        while (i < n && d[i] == 0xFF) ++i;
        if (i >= n) { why = "truncated marker stream"; return kParseFormat; }
        const int m = d[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) { why = "EOI before any scan"; return kParseFormat; }
        if (i + 2 > n) { why = "truncated marker segment"; return kParseFormat; }
        const size_t L = ((size_t)d[i] << 8) | d[i + 1];
        if (L < 2 || i + L > n) { why = "truncated marker segment"; return kParseFormat; }
        const uint8_t *s = d + i + 2;
        const size_t sl = L - 2;
        if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || s[0] != 8) { why = "sample precision other than 8 bits"; return kParseUnsupported; }
            P.h = (s[1] << 8) | s[2];
            P.w = (s[3] << 8) | s[4];
            P.nc = s[5];
            if (P.w <= 0 || P.h <= 0) { why = "empty image / DNL height"; return kParseUnsupported; }
            if ((P.nc != 1 && P.nc != 3) || sl < 6 + 3 * (size_t)P.nc) { why = "component count other than 1 or 3"; return kParseUnsupported; }
            int hsv[3] = {1, 1, 1}, vsv[3] = {1, 1, 1};
            for (int c = 0; c < P.nc; ++c) {
                id[c] = s[6 + 3 * c];
                hsv[c] = s[7 + 3 * c] >> 4;
                vsv[c] = s[7 + 3 * c] & 15;
                P.tq[c] = s[8 + 3 * c];
                if (P.tq[c] > 3) { why = "quantisation table index > 3"; return kParseFormat; }
            }
            if (P.nc == 1) {
                P.hs = P.vs = 1;   // a single-component scan is never interleaved (jdinput.c per_scan_setup)
            } else {
                if (hsv[1] != 1 || vsv[1] != 1 || hsv[2] != 1 || vsv[2] != 1) { why = "chroma sampling other than 1x1"; return kParseUnsupported; }
                if (!((hsv[0] == 1 && vsv[0] == 1) || (hsv[0] == 2 && vsv[0] == 1) || (hsv[0] == 2 && vsv[0] == 2))) {
                    why = "luma sampling other than 1x1, 2x1, 2x2";
                    return kParseUnsupported;
                }
                P.hs = hsv[0];
                P.vs = vsv[0];
            }
            sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            why = "progressive / lossless / arithmetic-coded JPEG";
            return kParseUnsupported;
        } else if (m == 0xC4) {
            size_t o = 0;
            while (o < sl) {
                if (o + 17 > sl) { why = "truncated DHT"; return kParseFormat; }
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3) { why = "bad DHT class / index"; return kParseFormat; }
                RawHuff &t = tc ? P.ac[th] : P.dc[th];
                memset(&t, 0, sizeof t);
                int cnt = 0;
                for (int l = 1; l <= 16; ++l) { t.bits[l] = s[o + l]; cnt += s[o + l]; }
                if (cnt > 256 || o + 17 + (size_t)cnt > sl) { why = "truncated DHT"; return kParseFormat; }
                memcpy(t.vals, s + o + 17, (size_t)cnt);
                t.ok = true;
                o += 17 + (size_t)cnt;
            }
        } else if (m == 0xDB) {
            size_t o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                if (tq > 3 || pq > 1 || o + 1 + (pq ? 128u : 64u) > sl) { why = "bad DQT"; return kParseFormat; }
                for (int k = 0; k < 64; ++k)
                    P.q[tq][natural_of(k)] = pq ? (uint16_t)((s[o + 1 + 2 * k] << 8) | s[o + 2 + 2 * k]) : s[o + 1 + k];
                P.qok[tq] = true;
                o += 1 + (pq ? 128u : 64u);
            }
        } else if (m == 0xDD) {
            if (sl < 2) { why = "bad DRI"; return kParseFormat; }
            P.ri = (s[0] << 8) | s[1];
        } else if (m == 0xE1) {
            const int o = exif_orientation(s, sl);
            if (o) P.orientation = o;
        } else if (m == 0xDA) {
            if (!sof) { why = "SOS before SOF"; return kParseFormat; }
            if (sl < 1 || s[0] != P.nc || sl < 1 + 2 * (size_t)P.nc + 3) { why = "non-interleaved (multi-scan) JPEG"; return kParseUnsupported; }
            for (int c = 0; c < P.nc; ++c) {
                if (s[1 + 2 * c] != id[c]) { why = "scan components out of frame order"; return kParseUnsupported; }
                P.td[c] = s[2 + 2 * c] >> 4;
                P.ta[c] = s[2 + 2 * c] & 15;
                if (P.td[c] > 3 || P.ta[c] > 3) { why = "bad table selector in SOS"; return kParseFormat; }
            }
            if (s[1 + 2 * P.nc] != 0 || s[2 + 2 * P.nc] != 63 || s[3 + 2 * P.nc] != 0) { why = "spectral selection / successive approximation"; return kParseUnsupported; }
            P.scan_off = i + L;
            break;
        }
        i += L;
    }
    for (int c = 0; c < P.nc; ++c)
        if (!P.qok[P.tq[c]] || !P.dc[P.td[c]].ok || !P.ac[P.ta[c]].ok) { why = "a table the scan refers to is missing"; return kParseFormat; }
    if (P.orientation < 1 || P.orientation > 8) P.orientation = 1;   // (an invalid tag is ignored, as cv2.imread does)
    return kParseOk;
}

inline bool make_hufftab(const RawHuff &r, HuffTab &T)
{
    memset(&T, 0, sizeof T);
    int p = 0, code = 0;
    for (int l = 1; l <= 16; ++l) {
        T.valoff[l] = p - code;
        // more codes of this length than the prefix tree has room for (attacker-controlled counts): refuse BEFORE filling the look-ahead
        // table -- the fill below indexes fast[] by the code (found by tests/native/jpeg_parse_fuzz.cpp under UBSan: index 256 ... 32 k)
        if (code + r.bits[l] > (1 << l) || p + r.bits[l] > 256) return false;
        for (int i = 0; i < r.bits[l]; ++i) {
            if (l <= 8) {
                const int c0 = (code + i) << (8 - l);
                for (int f = 0; f < (1 << (8 - l)); ++f) T.fast[c0 + f] = (uint16_t)((l << 8) | r.vals[p + i]);
            }
        }
        p += r.bits[l];
        code += r.bits[l];
        if (code > (1 << l)) return false;
        if (l >= 8) T.ub[l - 8] = (uint32_t)code << (16 - l);   // canonical codes: windows below this are codes of length <= l
        code <<= 1;
    }
    memcpy(T.vals, r.vals, 256);
    return true;
}

// Host statement of what the un-stuffing kernels do (tests/native/jpeg_emulate.cpp uses it; the product un-stuffs on the GPU): entropy-coded
// bytes of d[off ...) with 0xFF 0x00 -> 0xFF, RSTn markers dropped (their positions recorded) and everything from the next marker on left
// out.  dst needs n - off + 16 bytes; seg_byte receives nseg + 1 byte offsets.
inline size_t unstuff_scan(const uint8_t *d, size_t n, size_t off, uint8_t *dst, std::vector<uint32_t> &seg_byte)
{
    seg_byte.clear();
    seg_byte.push_back(0);
    size_t i = off, o = 0;
    while (i < n) {
        const uint8_t *f = (const uint8_t *)memchr(d + i, 0xFF, n - i);
        const size_t run = f ? (size_t)(f - (d + i)) : n - i;
        memcpy(dst + o, d + i, run);
        o += run;
        i += run;
        if (!f) break;
        size_t j = i + 1;
        while (j < n && d[j] == 0xFF) ++j;
        if (j >= n) break;
        const uint8_t m = d[j];
        if (m == 0) { dst[o++] = 0xFF; i = j + 1; }
        else if (m >= 0xD0 && m <= 0xD7) { seg_byte.push_back((uint32_t)o); i = j + 1; }
        else break;
    }
    seg_byte.push_back((uint32_t)o);
    memset(dst + o, 0, 16);
    return o;
}

// ---- the file header cv2.imwrite's libjpeg writes (jcmarker.c): SOI, JFIF APP0, DQT x 2, SOF0, DHT x 4, SOS --------------
// jcparam.c std_luminance_quant_tbl / std_chrominance_quant_tbl (ITU T.81 Annex K.1, K.2), row-major
inline const uint8_t *std_quant(int t)
{
    static const uint8_t q[2][64] = {
        {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,  69,  56,
         14, 17, 22, 29, 51,  87,  80,  62,  18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
         49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99},
        {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
         99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99}};
    return q[t];
}
// Annex K.3 - K.6 (jcparam.c std_huff_tables): bits[1..16] then the symbols
inline const uint8_t *std_huff(int cls, int t, int &nvals)
{
    static const uint8_t dc0[] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
    static const uint8_t dc1[] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
    static const uint8_t ac0[] = {
        0,    2,    1,    3,    3,    2,    4,    3,    5,    5,    4,    4,    0,    0,    1,    0x7d, 0x01, 0x02, 0x03, 0x00, 0x04, 0x11,
        0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1,
        0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a,
        0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
        0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87,
        0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2,
        0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
        0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
        0xf9, 0xfa};
    static const uint8_t ac1[] = {
        0,    2,    1,    2,    4,    4,    3,    4,    7,    5,    4,    4,    0,    1,    2,    0x77, 0x00, 0x01, 0x02, 0x03, 0x11, 0x04,
        0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09,
        0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28,
        0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
        0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85,
        0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9,
        0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
        0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
        0xf9, 0xfa};
    nvals = cls == 0 ? 12 : 162;
    return cls == 0 ? (t == 0 ? dc0 : dc1) : (t == 0 ? ac0 : ac1);
}

inline void make_enchuff(const uint8_t *bits_vals, EncHuff &E)
{
    memset(&E, 0, sizeof E);
    const uint8_t *bits = bits_vals - 1, *vals = bits_vals + 16;   // bits[1..16]
    int p = 0;
    uint32_t code = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l]; ++i, ++p) {
            E.code[vals[p]] = (uint16_t)code++;
            E.len[vals[p]] = (uint8_t)l;
        }
        code <<= 1;
    }
}

// jcparam.c jpeg_set_quality(quality, force_baseline = TRUE) + the derived Huffman tables
inline void make_enc_tables(int quality, EncTables &T)
{
    quality = quality <= 0 ? 1 : (quality > 100 ? 100 : quality);
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 64; ++i) {
            long v = ((long)std_quant(t)[i] * scale + 50L) / 100L;
            T.q[t][i] = (uint16_t)(v <= 0 ? 1 : (v > 255 ? 255 : v));
            T.recip[t][i] = (uint32_t)((((uint64_t)1 << 32) / ((uint32_t)T.q[t][i] << 3)) + 1u);
        }
    int nv;
    for (int t = 0; t < 2; ++t) {
        make_enchuff(std_huff(0, t, nv), T.dc[t]);
        make_enchuff(std_huff(1, t, nv), T.ac[t]);
    }
}

// jcmarker.c write_file_header / write_frame_header / write_scan_header for a 3-component baseline image
inline std::vector<uint8_t> make_file_header(int w, int h, int hs, int vs, const EncTables &T)
{
    std::vector<uint8_t> H;
    auto b = [&](int v) { H.push_back((uint8_t)v); };
    auto w16 = [&](int v) { b(v >> 8); b(v & 255); };
    w16(0xFFD8);
    w16(0xFFE0); w16(16); b('J'); b('F'); b('I'); b('F'); b(0); b(1); b(1); b(0); w16(1); w16(1); b(0); b(0);
    for (int t = 0; t < 2; ++t) {
        w16(0xFFDB); w16(67); b(t);
        for (int k = 0; k < 64; ++k) b(T.q[t][natural_of(k)]);
    }
    w16(0xFFC0); w16(17); b(8); w16(h); w16(w); b(3);
    b(1); b((hs << 4) | vs); b(0);
    b(2); b(0x11); b(1);
    b(3); b(0x11); b(1);
    for (int t = 0; t < 2; ++t)
        for (int cls = 0; cls < 2; ++cls) {
            int nv;
            const uint8_t *bv = std_huff(cls, t, nv);
            w16(0xFFC4); w16(2 + 1 + 16 + nv); b((cls << 4) | t);
            for (int i = 0; i < 16 + nv; ++i) b(bv[i]);
        }
    w16(0xFFDA); w16(12); b(3); b(1); b(0x00); b(2); b(0x11); b(3); b(0x11); b(0); b(63); b(0);
    return H;
}
}  // namespace jpg
}  // namespace bevw
