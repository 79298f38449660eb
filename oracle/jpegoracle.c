/*
 * jpegoracle.c -- CPU restatement of the JPEG codec that sits either side of the hot path.
 *
 * TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's library;
 * the product (cameracalibration_amd/) never does.
 *
 * What it restates.  The reference reads its camera frames with cv2.imread (main.py:74-77; Tools/undistort.py:65;
 * ExtrinsicCalibration/extrinsicCalib.py:203-204) and writes results with cv2.imwrite (SurroundBirdEyeView/surroundBEV.py:340,
 * Tools/undistort.py:73, ExtrinsicCalibration/extrinsicCalib.py:211).  For ".jpg" both are thin wrappers around libjpeg(-turbo) with the library's defaults
 * (opencv/modules/imgcodecs/src/grfmt_jpeg.cpp: jpeg_read_header + jpeg_start_decompress with out_color_space BGR;
 * jpeg_set_defaults + jpeg_set_quality(95, TRUE) + jpeg_start_compress).  libjpeg-turbo is a third-party dependency that is
 * absent from /root/reference (it is inside the opencv-python wheel); its published algorithm is restated here, function by
 * function, with the libjpeg-turbo source file each one follows:
 *
 *   decode  jdmarker.c (marker parsing), jdhuff.c (baseline Huffman decoding, DC prediction, restart intervals),
 *           jidctint.c (jpeg_idct_islow: the default JDCT_ISLOW inverse DCT, CONST_BITS 13 / PASS1_BITS 2),
 *           jdsample.c (h2v2_fancy_upsample / h2v1_fancy_upsample: do_fancy_upsampling defaults to TRUE),
 *           jdmainct.c (context rows: first / last real sample row duplicated), jdcolor.c (ycc_rgb_convert, SCALEBITS 16)
 *   encode  jccolor.c (rgb_ycc_convert), jcsample.c (h2v2_downsample, bias 1,2,1,2; fullsize_downsample; edge expansion),
 *           jcprepct.c (bottom-edge replication), jfdctint.c (jpeg_fdct_islow), jcdctmgr.c (quantisation: round half away from
 *           zero of coef / (8 q)), jccoefct.c (dummy blocks at the right / bottom edge: zero AC, DC of the previous block),
 *           jchuff.c (encode_one_block, flush with 1-bits, 0xFF stuffing), jcparam.c (quality scaling, the Annex K tables,
 *           4:2:0 default sampling), jcmarker.c (JFIF APP0, DQT, SOF0, DHT, SOS, EOI in that order)
 *
 * PARITY PINNED (this row, unlike the cv2 arithmetic of bevoracle.c): Pillow in this image bundles the very library
 * (libjpeg-turbo 3.1.x) and uses the same defaults, so tests/test_jpeg_oracle.py holds this file against it directly --
 * decode == PIL.Image.open(...) byte for byte on the reference's own camera JPEGs and on files of every supported sampling,
 * odd sizes, restart intervals and grayscale; encode == the exact FILE Pillow writes (quality 95, 4:2:0, and other
 * qualities / sizes).  What stays unverified is only that opencv-python's bundled libjpeg-turbo is configured like Pillow's
 * (both leave dct_method = JDCT_ISLOW and do_fancy_upsampling = TRUE).
 *
 * Scope: baseline / extended-sequential Huffman (SOF0 / SOF1), 8-bit, one interleaved scan, 1 component (grayscale, expanded
 * to BGR as cv2.IMREAD_COLOR does) or 3 components YCbCr with luma sampling 1x1, 2x1 or 2x2 and chroma 1x1.  Progressive,
 * arithmetic, CMYK, 12-bit, multi-scan and EXIF orientations other than 1 are reported as unsupported (negative return).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

#define JO_E_FORMAT (-1)      /* not a JPEG / truncated / corrupt */
#define JO_E_UNSUPPORTED (-2) /* valid JPEG outside the scope above */
#define JO_E_SPACE (-3)

/* jpeg_natural_order (jutils.c): zigzag index -> natural (row-major) index, 16 extra entries for corrupt run lengths */
static const uint8_t kNatural[64 + 16] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

/* ---------------------------------------------------------------------------------------------------------------------- */
/* decoder                                                                                                                  */
/* ---------------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int present;
    uint8_t bits[17];
    uint8_t vals[256];
    int32_t maxcode[18]; /* jdhuff.c jpeg_make_d_derived_tbl: largest code of length l, -1 if none; maxcode[17] = sentinel */
    int32_t valoffset[17];
} jo_huff;

typedef struct {
    int w, h, nc, orientation;
    int hs[4], vs[4], tq[4], td[4], ta[4], id[4];
    int hmax, vmax;
    int qpresent[4];
    uint16_t q[4][64]; /* natural order */
    jo_huff dc[4], ac[4];
    int ri;
    const uint8_t *scan;
    size_t scanlen; /* bytes from the first entropy-coded byte to the end of the buffer */
} jo_dec;

static int jo_make_huff(jo_huff *t)
{
    /* jdhuff.c jpeg_make_d_derived_tbl: canonical codes in order of increasing length */
    int p = 0, code = 0;
    for (int l = 1; l <= 16; ++l) {
        t->valoffset[l] = p - code;
        if (t->bits[l]) {
            p += t->bits[l];
            code += t->bits[l];
            t->maxcode[l] = code - 1;
            if (code > (1 << l)) return JO_E_FORMAT;
        } else {
            t->maxcode[l] = -1;
        }
        code <<= 1;
    }
    t->maxcode[17] = 0xFFFFF;
    t->present = 1;
    return 0;
}

static int jo_orientation(const uint8_t *p, size_t n)
{
    /* APP1 "Exif\0\0" + TIFF header; tag 0x0112 of IFD0 (cv2.imread applies it; anything but 1 is out of scope here) */
    if (n < 14 || memcmp(p, "Exif\0\0", 6) != 0) return 0;
    const uint8_t *t = p + 6;
    size_t tn = n - 6;
    int le;
    if (t[0] == 'I' && t[1] == 'I') le = 1;
    else if (t[0] == 'M' && t[1] == 'M') le = 0;
    else return 0;
#define RD16(o) (le ? (uint32_t)t[o] | ((uint32_t)t[(o) + 1] << 8) : ((uint32_t)t[o] << 8) | t[(o) + 1])
#define RD32(o) (le ? RD16(o) | (RD16((o) + 2) << 16) : (RD16(o) << 16) | RD16((o) + 2))
    uint32_t off = RD32(4);
    if ((size_t)off + 2 > tn) return 0;
    uint32_t cnt = RD16(off);
    for (uint32_t i = 0; i < cnt; ++i) {
        size_t e = (size_t)off + 2 + 12 * (size_t)i;
        if (e + 12 > tn) return 0;
        if (RD16(e) == 0x0112) return (int)RD16(e + 8);
    }
#undef RD16
#undef RD32
    return 0;
}

static int jo_parse(const uint8_t *d, size_t n, jo_dec *J)
{
    memset(J, 0, sizeof *J);
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return JO_E_FORMAT;
    size_t i = 2;
    int have_sof = 0;
    for (;;) {
        if (i + 4 > n) return JO_E_FORMAT;
        if (d[i] != 0xFF) return JO_E_FORMAT;
        while (i < n && d[i] == 0xFF) ++i; /* fill bytes */
        if (i >= n) return JO_E_FORMAT;
        const int m = d[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return JO_E_FORMAT;
        if (i + 2 > n) return JO_E_FORMAT;
        const size_t L = ((size_t)d[i] << 8) | d[i + 1];
        if (L < 2 || i + L > n) return JO_E_FORMAT;
        const uint8_t *s = d + i + 2;
        const size_t sl = L - 2;
        if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || s[0] != 8) return JO_E_UNSUPPORTED;
            J->h = (s[1] << 8) | s[2];
            J->w = (s[3] << 8) | s[4];
            J->nc = s[5];
            if (J->w <= 0 || J->h <= 0) return JO_E_UNSUPPORTED;
            if ((J->nc != 1 && J->nc != 3) || sl < 6 + 3 * (size_t)J->nc) return JO_E_UNSUPPORTED;
            for (int c = 0; c < J->nc; ++c) {
                J->id[c] = s[6 + 3 * c];
                J->hs[c] = s[7 + 3 * c] >> 4;
                J->vs[c] = s[7 + 3 * c] & 15;
                J->tq[c] = s[8 + 3 * c];
                if (J->tq[c] > 3) return JO_E_FORMAT;
            }
            have_sof = 1;
        } else if ((m >= 0xC2 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return JO_E_UNSUPPORTED; /* progressive, lossless, arithmetic */
        } else if (m == 0xC4) {
            size_t o = 0;
            while (o < sl) {
                if (o + 17 > sl) return JO_E_FORMAT;
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3) return JO_E_FORMAT;
                jo_huff *t = tc ? &J->ac[th] : &J->dc[th];
                memset(t, 0, sizeof *t);
                int cnt = 0;
                for (int l = 1; l <= 16; ++l) { t->bits[l] = s[o + l]; cnt += s[o + l]; }
                if (cnt > 256 || o + 17 + (size_t)cnt > sl) return JO_E_FORMAT;
                memcpy(t->vals, s + o + 17, (size_t)cnt);
                if (jo_make_huff(t)) return JO_E_FORMAT;
                o += 17 + (size_t)cnt;
            }
        } else if (m == 0xDB) {
            size_t o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                if (tq > 3 || pq > 1) return JO_E_FORMAT;
                if (o + 1 + (pq ? 128u : 64u) > sl) return JO_E_FORMAT;
                for (int k = 0; k < 64; ++k)
                    J->q[tq][kNatural[k]] = pq ? (uint16_t)((s[o + 1 + 2 * k] << 8) | s[o + 2 + 2 * k]) : s[o + 1 + k];
                J->qpresent[tq] = 1;
                o += 1 + (pq ? 128u : 64u);
            }
        } else if (m == 0xDD) {
            if (sl < 2) return JO_E_FORMAT;
            J->ri = (s[0] << 8) | s[1];
        } else if (m == 0xE1) {
            const int o = jo_orientation(s, sl);
            if (o) J->orientation = o;
        } else if (m == 0xDA) {
            if (!have_sof) return JO_E_FORMAT;
            if (sl < 1 || s[0] != J->nc || sl < 1 + 2 * (size_t)J->nc + 3) return JO_E_UNSUPPORTED; /* one interleaved scan only */
            for (int c = 0; c < J->nc; ++c) {
                if (s[1 + 2 * c] != J->id[c]) return JO_E_UNSUPPORTED;
                J->td[c] = s[2 + 2 * c] >> 4;
                J->ta[c] = s[2 + 2 * c] & 15;
                if (J->td[c] > 3 || J->ta[c] > 3) return JO_E_FORMAT;
            }
            if (s[1 + 2 * J->nc] != 0 || s[2 + 2 * J->nc] != 63 || s[3 + 2 * J->nc] != 0) return JO_E_UNSUPPORTED;
            J->scan = d + i + L;
            J->scanlen = n - (i + L);
            break;
        }
        i += L;
    }
    J->hmax = J->vmax = 1;
    for (int c = 0; c < J->nc; ++c) {
        if (J->hs[c] > J->hmax) J->hmax = J->hs[c];
        if (J->vs[c] > J->vmax) J->vmax = J->vs[c];
        if (!J->qpresent[J->tq[c]] || !J->dc[J->td[c]].present || !J->ac[J->ta[c]].present) return JO_E_FORMAT;
    }
    if (J->nc == 1) {
        J->hs[0] = J->vs[0] = J->hmax = J->vmax = 1; /* a single-component scan is never interleaved (jdinput.c per_scan_setup) */
    } else {
        if (J->hs[1] != 1 || J->vs[1] != 1 || J->hs[2] != 1 || J->vs[2] != 1) return JO_E_UNSUPPORTED;
        if (!((J->hs[0] == 1 && J->vs[0] == 1) || (J->hs[0] == 2 && J->vs[0] == 1) || (J->hs[0] == 2 && J->vs[0] == 2)))
            return JO_E_UNSUPPORTED;
    }
    if (J->orientation > 1) return JO_E_UNSUPPORTED;
    return 0;
}

/* jdhuff.c bit reader over the entropy-coded segment: 0xFF 0x00 -> 0xFF, a marker ends the data (zeros are fed after it) */
typedef struct {
    const uint8_t *p, *end;
    uint32_t acc;
    int nbits;
    int marker; /* pending marker byte, 0 if none */
} jo_bits;

static void jo_fill(jo_bits *b)
{
    while (b->nbits <= 24) {
        uint32_t c = 0;
        if (!b->marker && b->p < b->end) {
            c = *b->p++;
            if (c == 0xFF) {
                while (b->p < b->end && *b->p == 0xFF) ++b->p;
                if (b->p < b->end) {
                    const uint32_t c2 = *b->p++;
                    if (c2 != 0) { b->marker = (int)c2; c = 0; }
                } else {
                    c = 0;
                }
            }
        }
        b->acc |= c << (24 - b->nbits);
        b->nbits += 8;
    }
}
static inline uint32_t jo_peek(jo_bits *b, int n) { jo_fill(b); return b->acc >> (32 - n); }
static inline void jo_skip(jo_bits *b, int n) { b->acc <<= n; b->nbits -= n; }
static inline int jo_get(jo_bits *b, int n)
{
    if (n == 0) return 0;
    const uint32_t v = jo_peek(b, n);
    jo_skip(b, n);
    return (int)v;
}
static int jo_decode_sym(jo_bits *b, const jo_huff *t)
{
    /* jdhuff.c jpeg_huff_decode (the slow path is the definition; the look-ahead table is only a cache of it) */
    const uint32_t code16 = jo_peek(b, 16);
    for (int l = 1; l <= 16; ++l) {
        const int32_t code = (int32_t)(code16 >> (16 - l));
        if (code <= t->maxcode[l]) {
            jo_skip(b, l);
            return t->vals[(code + t->valoffset[l]) & 255];
        }
    }
    jo_skip(b, 16);
    return 0; /* corrupt data: libjpeg warns and returns 0 */
}
/* HUFF_EXTEND (jdhuff.c) */
static inline int jo_extend(int r, int s) { return r < (1 << (s - 1)) ? r + (int)((~0u) << s) + 1 : r; }

/* jidctint.c jpeg_idct_islow */
#define CONST_BITS 13
#define PASS1_BITS 2
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
#define DESCALE(x, n) (((x) + ((int64_t)1 << ((n) - 1))) >> (n))

static inline uint8_t jo_range_limit(int64_t x)
{
    /* sample_range_limit + CENTERJSAMPLE indexed with (x & RANGE_MASK), RANGE_MASK = 1023 (jdmaster.c prepare_range_limit_table) */
    const int i = (int)(x & 1023);
    if (i < 128) return (uint8_t)(i + 128);
    if (i < 512) return 255;
    if (i < 896) return 0;
    return (uint8_t)(i - 896);
}

static void jo_idct_islow(const int16_t *coef, const uint16_t *q, uint8_t *out, size_t pitch)
{
    int64_t ws[64];
    for (int c = 0; c < 8; ++c) {
#define DQ(r) ((int64_t)coef[8 * (r) + c] * (int64_t)q[8 * (r) + c])
        int64_t z2 = DQ(2), z3 = DQ(6);
        int64_t z1 = (z2 + z3) * FIX_0_541196100;
        int64_t tmp2 = z1 + z3 * (-FIX_1_847759065);
        int64_t tmp3 = z1 + z2 * FIX_0_765366865;
        z2 = DQ(0);
        z3 = DQ(4);
        int64_t tmp0 = (z2 + z3) * ((int64_t)1 << CONST_BITS);
        int64_t tmp1 = (z2 - z3) * ((int64_t)1 << CONST_BITS);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = DQ(7);
        tmp1 = DQ(5);
        tmp2 = DQ(3);
        tmp3 = DQ(1);
#undef DQ
        z1 = tmp0 + tmp3;
        z2 = tmp1 + tmp2;
        z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336;
        tmp1 *= FIX_2_053119869;
        tmp2 *= FIX_3_072711026;
        tmp3 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223;
        z2 *= -FIX_2_562915447;
        z3 *= -FIX_1_961570560;
        z4 *= -FIX_0_390180644;
        z3 += z5;
        z4 += z5;
        tmp0 += z1 + z3;
        tmp1 += z2 + z4;
        tmp2 += z2 + z3;
        tmp3 += z1 + z4;
        ws[8 * 0 + c] = DESCALE(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
        ws[8 * 7 + c] = DESCALE(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
        ws[8 * 1 + c] = DESCALE(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
        ws[8 * 6 + c] = DESCALE(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
        ws[8 * 2 + c] = DESCALE(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
        ws[8 * 5 + c] = DESCALE(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
        ws[8 * 3 + c] = DESCALE(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
        ws[8 * 4 + c] = DESCALE(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
    }
    for (int r = 0; r < 8; ++r) {
        const int64_t *w = ws + 8 * r;
        uint8_t *o = out + (size_t)r * pitch;
        int64_t z2 = w[2], z3 = w[6];
        int64_t z1 = (z2 + z3) * FIX_0_541196100;
        int64_t tmp2 = z1 + z3 * (-FIX_1_847759065);
        int64_t tmp3 = z1 + z2 * FIX_0_765366865;
        int64_t tmp0 = (w[0] + w[4]) * ((int64_t)1 << CONST_BITS);
        int64_t tmp1 = (w[0] - w[4]) * ((int64_t)1 << CONST_BITS);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7];
        tmp1 = w[5];
        tmp2 = w[3];
        tmp3 = w[1];
        z1 = tmp0 + tmp3;
        z2 = tmp1 + tmp2;
        z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336;
        tmp1 *= FIX_2_053119869;
        tmp2 *= FIX_3_072711026;
        tmp3 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223;
        z2 *= -FIX_2_562915447;
        z3 *= -FIX_1_961570560;
        z4 *= -FIX_0_390180644;
        z3 += z5;
        z4 += z5;
        tmp0 += z1 + z3;
        tmp1 += z2 + z4;
        tmp2 += z2 + z3;
        tmp3 += z1 + z4;
        o[0] = jo_range_limit(DESCALE(tmp10 + tmp3, CONST_BITS + PASS1_BITS + 3));
        o[7] = jo_range_limit(DESCALE(tmp10 - tmp3, CONST_BITS + PASS1_BITS + 3));
        o[1] = jo_range_limit(DESCALE(tmp11 + tmp2, CONST_BITS + PASS1_BITS + 3));
        o[6] = jo_range_limit(DESCALE(tmp11 - tmp2, CONST_BITS + PASS1_BITS + 3));
        o[2] = jo_range_limit(DESCALE(tmp12 + tmp1, CONST_BITS + PASS1_BITS + 3));
        o[5] = jo_range_limit(DESCALE(tmp12 - tmp1, CONST_BITS + PASS1_BITS + 3));
        o[3] = jo_range_limit(DESCALE(tmp13 + tmp0, CONST_BITS + PASS1_BITS + 3));
        o[4] = jo_range_limit(DESCALE(tmp13 - tmp0, CONST_BITS + PASS1_BITS + 3));
    }
}

static inline int jo_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* jdcolor.c build_ycc_rgb_table + ycc_rgb_convert for one pixel, written as B, G, R (JCS_EXT_BGR) */
static inline void jo_ycc_bgr(int y, int cb, int cr, uint8_t *o)
{
    const int32_t cr_r = (int32_t)((91881 * (int64_t)(cr - 128) + 32768) >> 16);   /* FIX(1.40200) */
    const int32_t cb_b = (int32_t)((116130 * (int64_t)(cb - 128) + 32768) >> 16);  /* FIX(1.77200) */
    const int64_t cr_g = -46802 * (int64_t)(cr - 128);                             /* FIX(0.71414) */
    const int64_t cb_g = -22554 * (int64_t)(cb - 128) + 32768;                     /* FIX(0.34414), + ONE_HALF */
    const int r = y + cr_r, g = y + (int)((cb_g + cr_g) >> 16), b = y + cb_b;
    o[0] = (uint8_t)jo_clampi(b, 0, 255);
    o[1] = (uint8_t)jo_clampi(g, 0, 255);
    o[2] = (uint8_t)jo_clampi(r, 0, 255);
}

/* info: width, height, components, luma h, luma v, restart interval, orientation (0 = absent), reserved */
EXPORT int jo_probe(const uint8_t *data, size_t len, int32_t info[8])
{
    jo_dec J;
    const int s = jo_parse(data, len, &J);
    if (s) return s;
    info[0] = J.w; info[1] = J.h; info[2] = J.nc; info[3] = J.hs[0]; info[4] = J.vs[0]; info[5] = J.ri; info[6] = J.orientation; info[7] = 0;
    return 0;
}

/* cv2.imread(path) of a JPEG file (main.py:74-77): out = BGR uint8 [h][w][3].
 * planes, if not NULL, receives the three sample planes after the inverse DCT (component c at planes + offsets of whole blocks:
 * Y [Hy][Wy], Cb [Hc][Wc], Cr [Hc][Wc] with W = blocks * 8) -- an intermediate the GPU tests compare as well. */
EXPORT int jo_decode_bgr(const uint8_t *data, size_t len, uint8_t *out, uint8_t *planes)
{
    jo_dec J;
    int s = jo_parse(data, len, &J);
    if (s) return s;
    const int mcuw = 8 * J.hmax, mcuh = 8 * J.vmax;
    const int mcux = (J.w + mcuw - 1) / mcuw, mcuy = (J.h + mcuh - 1) / mcuh;
    int pw[3], ph[3];
    size_t poff[3], ptotal = 0;
    for (int c = 0; c < J.nc; ++c) {
        pw[c] = mcux * J.hs[c] * 8;
        ph[c] = mcuy * J.vs[c] * 8;
        poff[c] = ptotal;
        ptotal += (size_t)pw[c] * ph[c];
    }
    uint8_t *P = (uint8_t *)malloc(ptotal ? ptotal : 1);
    if (!P) return JO_E_SPACE;
    jo_bits B = {J.scan, J.scan + J.scanlen, 0, 0, 0};
    int pred[3] = {0, 0, 0};
    int restarts_left = J.ri, next_rst = 0;
    for (int my = 0; my < mcuy; ++my) {
        for (int mx = 0; mx < mcux; ++mx) {
            if (J.ri) {
                if (restarts_left == 0) {
                    /* jdhuff.c process_restart: drop the partial byte, expect RSTn, reset the predictions */
                    B.acc = 0;
                    B.nbits = 0;
                    if (!B.marker) { /* the marker has not been reached by the bit reader yet: scan for it */
                        while (B.p + 1 < B.end && !(B.p[0] == 0xFF && B.p[1] != 0 && B.p[1] != 0xFF)) ++B.p;
                        if (B.p + 1 < B.end) { B.marker = B.p[1]; B.p += 2; }
                    }
                    if (B.marker != 0xD0 + next_rst) { free(P); return JO_E_FORMAT; }
                    B.marker = 0;
                    next_rst = (next_rst + 1) & 7;
                    pred[0] = pred[1] = pred[2] = 0;
                    restarts_left = J.ri;
                }
                --restarts_left;
            }
            for (int c = 0; c < J.nc; ++c) {
                for (int by = 0; by < J.vs[c]; ++by) {
                    for (int bx = 0; bx < J.hs[c]; ++bx) {
                        int16_t blk[64];
                        memset(blk, 0, sizeof blk);
                        /* jdhuff.c decode_mcu_slow */
                        int sz = jo_decode_sym(&B, &J.dc[J.td[c]]);
                        if (sz) {
                            if (sz > 16) sz = 16;
                            const int r = jo_get(&B, sz);
                            pred[c] += jo_extend(r, sz);
                        }
                        blk[0] = (int16_t)pred[c];
                        for (int k = 1; k < 64; ++k) {
                            const int rs = jo_decode_sym(&B, &J.ac[J.ta[c]]);
                            const int r = rs >> 4, sa = rs & 15;
                            if (sa) {
                                k += r;
                                const int v = jo_get(&B, sa);
                                blk[kNatural[k]] = (int16_t)jo_extend(v, sa);
                            } else {
                                if (r != 15) break;
                                k += 15;
                            }
                        }
                        const int X = (mx * J.hs[c] + bx) * 8, Y = (my * J.vs[c] + by) * 8;
                        jo_idct_islow(blk, J.q[J.tq[c]], P + poff[c] + (size_t)Y * pw[c] + X, (size_t)pw[c]);
                    }
                }
            }
        }
    }
    if (planes) memcpy(planes, P, ptotal);
    if (J.nc == 1) {
        /* JCS_GRAYSCALE -> BGR: cv2 replicates the sample (grfmt_jpeg.cpp, icvCvt_Gray2BGR_8u_C1C3R) */
        for (int y = 0; y < J.h; ++y)
            for (int x = 0; x < J.w; ++x) {
                const uint8_t v = P[(size_t)y * pw[0] + x];
                uint8_t *o = out + ((size_t)y * J.w + x) * 3;
                o[0] = o[1] = o[2] = v;
            }
        free(P);
        return 0;
    }
    /* jdsample.c: chroma up to the luma grid.  dw / dh = the component's downsampled_width / _height (jdmaster.c) */
    const int dw = (J.w * J.hs[1] + J.hmax - 1) / J.hmax, dh = (J.h * J.vs[1] + J.vmax - 1) / J.vmax;
    for (int y = 0; y < J.h; ++y) {
        for (int x = 0; x < J.w; ++x) {
            int cc[2];
            for (int c = 1; c <= 2; ++c) {
                const uint8_t *C = P + poff[c];
                const size_t cp = (size_t)pw[c];
                if (J.hmax == 1 && J.vmax == 1) {
                    cc[c - 1] = C[(size_t)y * cp + x];
                } else if (J.hmax == 2 && J.vmax == 1) {
                    /* h2v1_fancy_upsample: 3/4 nearer + 1/4 further, rounding 1 (even) / 2 (odd); edge columns are copies.
                     * (jdsample.c uses it only when downsampled_width > 2, else the box filter h2v1_upsample.) */
                    const int cx = x >> 1;
                    if (dw > 2) {
                        const int nb = jo_clampi((x & 1) ? cx + 1 : cx - 1, 0, dw - 1);
                        const int t = C[(size_t)y * cp + cx], o = C[(size_t)y * cp + nb];
                        cc[c - 1] = (x & 1) ? (3 * t + o + 2) >> 2 : (3 * t + o + 1) >> 2;
                    } else {
                        cc[c - 1] = C[(size_t)y * cp + cx];
                    }
                } else {
                    /* h2v2_fancy_upsample: vertical 3:1 of the nearer / further row (context rows of jdmainct.c: the first and the
                     * last REAL row are duplicated), then horizontal 3:1 with rounding 8 (even) / 7 (odd) */
                    const int cx = x >> 1, cy = y >> 1;
                    if (dw > 2) {
                        const int ny = jo_clampi((y & 1) ? cy + 1 : cy - 1, 0, dh - 1);
                        const int nx = jo_clampi((x & 1) ? cx + 1 : cx - 1, 0, dw - 1);
                        const int tcs = 3 * C[(size_t)cy * cp + cx] + C[(size_t)ny * cp + cx];
                        const int ocs = 3 * C[(size_t)cy * cp + nx] + C[(size_t)ny * cp + nx];
                        cc[c - 1] = (x & 1) ? (3 * tcs + ocs + 7) >> 4 : (3 * tcs + ocs + 8) >> 4;
                    } else {
                        cc[c - 1] = C[(size_t)cy * cp + cx]; /* h2v2_upsample (box) */
                    }
                }
            }
            jo_ycc_bgr(P[poff[0] + (size_t)y * pw[0] + x], cc[0], cc[1], out + ((size_t)y * J.w + x) * 3);
        }
    }
    free(P);
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------- */
/* encoder: cv2.imwrite("x.jpg", bgr) = libjpeg defaults at quality 95 (surroundBEV.py:340, Tools/undistort.py:73)                      */
/* ---------------------------------------------------------------------------------------------------------------------- */
/* jcparam.c std_luminance_quant_tbl / std_chrominance_quant_tbl (ITU T.81 Annex K.1 / K.2), natural order */
static const uint8_t kStdQ[2][64] = {
    {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,  69,  56,
     14, 17, 22, 29, 51,  87,  80,  62,  18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
     49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99},
    {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99}};
/* jcparam.c std_huff_tables (Annex K.3 - K.6): bits[1..16], values */
static const uint8_t kDcBits[2][17] = {{0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0}};
static const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kAcBits[2][17] = {{0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d}, {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77}};
static const uint8_t kAcVals[2][162] = {
    {0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
     0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
     0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
     0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
     0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
     0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
     0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
     0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa},
    {0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
     0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
     0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
     0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
     0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
     0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
     0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
     0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa}};

typedef struct { uint32_t code[256]; uint8_t len[256]; } jo_ehuff;

static void jo_make_ehuff(const uint8_t bits[17], const uint8_t *vals, jo_ehuff *t)
{
    /* jchuff.c jpeg_make_c_derived_tbl */
    memset(t, 0, sizeof *t);
    int p = 0;
    uint32_t code = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l]; ++i, ++p) {
            t->code[vals[p]] = code++;
            t->len[vals[p]] = (uint8_t)l;
        }
        code <<= 1;
    }
}

/* jcparam.c jpeg_quality_scaling + jpeg_add_quant_table(force_baseline = TRUE) */
static void jo_quant_tables(int quality, uint16_t q[2][64])
{
    if (quality <= 0) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 64; ++i) {
            long v = ((long)kStdQ[t][i] * scale + 50L) / 100L;
            if (v <= 0) v = 1;
            if (v > 255) v = 255;
            q[t][i] = (uint16_t)v;
        }
}

/* jfdctint.c jpeg_fdct_islow on centred samples; output scaled up by 8 */
static void jo_fdct_islow(int32_t *d)
{
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 8; ++i) {
            int32_t *p = pass == 0 ? d + 8 * i : d + i;
            const int st = pass == 0 ? 1 : 8;
            const int64_t tmp0 = p[0] + p[7 * st], tmp7 = p[0] - p[7 * st], tmp1 = p[st] + p[6 * st], tmp6 = p[st] - p[6 * st];
            const int64_t tmp2 = p[2 * st] + p[5 * st], tmp5 = p[2 * st] - p[5 * st], tmp3 = p[3 * st] + p[4 * st], tmp4 = p[3 * st] - p[4 * st];
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            int64_t z1 = (tmp12 + tmp13) * FIX_0_541196100;
            const int sh_even = pass == 0 ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS;
            if (pass == 0) {
                p[0] = (int32_t)((tmp10 + tmp11) * (1 << PASS1_BITS));
                p[4 * st] = (int32_t)((tmp10 - tmp11) * (1 << PASS1_BITS));
            } else {
                p[0] = (int32_t)DESCALE(tmp10 + tmp11, PASS1_BITS);
                p[4 * st] = (int32_t)DESCALE(tmp10 - tmp11, PASS1_BITS);
            }
            p[2 * st] = (int32_t)DESCALE(z1 + tmp13 * FIX_0_765366865, sh_even);
            p[6 * st] = (int32_t)DESCALE(z1 + tmp12 * (-FIX_1_847759065), sh_even);
            z1 = tmp4 + tmp7;
            int64_t z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
            const int64_t z5 = (z3 + z4) * FIX_1_175875602;
            const int64_t t4 = tmp4 * FIX_0_298631336, t5 = tmp5 * FIX_2_053119869, t6 = tmp6 * FIX_3_072711026, t7 = tmp7 * FIX_1_501321110;
            z1 *= -FIX_0_899976223;
            z2 *= -FIX_2_562915447;
            z3 *= -FIX_1_961570560;
            z4 *= -FIX_0_390180644;
            z3 += z5;
            z4 += z5;
            p[7 * st] = (int32_t)DESCALE(t4 + z1 + z3, sh_even);
            p[5 * st] = (int32_t)DESCALE(t5 + z2 + z4, sh_even);
            p[3 * st] = (int32_t)DESCALE(t6 + z2 + z3, sh_even);
            p[1 * st] = (int32_t)DESCALE(t7 + z1 + z4, sh_even);
        }
    }
}

typedef struct { uint8_t *p; size_t cap, n; uint64_t acc; int nbits; int overflow; } jo_out;
static void jo_putc(jo_out *o, int c) { if (o->n < o->cap) o->p[o->n] = (uint8_t)c; else o->overflow = 1; ++o->n; }
static void jo_put16(jo_out *o, int v) { jo_putc(o, v >> 8); jo_putc(o, v & 255); }
static void jo_emit(jo_out *o, uint32_t code, int len)
{
    /* jchuff.c emit_bits: MSB first, a 0xFF byte is followed by a stuffed 0x00 */
    if (!len) return;
    o->acc = (o->acc << len) | (code & ((1u << len) - 1));
    o->nbits += len;
    while (o->nbits >= 8) {
        const int c = (int)((o->acc >> (o->nbits - 8)) & 255);
        jo_putc(o, c);
        if (c == 0xFF) jo_putc(o, 0);
        o->nbits -= 8;
    }
}
static inline int jo_nbits(int v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }

EXPORT size_t jo_encode_bound(int w, int h)
{
    const size_t bw = ((size_t)w + 15) / 16 * 16, bh = ((size_t)h + 15) / 16 * 16;
    return 1024 + bw * bh * 3 / 64 * 416 / 1; /* headers + 2 x 208 bytes per block (worst case with stuffing) */
}

/* cv2.imwrite(path.jpg, bgr): baseline, 4:2:0 (sampling 0x22 = 4:2:0, 0x21 = 4:2:2, 0x11 = 4:4:4), standard Huffman tables, no restart markers.
 * Returns the file length or a negative code. */
EXPORT long jo_encode_bgr(const uint8_t *bgr, int w, int h, size_t pitch_bytes, int quality, int sampling, uint8_t *out, size_t cap)
{
    if (w <= 0 || h <= 0 || w > 65500 || h > 65500) return JO_E_UNSUPPORTED;
    const int hs = sampling >> 4, vs = sampling & 15;
    if (!((hs == 1 && vs == 1) || (hs == 2 && vs == 1) || (hs == 2 && vs == 2))) return JO_E_UNSUPPORTED;
    uint16_t q[2][64];
    jo_quant_tables(quality, q);
    jo_ehuff dch[2], ach[2];
    for (int t = 0; t < 2; ++t) {
        jo_make_ehuff(kDcBits[t] + 0, kDcVals, &dch[t]);
        jo_make_ehuff(kAcBits[t] + 0, kAcVals[t], &ach[t]);
    }
    jo_out O = {out, cap, 0, 0, 0, 0};
    /* jcmarker.c write_file_header / write_frame_header / write_scan_header */
    jo_put16(&O, 0xFFD8);
    jo_put16(&O, 0xFFE0); jo_put16(&O, 16);
    jo_putc(&O, 'J'); jo_putc(&O, 'F'); jo_putc(&O, 'I'); jo_putc(&O, 'F'); jo_putc(&O, 0);
    jo_putc(&O, 1); jo_putc(&O, 1); jo_putc(&O, 0); jo_put16(&O, 1); jo_put16(&O, 1); jo_putc(&O, 0); jo_putc(&O, 0);
    for (int t = 0; t < 2; ++t) {
        jo_put16(&O, 0xFFDB); jo_put16(&O, 67); jo_putc(&O, t);
        for (int k = 0; k < 64; ++k) jo_putc(&O, q[t][kNatural[k]]);
    }
    jo_put16(&O, 0xFFC0); jo_put16(&O, 17); jo_putc(&O, 8); jo_put16(&O, h); jo_put16(&O, w); jo_putc(&O, 3);
    jo_putc(&O, 1); jo_putc(&O, (hs << 4) | vs); jo_putc(&O, 0);
    jo_putc(&O, 2); jo_putc(&O, 0x11); jo_putc(&O, 1);
    jo_putc(&O, 3); jo_putc(&O, 0x11); jo_putc(&O, 1);
    for (int t = 0; t < 2; ++t) {
        jo_put16(&O, 0xFFC4); jo_put16(&O, 2 + 1 + 16 + 12); jo_putc(&O, t);
        for (int l = 1; l <= 16; ++l) jo_putc(&O, kDcBits[t][l]);
        for (int i = 0; i < 12; ++i) jo_putc(&O, kDcVals[i]);
        jo_put16(&O, 0xFFC4); jo_put16(&O, 2 + 1 + 16 + 162); jo_putc(&O, 0x10 | t);
        for (int l = 1; l <= 16; ++l) jo_putc(&O, kAcBits[t][l]);
        for (int i = 0; i < 162; ++i) jo_putc(&O, kAcVals[t][i]);
    }
    jo_put16(&O, 0xFFDA); jo_put16(&O, 12); jo_putc(&O, 3);
    jo_putc(&O, 1); jo_putc(&O, 0x00); jo_putc(&O, 2); jo_putc(&O, 0x11); jo_putc(&O, 3); jo_putc(&O, 0x11);
    jo_putc(&O, 0); jo_putc(&O, 63); jo_putc(&O, 0);

    /* sample planes: Y at full resolution, Cb / Cr downsampled; every read position is clamped to the image (jcsample.c
     * expand_right_edge, jcprepct.c expand_bottom_edge: edge pixels are replicated BEFORE the averaging) */
    const int mcuw = 8 * hs, mcuh = 8 * vs;
    const int mcux = (w + mcuw - 1) / mcuw, mcuy = (h + mcuh - 1) / mcuh;
    const int yw = mcux * mcuw, yh = mcuy * mcuh, cw = mcux * 8, ch = mcuy * 8;
    uint8_t *Y = (uint8_t *)malloc((size_t)yw * yh), *Cb = (uint8_t *)malloc((size_t)cw * ch), *Cr = (uint8_t *)malloc((size_t)cw * ch);
    uint8_t *fcb = (uint8_t *)malloc((size_t)yw * yh), *fcr = (uint8_t *)malloc((size_t)yw * yh);
    if (!Y || !Cb || !Cr || !fcb || !fcr) { free(Y); free(Cb); free(Cr); free(fcb); free(fcr); return JO_E_SPACE; }
    for (int y = 0; y < yh; ++y)
        for (int x = 0; x < yw; ++x) {
            const uint8_t *px = bgr + (size_t)(y < h ? y : h - 1) * pitch_bytes + (size_t)(x < w ? x : w - 1) * 3;
            const int64_t b = px[0], g = px[1], r = px[2];
            /* jccolor.c rgb_ycc_convert: FIX(x) = x * 65536 + 0.5, CBCR_OFFSET = 128 << 16, ONE_HALF = 32768 */
            Y[(size_t)y * yw + x] = (uint8_t)((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
            fcb[(size_t)y * yw + x] = (uint8_t)((-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16);
            fcr[(size_t)y * yw + x] = (uint8_t)((32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16);
        }
    const int dh = (h + vs - 1) / vs; /* the chroma components' real height: rows below it are COPIES of row dh - 1 (jcprepct.c pads the
                                         DOWNSAMPLED rows to a whole iMCU row; an odd image height is completed before the averaging) */
    for (int yo = 0; yo < ch; ++yo)
        for (int x = 0; x < cw; ++x) {
            const int y = yo < dh ? yo : dh - 1;
            if (hs == 2 && vs == 2) {
                /* h2v2_downsample: bias 1, 2, 1, 2, ... along the row */
                const size_t o = (size_t)(2 * y) * yw + 2 * x;
                const int bias = (x & 1) ? 2 : 1;
                Cb[(size_t)yo * cw + x] = (uint8_t)((fcb[o] + fcb[o + 1] + fcb[o + yw] + fcb[o + yw + 1] + bias) >> 2);
                Cr[(size_t)yo * cw + x] = (uint8_t)((fcr[o] + fcr[o + 1] + fcr[o + yw] + fcr[o + yw + 1] + bias) >> 2);
            } else if (hs == 2) {
                /* h2v1_downsample: bias 0, 1, 0, 1, ... */
                const size_t o = (size_t)y * yw + 2 * x;
                const int bias = x & 1;
                Cb[(size_t)yo * cw + x] = (uint8_t)((fcb[o] + fcb[o + 1] + bias) >> 1);
                Cr[(size_t)yo * cw + x] = (uint8_t)((fcr[o] + fcr[o + 1] + bias) >> 1);
            } else {
                Cb[(size_t)yo * cw + x] = fcb[(size_t)y * yw + x];
                Cr[(size_t)yo * cw + x] = fcr[(size_t)y * yw + x];
            }
        }
    free(fcb);
    free(fcr);
    /* jccoefct.c compress_data: blocks beyond the component's width_in_blocks / height_in_blocks inside the last MCU column /
     * row are dummies: AC = 0, DC = the (quantised) DC of the previous block of the MCU */
    const int ywb = (w + 7) / 8, yhb = (h + 7) / 8;          /* luma width_in_blocks / height_in_blocks */
    int last_dc[3] = {0, 0, 0};
    for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
            int16_t mcu[6][64];
            int nb = 0;
            for (int c = 0; c < 3; ++c) {
                const int bh_ = c == 0 ? vs : 1, bw_ = c == 0 ? hs : 1;
                const uint8_t *P = c == 0 ? Y : (c == 1 ? Cb : Cr);
                const int pw = c == 0 ? yw : cw;
                const int wb = c == 0 ? ywb : (((w + hs - 1) / hs) + 7) / 8, hb = c == 0 ? yhb : (((h + vs - 1) / vs) + 7) / 8;
                for (int by = 0; by < bh_; ++by)
                    for (int bx = 0; bx < bw_; ++bx, ++nb) {
                        const int X = mx * bw_ + bx, Yb = my * bh_ + by;
                        if (Yb >= hb) { /* a row of dummy blocks at the bottom: DC of the block before this ROW of the MCU */
                            memset(mcu[nb], 0, sizeof mcu[nb]);
                            mcu[nb][0] = mcu[nb - bx - 1][0];
                            continue;
                        }
                        if (X >= wb) { /* dummy block at the right edge */
                            memset(mcu[nb], 0, sizeof mcu[nb]);
                            mcu[nb][0] = mcu[nb - 1][0];
                            continue;
                        }
                        int32_t d[64];
                        for (int yy = 0; yy < 8; ++yy)
                            for (int xx = 0; xx < 8; ++xx) d[8 * yy + xx] = (int32_t)P[(size_t)(Yb * 8 + yy) * pw + X * 8 + xx] - 128;
                        jo_fdct_islow(d);
                        const uint16_t *qt = q[c ? 1 : 0];
                        for (int i = 0; i < 64; ++i) {
                            /* jcdctmgr.c quantize: divisor 8 q, round half away from zero */
                            const int32_t qv = (int32_t)qt[i] << 3;
                            int32_t t = d[i];
                            if (t < 0) { t = -t; t += qv >> 1; t = t >= qv ? t / qv : 0; t = -t; }
                            else { t += qv >> 1; t = t >= qv ? t / qv : 0; }
                            mcu[nb][i] = (int16_t)t;
                        }
                    }
            }
            /* jchuff.c encode_one_block */
            nb = 0;
            for (int c = 0; c < 3; ++c) {
                const int cnt = c == 0 ? hs * vs : 1;
                const jo_ehuff *dt = &dch[c ? 1 : 0], *at = &ach[c ? 1 : 0];
                for (int b = 0; b < cnt; ++b, ++nb) {
                    const int16_t *blk = mcu[nb];
                    int temp = blk[0] - last_dc[c], temp2 = temp;
                    last_dc[c] = blk[0];
                    if (temp < 0) { temp = -temp; --temp2; }
                    int nbits = jo_nbits(temp);
                    jo_emit(&O, dt->code[nbits], dt->len[nbits]);
                    if (nbits) jo_emit(&O, (uint32_t)temp2, nbits);
                    int r = 0;
                    for (int k = 1; k < 64; ++k) {
                        temp = blk[kNatural[k]];
                        if (temp == 0) { ++r; continue; }
                        while (r > 15) { jo_emit(&O, at->code[0xF0], at->len[0xF0]); r -= 16; }
                        temp2 = temp;
                        if (temp < 0) { temp = -temp; --temp2; }
                        nbits = jo_nbits(temp);
                        const int sym = (r << 4) + nbits;
                        jo_emit(&O, at->code[sym], at->len[sym]);
                        jo_emit(&O, (uint32_t)temp2, nbits);
                        r = 0;
                    }
                    if (r > 0) jo_emit(&O, at->code[0], at->len[0]);
                }
            }
        }
    jo_emit(&O, 0x7F, 7); /* flush_bits: fill the last byte with ones */
    O.acc = 0;
    O.nbits = 0;
    jo_put16(&O, 0xFFD9);
    free(Y); free(Cb); free(Cr);
    if (O.overflow) return JO_E_SPACE;
    return (long)O.n;
}
