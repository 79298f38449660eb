/*
 * bevoracle.c -- CPU ORACLE for the surround-BEV warping hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The shipped path is cameracalibration_amd/csrc (HIP).
 *
 * PARITY UNPINNED.  The arithmetic of the reference path lives in the un-vendored third-party module
 * opencv-python (README.md:14 of the reference: "opencv(>=3.4.2)", unpinned), which is not installable in this
 * image, and the reference has no tests or golden outputs.  Every function below is a
 * restatement of the PUBLISHED OpenCV algorithm for the call the reference makes, written from the documented
 * semantics of the classic fixed-point code paths (OpenCV 3.4.2 ... 4.10).  Each function cites the reference
 * call site (file:line under /root/reference) whose behaviour it restates.  tests/golden/ holds the hook
 * (make_goldens_with_cv2.py) that pins it against a real cv2 wherever one exists.
 *
 * Build: see oracle/Makefile   (gcc -O2 -ffp-contract=off: no FMA contraction -- OpenCV's x86-64 baseline
 * code has none, and the HIP side is built with the same flag so fp64/fp32 results are bit-comparable).
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* OpenCV interpolation constants (imgproc: INTER_BITS etc.). */
enum { Q_BITS = 5, Q_ONE = 32, Q_TAB2 = 1024, COEF_BITS = 15 };

/* ------------------------------------------------------------------------------------------------ */
/* rounding / saturation helpers (cvRound = round-half-to-even; saturate_cast<>)                      */
/* ------------------------------------------------------------------------------------------------ */
static inline int rne_d(double v)
{
    /* cvRound(double): cvtsd2si -> nearest-even; out-of-range / NaN give INT_MIN on x86. */
    if (!(v > -2147483648.5 && v < 2147483647.5)) return INT_MIN;
    return (int)lrint(v);
}
static inline int rne_f(float v)
{
    if (!(v > -2147483904.0f && v < 2147483520.0f)) return INT_MIN;
    return (int)lrintf(v);
}
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
static inline int16_t sat_s16(int v) { return (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static inline uint16_t sat_u16(int v) { return (uint16_t)(v < 0 ? 0 : v > 65535 ? 65535 : v); }

ORC_API void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
ORC_API int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------ */
/* A.1  cv2.fisheye.initUndistortRectifyMap(K, D, I, K', size, CV_16SC2)                              */
/*      reference: surroundBEV.py:98-103, intrinsicCalib.py:98-103, Tools/undistort.py:50-52          */
/* ------------------------------------------------------------------------------------------------ */
/* iR = inv(K' * I).  OpenCV uses an SVD inverse; for the skew-free camera matrix the reference builds
 * (surroundBEV.py:90-96) that equals this closed form to a few ulp (documented deviation). */
ORC_API void orc_newcam_inverse(const double Knew[9], double iR[9])
{
    double fx = Knew[0], fy = Knew[4], cx = Knew[2], cy = Knew[5];
    iR[0] = 1.0 / fx; iR[1] = 0.0;      iR[2] = -cx / fx;
    iR[3] = 0.0;      iR[4] = 1.0 / fy; iR[5] = -cy / fy;
    iR[6] = 0.0;      iR[7] = 0.0;      iR[8] = 1.0;
}

ORC_API void orc_fisheye_undistort_map(const double K[9], const double D[4], const double iR[9],
                                       int width, int height, int16_t *map1, uint16_t *map2)
{
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5]; /* skew K[1] is ignored by OpenCV here */
    const double k0 = D[0], k1 = D[1], k2 = D[2], k3 = D[3];
#pragma omp parallel for schedule(static)
    for (int i = 0; i < height; ++i) {
        double _x = i * iR[1] + iR[2], _y = i * iR[4] + iR[5], _w = i * iR[7] + iR[8];
        int16_t *m1 = map1 + (size_t)i * width * 2;
        uint16_t *m2 = map2 + (size_t)i * width;
        for (int j = 0; j < width; ++j) {
            double u, v;
            if (_w <= 0) {
                u = (_x > 0) ? -INFINITY : INFINITY;
                v = (_y > 0) ? -INFINITY : INFINITY;
            } else {
                double x = _x / _w, y = _y / _w;
                double r = sqrt(x * x + y * y);
                double theta = atan(r);
                double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
                double theta_d = theta * (1 + k0 * t2 + k1 * t4 + k2 * t6 + k3 * t8);
                double scale = (r == 0) ? 1.0 : theta_d / r;
                u = fx * x * scale + cx;
                v = fy * y * scale + cy;
            }
            int iu = rne_d(u * Q_ONE), iv = rne_d(v * Q_ONE);
            m1[j * 2 + 0] = (int16_t)(iu >> Q_BITS);
            m1[j * 2 + 1] = (int16_t)(iv >> Q_BITS);
            m2[j] = (uint16_t)((iv & (Q_ONE - 1)) * Q_ONE + (iu & (Q_ONE - 1)));
            _x += iR[0]; _y += iR[3]; _w += iR[6]; /* accumulated, not j*iR: order matters in fp64 */
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* A.1b cv2.initUndistortRectifyMap(K, D, I, K', size, CV_16SC2)  -- the pinhole ("normal") camera model   */
/*      reference: Normal._get_undistort_maps intrinsicCalib.py:158-163                                    */
/* ------------------------------------------------------------------------------------------------ */
/* iR = inv(K' * I) with DECOMP_LU, which for a 3x3 matrix is the cofactor formula (orc_invert3x3, defined below).
 * D holds k1 k2 p1 p2 k3 [k4 k5 k6]; thin-prism and tilt terms are zero in the reference's calibration output
 * (the tilt matrix is the identity, so xd, yd pass through unchanged).  Scalar C++ path of OpenCV (no FMA);
 * builds with an AVX2/FMA dispatch of this loop may differ in the last ulp (version-sensitive, see header). */
ORC_API int orc_invert3x3(const double m[9], double t[9]);
ORC_API void orc_pinhole_undistort_map(const double K[9], const double D[8], const double iR[9], int width, int height,
                                       int16_t *map1, uint16_t *map2)
{
    const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4], k4 = D[5], k5 = D[6], k6 = D[7];
#pragma omp parallel for schedule(static)
    for (int i = 0; i < height; ++i) {
        double _x = i * iR[1] + iR[2], _y = i * iR[4] + iR[5], _w = i * iR[7] + iR[8];
        int16_t *m1 = map1 + (size_t)i * width * 2;
        uint16_t *m2 = map2 + (size_t)i * width;
        for (int j = 0; j < width; ++j, _x += iR[0], _y += iR[3], _w += iR[6]) {
            double w = 1. / _w, x = _x * w, y = _y * w;
            double x2 = x * x, y2 = y * y;
            double r2 = x2 + y2, _2xy = 2 * x * y;
            double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
            double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
            double u = fx * xd + u0, v = fy * yd + v0;
            int iu = rne_d(u * Q_ONE), iv = rne_d(v * Q_ONE);
            m1[j * 2 + 0] = (int16_t)(iu >> Q_BITS);
            m1[j * 2 + 1] = (int16_t)(iv >> Q_BITS);
            m2[j] = (uint16_t)((iv & (Q_ONE - 1)) * Q_ONE + (iu & (Q_ONE - 1)));
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* A.2  cv2.warpPerspective coordinate generation (inverse map, Q5 quantisation)                      */
/*      reference: surroundBEV.py:113-114, extrinsicCalib.py:166-169                                  */
/* ------------------------------------------------------------------------------------------------ */
/* cv::invert of a 3x3 CV_64F matrix (cofactor form). Returns 0 if singular (OpenCV then zero-fills). */
ORC_API int orc_invert3x3(const double m[9], double t[9])
{
    double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
               m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (d == 0.0) {
        memset(t, 0, 9 * sizeof(double));
        return 0;
    }
    d = 1.0 / d;
    t[0] = (m[4] * m[8] - m[5] * m[7]) * d;
    t[1] = (m[2] * m[7] - m[1] * m[8]) * d;
    t[2] = (m[1] * m[5] - m[2] * m[4]) * d;
    t[3] = (m[5] * m[6] - m[3] * m[8]) * d;
    t[4] = (m[0] * m[8] - m[2] * m[6]) * d;
    t[5] = (m[2] * m[3] - m[0] * m[5]) * d;
    t[6] = (m[3] * m[7] - m[4] * m[6]) * d;
    t[7] = (m[1] * m[6] - m[0] * m[7]) * d;
    t[8] = (m[0] * m[4] - m[1] * m[3]) * d;
    return 1;
}

/* Destination is walked in column blocks of min(64, width) pixels (OpenCV: BLOCK_SZ 32 -> 64x16 tiles); the
 * block origin x0 enters the fp64 expression, so the association (X0 + M0*x1) is reproduced here. */
static int g_variant[4];   /* defined with orc_set_variant below */
ORC_API void orc_perspective_coords(const double M[9], int dw, int dh, int16_t *xy, uint16_t *alpha)
{
    int bh0 = dh < 16 ? dh : 16;
    int bw0 = 1024 / bh0;
    if (bw0 > dw) bw0 = dw;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        for (int x0 = 0; x0 < dw; x0 += bw0) {
            int bw = dw - x0 < bw0 ? dw - x0 : bw0;
            double X0 = M[0] * x0 + M[1] * y + M[2];
            double Y0 = M[3] * x0 + M[4] * y + M[5];
            double W0 = M[6] * x0 + M[7] * y + M[8];
            for (int x1 = 0; x1 < bw; ++x1) {
                double W = W0 + M[6] * x1;
                W = W ? Q_ONE / W : 0;
                double fX = (X0 + M[0] * x1) * W, fY = (Y0 + M[3] * x1) * W;
                fX = fmax((double)INT_MIN, fmin((double)INT_MAX, fX));
                fY = fmax((double)INT_MIN, fmin((double)INT_MAX, fY));
                int X = rne_d(fX), Y = rne_d(fY);
                size_t o = (size_t)y * dw + x0 + x1;
                xy[o * 2 + 0] = sat_s16(X >> Q_BITS);
                xy[o * 2 + 1] = sat_s16(Y >> Q_BITS);
                alpha[o] = (uint16_t)((Y & (Q_ONE - 1)) * Q_ONE + (X & (Q_ONE - 1)));
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* A.3  bilinear remap of 16S / 16U sources with float32 weights (the LUT quirk)                      */
/*      reference: Camera.get_bev_maps surroundBEV.py:105-108 (warpPerspective over the CV_16SC2 /    */
/*      CV_16UC1 undistort maps)                                                                      */
/* ------------------------------------------------------------------------------------------------ */
static inline void f32_weights(unsigned code, float w[4])
{
    /* initInterTab2D(INTER_LINEAR, fixpt=false): 1-D tab {1 - t, t}, t = i * (1/32); products are exact. */
    float fx = (float)(code & 31) * (1.f / 32), fy = (float)((code >> 5) & 31) * (1.f / 32);
    float ax = 1.f - fx, ay = 1.f - fy;
    w[0] = ay * ax; w[1] = ay * fx; w[2] = fy * ax; w[3] = fy * fx;
}

#define DEFINE_REMAP_F32(NAME, T, CAST)                                                                        \
    ORC_API void NAME(const T *src, int sw, int sh, int cn, const int16_t *xy, const uint16_t *alpha, int dw,   \
                      int dh, T *dst)                                                                           \
    {                                                                                                           \
        _Pragma("omp parallel for schedule(static)") for (int dy = 0; dy < dh; ++dy)                            \
        {                                                                                                       \
            for (int dx = 0; dx < dw; ++dx) {                                                                   \
                size_t o = (size_t)dy * dw + dx;                                                                \
                int sx = xy[o * 2], sy = xy[o * 2 + 1];                                                         \
                float w[4];                                                                                     \
                f32_weights(alpha[o] & (Q_TAB2 - 1), w);                                                        \
                T *D = dst + o * cn;                                                                            \
                if ((unsigned)sx < (unsigned)(sw > 1 ? sw - 1 : 0) &&                                           \
                    (unsigned)sy < (unsigned)(sh > 1 ? sh - 1 : 0)) {                                           \
                    const T *S = src + ((size_t)sy * sw + sx) * cn;                                             \
                    size_t st = (size_t)sw * cn;                                                                \
                    for (int k = 0; k < cn; ++k)                                                                \
                        D[k] = CAST(rne_f(S[k] * w[0] + S[k + cn] * w[1] + S[k + st] * w[2] +                   \
                                          S[k + st + cn] * w[3]));                                              \
                } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {                                  \
                    for (int k = 0; k < cn; ++k) D[k] = 0;                                                      \
                } else {                                                                                        \
                    int x0ok = sx >= 0 && sx < sw, x1ok = sx + 1 >= 0 && sx + 1 < sw;                           \
                    int y0ok = sy >= 0 && sy < sh, y1ok = sy + 1 >= 0 && sy + 1 < sh;                           \
                    for (int k = 0; k < cn; ++k) {                                                              \
                        T v0 = (x0ok && y0ok) ? src[((size_t)sy * sw + sx) * cn + k] : 0;                       \
                        T v1 = (x1ok && y0ok) ? src[((size_t)sy * sw + sx + 1) * cn + k] : 0;                   \
                        T v2 = (x0ok && y1ok) ? src[((size_t)(sy + 1) * sw + sx) * cn + k] : 0;                 \
                        T v3 = (x1ok && y1ok) ? src[((size_t)(sy + 1) * sw + sx + 1) * cn + k] : 0;             \
                        D[k] = CAST(rne_f(v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3]));                      \
                    }                                                                                           \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
    }
DEFINE_REMAP_F32(orc_remap_s16_f32, int16_t, sat_s16)
DEFINE_REMAP_F32(orc_remap_u16_f32, uint16_t, sat_u16)

/* ------------------------------------------------------------------------------------------------ */
/* A.4  cv2.remap(src 8U, map1 16SC2, map2 16UC1, INTER_LINEAR), BORDER_CONSTANT 0                    */
/*      reference: surroundBEV.py:110-111,116-117; intrinsicCalib.py:193-195; Tools/undistort.py:66   */
/* ------------------------------------------------------------------------------------------------ */
static int16_t g_wtab[Q_TAB2][4];
static int g_wtab_ready;
static void build_fixed_weights(void)
{
    /* initInterTab2D(INTER_LINEAR, fixpt=true): saturate_cast<short>(v * 32768), then the sum fix-up that
     * moves the residue onto the largest (deficit) or smallest (excess) tap; only code 0 needs it. */
    for (int fy = 0; fy < Q_ONE; ++fy)
        for (int fx = 0; fx < Q_ONE; ++fx) {
            int16_t *t = g_wtab[fy * Q_ONE + fx];
            int w[4] = {(Q_ONE - fx) * (Q_ONE - fy) * 32, fx * (Q_ONE - fy) * 32, (Q_ONE - fx) * fy * 32,
                        fx * fy * 32};
            int sum = 0;
            for (int k = 0; k < 4; ++k) sum += t[k] = sat_s16(w[k]);
            if (sum != (1 << COEF_BITS)) t[3] = (int16_t)(t[3] - (sum - (1 << COEF_BITS)));
        }
    g_wtab_ready = 1;
}

/* the sum of the four taps times their 15-bit weights, rounded to the pixel value: (S + 2^14) >> 15 in the classic kernels (half up);
 * variant 3 = 1: the same exact quotient S / 2^15 rounded half to even, as a float kernel that ends in cvRound does (the weights are
 * 5-bit x 5-bit products times 32, the taps 8 bit: a float32 evaluation of the same sum is exact in any order) */
static inline int remap_round15(int S)
{
    int r = (S + (1 << (COEF_BITS - 1))) >> COEF_BITS;
    if (g_variant[3] && (S & ((1 << COEF_BITS) - 1)) == (1 << (COEF_BITS - 1)) && (r & 1)) --r;
    return r;
}

ORC_API void orc_remap_u8(const uint8_t *src, int sw, int sh, int cn, const int16_t *xy, const uint16_t *alpha,
                          int dw, int dh, uint8_t *dst)
{
    if (!g_wtab_ready) {
#pragma omp critical
        if (!g_wtab_ready) build_fixed_weights();
    }
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < dh; ++dy) {
        for (int dx = 0; dx < dw; ++dx) {
            size_t o = (size_t)dy * dw + dx;
            int sx = xy[o * 2], sy = xy[o * 2 + 1];
            const int16_t *w = g_wtab[alpha[o] & (Q_TAB2 - 1)];
            uint8_t *D = dst + o * cn;
            if ((unsigned)sx < (unsigned)(sw > 1 ? sw - 1 : 0) && (unsigned)sy < (unsigned)(sh > 1 ? sh - 1 : 0)) {
                const uint8_t *S = src + ((size_t)sy * sw + sx) * cn;
                size_t st = (size_t)sw * cn;
                for (int k = 0; k < cn; ++k)
                    D[k] = sat_u8(remap_round15(S[k] * w[0] + S[k + cn] * w[1] + S[k + st] * w[2] + S[k + st + cn] * w[3]));
            } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
                for (int k = 0; k < cn; ++k) D[k] = 0;
            } else {
                int x0ok = sx >= 0 && sx < sw, x1ok = sx + 1 >= 0 && sx + 1 < sw;
                int y0ok = sy >= 0 && sy < sh, y1ok = sy + 1 >= 0 && sy + 1 < sh;
                for (int k = 0; k < cn; ++k) {
                    int v0 = (x0ok && y0ok) ? src[((size_t)sy * sw + sx) * cn + k] : 0;
                    int v1 = (x1ok && y0ok) ? src[((size_t)sy * sw + sx + 1) * cn + k] : 0;
                    int v2 = (x0ok && y1ok) ? src[((size_t)(sy + 1) * sw + sx) * cn + k] : 0;
                    int v3 = (x1ok && y1ok) ? src[((size_t)(sy + 1) * sw + sx + 1) * cn + k] : 0;
                    D[k] = sat_u8(remap_round15(v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3]));
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* A.4b cv2.warpPerspective(INTER_LINEAR, BORDER_CONSTANT 0) in the style of OpenCV >= 4.11: float32   */
/*      linear kernels for 8U / 16U images with 1, 3 or 4 channels.  A FAMILY of candidates, not a     */
/*      restatement of one known operation order (the library is not at hand; its SIMD body and scalar */
/*      tail differ in their use of fused multiply-adds): `mode` picks the member, bit by bit --       */
/*        1  the float family (0 = the classic path A.2 - A.4)                                         */
/*        2  source position: fma(M0, x, fma(M1, y, M2)) instead of (x M0 + y M1) + M2                 */
/*        4  interpolation with fused multiply-adds                                                    */
/*        8  interpolation (1 - t) a + t b instead of a + t (b - a)                                    */
/*       16  numerators and denominator in double, the quotient rounded to float                       */
/*       32  multiply by 1 / w instead of dividing by w                                                */
/*      reference call sites: extrinsicCalib.py:166-169 (8UC3), surroundBEV.py:105-108 (16UC1 map).    */
/* ------------------------------------------------------------------------------------------------ */
static inline float family_mix(float lo, float hi, float t, int mode)
{
    if (mode & 8) {
        float keep = 1.f - t;
        if (mode & 4) return fmaf(t, hi, keep * lo);
        return keep * lo + t * hi;
    }
    if (mode & 4) return fmaf(t, hi - lo, lo);
    return lo + t * (hi - lo);
}
static inline void family_position(const double M[9], int dx, int dy, int mode, float *px, float *py)
{
    if (mode & 16) {
        double num_x = M[0] * dx + M[1] * dy + M[2], num_y = M[3] * dx + M[4] * dy + M[5], den = M[6] * dx + M[7] * dy + M[8];
        *px = (float)(num_x / den);
        *py = (float)(num_y / den);
        return;
    }
    float m[9], fx = (float)dx, fy = (float)dy, num_x, num_y, den;
    for (int i = 0; i < 9; ++i) m[i] = (float)M[i];
    if (mode & 2) {
        num_x = fmaf(m[0], fx, fmaf(m[1], fy, m[2]));
        num_y = fmaf(m[3], fx, fmaf(m[4], fy, m[5]));
        den = fmaf(m[6], fx, fmaf(m[7], fy, m[8]));
    } else {
        num_x = (fx * m[0] + fy * m[1]) + m[2];
        num_y = (fx * m[3] + fy * m[4]) + m[5];
        den = (fx * m[6] + fy * m[7]) + m[8];
    }
    if (mode & 32) {
        float inv = 1.f / den;
        *px = num_x * inv;
        *py = num_y * inv;
    } else {
        *px = num_x / den;
        *py = num_y / den;
    }
}
#define DEFINE_WARP_F32(NAME, T, CAST)                                                                                   \
    ORC_API void NAME(const T *src, int sw, int sh, int cn, const double M[9], int mode, int dw, int dh, T *dst)          \
    {                                                                                                                     \
        _Pragma("omp parallel for schedule(static)") for (int dy = 0; dy < dh; ++dy)                                      \
        {                                                                                                                 \
            for (int dx = 0; dx < dw; ++dx) {                                                                             \
                T *D = dst + ((size_t)dy * dw + dx) * cn;                                                                 \
                float px, py;                                                                                             \
                family_position(M, dx, dy, mode, &px, &py);                                                               \
                for (int k = 0; k < cn; ++k) D[k] = 0;                                                                    \
                if (!(px > -2.f && px < (float)sw + 1.f && py > -2.f && py < (float)sh + 1.f)) continue;                  \
                float bx = floorf(px), by = floorf(py), tx = px - bx, ty = py - by;                                       \
                int ix = (int)bx, iy = (int)by;                                                                           \
                for (int k = 0; k < cn; ++k) {                                                                            \
                    float tap[2][2];                                                                                      \
                    for (int r = 0; r < 2; ++r)                                                                           \
                        for (int c = 0; c < 2; ++c) {                                                                     \
                            int xx = ix + c, yy = iy + r;                                                                 \
                            tap[r][c] = (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? (float)src[((size_t)yy * sw + xx) * cn + k] : 0.f; \
                        }                                                                                                 \
                    float top = family_mix(tap[0][0], tap[0][1], tx, mode), bot = family_mix(tap[1][0], tap[1][1], tx, mode); \
                    D[k] = CAST(rne_f(family_mix(top, bot, ty, mode)));                                                   \
                }                                                                                                         \
            }                                                                                                             \
        }                                                                                                                 \
    }
DEFINE_WARP_F32(orc_warp_f32_u8, uint8_t, sat_u8)
DEFINE_WARP_F32(orc_warp_f32_u16, uint16_t, sat_u16)

/* ------------------------------------------------------------------------------------------------ */
/* A.5  cv2.fillPoly(mask, [pts], 255)  (LINE_8, shift 0)   reference: surroundBEV.py:156-159,231-234 */
/* ------------------------------------------------------------------------------------------------ */
/* cv::clipLine on 64-bit points. Returns 1 when some part of the segment is inside [0,w)x[0,h). */
static int clip_segment(int w, int h, int64_t *px1, int64_t *py1, int64_t *px2, int64_t *py2)
{
    int64_t x1 = *px1, y1 = *py1, x2 = *px2, y2 = *py2;
    int64_t right = w - 1, bottom = h - 1;
    if (w <= 0 || h <= 0) return 0;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        int64_t a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (int64_t)((double)(a - y1) * (x2 - x1) / (y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (int64_t)((double)(a - y2) * (x2 - x1) / (y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (int64_t)((double)(a - x1) * (y2 - y1) / (x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (int64_t)((double)(a - x2) * (y2 - y1) / (x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    *px1 = x1; *py1 = y1; *px2 = x2; *py2 = y2;
    return (c1 | c2) == 0;
}

/* 8-connected line, endpoints normalised left-to-right (cv::LineIterator with leftToRight=true). */
static void draw_line8(uint8_t *img, int w, int h, int64_t x1, int64_t y1, int64_t x2, int64_t y2, uint8_t color)
{
    if ((uint64_t)x1 >= (uint64_t)w || (uint64_t)x2 >= (uint64_t)w || (uint64_t)y1 >= (uint64_t)h ||
        (uint64_t)y2 >= (uint64_t)h) {
        if (!clip_segment(w, h, &x1, &y1, &x2, &y2)) return;
    }
    int64_t dx = x2 - x1, dy = y2 - y1;
    int64_t x = x1, y = y1;
    if (dx < 0) { dx = -dx; dy = -dy; x = x2; y = y2; }
    int ystep = dy < 0 ? -1 : 1;
    if (dy < 0) dy = -dy;
    int steep = dy > dx;
    int64_t major = steep ? dy : dx, minor = steep ? dx : dy;
    int64_t err = major - 2 * minor;
    for (int64_t i = 0; i <= major; ++i) {
        img[(size_t)y * w + x] = color;
        int neg = err < 0;
        err += -2 * minor + (neg ? 2 * major : 0);
        if (steep) { y += ystep; if (neg) x += 1; }
        else       { x += 1;     if (neg) y += ystep; }
    }
}

/* OpenCV-version-sensitive choices of this restatement, selectable so that a golden file from a real cv2 can flip them
 * (tests/golden/README.md).  Keys as in include/bevwarp.h (BEVW_COMPAT_*):
 *   0 fillPoly edge rule   : 1 = OpenCV >= 4.5.2 (edges from the clipped end points, x + 1/2 pixel, both span ends floored),
 *                            0 = OpenCV 2.4 .. 4.5.1 (edges from the raw vertices, left span end rounded up, right end floored)
 *   1 addWeighted work type: 1 = CV_64F (arithm_op picks the scalar's depth), 0 = CV_32F (float(ch) * float(k))
 *   2 warpPerspective      : 0 = the classic kernels of OpenCV 2.4 .. 4.10 (A.2 - A.4), odd = a member of the float32 family below
 *                            (orc_warp_f32_*: candidates for OpenCV >= 4.11's linear kernels, decided by the implementation probes)
 *   3 remap tie rule       : 0 = (S + 512) >> 10, half up (classic), 1 = half to even (a float kernel ending in cvRound)
 * g_variant is declared before its first use further up in this file. */
static int g_variant[4] = {1, 1, 0, 0};   /* (tentative definition above, initialised here) */
ORC_API void orc_set_variant(int key, int value) { if (key >= 0 && key < 4) g_variant[key] = value; }
ORC_API int orc_get_variant(int key) { return (key >= 0 && key < 4) ? g_variant[key] : -1; }

typedef struct { int y0, y1; int64_t x, dx; } orc_edge;

static int edge_less(const orc_edge *a, const orc_edge *b)
{
    if (a->y0 != b->y0) return a->y0 < b->y0;
    if (a->x != b->x) return a->x < b->x;
    return a->dx < b->dx;
}

ORC_API void orc_fill_poly(uint8_t *img, int w, int h, const int32_t *pts, int npts, uint8_t color)
{
    enum { XY_SHIFT = 16, MAXE = 64 };
    const int64_t XY_HALF = 1 << (XY_SHIFT - 1);
    orc_edge edges[MAXE];
    int ne = 0;
    if (npts <= 0 || npts > MAXE) return;
    /* ---- collect edges, drawing each boundary segment (CollectPolyEdges, OpenCV >= 4.5.2 form) ---- */
    int64_t p0x = (int64_t)pts[(npts - 1) * 2] << XY_SHIFT, p0y = pts[(npts - 1) * 2 + 1];
    for (int i = 0; i < npts; ++i) {
        int64_t p1x = (int64_t)pts[i * 2] << XY_SHIFT, p1y = pts[i * 2 + 1];
        int64_t c0x = p0x, c0y = p0y, c1x = p1x, c1y = p1y;
        int64_t t0x = (p0x + XY_HALF) >> XY_SHIFT, t0y = p0y, t1x = (p1x + XY_HALF) >> XY_SHIFT, t1y = p1y;
        draw_line8(img, w, h, t0x, t0y, t1x, t1y, color);
        if (!g_variant[0]) {
            /* CollectPolyEdges before 4.5.2: the edge runs between the raw vertices, no half-pixel offset */
        } else if ((uint64_t)t0x >= (uint64_t)w || (uint64_t)t1x >= (uint64_t)w || (uint64_t)t0y >= (uint64_t)h ||
            (uint64_t)t1y >= (uint64_t)h) {
            clip_segment(w, h, &t0x, &t0y, &t1x, &t1y);
            if (t0y != t1y) {
                c0y = t0y; c1y = t1y;
                c0x = t0x << XY_SHIFT; c1x = t1x << XY_SHIFT;
            }
        } else {
            c0x += XY_HALF; c1x += XY_HALF;
        }
        if (p0y != p1y) {
            orc_edge e;
            e.dx = (c1x - c0x) / (c1y - c0y);
            if (p0y < p1y) { e.y0 = (int)p0y; e.y1 = (int)p1y; e.x = c0x + (p0y - c0y) * e.dx; }
            else           { e.y0 = (int)p1y; e.y1 = (int)p0y; e.x = c1x + (p1y - c1y) * e.dx; }
            edges[ne++] = e;
        }
        p0x = p1x; p0y = p1y;
    }
    if (ne < 2) return;
    /* ---- scan conversion (FillEdgeCollection) ---- */
    int y_max = INT_MIN, y_min = INT_MAX;
    int64_t x_max = -1, x_min = INT64_MAX;
    for (int i = 0; i < ne; ++i) {
        int64_t x1 = edges[i].x + (int64_t)(edges[i].y1 - edges[i].y0) * edges[i].dx;
        if (edges[i].y0 < y_min) y_min = edges[i].y0;
        if (edges[i].y1 > y_max) y_max = edges[i].y1;
        if (edges[i].x < x_min) x_min = edges[i].x;
        if (edges[i].x > x_max) x_max = edges[i].x;
        if (x1 < x_min) x_min = x1;
        if (x1 > x_max) x_max = x1;
    }
    if (y_max < 0 || y_min >= h || x_max < 0 || x_min >= ((int64_t)w << XY_SHIFT)) return;
    for (int i = 1; i < ne; ++i) { /* insertion sort by (y0, x, dx) */
        orc_edge k = edges[i];
        int j = i - 1;
        while (j >= 0 && edge_less(&k, &edges[j])) { edges[j + 1] = edges[j]; --j; }
        edges[j + 1] = k;
    }
    int active[MAXE], na = 0, next = 0;
    if (y_max > h) y_max = h;
    for (int y = edges[0].y0; y < y_max; ++y) {
        /* drop finished edges, then merge edges starting on this row into the x-ordered active list */
        int merged[MAXE], nm = 0, ai = 0;
        int kept[MAXE], nk = 0;
        for (int i = 0; i < na; ++i)
            if (edges[active[i]].y1 != y) kept[nk++] = active[i];
        while (ai < nk || (next < ne && edges[next].y0 == y)) {
            int take_new = !(ai < nk) || ((next < ne && edges[next].y0 == y) && !(edges[kept[ai]].x < edges[next].x));
            merged[nm++] = take_new ? next++ : kept[ai++];
        }
        for (int i = 0; i + 1 < nm; i += 2) {
            orc_edge *a = &edges[merged[i]], *b = &edges[merged[i + 1]];
            if (y >= 0) {
                int x1, x2;
                const int64_t up = g_variant[0] ? 0 : ((int64_t)1 << XY_SHIFT) - 1;   /* before 4.5.2: left end rounded up */
                if (a->x > b->x) { x1 = (int)((b->x + up) >> XY_SHIFT); x2 = (int)(a->x >> XY_SHIFT); }
                else             { x1 = (int)((a->x + up) >> XY_SHIFT); x2 = (int)(b->x >> XY_SHIFT); }
                if (x1 < w && x2 >= 0) {
                    if (x1 < 0) x1 = 0;
                    if (x2 >= w) x2 = w - 1;
                    for (int x = x1; x <= x2; ++x) img[(size_t)y * w + x] = color;
                }
            }
            a->x += a->dx;
            b->x += b->dx;
        }
        /* re-order by the advanced x (stable bubble sort, strict >) */
        na = nm;
        memcpy(active, merged, sizeof(int) * nm);
        for (int pass = 0; pass < na; ++pass) {
            int swapped = 0;
            for (int i = 0; i + 1 < na; ++i)
                if (edges[active[i]].x > edges[active[i + 1]].x) {
                    int t = active[i]; active[i] = active[i + 1]; active[i + 1] = t;
                    swapped = 1;
                }
            if (!swapped) break;
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* A.6  blend weights  (cv2.pointPolygonTest on a 2-point contour, measureDist=True)                  */
/*      reference: BlendMask.get_blend_mask surroundBEV.py:270-277                                    */
/* ------------------------------------------------------------------------------------------------ */
static double segment_distance(const int32_t line[4], double px, double py)
{
    /* contour = {P0, P1}; the traversal visits edge P1->P0 then P0->P1 (both the same segment). */
    double min_num = 3.402823466e+38 /* FLT_MAX */, min_den = 1;
    float vx = (float)line[2], vy = (float)line[3];
    float ptx = (float)px, pty = (float)py;
    for (int i = 0; i < 2; ++i) {
        float v0x = vx, v0y = vy;
        vx = (float)line[i * 2]; vy = (float)line[i * 2 + 1];
        double dx = vx - v0x, dy = vy - v0y;
        double dx1 = ptx - v0x, dy1 = pty - v0y;
        double dx2 = ptx - vx, dy2 = pty - vy;
        double num, den = 1;
        if (dx1 * dx + dy1 * dy <= 0) num = dx1 * dx1 + dy1 * dy1;
        else if (dx2 * dx + dy2 * dy >= 0) num = dx2 * dx2 + dy2 * dy2;
        else { num = dy1 * dx - dx1 * dy; num *= num; den = dx * dx + dy * dy; }
        if (num * min_den < min_num * den) {
            min_num = num; min_den = den;
            if (min_num == 0) break;
        }
    }
    return sqrt(min_num / min_den); /* sign dropped: the caller squares it */
}

/* |cv2.pointPolygonTest({P0, P1}, (px, py), True)| -- exported for tests that drive the primitive one point at a time */
ORC_API double orc_segment_distance(const int32_t line[4], double px, double py) { return segment_distance(line, px, py); }

/* maskA[y,x] = uint8(dA**2 / (dA**2 + dB**2 + 1e-6) * 255) on every pixel where (maskA & maskB) != 0.
 * CPython evaluates d**2 with libm pow(); pow(d, 2.0) is used here for the same reason. */
ORC_API void orc_blend_mask(uint8_t *maskA, const uint8_t *maskB, int w, int h, const int32_t lineA[4],
                            const int32_t lineB[4])
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            size_t o = (size_t)y * w + x;
            if ((maskA[o] & maskB[o]) == 0) continue;
            double dA = segment_distance(lineA, x, y), dB = segment_distance(lineB, x, y);
            double a2 = pow(dA, 2.0), b2 = pow(dB, 2.0);
            double v = a2 / (a2 + b2 + 1e-6) * 255;
            maskA[o] = (uint8_t)v; /* numpy float64 -> uint8 store: C truncation */
        }
}

/* ------------------------------------------------------------------------------------------------ */
/* A.7  luminance balance pieces: cvtColor BGR<->HSV (8-bit, H in [0,180)), V shift                   */
/*      reference: luminance_balance surroundBEV.py:57-79                                             */
/* ------------------------------------------------------------------------------------------------ */
static int g_sdiv[256], g_hdiv[256];
static int g_hsv_ready;
static void build_hsv_tables(void)
{
    g_sdiv[0] = g_hdiv[0] = 0;
    for (int i = 1; i < 256; ++i) {
        g_sdiv[i] = rne_d((255 << 12) / (1. * i));
        g_hdiv[i] = rne_d((180 << 12) / (6. * i));
    }
    g_hsv_ready = 1;
}
ORC_API void orc_hsv_tables(int32_t *sdiv, int32_t *hdiv)
{
    if (!g_hsv_ready) build_hsv_tables();
    memcpy(sdiv, g_sdiv, sizeof g_sdiv);
    memcpy(hdiv, g_hdiv, sizeof g_hdiv);
}

static inline void bgr2hsv_px(const uint8_t *s, uint8_t *d)
{
    int b = s[0], g = s[1], r = s[2];
    int v = b > g ? b : g; v = v > r ? v : r;
    int vmin = b < g ? b : g; vmin = vmin < r ? vmin : r;
    int diff = v - vmin;
    int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    int sat = (diff * g_sdiv[v] + (1 << 11)) >> 12;
    int hue = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    hue = (hue * g_hdiv[diff] + (1 << 11)) >> 12;
    hue += hue < 0 ? 180 : 0;
    d[0] = sat_u8(hue); d[1] = (uint8_t)sat; d[2] = (uint8_t)v;
}

static inline void hsv2bgr_px(const uint8_t *s, uint8_t *d)
{
    static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    const float hscale = 6.f / 180.f;
    float h = (float)s[0], sat = s[1] * (1.f / 255.f), v = s[2] * (1.f / 255.f);
    float b, g, r;
    if (sat == 0) b = g = r = v;
    else {
        float tab[4];
        h *= hscale;
        h = fmodf(h, 6.f);
        int sector = (int)floorf(h);
        h -= sector;
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        tab[0] = v;
        tab[1] = v * (1.f - sat);
        tab[2] = v * (1.f - sat * h);
        tab[3] = v * (1.f - sat * (1.f - h));
        b = tab[sector_data[sector][0]];
        g = tab[sector_data[sector][1]];
        r = tab[sector_data[sector][2]];
    }
    d[0] = sat_u8(rne_f(b * 255.f)); d[1] = sat_u8(rne_f(g * 255.f)); d[2] = sat_u8(rne_f(r * 255.f));
}

ORC_API void orc_bgr2hsv(const uint8_t *src, size_t npx, uint8_t *dst)
{
    if (!g_hsv_ready) build_hsv_tables();
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < npx; ++i) bgr2hsv_px(src + i * 3, dst + i * 3);
}
ORC_API void orc_hsv2bgr(const uint8_t *src, size_t npx, uint8_t *dst)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < npx; ++i) hsv2bgr_px(src + i * 3, dst + i * 3);
}
/* sum of the V plane (= sum of max(B,G,R)); np.mean(v) = sum / N exactly in fp64 (surroundBEV.py:64-67). */
ORC_API uint64_t orc_sum_v(const uint8_t *bgr, size_t npx)
{
    uint64_t s = 0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (size_t i = 0; i < npx; ++i) {
        const uint8_t *p = bgr + i * 3;
        int v = p[0] > p[1] ? p[0] : p[1];
        s += (uint64_t)(v > p[2] ? v : p[2]);
    }
    return s;
}
/* cv2.add(v_u8, python_float): the non-integer scalar makes arithm_op pick a 32-bit-integer work type for an
 * 8U destination, i.e. v' = sat_u8(v + cvRound(delta)).  (surroundBEV.py:69-72) */
ORC_API int orc_round_delta(double delta) { return rne_d(delta); }

/* one camera of luminance_balance: BGR -> HSV, V += idelta (saturating), HSV -> BGR */
ORC_API void orc_luminance_shift(const uint8_t *src, size_t npx, int idelta, uint8_t *dst)
{
    if (!g_hsv_ready) build_hsv_tables();
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < npx; ++i) {
        uint8_t hsv[3];
        bgr2hsv_px(src + i * 3, hsv);
        hsv[2] = sat_u8(hsv[2] + idelta);
        hsv2bgr_px(hsv, dst + i * 3);
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* A.8  stitch arithmetic   reference: surroundBEV.py:161-162, 279-280, 318-324, 43-55               */
/* ------------------------------------------------------------------------------------------------ */
ORC_API void orc_mask_select(const uint8_t *img, const uint8_t *mask, size_t npx, uint8_t *dst)
{ /* cv2.bitwise_and(img, img, mask=mask) into a fresh (zeroed) destination */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < npx; ++i) {
        int on = mask[i] != 0;
        dst[i * 3] = on ? img[i * 3] : 0; dst[i * 3 + 1] = on ? img[i * 3 + 1] : 0; dst[i * 3 + 2] = on ? img[i * 3 + 2] : 0;
    }
}
ORC_API void orc_weight_mul(const uint8_t *img, const float *weight, size_t n, uint8_t *dst)
{ /* (img * weight_f32).astype(np.uint8): one RN float32 multiply, then truncation */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) dst[i] = (uint8_t)((float)img[i] * weight[i]);
}
ORC_API void orc_add_sat(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *dst)
{ /* cv2.add on 8U */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) { int s = a[i] + b[i]; dst[i] = (uint8_t)(s > 255 ? 255 : s); }
}
ORC_API void orc_channel_sums(const uint8_t *img, size_t npx, uint64_t sums[3])
{
    uint64_t s0 = 0, s1 = 0, s2 = 0;
#pragma omp parallel for schedule(static) reduction(+ : s0, s1, s2)
    for (size_t i = 0; i < npx; ++i) { s0 += img[i * 3]; s1 += img[i * 3 + 1]; s2 += img[i * 3 + 2]; }
    sums[0] = s0; sums[1] = s1; sums[2] = s2;
}
/* cv2.addWeighted(ch, k, 0, 0, 0, ch): the second operand is a scalar, so arithm_op (muldiv=true) works in
 * CV_64F: ch' = sat_u8(cvRound(double(ch) * k + 0*0 + 0)). Applied per channel with gains[3]. */
ORC_API void orc_gain(uint8_t *img, size_t npx, const double gains[3])
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < npx; ++i)
        for (int c = 0; c < 3; ++c)
            img[i * 3 + c] = g_variant[1] ? sat_u8(rne_d((double)img[i * 3 + c] * gains[c] + 0.0 * 0.0 + 0.0))
                                          : sat_u8(rne_f((float)img[i * 3 + c] * (float)gains[c] + 0.0f * 0.0f + 0.0f));
}

/* ------------------------------------------------------------------------------------------------ */
/* Whole per-frame path in the REFERENCE's operation order (BevGenerator.__call__, surroundBEV.py:312-325)
 * -- used for the timed CPU baseline and for full-size parity.  scratch: 5 * bw*bh*3 + 4*fw*fh*3 bytes. */
/* ------------------------------------------------------------------------------------------------ */
ORC_API void orc_bev_call(const uint8_t *const frames[4], int fw, int fh, const int16_t *const lut_xy[4],
                          const uint16_t *const lut_a[4], int bw, int bh, const uint8_t *const masks[4],
                          const float *const weights[4], int blend, int balance, const uint8_t *car,
                          uint8_t *scratch, uint8_t *out)
{
    size_t bpx = (size_t)bw * bh, fpx = (size_t)fw * fh;
    uint8_t *warped = scratch, *part[4];
    for (int c = 0; c < 4; ++c) part[c] = scratch + (size_t)(1 + c) * bpx * 3;
    uint8_t *bal = scratch + 5 * bpx * 3;
    const uint8_t *in[4] = {frames[0], frames[1], frames[2], frames[3]};
    if (balance) {
        double vm[4], vmean;
        for (int c = 0; c < 4; ++c) vm[c] = (double)orc_sum_v(frames[c], fpx) / (double)fpx;
        vmean = (vm[0] + vm[1] + vm[2] + vm[3]) / 4;
        for (int c = 0; c < 4; ++c) {
            orc_luminance_shift(frames[c], fpx, rne_d(vmean - vm[c]), bal + (size_t)c * fpx * 3);
            in[c] = bal + (size_t)c * fpx * 3;
        }
    }
    for (int c = 0; c < 4; ++c) {
        orc_remap_u8(in[c], fw, fh, 3, lut_xy[c], lut_a[c], bw, bh, warped);
        if (blend) orc_weight_mul(warped, weights[c], bpx * 3, part[c]);
        else orc_mask_select(warped, masks[c], bpx, part[c]);
    }
    orc_add_sat(part[0], part[1], bpx * 3, out);
    orc_add_sat(out, part[2], bpx * 3, out);
    orc_add_sat(out, part[3], bpx * 3, out);
    if (balance) {
        uint64_t s[3];
        orc_channel_sums(out, bpx, s);
        double B = (double)s[0] / (double)bpx, G = (double)s[1] / (double)bpx, R = (double)s[2] / (double)bpx;
        double K = (R + G + B) / 3;
        double gains[3] = {K / B, K / G, K / R};
        orc_gain(out, bpx, gains);
    }
    if (car) orc_add_sat(out, car, bpx * 3, out);
}

/* ------------------------------------------------------------------------------------------------ */
/* ExCalibrator pre-processing warps (extrinsicCalib.py:54-59, 122-130)                               */
/* ------------------------------------------------------------------------------------------------ */
/* CenterImage.translate (extrinsicCalib.py:54-59): cv2.warpAffine(img, [[1,0,sx],[0,1,sy]], (w,h)), INTER_LINEAR,
 * BORDER_CONSTANT 0.  warpAffine inverts M and walks fixed-point coordinates (AB_BITS 10, round_delta 16, then >> 5);
 * with integer shifts every coordinate lands on an integer texel with fraction 0, so all the bilinear weight sits on
 * one tap: dst(x, y) = src(x - sx, y - sy), 0 where that tap is outside. */
ORC_API void orc_translate_u8c3(const uint8_t *src, int w, int h, int shift_x, int shift_y, uint8_t *dst)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int u = x - shift_x, v = y - shift_y;
            uint8_t *d = dst + ((size_t)y * w + x) * 3;
            if (u >= 0 && u < w && v >= 0 && v < h) memcpy(d, src + ((size_t)v * w + u) * 3, 3);
            else d[0] = d[1] = d[2] = 0;
        }
}

/* cv2.resize(src, (0,0), fx, fy) with INTER_LINEAR on 8UC3 (ScaleImage.__call__, extrinsicCalib.py:125).
 * dsize = (cvRound(w*fx), cvRound(h*fy)); scale = 1/fx (not re-derived from dsize when dsize is empty).
 * Per destination column: f = (float)((dx+0.5)*scale - 0.5); s = floor(f); f -= s; s < 0 -> (0, 0);
 * s >= w-1 -> (w-1, 0); alpha = saturate_cast<short>({1-f, f} * 2048).  Rows: same without the clamping of f
 * (the two source rows are clipped into the image instead).  Horizontal pass in int32 (S[s]*a0 + S[s+1]*a1),
 * vertical pass of the 8U specialisation: (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
 * (For fx = fy = 0.5 OpenCV switches to its INTER_AREA fast path, (a+b+c+d+2)>>2, which this formula reproduces.) */
ORC_API void orc_resize_dsize(int w, int h, double fx, double fy, int *dw, int *dh)
{
    *dw = rne_d((double)w * fx);
    *dh = rne_d((double)h * fy);
}
static void resize_axis(int n_src, int n_dst, double scale, int clamp_frac, int *ofs, short *coef)
{
    for (int d = 0; d < n_dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (clamp_frac) {
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
        }
        ofs[d] = s;
        const float c0 = 1.f - f;
        int a0 = (int)lrintf(c0 * 2048.f), a1 = (int)lrintf(f * 2048.f);
        coef[d * 2] = (short)(a0 > 32767 ? 32767 : a0 < -32768 ? -32768 : a0);
        coef[d * 2 + 1] = (short)(a1 > 32767 ? 32767 : a1 < -32768 ? -32768 : a1);
    }
}
ORC_API void orc_resize_linear_u8c3(const uint8_t *src, int w, int h, double fx, double fy, uint8_t *dst, int dw, int dh)
{
    int *xofs = (int *)malloc(sizeof(int) * (size_t)(dw + dh));
    short *alpha = (short *)malloc(sizeof(short) * 2 * (size_t)(dw + dh));
    int *yofs = xofs + dw;
    short *beta = alpha + 2 * (size_t)dw;
    resize_axis(w, dw, 1.0 / fx, 1, xofs, alpha);
    resize_axis(h, dh, 1.0 / fy, 0, yofs, beta);
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < dh; ++dy) {
        int r0 = yofs[dy], r1 = yofs[dy] + 1;
        r0 = r0 < 0 ? 0 : (r0 < h ? r0 : h - 1);
        r1 = r1 < 0 ? 0 : (r1 < h ? r1 : h - 1);
        const int b0 = beta[dy * 2], b1 = beta[dy * 2 + 1];
        const uint8_t *p0 = src + (size_t)r0 * w * 3, *p1 = src + (size_t)r1 * w * 3;
        for (int dx = 0; dx < dw; ++dx) {
            const int s0 = xofs[dx], s1 = s0 + 1 < w ? s0 + 1 : w - 1;
            const int a0 = alpha[dx * 2], a1 = alpha[dx * 2 + 1];
            for (int c = 0; c < 3; ++c) {
                const int S0 = p0[s0 * 3 + c] * a0 + p0[s1 * 3 + c] * a1;
                const int S1 = p1[s0 * 3 + c] * a0 + p1[s1 * 3 + c] * a1;
                dst[((size_t)dy * dw + dx) * 3 + c] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
    free(xofs);
    free(alpha);
}
