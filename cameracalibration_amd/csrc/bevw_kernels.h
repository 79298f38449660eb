// bevw_kernels.h -- HIP kernels of libbevwarp (gfx950).  Table builders (run once per calibration) and the
// per-frame kernels of the first, always-valid schedule (BEVW_SCHED_PER_PIXEL).  The tile-plan schedule lives in
// bevw_plan.h.
#pragma once
#include "bevw_device.h"

namespace bevw {

// =============================================================================================================
// Table builders
// =============================================================================================================

// cv2.fisheye.initUndistortRectifyMap(K, D, I, K', size, CV_16SC2)
// reference: surroundBEV.py:98-103, intrinsicCalib.py:98-103, Tools/undistort.py:50-52
//
// OpenCV walks a row accumulating _x += iR[0] per column (an fp64 chain, not j*iR[0]).  With R = I and the skew-free
// K' the reference always builds, iR = [[1/fx',0,-cx'/fx'],[0,1/fy',-cy'/fy'],[0,0,1]], so the chain of _x is the
// same for every row (xs[], produced once by the host in the same serial order), _y is constant along a row
// (it accumulates iR[3] = 0) and _w == 1.  That makes the per-pixel work independent: one thread per map entry.
struct FisheyeParams {
    double fx, fy, cx, cy;  // K
    double k0, k1, k2, k3;  // D
    double iR4, iR5;        // 1/fy', -cy'/fy'
};

static __global__ void k_fisheye_map(FisheyeParams p, const double *__restrict__ xs, int width, int height,
                              int16_t *__restrict__ map1, uint16_t *__restrict__ map2)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= width || i >= height) return;
    const double _w = 1.0;
    const double x = xs[j] / _w;
    const double y = ((double)i * p.iR4 + p.iR5) / _w;
    const double r = sqrt(x * x + y * y);
    const double theta = atan(r);
    const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const double theta_d = theta * (1 + p.k0 * t2 + p.k1 * t4 + p.k2 * t6 + p.k3 * t8);
    const double scale = (r == 0) ? 1.0 : theta_d / r;
    const double u = p.fx * x * scale + p.cx;
    const double v = p.fy * y * scale + p.cy;
    const int iu = rne_d(u * kQOne), iv = rne_d(v * kQOne);
    const size_t o = (size_t)i * width + j;
    map1[o * 2 + 0] = (int16_t)(iu >> kQBits);
    map1[o * 2 + 1] = (int16_t)(iv >> kQBits);
    map2[o] = (uint16_t)((iv & (kQOne - 1)) * kQOne + (iu & (kQOne - 1)));
}

// cv2.initUndistortRectifyMap(K, D, I, K', size, CV_16SC2) -- pinhole model of Normal._get_undistort_maps
// (intrinsicCalib.py:158-163).  iR = inv(K') by cofactors (cv::invert, DECOMP_LU, 3x3); for the skew-free K' its
// off-diagonal terms iR[1], iR[3], iR[6], iR[7] are exactly 0, so the per-column accumulation _x += iR[0] is the same
// chain for every row (xs[], made by the host in the same serial order) and _y, _w are constant along a row.
struct PinholeParams {
    double fx, fy, u0, v0;
    double k1, k2, p1, p2, k3, k4, k5, k6;
    double iR4, iR5, iR8;
};

static __global__ void k_pinhole_map(PinholeParams p, const double *__restrict__ xs, int width, int height,
                              int16_t *__restrict__ map1, uint16_t *__restrict__ map2)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= width || i >= height) return;
    const double _x = xs[j], _y = (double)i * p.iR4 + p.iR5, _w = (double)i * 0.0 + p.iR8;
    const double w = 1. / _w, x = _x * w, y = _y * w;
    const double x2 = x * x, y2 = y * y;
    const double r2 = x2 + y2, _2xy = 2 * x * y;
    const double kr = (1 + ((p.k3 * r2 + p.k2) * r2 + p.k1) * r2) / (1 + ((p.k6 * r2 + p.k5) * r2 + p.k4) * r2);
    const double xd = (x * kr + p.p1 * _2xy + p.p2 * (r2 + 2 * x2));
    const double yd = (y * kr + p.p1 * (r2 + 2 * y2) + p.p2 * _2xy);
    const double u = p.fx * xd + p.u0, v = p.fy * yd + p.v0;
    const int iu = rne_d(u * kQOne), iv = rne_d(v * kQOne);
    const size_t o = (size_t)i * width + j;
    map1[o * 2 + 0] = (int16_t)(iu >> kQBits);
    map1[o * 2 + 1] = (int16_t)(iv >> kQBits);
    map2[o] = (uint16_t)((iv & (kQOne - 1)) * kQOne + (iu & (kQOne - 1)));
}

// Camera.get_bev_maps (surroundBEV.py:105-108): cv2.warpPerspective over the CV_16SC2 and CV_16UC1 undistort maps.
struct Mat3 { double m[9]; };

// warp_mode (BEVW_COMPAT_WARP): 0 = classic; else the 16UC1 map goes through member `warp_mode` of the float32 family (bevw_device.h:
// OpenCV >= 4.11 has float32 linear kernels for one-channel 16U images; the two-channel 16S map is not a type they take and stays classic)
static __global__ void k_bev_lut(Mat3 Minv, const int16_t *__restrict__ und1, const uint16_t *__restrict__ und2, int uw,
                          int uh, int bw, int bh, int bw0, int16_t *__restrict__ lut1, uint16_t *__restrict__ lut2, int warp_mode = 0)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= bw || y >= bh) return;
    int sx, sy;
    unsigned code;
    perspective_coord(Minv.m, x, y, bw0, sx, sy, code);
    int o1[2], o2[1];
    remap_f32_px<int16_t, 2>(und1, uw, uh, sx, sy, code, o1);
    if (warp_mode & kWarpF32) warp_f32_px<uint16_t, 1>(und2, uw, uh, Minv.m, x, y, warp_mode, o2);
    else remap_f32_px<uint16_t, 1>(und2, uw, uh, sx, sy, code, o2);
    const size_t o = (size_t)y * bw + x;
    lut1[o * 2 + 0] = (int16_t)sat_s16(o1[0]);
    lut1[o * 2 + 1] = (int16_t)sat_s16(o1[1]);
    lut2[o] = (uint16_t)sat_u16(o2[0]);
}

// cv2.fillPoly(mask, [pts], 255), LINE_8, shift 0  (surroundBEV.py:156-159, 231-234)
// = Bresenham boundary of every edge  UNION  even-odd scanline spans between XY_SHIFT=16 fixed-point edges.
struct PolyEdge { int y0, y1; long long x, dx; };
struct PolyJob {
    int ceil_left;       // OpenCV < 4.5.2 span rule: left end rounded up (BEVW_COMPAT_FILLPOLY 0)
    int npts;
    int pts[8][2];       // integer vertices (after .astype(np.int32))
    int nedges;
    PolyEdge edges[8];   // non-horizontal edges, unsorted
};

__device__ inline bool clip_segment(int w, int h, long long &x1, long long &y1, long long &x2, long long &y2)
{
    const long long right = w - 1, bottom = h - 1;
    if (w <= 0 || h <= 0) return false;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// One thread per polygon edge: 8-connected Bresenham, left-to-right, after clipping to the image.
static __global__ void k_poly_outline(PolyJob job, uint8_t *__restrict__ img, int w, int h, uint8_t color)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= job.npts) return;
    const int p = (e + job.npts - 1) % job.npts;
    long long x1 = job.pts[p][0], y1 = job.pts[p][1], x2 = job.pts[e][0], y2 = job.pts[e][1];
    if ((unsigned long long)x1 >= (unsigned long long)w || (unsigned long long)x2 >= (unsigned long long)w ||
        (unsigned long long)y1 >= (unsigned long long)h || (unsigned long long)y2 >= (unsigned long long)h) {
        if (!clip_segment(w, h, x1, y1, x2, y2)) return;
    }
    long long dx = x2 - x1, dy = y2 - y1, x = x1, y = y1;
    if (dx < 0) { dx = -dx; dy = -dy; x = x2; y = y2; }
    const int ystep = dy < 0 ? -1 : 1;
    if (dy < 0) dy = -dy;
    const bool steep = dy > dx;
    const long long major = steep ? dy : dx, minor = steep ? dx : dy;
    long long err = major - 2 * minor;
    for (long long i = 0; i <= major; ++i) {
        img[(size_t)y * w + x] = color;
        const bool neg = err < 0;
        err += -2 * minor + (neg ? 2 * major : 0);
        if (steep) { y += ystep; if (neg) x += 1; }
        else       { x += 1;     if (neg) y += ystep; }
    }
}

// One thread per scanline.  An edge alive on row y (y0 <= y < y1) sits at x + (y - y0) * dx: the reference advances
// every paired edge by dx once per row, which is this closed form.  Spans are [xa >> 16, xb >> 16] between
// x-sorted pairs.
static __global__ void k_poly_fill(PolyJob job, uint8_t *__restrict__ img, int w, int h, uint8_t color)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= h || job.nedges < 2) return;
    long long xs[8];
    int n = 0;
    for (int e = 0; e < job.nedges; ++e) {
        const PolyEdge &E = job.edges[e];
        if (E.y0 <= y && y < E.y1) {
            const long long xv = E.x + (long long)(y - E.y0) * E.dx;
            int k = n++;
            while (k > 0 && xs[k - 1] > xv) { xs[k] = xs[k - 1]; --k; }
            xs[k] = xv;
        }
    }
    for (int i = 0; i + 1 < n; i += 2) {
        int xa = (int)((xs[i] + (job.ceil_left ? 65535 : 0)) >> 16), xb = (int)(xs[i + 1] >> 16);
        if (xa < w && xb >= 0) {
            if (xa < 0) xa = 0;
            if (xb >= w) xb = w - 1;
            for (int x = xa; x <= xb; ++x) img[(size_t)y * w + x] = color;
        }
    }
}

// BlendMask.get_blend_mask (surroundBEV.py:270-277): weight = uint8(dA**2 / (dA**2 + dB**2 + 1e-6) * 255) with
// d = |cv2.pointPolygonTest(2-point contour, (x, y), True)| = distance to the seam segment.
struct Seam { int p[4]; };

__device__ inline double segment_distance(const Seam &s, double px, double py)
{
    double min_num = 3.402823466e+38, min_den = 1;
    float vx = (float)s.p[2], vy = (float)s.p[3];
    const float ptx = (float)px, pty = (float)py;
    for (int i = 0; i < 2; ++i) {
        const float v0x = vx, v0y = vy;
        vx = (float)s.p[i * 2]; vy = (float)s.p[i * 2 + 1];
        const double dx = vx - v0x, dy = vy - v0y;
        const double dx1 = ptx - v0x, dy1 = pty - v0y;
        const double dx2 = ptx - vx, dy2 = pty - vy;
        double num, den = 1;
        if (dx1 * dx + dy1 * dy <= 0) num = dx1 * dx1 + dy1 * dy1;
        else if (dx2 * dx + dy2 * dy >= 0) num = dx2 * dx2 + dy2 * dy2;
        else { num = dy1 * dx - dx1 * dy; num *= num; den = dx * dx + dy * dy; }
        if (num * min_den < min_num * den) {
            min_num = num; min_den = den;
            if (min_num == 0) break;
        }
    }
    return sqrt(min_num / min_den);
}

static __global__ void k_blend_weights(uint8_t *__restrict__ maskA, const uint8_t *__restrict__ maskB, int w, int h, Seam lineA,
                                Seam lineB)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= h) return;
    const size_t o = (size_t)y * w + x;
    if ((maskA[o] & maskB[o]) == 0) return;
    const double dA = segment_distance(lineA, x, y), dB = segment_distance(lineB, x, y);
    const double a2 = dA * dA, b2 = dB * dB;  // Python's d**2
    const double v = a2 / (a2 + b2 + 1e-6) * 255;
    maskA[o] = (uint8_t)v;  // float64 -> uint8 store: truncation
}

// =============================================================================================================
// Per-frame kernels
// =============================================================================================================

// cv2.remap(src, map1, map2, INTER_LINEAR) for a batch: one thread per destination pixel.
// grid = (ceil(dw / 256), dh, batch)
static __global__ void k_remap_lut(const uint8_t *__restrict__ src, int sw, int sh, const int16_t *__restrict__ map1,
                            const uint16_t *__restrict__ map2, int dw, int dh, uint8_t *__restrict__ dst, int ties_even = 0)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= dw) return;
    const size_t o = (size_t)y * dw + x;
    const uint8_t *s = src + (size_t)blockIdx.z * sw * sh * 3;
    uint8_t *d = dst + ((size_t)blockIdx.z * dw * dh + o) * 3;
    const int sx = map1[o * 2], sy = map1[o * 2 + 1];
    int out[3];
    remap_u8c3_px<false>(s, sw, sh, sx, sy, map2[o] & (kQTab2 - 1), out, 0, nullptr, ties_even);
    d[0] = (uint8_t)out[0]; d[1] = (uint8_t)out[1]; d[2] = (uint8_t)out[2];
}

// cv2.warpPerspective(src_8UC3, H, dsize): coordinates made on the fly (extrinsicCalib.py:166-169).
// warp_mode (BEVW_COMPAT_WARP): 0 = the classic fixed-point kernels, else member `warp_mode` of the float32 family (bevw_device.h)
static __global__ void k_warp_perspective(const uint8_t *__restrict__ src, int sw, int sh, Mat3 Minv, int bw0, int dw, int dh,
                                   uint8_t *__restrict__ dst, int warp_mode = 0)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= dw) return;
    const size_t o = (size_t)y * dw + x;
    const uint8_t *s = src + (size_t)blockIdx.z * sw * sh * 3;
    uint8_t *d = dst + ((size_t)blockIdx.z * dw * dh + o) * 3;
    int sx, sy, out[3];
    unsigned code;
    if (warp_mode & kWarpF32) {
        warp_f32_px<uint8_t, 3>(s, sw, sh, Minv.m, x, y, warp_mode, out);
        d[0] = (uint8_t)sat_u8(out[0]); d[1] = (uint8_t)sat_u8(out[1]); d[2] = (uint8_t)sat_u8(out[2]);
        return;
    }
    perspective_coord(Minv.m, x, y, bw0, sx, sy, code);
    remap_u8c3_px<false>(s, sw, sh, sx, sy, code, out, 0, nullptr);
    d[0] = (uint8_t)out[0]; d[1] = (uint8_t)out[1]; d[2] = (uint8_t)out[2];
}

// sum of V = max(B,G,R) over one frame; np.mean(v) = sum / N exactly in fp64 (surroundBEV.py:64-67).
// grid = (blocks_per_frame, n_frames).  A lane eats 12-byte pieces (4 whole texels, one global_load_dwordx3): consecutive lanes read
// consecutive pieces, so every wave instruction covers 768 contiguous bytes, and 4 pieces per lane are in flight per trip.
// vec_ok: the frames are 4-byte aligned and a whole number of dwords.
struct __attribute__((packed, aligned(4))) VsumPiece { uint32_t x, y, z; };
__device__ __forceinline__ unsigned vsum_piece(const VsumPiece &p)
{
    // texels (B G R): bytes 0..2 | 3..5 | 6..8 | 9..11 of x y z
    const unsigned t1 = __builtin_amdgcn_alignbyte(p.y, p.x, 3), t2 = __builtin_amdgcn_alignbyte(p.z, p.y, 2), t3 = p.z >> 8;
    auto v = [](unsigned t) { return max(t & 255u, max((t >> 8) & 255u, (t >> 16) & 255u)); };
    return v(p.x) + v(t1) + v(t2) + v(t3);
}
// part_stride 0: the blocks of a frame add their sums atomically into sums[frame] (zeroed by the caller); > 0: block x of frame y stores
// its sum at sums[y * part_stride + x] -- no atomics, no zeroing pass in front of the kernel; k_lum_delta adds the parts.
static __global__ void k_vsum(const uint8_t *__restrict__ frames, size_t frame_bytes, int vec_ok,
                       unsigned long long *__restrict__ sums, int part_stride = 0)
{
    const uint8_t *f = frames + (size_t)blockIdx.y * frame_bytes;
    const size_t npieces = vec_ok ? frame_bytes / 12 : 0;
    const VsumPiece *fp = reinterpret_cast<const VsumPiece *>(f);
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    size_t i = tid;
    for (; i + 3 * nthreads < npieces; i += 4 * nthreads) {
#if BEVW_VSUM_NT   // (bevw_device.h: the frames pass once)
        auto piece = [](const VsumPiece *q) {
            const uint32_t *w = reinterpret_cast<const uint32_t *>(q);
            return VsumPiece{once_load<1>(w), once_load<1>(w + 1), once_load<1>(w + 2)};
        };
        const VsumPiece a = piece(fp + i), b = piece(fp + i + nthreads), c = piece(fp + i + 2 * nthreads), d = piece(fp + i + 3 * nthreads);
#else
        const VsumPiece a = fp[i], b = fp[i + nthreads], c = fp[i + 2 * nthreads], d = fp[i + 3 * nthreads];
#endif
        acc += vsum_piece(a) + vsum_piece(b) + vsum_piece(c) + vsum_piece(d);
    }
    for (; i < npieces; i += nthreads) acc += vsum_piece(fp[i]);
    // texels not covered by whole pieces
    for (size_t t = npieces * 4 + tid; t * 3 + 2 < frame_bytes; t += nthreads)
        acc += max((unsigned)f[t * 3], max((unsigned)f[t * 3 + 1], (unsigned)f[t * 3 + 2]));
    __shared__ unsigned long long part[16];
    unsigned long long s = wave_sum_u64(acc);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) part[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i2 = 0; i2 < (int)(blockDim.x >> 6); ++i2) t += part[i2];
        if (part_stride > 0) sums[(size_t)blockIdx.y * part_stride + blockIdx.x] = t;
        else atomicAdd(&sums[blockIdx.y], t);
    }
}

// luminance_balance scalars (surroundBEV.py:64-72): delta_c = cvRound(V_mean - V_c), V_mean = (Vf+Vb+Vl+Vr)/4.
// one thread per 4-camera frame set.
// nparts / part_stride: every frame's V sum arrives as `nparts` partial sums, part_stride entries apart per frame (k_vsum); 1 / 1: whole sums.
static __global__ void k_lum_delta(const unsigned long long *__restrict__ vsums, double npx, int nsets, int *__restrict__ deltas, int nparts = 1,
                                   int part_stride = 1)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nsets) return;
    double m[4];
    for (int c = 0; c < 4; ++c) {
        unsigned long long t = 0;   // exact: integer sums, whatever the number of parts
        for (int p = 0; p < nparts; ++p) t += vsums[(size_t)(b * 4 + c) * part_stride + p];
        m[c] = (double)t / npx;
    }
    const double vmean = (m[0] + m[1] + m[2] + m[3]) / 4;
    for (int c = 0; c < 4; ++c) deltas[b * 4 + c] = rne_d(vmean - m[c]);
}

// luminance_balance applied to whole frames (the exported helper; the stitch kernels apply it per fetched texel).
// grid = (blocks, n_frames)
static __global__ void k_lum_shift(const uint8_t *__restrict__ frames, size_t frame_px, const int *__restrict__ deltas,
                            const HsvTables *__restrict__ tab, uint8_t *__restrict__ out)
{
    __shared__ HsvTables hsv;
    hsv_tables_to_lds(hsv, tab);
    __syncthreads();
    const size_t base = (size_t)blockIdx.y * frame_px * 3;
    const int delta = deltas[blockIdx.y];
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < frame_px; p += (size_t)gridDim.x * blockDim.x) {
        int b = frames[base + p * 3], g = frames[base + p * 3 + 1], r = frames[base + p * 3 + 2];
        luminance_shift_px(b, g, r, delta, hsv);
        out[base + p * 3] = (uint8_t)b; out[base + p * 3 + 1] = (uint8_t)g; out[base + p * 3 + 2] = (uint8_t)r;
    }
}

// Static tables of one BevGenerator as the per-pixel schedule reads them.
struct StitchTables {
    const int16_t *lut1[4];
    const uint16_t *lut2[4];
    const uint8_t *mask[4];
};

// BevGenerator.__call__ (surroundBEV.py:312-325), schedule BEVW_SCHED_PER_PIXEL: one thread per BEV pixel loops the
// four cameras: mask test -> LUT fetch -> fixed-point bilinear gather (with the luminance round trip on the fetched
// texels when BAL) -> Mask select or BlendMask truncating multiply -> saturating sum.  Without balance the car
// sprite is added here; with balance the pre-gain value is stored and per-frame channel sums are accumulated
// (integer, so the result does not depend on the order of the atomics).
// grid = (ceil(bw / 256), bh, batch)
template <bool BLEND, bool BAL>
static __global__ void k_stitch_pp(const uint8_t *__restrict__ frames, int fw, int fh, StitchTables T, int bw, int bh,
                            const int *__restrict__ deltas, const HsvTables *__restrict__ tab,
                            const uint8_t *__restrict__ car, unsigned long long *__restrict__ chsums,
                            uint8_t *__restrict__ out, int ties_even = 0)
{
    __shared__ HsvTables hsv;
    __shared__ unsigned long long part[3][4];
    if (BAL) {
        hsv_tables_to_lds(hsv, tab);
        __syncthreads();
    }
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int b = blockIdx.z;
    const size_t frame_bytes = (size_t)fw * fh * 3;
    int acc[3] = {0, 0, 0};
    if (x < bw) {
        const size_t o = (size_t)y * bw + x;
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            const int m = T.mask[c][o];
            if (m == 0) continue;
            const uint8_t *src = frames + ((size_t)b * 4 + c) * frame_bytes;
            const int sx = T.lut1[c][o * 2], sy = T.lut1[c][o * 2 + 1];
            int v[3];
            remap_u8c3_px<BAL>(src, fw, fh, sx, sy, T.lut2[c][o] & (kQTab2 - 1), v, BAL ? deltas[b * 4 + c] : 0, &hsv, ties_even);
            if (BLEND) {
                const float wgt = blend_weight_f32(m);
                v[0] = blend_mul(v[0], wgt); v[1] = blend_mul(v[1], wgt); v[2] = blend_mul(v[2], wgt);
            }
            acc[0] = min(255, acc[0] + v[0]); acc[1] = min(255, acc[1] + v[1]); acc[2] = min(255, acc[2] + v[2]);
        }
        uint8_t *d = out + ((size_t)b * bw * bh + o) * 3;
        if (!BAL && car != nullptr) {
            acc[0] = min(255, acc[0] + car[o * 3]); acc[1] = min(255, acc[1] + car[o * 3 + 1]);
            acc[2] = min(255, acc[2] + car[o * 3 + 2]);
        }
        d[0] = (uint8_t)acc[0]; d[1] = (uint8_t)acc[1]; d[2] = (uint8_t)acc[2];
    }
    if (BAL) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            unsigned s = wave_sum_u32((unsigned)acc[k]);
            if (lane == 0) part[k][wv] = s;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            unsigned long long t = 0;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += part[threadIdx.x][i];
            atomicAdd(&chsums[b * 3 + threadIdx.x], t);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Analytic projection (BASELINE north star): the same stitch WITHOUT look-up tables.  Per frame and
// BEV pixel: inverse homography -> undistorted pixel -> fisheye model (the formulas of cv2.fisheye.initUndistortRectifyMap,
// k_fisheye_map above) -> raw frame position, fp64; bilinear interpolation of the four texels in fp64, round half to even.
// This is NOT the reference's arithmetic (the reference samples through two fixed-point tables, quantised to 1/32 pixel
// twice -- the LUT quirk); it is what those tables approximate, judged against the table path by PSNR
// (tests/test_analytic.py).  Masks, blend weights, the luminance round trip on the taps, the saturating sums, the white
// balance and the car are the reference's, shared with k_stitch_pp.  Pixels whose undistorted position falls outside the
// undistorted image contribute 0 (warp_homography of an image has BORDER_CONSTANT 0 there).
// grid = (ceil(bw / 256), bh, batch)
struct AnalyticRig {
    double Minv[4][9];          // inverse homographies (BEV -> undistorted image)
    double fx[4], fy[4], cx[4], cy[4];     // K
    double d[4][4];             // D
    double nfx[4], nfy[4], ncx[4], ncy[4]; // K' of the undistorted image (get_camera_mat_dst)
    int uw, uh;                 // undistorted image size
    // the same calibration rounded to float32 ONCE on the host (bevw_build), with 1 / nfx, 1 / nfy: what the fp32 mode computes with
    // (round 6: it used to convert every fp64 field per pixel)
    float fMinv[4][9], ffx[4], ffy[4], fcx[4], fcy[4], fd[4][4], finv_nfx[4], finv_nfy[4], fncx[4], fncy[4];
};

// F = double: the specification's arithmetic (oracle/np_analytic.py).  F = float: the same formulas in fp32 (atanf, sqrtf; positions good
// to ~1e-4 pixel), faster; held against the fp64 result by PSNR, not byte for byte.
template <typename F>
struct AnalyticTap { int sx, sy; F ax, ay; };   // raw-frame footprint of one (pixel, camera): top-left texel and the fractions

// BEV pixel (x, y) of camera c -> footprint; false when the pixel samples nothing (outside the undistorted image or the frame)
// fp32 mode: the formulas below with float32 parameters, fused multiply-adds and reciprocals (v_rcp_f32, 1 ulp) instead of IEEE divisions --
// an arithmetic of its own, held against the fp64 specification by PSNR and by the share of identical bytes (tests/test_analytic.py), not
// bit for bit.  The same function serves the once-per-handle map (k_analytic_map<float>) and the per-pixel kernel (k_stitch_analytic<.., float>).
__device__ __forceinline__ bool analytic_project_f32(const AnalyticRig &R, int c, int x, int y, int fw, int fh, AnalyticTap<float> &t)
{
    const float *M = R.fMinv[c];
    const float xf = (float)x, yf = (float)y;
    const float X = fmaf(M[0], xf, fmaf(M[1], yf, M[2])), Y = fmaf(M[3], xf, fmaf(M[4], yf, M[5])), Wd = fmaf(M[6], xf, fmaf(M[7], yf, M[8]));
    if (Wd == 0.f) return false;
    const float iw = __builtin_amdgcn_rcpf(Wd);
    const float u = X * iw, v = Y * iw;
    if (!(u >= 0.f && u <= (float)(R.uw - 1) && v >= 0.f && v <= (float)(R.uh - 1))) return false;
    const float xn = (u - R.fncx[c]) * R.finv_nfx[c], yn = (v - R.fncy[c]) * R.finv_nfy[c];
    const float r = __builtin_amdgcn_sqrtf(fmaf(xn, xn, yn * yn));
    const float theta = atanf(r);
    const float t2 = theta * theta;
    const float poly = fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, R.fd[c][3], R.fd[c][2]), R.fd[c][1]), R.fd[c][0]), 1.f);   // 1 + k1 t^2 + k2 t^4 + k3 t^6 + k4 t^8
    const float scale = (r == 0.f) ? 1.f : theta * poly * __builtin_amdgcn_rcpf(r);
    const float px = fmaf(R.ffx[c] * xn, scale, R.fcx[c]), py = fmaf(R.ffy[c] * yn, scale, R.fcy[c]);
    if (!(px > -1.f && px < (float)fw && py > -1.f && py < (float)fh)) return false;   // the whole footprint is outside
    const float fpx = floorf(px), fpy = floorf(py);
    t.sx = (int)fpx; t.sy = (int)fpy;
    t.ax = px - fpx; t.ay = py - fpy;
    return true;
}
template <typename F>
__device__ __forceinline__ bool analytic_project(const AnalyticRig &R, int c, int x, int y, int fw, int fh, AnalyticTap<F> &t)
{
    if constexpr (sizeof(F) == 4) return analytic_project_f32(R, c, x, y, fw, fh, t);
    const double *M = R.Minv[c];
    const F X = (F)M[0] * x + (F)M[1] * y + (F)M[2], Y = (F)M[3] * x + (F)M[4] * y + (F)M[5], Wd = (F)M[6] * x + (F)M[7] * y + (F)M[8];
    if (Wd == (F)0) return false;
    const F u = X / Wd, v = Y / Wd;
    if (!(u >= (F)0 && u <= (F)(R.uw - 1) && v >= (F)0 && v <= (F)(R.uh - 1))) return false;
    const F xn = (u - (F)R.ncx[c]) / (F)R.nfx[c], yn = (v - (F)R.ncy[c]) / (F)R.nfy[c];
    const F r = sqrt(xn * xn + yn * yn);
    const F theta = atan(r);
    const F t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const F theta_d = theta * ((F)1 + (F)R.d[c][0] * t2 + (F)R.d[c][1] * t4 + (F)R.d[c][2] * t6 + (F)R.d[c][3] * t8);
    const F scale = (r == (F)0) ? (F)1 : theta_d / r;
    const F px = (F)R.fx[c] * xn * scale + (F)R.cx[c], py = (F)R.fy[c] * yn * scale + (F)R.cy[c];
    if (!(px > (F)-1 && px < (F)fw && py > (F)-1 && py < (F)fh)) return false;   // the whole footprint is outside
    const F fpx = floor(px), fpy = floor(py);
    t.sx = (int)fpx; t.sy = (int)fpy;
    t.ax = px - fpx; t.ay = py - fpy;
    return true;
}

// The projection of every BEV pixel of camera c as a table: top-left texel (sx, sy) and the fractions as 21-bit fixed point
// (bevw_unit.h: kUnitFracBits).  Pixels that sample nothing get sx = sy = INT16_MIN.  Evaluated once per handle and projection mode:
// the host compiles the unit schedule (bevw_unit.h, wide plan) from it, and the table is dropped again.
template <typename F>
static __global__ void k_analytic_map(AnalyticRig R, int c, int fw, int fh, int bw, int bh, int16_t *__restrict__ sxy, uint32_t *__restrict__ frac)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= bw || y >= bh) return;
    const size_t o = (size_t)y * bw + x;
    AnalyticTap<F> t;
    if (!analytic_project<F>(R, c, x, y, fw, fh, t)) {
        sxy[o * 2] = sxy[o * 2 + 1] = (int16_t)-32768;
        frac[o * 2] = frac[o * 2 + 1] = 0u;
        return;
    }
    const F one = (F)(1u << 21);
    const uint32_t top = (1u << 21) - 1u;
    sxy[o * 2] = (int16_t)t.sx; sxy[o * 2 + 1] = (int16_t)t.sy;
    frac[o * 2] = min(top, (uint32_t)rint(t.ax * one));
    frac[o * 2 + 1] = min(top, (uint32_t)rint(t.ay * one));
}

// bilinear interpolation of the footprint in F, BORDER_CONSTANT 0 per tap, round half to even.
// Interior footprints of 4-byte aligned frames (every footprint but the ones on the frame border) are fetched as two aligned 12-byte
// windows, one per footprint row, and realigned with v_alignbyte (two vector loads per contributor instead of twelve byte loads).
template <bool BAL, typename F>
__device__ __forceinline__ void analytic_sample(const uint8_t *__restrict__ src, int fw, int fh, const AnalyticTap<F> &tp, int out[3], int delta,
                                                const HsvTables &hsv, bool aligned)
{
    F t[4][3];
    const size_t toff = ((size_t)tp.sy * fw + tp.sx) * 3;
    if (aligned && (unsigned)tp.sx < (unsigned)(fw - 1) && (unsigned)tp.sy < (unsigned)(fh - 1) &&
        (toff & ~(size_t)3) + (size_t)fw * 3 + 12 <= (size_t)fw * fh * 3) {
        const uint32_t mis = (uint32_t)toff & 3u;
        const uint32_t *p0 = reinterpret_cast<const uint32_t *>(src + (toff & ~(size_t)3));
        const uint32_t *p1 = reinterpret_cast<const uint32_t *>(src + ((toff + (size_t)fw * 3) & ~(size_t)3));
        const uint32_t mis1 = (uint32_t)(toff + (size_t)fw * 3) & 3u;
        const uint32_t a0 = p0[0], a1 = p0[1], a2 = p0[2], b0 = p1[0], b1 = p1[1], b2 = p1[2];
        // 8 footprint bytes of a row (B0 G0 R0 B1 G1 R1 x x) from the 12-byte window around them
        const uint32_t r0x = __builtin_amdgcn_alignbyte(a1, a0, mis), r0y = __builtin_amdgcn_alignbyte(a2, a1, mis);
        const uint32_t r1x = __builtin_amdgcn_alignbyte(b1, b0, mis1), r1y = __builtin_amdgcn_alignbyte(b2, b1, mis1);
        int px[4][3] = {{(int)(r0x & 255u), (int)((r0x >> 8) & 255u), (int)((r0x >> 16) & 255u)},
                        {(int)(r0x >> 24), (int)(r0y & 255u), (int)((r0y >> 8) & 255u)},
                        {(int)(r1x & 255u), (int)((r1x >> 8) & 255u), (int)((r1x >> 16) & 255u)},
                        {(int)(r1x >> 24), (int)(r1y & 255u), (int)((r1y >> 8) & 255u)}};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (BAL) luminance_shift_px(px[q][0], px[q][1], px[q][2], delta, hsv);
            t[q][0] = (F)px[q][0]; t[q][1] = (F)px[q][1]; t[q][2] = (F)px[q][2];
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tx = tp.sx + (q & 1), ty = tp.sy + (q >> 1);
            if ((unsigned)tx < (unsigned)fw && (unsigned)ty < (unsigned)fh) {
                const uint8_t *p = src + ((size_t)ty * fw + tx) * 3;
                int b = p[0], g = p[1], rr = p[2];
                if (BAL) luminance_shift_px(b, g, rr, delta, hsv);
                t[q][0] = (F)b; t[q][1] = (F)g; t[q][2] = (F)rr;
            } else {
                t[q][0] = t[q][1] = t[q][2] = (F)0;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const F top = ((F)1 - tp.ax) * t[0][k] + tp.ax * t[1][k], bot = ((F)1 - tp.ax) * t[2][k] + tp.ax * t[3][k];
        if constexpr (sizeof(F) == 4) out[k] = sat_u8(__float2int_rn(((F)1 - tp.ay) * top + tp.ay * bot));   // (a convex combination of bytes: no range check needed)
        else out[k] = sat_u8(rne_d((double)(((F)1 - tp.ay) * top + tp.ay * bot)));
    }
}

// grid = (ceil(bw / 256), bh, ceil(batch / fpt)): a thread evaluates the projection of its pixel and samples `fpt` frames with it (the
// calibration of a handle is the same for every frame of a call; no table ever reaches memory).  fpt = kAnalyticFrames by default;
// fpt = 1 (BEVW_ANALYTIC_FRAMES=1) is north_star's wording taken literally: inverse homography + fisheye model per output pixel PER FRAME.
// tiles != nullptr: grid = (number of listed tiles, 1, chunks) -- a block takes one 32 x 8 base tile of the list (the tiles the unit
// schedule of the analytic mode leaves over: frame-border footprints)
constexpr int kAnalyticFrames = 32;   // (8: the projection was a third of a 64-frame call; profiles/r03/sweeps.log)
template <bool BLEND, bool BAL, typename F>
static __global__ void k_stitch_analytic(const uint8_t *__restrict__ frames, int fw, int fh, AnalyticRig R, StitchTables T, int bw, int bh, int batch,
                                  const int *__restrict__ deltas, const HsvTables *__restrict__ tab,
                                  const uint8_t *__restrict__ car, unsigned long long *__restrict__ chsums,
                                  uint8_t *__restrict__ out, const uint32_t *__restrict__ tiles = nullptr, int tiles_x = 0, int fpt = kAnalyticFrames)
{
    __shared__ HsvTables hsv;
    __shared__ unsigned long long part[3][4];
    if (BAL) {
        hsv_tables_to_lds(hsv, tab);
        __syncthreads();
    }
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (tiles != nullptr) {
        const int t = (int)tiles[blockIdx.x];
        x = (t % tiles_x) * 32 + (int)(threadIdx.x & 31u);
        y = (t / tiles_x) * 8 + (int)(threadIdx.x >> 5);
        if (y >= bh) { x = bw; y = 0; }     // below the image: an idle lane
    }
    const int b_begin = blockIdx.z * fpt, b_end = min(batch, b_begin + fpt);
    const size_t frame_bytes = (size_t)fw * fh * 3;
    const size_t o = (size_t)y * bw + (x < bw ? x : 0);
    // dword accesses: frames on 4-byte boundaries (window loads) / rows of whole pixel quads on 4-byte boundaries (12-byte stores)
    const bool aligned = (frame_bytes & 3) == 0 && (((uintptr_t)frames) & 3) == 0;
    const bool quad_store = (bw & 3) == 0 && (((uintptr_t)out) & 3) == 0;
    AnalyticTap<F> tap[4];
    int cam[4], msk[4], n = 0;
    if (x < bw) {
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            const int m = T.mask[c][o];
            if (m == 0) continue;
            AnalyticTap<F> t;
            if (!analytic_project<F>(R, c, x, y, fw, fh, t)) continue;   // contributes 0
            // (the slots are filled in camera order, as the reference adds front, back, left, right)
            if (n == 0) { tap[0] = t; cam[0] = c; msk[0] = m; }
            else if (n == 1) { tap[1] = t; cam[1] = c; msk[1] = m; }
            else if (n == 2) { tap[2] = t; cam[2] = c; msk[2] = m; }
            else { tap[3] = t; cam[3] = c; msk[3] = m; }
            ++n;
        }
    }
#pragma unroll 4
    for (int b = b_begin; b < b_end; ++b) {   // (unrolled: the footprint loads of four frames are in flight together)
        int acc[3] = {0, 0, 0};
        if (x < bw) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k >= n) break;
                const uint8_t *src = frames + ((size_t)b * 4 + cam[k]) * frame_bytes;
                int v[3];
                analytic_sample<BAL, F>(src, fw, fh, tap[k], v, BAL ? deltas[b * 4 + cam[k]] : 0, hsv, aligned);
                if (BLEND) {
                    const float wgt = blend_weight_f32(msk[k]);
                    v[0] = blend_mul(v[0], wgt); v[1] = blend_mul(v[1], wgt); v[2] = blend_mul(v[2], wgt);
                }
                acc[0] = min(255, acc[0] + v[0]); acc[1] = min(255, acc[1] + v[1]); acc[2] = min(255, acc[2] + v[2]);
            }
            if (!BAL && car != nullptr) {
                acc[0] = min(255, acc[0] + car[o * 3]); acc[1] = min(255, acc[1] + car[o * 3 + 1]);
                acc[2] = min(255, acc[2] + car[o * 3 + 2]);
            }
        }
        if (quad_store) {
            // the 4 lanes of a pixel quad hand their pixels to the first one, which stores 12 bytes (bw % 4 == 0: a quad is inside the
            // image or outside as a whole; blockDim.x is a multiple of 4)
            const uint32_t P = (uint32_t)acc[0] | ((uint32_t)acc[1] << 8) | ((uint32_t)acc[2] << 16);
            const int l0 = (int)(threadIdx.x & 63u) & ~3;
            const uint32_t P0 = __shfl(P, l0, 64), P1 = __shfl(P, l0 + 1, 64), P2 = __shfl(P, l0 + 2, 64), P3 = __shfl(P, l0 + 3, 64);
            if ((threadIdx.x & 3u) == 0 && x < bw) {
                uint32_t *d = reinterpret_cast<uint32_t *>(out + ((size_t)b * bw * bh + o) * 3);
                d[0] = P0 | (P1 << 24); d[1] = (P1 >> 8) | (P2 << 16); d[2] = (P2 >> 16) | (P3 << 8);
            }
        } else if (x < bw) {
            uint8_t *d = out + ((size_t)b * bw * bh + o) * 3;
            d[0] = (uint8_t)acc[0]; d[1] = (uint8_t)acc[1]; d[2] = (uint8_t)acc[2];
        }
        if (BAL) {
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            __syncthreads();   // part[] of the previous frame has been consumed
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                unsigned s = wave_sum_u32((unsigned)acc[k]);
                if (lane == 0) part[k][wv] = s;
            }
            __syncthreads();
            if (threadIdx.x < 3) {
                unsigned long long t = 0;
                for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += part[threadIdx.x][i];
                atomicAdd(&chsums[b * 3 + threadIdx.x], t);
            }
        }
    }
}

// north_star's wording taken literally -- "a fused per-output-pixel kernel that inverts the 3x3 homography, applies the K/D fisheye forward
// projection and bilinear-samples the source image" -- as a kernel of its own (round 6): ONE thread = one BEV pixel of ONE frame; per
// contributing camera the inverse homography + fisheye model (analytic_project) and at once the fp32 / fp64 bilinear sample of the 2 x 2
// footprint; blend weight, saturating sum, car sprite; the 4 lanes of a pixel quad hand their pixels to one 12-byte store.  Nothing is kept
// across cameras or frames (k_stitch_analytic above amortises the projection over `fpt` frames and carries up to four taps in registers: 546
// wave-level VALU instructions per wave and pixel at fpt = 1, a third of them register moves and 64-bit address arithmetic).  Interior
// footprints of 4-byte aligned frame sets are two buffer_load_dwordx3 with 32-bit offsets inside the frame set of frame b; the others take
// analytic_sample's per-byte path.  No balance variant (the luminance statistics need the kernel above).
// grid = (ceil(bw / 256), bh, batch); bench.py: direct_stitch_analytic_perpixel_b64 (BEVW_ANALYTIC_UNITS=0, BEVW_ANALYTIC_FRAMES=1).
typedef uint32_t an_u32x3 __attribute__((ext_vector_type(3)));
template <bool BLEND, typename F>
static __global__ void __launch_bounds__(256) k_stitch_perpixel(const uint8_t *__restrict__ frames, int fw, int fh, AnalyticRig R, StitchTables T, int bw, int bh,
                                                               const uint8_t *__restrict__ car, uint8_t *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    const uint32_t fb32 = (uint32_t)fw * (uint32_t)fh * 3u, row32 = (uint32_t)fw * 3u;
    const size_t set_bytes = (size_t)fb32 * 4;
    const uint8_t *set_base = frames + (size_t)b * set_bytes;
    const bool set32 = (fb32 & 3u) == 0 && (((uintptr_t)frames) & 3u) == 0 && set_bytes < ((size_t)1 << 31);
    const __amdgpu_buffer_rsrc_t set = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(set_base), 0, set32 ? (uint32_t)set_bytes : 0u, 0x00020000u);
    const uint32_t o = (uint32_t)y * (uint32_t)bw + (uint32_t)(x < bw ? x : 0);
    int acc[3] = {0, 0, 0};
    if (x < bw) {
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {   // (camera order: the reference adds front, back, left, right)
            const int m = T.mask[c][o];
            if (m == 0) continue;
            AnalyticTap<F> t;
            if (!analytic_project<F>(R, c, x, y, fw, fh, t)) continue;   // contributes 0
            int v[3];
            const uint32_t toff = ((uint32_t)t.sy * (uint32_t)fw + (uint32_t)t.sx) * 3u;
            if (set32 && (unsigned)t.sx < (unsigned)(fw - 1) && (unsigned)t.sy < (unsigned)(fh - 1) && (toff & ~3u) + row32 + 12u <= fb32) {
                const uint32_t w0 = (uint32_t)c * fb32 + toff, w1 = w0 + row32;
                const an_u32x3 a = __builtin_amdgcn_raw_buffer_load_b96(set, (int)(w0 & ~3u), 0, 0), bb = __builtin_amdgcn_raw_buffer_load_b96(set, (int)(w1 & ~3u), 0, 0);
                // 8 footprint bytes of a row (B0 G0 R0 B1 G1 R1 x x) from the 12-byte window around them
                const uint32_t r0x = __builtin_amdgcn_alignbyte(a.y, a.x, w0 & 3u), r0y = __builtin_amdgcn_alignbyte(a.z, a.y, w0 & 3u);
                const uint32_t r1x = __builtin_amdgcn_alignbyte(bb.y, bb.x, w1 & 3u), r1y = __builtin_amdgcn_alignbyte(bb.z, bb.y, w1 & 3u);
                const uint32_t t00[3] = {r0x & 255u, (r0x >> 8) & 255u, (r0x >> 16) & 255u}, t01[3] = {r0x >> 24, r0y & 255u, (r0y >> 8) & 255u};
                const uint32_t t10[3] = {r1x & 255u, (r1x >> 8) & 255u, (r1x >> 16) & 255u}, t11[3] = {r1x >> 24, r1y & 255u, (r1y >> 8) & 255u};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const F top = ((F)1 - t.ax) * (F)t00[k] + t.ax * (F)t01[k], bot = ((F)1 - t.ax) * (F)t10[k] + t.ax * (F)t11[k];
                    if constexpr (sizeof(F) == 4) v[k] = sat_u8(__float2int_rn(((F)1 - t.ay) * top + t.ay * bot));
                    else v[k] = sat_u8(rne_d((double)(((F)1 - t.ay) * top + t.ay * bot)));
                }
            } else {
                // (frame border; BAL = false: the table argument is never read -- any valid object will do)
                analytic_sample<false, F>(set_base + (size_t)c * fb32, fw, fh, t, v, 0, *reinterpret_cast<const HsvTables *>(&R), false);
            }
            if (BLEND) {
                const float wgt = blend_weight_f32(m);
                v[0] = blend_mul(v[0], wgt); v[1] = blend_mul(v[1], wgt); v[2] = blend_mul(v[2], wgt);
            }
            acc[0] = min(255, acc[0] + v[0]); acc[1] = min(255, acc[1] + v[1]); acc[2] = min(255, acc[2] + v[2]);
        }
        if (car != nullptr) {
            acc[0] = min(255, acc[0] + car[(size_t)o * 3]); acc[1] = min(255, acc[1] + car[(size_t)o * 3 + 1]);
            acc[2] = min(255, acc[2] + car[(size_t)o * 3 + 2]);
        }
    }
    uint8_t *img = out + (size_t)b * bw * bh * 3;
    if ((bw & 3) == 0 && (((uintptr_t)out) & 3) == 0) {
        // the 4 lanes of a pixel quad hand their pixels to the first one, which stores 12 bytes (bw % 4 == 0: a quad is inside the image or
        // outside as a whole; blockDim.x is a multiple of 4).  quad_perm DPP moves: no LDS traffic
        const uint32_t P = (uint32_t)acc[0] | ((uint32_t)acc[1] << 8) | ((uint32_t)acc[2] << 16);
        const uint32_t P1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x55, 0xf, 0xf, false);   // quad_perm [1, 1, 1, 1]
        const uint32_t P2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0xAA, 0xf, 0xf, false);   // [2, 2, 2, 2]
        const uint32_t P3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0xFF, 0xf, 0xf, false);   // [3, 3, 3, 3]
        if ((threadIdx.x & 3u) == 0 && x < bw) {
            uint32_t *d = reinterpret_cast<uint32_t *>(img + (size_t)o * 3);
            d[0] = P | (P1 << 24); d[1] = (P1 >> 8) | (P2 << 16); d[2] = (P2 >> 16) | (P3 << 8);
        }
    } else if (x < bw) {
        uint8_t *d = img + (size_t)o * 3;
        d[0] = (uint8_t)acc[0]; d[1] = (uint8_t)acc[1]; d[2] = (uint8_t)acc[2];
    }
}

// per-channel sums of a batch of images (color_balance as an exported helper). grid = (blocks, batch)
static __global__ void k_channel_sums(const uint8_t *__restrict__ img, size_t npx, unsigned long long *__restrict__ chsums)
{
    const uint8_t *p = img + (size_t)blockIdx.y * npx * 3;
    unsigned acc[3] = {0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (size_t)gridDim.x * blockDim.x) {
        acc[0] += p[i * 3]; acc[1] += p[i * 3 + 1]; acc[2] += p[i * 3 + 2];
    }
    __shared__ unsigned long long part[3][16];
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
        atomicAdd(&chsums[blockIdx.y * 3 + threadIdx.x], t);
    }
}

// color_balance (surroundBEV.py:43-55) gain step + the trailing cv2.add(surround, car) (:323-324).
// B,G,R means in fp64 from the integer sums; K = (R + G + B) / 3; gain_c = K / mean_c;
// cv2.addWeighted(ch, gain, 0, 0, 0, ch) = sat_u8(cvRound(double(ch) * gain + 0*0 + 0)).
// grid = (blocks, batch); in place when in == out.
// f32: cv2.addWeighted evaluated in CV_32F instead of CV_64F (BEVW_COMPAT_ADDWEIGHTED 0)
__device__ __forceinline__ int gain_px(int v, double gain, int f32)
{
    return f32 ? sat_u8(rne_f((float)v * (float)gain + 0.0f * 0.0f + 0.0f)) : sat_u8(rne_d((double)v * gain + 0.0 * 0.0 + 0.0));
}
static __global__ void k_gain(const uint8_t *in, size_t npx, const unsigned long long *__restrict__ chsums,
                       const uint8_t *__restrict__ car, uint8_t *out, int f32 = 0)
{
    const double n = (double)npx;
    const double B = (double)chsums[blockIdx.y * 3 + 0] / n, G = (double)chsums[blockIdx.y * 3 + 1] / n,
                 R = (double)chsums[blockIdx.y * 3 + 2] / n;
    const double K = (R + G + B) / 3;
    const double gain[3] = {K / B, K / G, K / R};
    const size_t base = (size_t)blockIdx.y * npx * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx * 3; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        int v = gain_px((int)in[base + i], gain[c], f32);
        if (car != nullptr) v = min(255, v + car[i]);
        out[base + i] = (uint8_t)v;
    }
}

// Mask.__call__ = cv2.bitwise_and(img, img, mask=mask) (surroundBEV.py:161-162) and
// BlendMask.__call__ = (img * float32(mask / 255.0)).astype(uint8) (surroundBEV.py:279-280) as stand-alone operations
// (inside BevGenerator.__call__ they are fused into the stitch kernels).  grid = (blocks, batch)
static __global__ void k_apply_mask(const uint8_t *__restrict__ img, const uint8_t *__restrict__ mask, size_t npx, int blend,
                             uint8_t *__restrict__ out)
{
    const size_t base = (size_t)blockIdx.y * npx * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (size_t)gridDim.x * blockDim.x) {
        const int m = mask[i];
        const float w = blend_weight_f32(m);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int v = img[base + i * 3 + k];
            out[base + i * 3 + k] = (uint8_t)(blend ? blend_mul(v, w) : (m != 0 ? v : 0));
        }
    }
}

// k_gain with a per-frame 3 x 256 look-up table (the gain is one fp64 multiply + cvRound per byte VALUE, so 768
// table entries per frame replace 3.5 M fp64 operations) and 12-byte vector accesses (4 pixels per lane).
// Needs npx % 4 == 0 and 4-byte aligned images.  grid = (blocks, batch), block = 256; in place when in == out.
// psums != nullptr: the channel sums of a frame arrive as `nsum` partial sums (one per unit / border tile of the tile plan:
// psums[frame][nsum][3], bevw_plan.h) and every block adds them up itself -- integer sums, the same value k_reduce_psums would have left in
// chsums, without that kernel between the stitch and this pass.
static __global__ void __launch_bounds__(256) k_gain_lut(const uint8_t *in, size_t npx, const unsigned long long *__restrict__ chsums,
                                                   const uint8_t *__restrict__ car, uint8_t *out, uint32_t blocks_per_frame,
                                                   uint32_t nframes, int f32 = 0, size_t npx_mean = 0, const uint32_t *__restrict__ psums = nullptr,
                                                   int nsum = 0)
{
    __shared__ uint8_t lut[3][256];
    __shared__ unsigned long long part[3][4];
    uint32_t frame, blk;
    if (!xcd_frame_map(blockIdx.x, blocks_per_frame, nframes, frame, blk)) return;   // grid: xcd_frame_grid()
    unsigned long long total[3];
    if (psums != nullptr) {
        const uint32_t *p = psums + (size_t)frame * nsum * 3;
        unsigned long long acc[3] = {0, 0, 0};
        for (int t = threadIdx.x; t < nsum; t += blockDim.x) { acc[0] += p[t * 3]; acc[1] += p[t * 3 + 1]; acc[2] += p[t * 3 + 2]; }
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned long long w = wave_sum_u64(acc[k]);
            if (lane == 0) part[k][wv] = w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k) total[k] = part[k][0] + part[k][1] + part[k][2] + part[k][3];
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) total[k] = chsums[frame * 3 + k];
    }
    {
        // npx_mean: pixels the channel means are taken over when the images carry padding columns (bevw_set_output_pitch); 0 = npx
        const double n = (double)(npx_mean ? npx_mean : npx);
        const double B = (double)total[0] / n, G = (double)total[1] / n, R = (double)total[2] / n;
        const double K = (R + G + B) / 3;
        const double gain[3] = {K / B, K / G, K / R};
        for (int i = threadIdx.x; i < 768; i += blockDim.x) {
            const int c = i >> 8, v = i & 255;
            lut[c][v] = (uint8_t)gain_px(v, gain[c], f32);
        }
    }
    __syncthreads();
    const size_t base = (size_t)frame * npx * 3, nq = npx / 4;
    for (size_t q = (size_t)blk * blockDim.x + threadIdx.x; q < nq; q += (size_t)blocks_per_frame * blockDim.x) {
        const uint32_t *ip = reinterpret_cast<const uint32_t *>(in + base + q * 12);
        uint32_t w[3] = {once_load<BEVW_GAIN_NT>(ip), once_load<BEVW_GAIN_NT>(ip + 1), once_load<BEVW_GAIN_NT>(ip + 2)}, cw[3] = {0, 0, 0}, o[3] = {0, 0, 0};
        if (car != nullptr) {
            const uint32_t *cp = reinterpret_cast<const uint32_t *>(car + q * 12);
            cw[0] = cp[0]; cw[1] = cp[1]; cw[2] = cp[2];
        }
#pragma unroll
        for (int bi = 0; bi < 12; ++bi) {
            const int c = bi % 3;
            uint32_t v = lut[c][(w[bi >> 2] >> ((bi & 3) * 8)) & 255u];
            v = min(255u, v + ((cw[bi >> 2] >> ((bi & 3) * 8)) & 255u));
            o[bi >> 2] |= v << ((bi & 3) * 8);
        }
        uint32_t *op = reinterpret_cast<uint32_t *>(out + base + q * 12);
        once_store<BEVW_GAIN_NT>(op, o[0]); once_store<BEVW_GAIN_NT>(op + 1, o[1]); once_store<BEVW_GAIN_NT>(op + 2, o[2]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Camera-per-GPU mode: a rank stitches only the cameras it owns, sends the bounding box of its
// masks, and the stitch rank adds the parts.  cv2.add saturates, so front + back + left + right (+ car) is
// min(255, sum) however the terms are grouped (surroundBEV.py:318-320, 323-324).
// ---------------------------------------------------------------------------------------------------------------

// deltas of the owned cameras in plan order: out[b][k] = deltas[b][cams[k]]
struct ShardCams { int cam[4]; int n; };
static __global__ void k_delta_select(const int *__restrict__ deltas, ShardCams sc, int nsets, int *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsets * 4) return;
    const int b = i >> 2, k = i & 3;
    out[i] = k < sc.n ? deltas[b * 4 + sc.cam[k]] : 0;
}

// full BEV [batch][bh][bw][3] -> packed box [batch][y1-y0][x1-x0][3].  U = uint32_t when every row start is dword
// aligned (bw % 4 == 0 and x0 % 4 == 0), else uint8_t.  grid = (ceil(row units / 256), box rows, batch)
template <typename U>
static __global__ void k_pack_box(const uint8_t *__restrict__ full, int bw, int bh, int x0, int y0, int x1, int y1, uint8_t *__restrict__ packed)
{
    const int row_units = (x1 - x0) * 3 / (int)sizeof(U);
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= row_units) return;
    const int y = blockIdx.y;
    const size_t b = blockIdx.z;
    const U *src = reinterpret_cast<const U *>(full + (b * bh + (size_t)(y0 + y)) * bw * 3 + (size_t)x0 * 3);
    U *dst = reinterpret_cast<U *>(packed + (b * (size_t)(y1 - y0) + y) * (size_t)(x1 - x0) * 3);
    dst[u] = src[u];
}

struct CombineParts {
    const uint8_t *p[8];
    int box[8][4];   // x0, y0, x1, y1 of each packed part
    int n;
};

__device__ inline uint32_t sat_add_u8x4(uint32_t a, uint32_t b)
{
    const uint32_t lo = (a & 0x00ff00ffu) + (b & 0x00ff00ffu), hi = ((a >> 8) & 0x00ff00ffu) + ((b >> 8) & 0x00ff00ffu);
    const uint32_t slo = (lo | (((lo >> 8) & 0x00010001u) * 0xffu)) & 0x00ff00ffu;
    const uint32_t shi = (hi | (((hi >> 8) & 0x00010001u) * 0xffu)) & 0x00ff00ffu;
    return slo | (shi << 8);
}

// out = min(255, sum of the parts covering the pixel (+ car)).  PX = 4: one thread = 4 pixels = 3 dwords (needs
// bw % 4 == 0 and boxes aligned to 4 pixels in x); PX = 1: one thread = one pixel, byte accesses.
// grid = (ceil(bw / PX / 256), bh, batch)
template <int PX>
static __global__ void k_combine(CombineParts parts, int bw, int bh, const uint8_t *__restrict__ car, uint8_t *__restrict__ out)
{
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * PX, y = blockIdx.y;
    if (x >= bw) return;
    const size_t b = blockIdx.z;
    const size_t o = ((size_t)y * bw + x) * 3;
    if (PX == 4) {
        uint32_t acc[3] = {0, 0, 0};
        for (int k = 0; k < parts.n; ++k) {
            const int *bx = parts.box[k];
            if (x < bx[0] || x >= bx[2] || y < bx[1] || y >= bx[3]) continue;
            const size_t pw = (size_t)(bx[2] - bx[0]), ph = (size_t)(bx[3] - bx[1]);
            const uint32_t *pp = reinterpret_cast<const uint32_t *>(parts.p[k] + ((b * ph + (size_t)(y - bx[1])) * pw + (size_t)(x - bx[0])) * 3);
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[i] = sat_add_u8x4(acc[i], pp[i]);
        }
        if (car != nullptr) {
            const uint32_t *cp = reinterpret_cast<const uint32_t *>(car + o);
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[i] = sat_add_u8x4(acc[i], cp[i]);
        }
        uint32_t *op = reinterpret_cast<uint32_t *>(out + b * (size_t)bw * bh * 3 + o);
        op[0] = acc[0]; op[1] = acc[1]; op[2] = acc[2];
    } else {
        unsigned acc[3] = {0, 0, 0};
        for (int k = 0; k < parts.n; ++k) {
            const int *bx = parts.box[k];
            if (x < bx[0] || x >= bx[2] || y < bx[1] || y >= bx[3]) continue;
            const size_t pw = (size_t)(bx[2] - bx[0]), ph = (size_t)(bx[3] - bx[1]);
            const uint8_t *pp = parts.p[k] + ((b * ph + (size_t)(y - bx[1])) * pw + (size_t)(x - bx[0])) * 3;
            for (int i = 0; i < 3; ++i) acc[i] = min(255u, acc[i] + pp[i]);
        }
        uint8_t *op = out + b * (size_t)bw * bh * 3 + o;
        for (int i = 0; i < 3; ++i) op[i] = (uint8_t)min(255u, acc[i] + (car != nullptr ? car[o + i] : 0u));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// ExCalibrator pre-processing warps (extrinsicCalib.py:54-59, 122-130)
// ---------------------------------------------------------------------------------------------------------------
// CenterImage.translate: cv2.warpAffine with an integer shift = one tap per pixel, zeros outside.
// grid = (ceil(w / 256), h, batch)
static __global__ void k_translate(const uint8_t *__restrict__ src, int w, int h, int shift_x, int shift_y, uint8_t *__restrict__ dst)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const size_t img = (size_t)blockIdx.z * w * h * 3;
    const int u = x - shift_x, v = y - shift_y;
    uint8_t *d = dst + img + ((size_t)y * w + x) * 3;
    if (u >= 0 && u < w && v >= 0 && v < h) {
        const uint8_t *p = src + img + ((size_t)v * w + u) * 3;
        d[0] = p[0]; d[1] = p[1]; d[2] = p[2];
    } else {
        d[0] = d[1] = d[2] = 0;
    }
}

// One axis of cv2.resize INTER_LINEAR (8U fixed point): source index and the two 11-bit weights of destination
// index d.  clamp_frac: columns zero the fraction when the tap pair leaves the image, rows clip the pair instead.
__device__ inline void resize_tap(int d, double scale, int n_src, bool clamp_frac, int &s, int &c0, int &c1)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (clamp_frac) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
    }
    c0 = rne_f((1.f - f) * 2048.f);
    c1 = rne_f(f * 2048.f);
}

// cv2.resize(src, (0,0), fx, fy), INTER_LINEAR, 8UC3 (ScaleImage.__call__, extrinsicCalib.py:125).
// grid = (ceil(dw / 256), dh, batch)
static __global__ void k_resize_linear(const uint8_t *__restrict__ src, int w, int h, double scale_x, double scale_y,
                                uint8_t *__restrict__ dst, int dw, int dh)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (dx >= dw) return;
    int s0, a0, a1, r0, b0, b1;
    resize_tap(dx, scale_x, w, true, s0, a0, a1);
    resize_tap(dy, scale_y, h, false, r0, b0, b1);
    const int s1 = min(s0 + 1, w - 1);
    const int r1 = min(max(r0 + 1, 0), h - 1);
    r0 = min(max(r0, 0), h - 1);
    const uint8_t *img = src + (size_t)blockIdx.z * w * h * 3;
    const uint8_t *p0 = img + (size_t)r0 * w * 3, *p1 = img + (size_t)r1 * w * 3;
    uint8_t *d = dst + ((size_t)blockIdx.z * dh * dw + (size_t)dy * dw + dx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int S0 = p0[s0 * 3 + c] * a0 + p0[s1 * 3 + c] * a1;
        const int S1 = p1[s0 * 3 + c] * a0 + p1[s1 * 3 + c] * a1;
        d[c] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
    }
}

// a plain copy: the measured yardstick of bevw_device_copy_rate (csrc/bevwarp.hip).  Round 6: kCopyDepth independent 16-byte loads in flight per
// lane before the first store (round 5's one-load-per-trip loop with dword-wise non-temporal accesses gave 4.8 - 5.1 TB/s where
// tools/hbm_stream.hip's unrolled copy gives 5.35); every block-trip moves kCopyDepth consecutive 4 KB pieces, blocks interleaved.
constexpr int kCopyDepth = 8;
typedef uint32_t copy_u32x4 __attribute__((ext_vector_type(4)));
template <int NT>
static __global__ void __launch_bounds__(256) k_copy16(const copy_u32x4 *__restrict__ src, copy_u32x4 *__restrict__ dst, size_t n)
{
    const size_t span = (size_t)kCopyDepth * 256;
    size_t base = (size_t)blockIdx.x * span;
    for (; base + span <= n; base += (size_t)gridDim.x * span) {
        copy_u32x4 v[kCopyDepth];
#pragma unroll
        for (int u = 0; u < kCopyDepth; ++u) {
            const copy_u32x4 *p = src + base + (size_t)u * 256 + threadIdx.x;
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < kCopyDepth; ++u) {
            copy_u32x4 *q = dst + base + (size_t)u * 256 + threadIdx.x;
            if (NT) __builtin_nontemporal_store(v[u], q);
            else *q = v[u];
        }
    }
    for (size_t i = base + threadIdx.x; i < n && base < n; i += 256) {   // the last, partial span (at most one block gets here with work)
        if (i >= base + span) break;
        dst[i] = src[i];
    }
}

}  // namespace bevw
