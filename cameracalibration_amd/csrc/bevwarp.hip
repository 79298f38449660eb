// bevwarp.hip -- libbevwarp.so: the C-ABI of include/bevwarp.h over the HIP kernels (gfx950 / MI355X only).
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared   (cameracalibration_amd/build.py: this file, bevwarp_plan.hip --
// the tile plan and its per-frame kernels -- and bevwarp_jpeg.hip -- the JPEG codec -- are compiled in parallel and linked together)
// There is no CPU path in this library: every pixel and every table entry is produced by a kernel.  The host
// code below only derives a handful of scalars per calibration (3x3 inverses, polygon vertices and edge slopes,
// the 512-entry HSV divisor tables, the per-column fp64 chain of the fisheye map) exactly the way the reference's
// Python / OpenCV host code does.
#include "bevw_host.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <new>
#include <string>
#include <thread>

#include "bevw_kernels.h"
#include "bevw_planapi.h"
#include "bevw_comm.h"

using namespace bevw;

// ---------------------------------------------------------------------------------------------------------------
// small host-side scalars
// ---------------------------------------------------------------------------------------------------------------
static int host_rne(double v)
{
    if (!(v > -2147483648.5 && v < 2147483647.5)) return INT_MIN;
    return (int)std::lrint(v);
}

// cv::invert on a 3x3 CV_64F matrix: cofactors scaled by 1/det (cv2.warpPerspective inverts H internally).
static bool invert3x3(const double m[9], double t[9])
{
    double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
               m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (d == 0.0) {
        memset(t, 0, 9 * sizeof(double));
        return false;
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
    return true;
}

// warpPerspective walks the destination in blocks of min(1024 / min(16, dh), dw) columns.
static int persp_block_width(int dw, int dh)
{
    int bh0 = dh < 16 ? dh : 16;
    if (bh0 < 1) bh0 = 1;
    int bw0 = 1024 / bh0;
    if (bw0 > dw) bw0 = dw;
    return bw0 < 1 ? 1 : bw0;
}

static HsvTables make_hsv_tables()
{
    HsvTables t;
    t.sdiv[0] = t.hdiv[0] = 0;
    for (int i = 1; i < 256; ++i) {
        t.sdiv[i] = host_rne((255 << 12) / (1. * i));
        t.hdiv[i] = host_rne((180 << 12) / (6. * i));
    }
    for (int i = 0; i < 256; ++i) hsv_hue_entry(i, t.hue[i].x, t.hue[i].y);
    return t;
}

static bool host_clip_segment(int w, int h, long long &x1, long long &y1, long long &x2, long long &y2)
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

// Edge table of cv2.fillPoly for one polygon (XY_SHIFT = 16): slopes come from the image-clipped end points,
// the y extent from the original ones.  Only scalars are produced here; pixels are written by k_poly_*.
// OpenCV-version-sensitive choices (include/bevwarp.h: bevw_set_compat); process-wide, read when tables are built / gains applied
static std::atomic<int> g_compat[BEVW_COMPAT_KEYS] = {{1}, {1}, {0}, {0}};   // the DEFAULTS of new handles; a handle snapshots them in bevw_build

static PolyJob make_poly_job(const int (*pts)[2], int npts, int w, int h, bool modern)
{
    PolyJob job;
    memset(&job, 0, sizeof job);
    job.npts = npts;
    // modern: OpenCV >= 4.5.2 edge collection (BEVW_COMPAT_FILLPOLY)
    job.ceil_left = modern ? 0 : 1;
    for (int i = 0; i < npts; ++i) { job.pts[i][0] = pts[i][0]; job.pts[i][1] = pts[i][1]; }
    const long long HALF = 1 << 15;
    long long p0x = (long long)pts[npts - 1][0] << 16, p0y = pts[npts - 1][1];
    for (int i = 0; i < npts; ++i) {
        long long p1x = (long long)pts[i][0] << 16, p1y = pts[i][1];
        long long c0x = p0x, c0y = p0y, c1x = p1x, c1y = p1y;
        long long t0x = (p0x + HALF) >> 16, t0y = p0y, t1x = (p1x + HALF) >> 16, t1y = p1y;
        if (!modern) {
            // before 4.5.2 the edge runs between the raw vertices, without the half-pixel offset
        } else if ((unsigned long long)t0x >= (unsigned long long)w || (unsigned long long)t1x >= (unsigned long long)w ||
            (unsigned long long)t0y >= (unsigned long long)h || (unsigned long long)t1y >= (unsigned long long)h) {
            host_clip_segment(w, h, t0x, t0y, t1x, t1y);
            if (t0y != t1y) {
                c0y = t0y; c1y = t1y;
                c0x = t0x << 16; c1x = t1x << 16;
            }
        } else {
            c0x += HALF; c1x += HALF;
        }
        if (p0y != p1y) {
            PolyEdge e;
            e.dx = (c1x - c0x) / (c1y - c0y);
            if (p0y < p1y) { e.y0 = (int)p0y; e.y1 = (int)p1y; e.x = c0x + (p0y - c0y) * e.dx; }
            else           { e.y0 = (int)p1y; e.y1 = (int)p0y; e.x = c1x + (p1y - c1y) * e.dx; }
            job.edges[job.nedges++] = e;
        }
        p0x = p1x; p0y = p1y;
    }
    return job;
}

// Mask.get_points / BlendMask.get_points / BlendMask.get_lines (surroundBEV.py:123-154, 190-229, 236-268):
// Python float expressions truncated by .astype(np.int32).
struct MaskGeometry {
    int bw, bh, cw, ch;
    int tr(double v) const { return (int)v; }
    void anchor(const char *name, int out[2]) const
    {
        const double BW = bw, BH = bh, CW = cw, CH = ch;
        struct { const char *n; double x, y; } tab[] = {
            {"O", 0, 0}, {"X", BW, 0}, {"Y", 0, BH}, {"XY", BW, BH},
            {"cTL", (BW - CW) / 2, (BH - CH) / 2}, {"cTR", (BW + CW) / 2, (BH - CH) / 2},
            {"cBL", (BW - CW) / 2, (BH + CH) / 2}, {"cBR", (BW + CW) / 2, (BH + CH) / 2},
            {"lT", 0, BH / 5}, {"rT", BW, BH / 5}, {"lB", 0, BH - BH / 5}, {"rB", BW, BH - BH / 5},
            {"tL", BW / 5, 0}, {"bL", BW / 5, BH}, {"tR", BW - BW / 5, 0}, {"bR", BW - BW / 5, BH},
        };
        for (auto &t : tab)
            if (strcmp(t.n, name) == 0) { out[0] = tr(t.x); out[1] = tr(t.y); return; }
        out[0] = out[1] = 0;
    }
    int polygon(int cam, bool blend, int pts[8][2]) const
    {
        static const char *direct[4][4] = {{"O", "X", "cTR", "cTL"}, {"Y", "XY", "cBR", "cBL"},
                                           {"O", "Y", "cBL", "cTL"}, {"X", "XY", "cBR", "cTR"}};
        static const char *hexa[4][6] = {{"O", "X", "rT", "cTR", "cTL", "lT"}, {"Y", "XY", "rB", "cBR", "cBL", "lB"},
                                         {"O", "Y", "bL", "cBL", "cTL", "tL"}, {"X", "XY", "bR", "cBR", "cTR", "tR"}};
        const int n = blend ? 6 : 4;
        for (int i = 0; i < n; ++i) anchor(blend ? hexa[cam][i] : direct[cam][i], pts[i]);
        return n;
    }
    Seam seam(const char *a, const char *b) const
    {
        Seam s;
        anchor(a, &s.p[0]);
        anchor(b, &s.p[2]);
        return s;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// device buffer helper
// ---------------------------------------------------------------------------------------------------------------
// cv2.fisheye.initUndistortRectifyMap on the device (see k_fisheye_map for why xs[] is a host-made chain).
static int build_fisheye_maps(hipStream_t st, const double K[9], const double D[4], const double Knew[9], int w, int h,
                              int16_t *d_map1, uint16_t *d_map2)
{
    const double fxn = Knew[0], fyn = Knew[4], cxn = Knew[2], cyn = Knew[5];
    const double iR0 = 1.0 / fxn, iR2 = -cxn / fxn;
    FisheyeParams p;
    p.fx = K[0]; p.fy = K[4]; p.cx = K[2]; p.cy = K[5];
    p.k0 = D[0]; p.k1 = D[1]; p.k2 = D[2]; p.k3 = D[3];
    p.iR4 = 1.0 / fyn; p.iR5 = -cyn / fyn;
    std::vector<double> xs((size_t)w);
    double x = 0 * 0.0 + iR2;
    for (int j = 0; j < w; ++j) { xs[j] = x; x += iR0; }
    DevBuf dxs;
    BEVW_TRY(dxs.reserve(sizeof(double) * (size_t)w));
    HIP_TRY(hipMemcpyAsync(dxs.p, xs.data(), sizeof(double) * (size_t)w, hipMemcpyHostToDevice, st));
    dim3 grid((w + 255) / 256, h);
    hipLaunchKernelGGL(k_fisheye_map, grid, dim3(256), 0, st, p, dxs.as<double>(), w, h, d_map1, d_map2);
    BEVW_TRY(launch_check("k_fisheye_map"));
    HIP_TRY(hipStreamSynchronize(st));
    dxs.release();
    return BEVW_OK;
}

// cv2.initUndistortRectifyMap (pinhole) on the device; returns BEVW_E_INVALID when K' has skew (the reference never
// builds one: intrinsicCalib.py:150-156)
static int build_pinhole_maps(hipStream_t st, const double K[9], const double D[8], const double Knew[9], int w, int h,
                              int16_t *d_map1, uint16_t *d_map2)
{
    double iR[9];
    if (!invert3x3(Knew, iR)) return fail(BEVW_E_INVALID, "new camera matrix is singular");
    if (iR[1] != 0.0 || iR[3] != 0.0 || iR[6] != 0.0 || iR[7] != 0.0)
        return fail(BEVW_E_INVALID, "new camera matrix with skew / perspective terms is not supported");
    PinholeParams p;
    p.fx = K[0]; p.fy = K[4]; p.u0 = K[2]; p.v0 = K[5];
    p.k1 = D[0]; p.k2 = D[1]; p.p1 = D[2]; p.p2 = D[3]; p.k3 = D[4]; p.k4 = D[5]; p.k5 = D[6]; p.k6 = D[7];
    p.iR4 = iR[4]; p.iR5 = iR[5]; p.iR8 = iR[8];
    std::vector<double> xs((size_t)w);
    double x = 0 * iR[1] + iR[2];
    for (int j = 0; j < w; ++j) { xs[j] = x; x += iR[0]; }
    DevBuf dxs;
    BEVW_TRY(dxs.reserve(sizeof(double) * (size_t)w));
    HIP_TRY(hipMemcpyAsync(dxs.p, xs.data(), sizeof(double) * (size_t)w, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_pinhole_map, dim3((w + 255) / 256, h), dim3(256), 0, st, p, dxs.as<double>(), w, h, d_map1, d_map2);
    BEVW_TRY(launch_check("k_pinhole_map"));
    HIP_TRY(hipStreamSynchronize(st));
    dxs.release();
    return BEVW_OK;
}

static void camera_mat_dst(const double K[9], int fw, int fh, double fs, double ss, double off_h, double off_v,
                           double Kd[9])
{
    memcpy(Kd, K, 9 * sizeof(double));
    Kd[0] *= fs;
    Kd[4] *= fs;
    Kd[2] = (double)fw / 2 * ss + off_h;
    Kd[5] = (double)fh / 2 * ss + off_v;
}

// ---------------------------------------------------------------------------------------------------------------
// bevw_remapper
// ---------------------------------------------------------------------------------------------------------------
struct bevw_remapper {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    LapTimer laps;
    int sw = 0, sh = 0, dw = 0, dh = 0;
    DevBuf map1, map2, in, out, ones;
    Plan plan;            // single-image contributor plan (same kernels as the BEV stitch, ncams = 1)
    bool plan_ready = false;
    int ties_even = 0;    // BEVW_COMPAT_REMAP at creation: half-to-even ties -> the per-pixel kernel (the plan's arithmetic rounds half up)
};

// cv2.remap as a 1-camera stitch: every destination pixel has exactly one contributor with mask 255.
static int remapper_build_plan(bevw_remapper *r)
{
    static const int use_plan = [] { const char *s = getenv("BEVW_REMAP_PLAN"); return s ? atoi(s) : 1; }();
    r->plan_ready = false;
    if (!use_plan || r->ties_even) return BEVW_OK;
    const size_t npx = (size_t)r->dw * r->dh;
    BEVW_TRY(r->ones.reserve(npx));
    HIP_TRY(hipMemsetAsync(r->ones.p, 0xff, npx, r->stream));
    StitchTables T;
    for (int i = 0; i < 4; ++i) { T.lut1[i] = r->map1.as<int16_t>(); T.lut2[i] = r->map2.as<uint16_t>(); T.mask[i] = r->ones.as<uint8_t>(); }
    BEVW_TRY(plan_build(r->plan, r->stream, T, r->sw, r->sh, r->dw, r->dh, 1, 0, false));
    r->ones.release();
    r->plan_ready = r->plan.usable;
    return BEVW_OK;
}

static int remapper_alloc(int device, int sw, int sh, int dw, int dh, bevw_remapper **out)
{
    if (!out) return fail(BEVW_E_INVALID, "null output pointer");
    *out = nullptr;
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return fail(BEVW_E_INVALID, "non-positive image size");
    if (dh > 65535 || sh > 65535) return fail(BEVW_E_INVALID, "image height > 65535 (rows ride in grid.y)");
    BEVW_TRY(use_device(device));
    bevw_remapper *r = new (std::nothrow) bevw_remapper();
    if (!r) return fail(BEVW_E_NOMEM, "out of host memory");
    r->device = device; r->sw = sw; r->sh = sh; r->dw = dw; r->dh = dh;
    r->ties_even = g_compat[BEVW_COMPAT_REMAP].load();
    int s = BEVW_OK;
    if (hipStreamCreate(&r->stream) != hipSuccess || hipEventCreate(&r->ev0) != hipSuccess ||
        hipEventCreate(&r->ev1) != hipSuccess)
        s = fail(BEVW_E_HIP, "stream/event creation failed");
    if (s == BEVW_OK) s = r->map1.reserve((size_t)dw * dh * 2 * sizeof(int16_t));
    if (s == BEVW_OK) s = r->map2.reserve((size_t)dw * dh * sizeof(uint16_t));
    if (s != BEVW_OK) { bevw_remapper_destroy(r); return s; }
    *out = r;
    return BEVW_OK;
}

static int remap_launch(hipStream_t st, const uint8_t *d_src, int sw, int sh, const int16_t *m1, const uint16_t *m2,
                        int dw, int dh, int batch, uint8_t *d_dst, int ties_even = 0)
{
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        dim3 grid((dw + 255) / 256, dh, nb);
        hipLaunchKernelGGL(k_remap_lut, grid, dim3(256), 0, st, d_src + (size_t)b0 * sw * sh * 3, sw, sh, m1, m2, dw, dh,
                           d_dst + (size_t)b0 * dw * dh * 3, ties_even);
    }
    return launch_check("k_remap_lut");
}

extern "C" {

int bevw_abi_version(void) { return BEVW_ABI_VERSION; }

int bevw_set_compat(int key, int value)
{
    if (key < 0 || key >= BEVW_COMPAT_KEYS) return fail(BEVW_E_INVALID, "unknown compatibility key %d", key);
    if (key == BEVW_COMPAT_WARP) {
        if (value < 0 || value >= kWarpModes || (value != 0 && !(value & kWarpF32)))
            return fail(BEVW_E_INVALID, "BEVW_COMPAT_WARP: 0 (classic) or an odd member number below %d", kWarpModes);
    } else if (value != 0 && value != 1) return fail(BEVW_E_INVALID, "compatibility value must be 0 or 1");
    g_compat[key].store(value);
    return BEVW_OK;
}

int bevw_get_compat(int key)
{
    if (key < 0 || key >= BEVW_COMPAT_KEYS) return fail(BEVW_E_INVALID, "unknown compatibility key %d", key);
    return g_compat[key].load();
}

const char *bevw_last_error(void) { return g_bevw_err; }

int bevw_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n < 0 ? 0 : n;
}

int bevw_device_name(int device, char *buf, size_t buflen)
{
    if (!buf || buflen == 0) return fail(BEVW_E_INVALID, "null buffer");
    BEVW_TRY(use_device(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return BEVW_OK;
}

int bevw_malloc(int device, size_t nbytes, void **dptr)
{
    if (!dptr) return fail(BEVW_E_INVALID, "null output pointer");
    *dptr = nullptr;
    BEVW_TRY(use_device(device));
    hipError_t e = hipMalloc(dptr, nbytes ? nbytes : 1);
    if (e != hipSuccess) return fail(BEVW_E_NOMEM, "hipMalloc(%zu) failed: %s", nbytes, hipGetErrorString(e));
    return BEVW_OK;
}
int bevw_free(int device, void *dptr)
{
    if (!dptr) return BEVW_OK;
    BEVW_TRY(use_device(device));
    HIP_TRY(hipFree(dptr));
    return BEVW_OK;
}
int bevw_memcpy_h2d(int device, void *dst, const void *src, size_t nbytes)
{
    BEVW_TRY(use_device(device));
    HIP_TRY(hipMemcpy(dst, src, nbytes, hipMemcpyHostToDevice));
    return BEVW_OK;
}
int bevw_memcpy_d2h(int device, void *dst, const void *src, size_t nbytes)
{
    BEVW_TRY(use_device(device));
    HIP_TRY(hipMemcpy(dst, src, nbytes, hipMemcpyDeviceToHost));
    return BEVW_OK;
}
int bevw_memset(int device, void *dst, int value, size_t nbytes)
{
    BEVW_TRY(use_device(device));
    HIP_TRY(hipMemset(dst, value, nbytes));
    HIP_TRY(hipDeviceSynchronize());
    return BEVW_OK;
}

// A plain device-to-device copy kernel and its rate: the yardstick SURVEY.md 8(d) asks for beside the 8 TB/s specification figure ("also
// report fraction of a measured device-copy kernel").  Eight 16-byte loads in flight per lane, blocks interleaved; streaming == 1: non-temporal
// loads and stores.
int bevw_device_copy_rate(int device, size_t nbytes, int reps, int streaming, double *gb_per_s_moved)
{
    if (!gb_per_s_moved || reps <= 0 || nbytes < 16) return fail(BEVW_E_INVALID, "bad argument");
    BEVW_TRY(use_device(device));
    DevBuf a, b;
    int s = a.reserve(nbytes);
    if (s == BEVW_OK) s = b.reserve(nbytes);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (s == BEVW_OK && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) s = fail(BEVW_E_HIP, "event creation failed");
    if (s == BEVW_OK && hipMemset(a.p, 0x5a, nbytes) != hipSuccess) s = fail(BEVW_E_HIP, "memset failed");
    if (s == BEVW_OK) {
        const size_t n = nbytes / 16;
        const dim3 grid(256 * 32), block(256);
        auto launch = [&] {
            if (streaming) hipLaunchKernelGGL(k_copy16<1>, grid, block, 0, nullptr, a.as<copy_u32x4>(), b.as<copy_u32x4>(), n);
            else hipLaunchKernelGGL(k_copy16<0>, grid, block, 0, nullptr, a.as<copy_u32x4>(), b.as<copy_u32x4>(), n);
        };
        launch();   // warm-up
        (void)hipEventRecord(e0, nullptr);
        for (int r = 0; r < reps; ++r) launch();
        (void)hipEventRecord(e1, nullptr);
        s = launch_check("k_copy16");
        float ms = 0.f;
        if (s == BEVW_OK && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)))
            s = fail(BEVW_E_HIP, "timing the copy failed");
        if (s == BEVW_OK) *gb_per_s_moved = 2.0 * (double)(n * 16) * reps / ((double)ms * 1e-3) / 1e9;   // bytes read + bytes written
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    a.release(); b.release();
    return s;
}

// ---- remapper -------------------------------------------------------------------------------------------------
int bevw_fisheye_remapper_create(int device, int frame_width, int frame_height, const double K[9], const double D[4],
                                 double focal_scale, double size_scale, double offset_h, double offset_v,
                                 bevw_remapper **out)
{
    if (!K || !D) return fail(BEVW_E_INVALID, "null K/D");
    const int dw = (int)(frame_width * size_scale), dh = (int)(frame_height * size_scale);
    bevw_remapper *r = nullptr;
    BEVW_TRY(remapper_alloc(device, frame_width, frame_height, dw, dh, &r));
    double Kd[9];
    camera_mat_dst(K, frame_width, frame_height, focal_scale, size_scale, offset_h, offset_v, Kd);
    int s = build_fisheye_maps(r->stream, K, D, Kd, dw, dh, r->map1.as<int16_t>(), r->map2.as<uint16_t>());
    if (s == BEVW_OK) s = remapper_build_plan(r);
    if (s != BEVW_OK) { bevw_remapper_destroy(r); return s; }
    *out = r;
    return BEVW_OK;
}

int bevw_pinhole_remapper_create(int device, int frame_width, int frame_height, const double K[9], const double *D, int n_dist,
                                 double focal_scale, double size_scale, double offset_h, double offset_v, bevw_remapper **out)
{
    if (!K || (!D && n_dist > 0) || n_dist < 0) return fail(BEVW_E_INVALID, "null K/D");
    if (n_dist > 8) return fail(BEVW_E_INVALID, "thin-prism / tilt distortion terms (more than 8 coefficients) are not supported");
    const int dw = (int)(frame_width * size_scale), dh = (int)(frame_height * size_scale);
    bevw_remapper *r = nullptr;
    BEVW_TRY(remapper_alloc(device, frame_width, frame_height, dw, dh, &r));
    double Kd[9], d8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n_dist; ++i) d8[i] = D[i];
    camera_mat_dst(K, frame_width, frame_height, focal_scale, size_scale, offset_h, offset_v, Kd);
    int s = build_pinhole_maps(r->stream, K, d8, Kd, dw, dh, r->map1.as<int16_t>(), r->map2.as<uint16_t>());
    if (s == BEVW_OK) s = remapper_build_plan(r);
    if (s != BEVW_OK) { bevw_remapper_destroy(r); return s; }
    *out = r;
    return BEVW_OK;
}

int bevw_remapper_from_maps(int device, int src_w, int src_h, const int16_t *map1, const uint16_t *map2, int dst_w,
                            int dst_h, bevw_remapper **out)
{
    if (!map1 || !map2) return fail(BEVW_E_INVALID, "null map");
    bevw_remapper *r = nullptr;
    BEVW_TRY(remapper_alloc(device, src_w, src_h, dst_w, dst_h, &r));
    hipError_t e = hipMemcpy(r->map1.p, map1, (size_t)dst_w * dst_h * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(r->map2.p, map2, (size_t)dst_w * dst_h * 2, hipMemcpyHostToDevice);
    if (e != hipSuccess) { bevw_remapper_destroy(r); return fail(BEVW_E_HIP, "map upload failed: %s", hipGetErrorString(e)); }
    int s = remapper_build_plan(r);
    if (s != BEVW_OK) { bevw_remapper_destroy(r); return s; }
    *out = r;
    return BEVW_OK;
}

int bevw_remapper_dims(bevw_remapper *r, int32_t dims[4])
{
    if (!r || !dims) return fail(BEVW_E_INVALID, "null argument");
    dims[0] = r->sw; dims[1] = r->sh; dims[2] = r->dw; dims[3] = r->dh;
    return BEVW_OK;
}

int bevw_remapper_get_maps(bevw_remapper *r, int16_t *map1, uint16_t *map2)
{
    if (!r) return fail(BEVW_E_INVALID, "null remapper");
    BEVW_TRY(use_device(r->device));
    HIP_TRY(hipStreamSynchronize(r->stream));
    if (map1) HIP_TRY(hipMemcpy(map1, r->map1.p, (size_t)r->dw * r->dh * 4, hipMemcpyDeviceToHost));
    if (map2) HIP_TRY(hipMemcpy(map2, r->map2.p, (size_t)r->dw * r->dh * 2, hipMemcpyDeviceToHost));
    return BEVW_OK;
}

int bevw_remap_device(bevw_remapper *r, const void *d_src, int batch, void *d_dst)
{
    if (!r || !d_src || !d_dst || batch < 0) return fail(BEVW_E_INVALID, "bad argument");
    if (batch == 0) return BEVW_OK;
    BEVW_TRY(use_device(r->device));
    if (r->plan_ready && ((((uintptr_t)d_src) | ((uintptr_t)d_dst)) & 3u) == 0)
        return plan_stitch(r->plan, r->stream, (const uint8_t *)d_src, batch, false, false, nullptr, nullptr, nullptr, nullptr,
                           (uint8_t *)d_dst);
    return remap_launch(r->stream, (const uint8_t *)d_src, r->sw, r->sh, r->map1.as<int16_t>(), r->map2.as<uint16_t>(),
                        r->dw, r->dh, batch, (uint8_t *)d_dst, r->ties_even);
}

int bevw_remap(bevw_remapper *r, const uint8_t *src, int batch, uint8_t *dst)
{
    if (!r || !src || !dst || batch < 0) return fail(BEVW_E_INVALID, "bad argument");
    if (batch == 0) return BEVW_OK;
    BEVW_TRY(use_device(r->device));
    const size_t nin = (size_t)batch * r->sw * r->sh * 3, nout = (size_t)batch * r->dw * r->dh * 3;
    BEVW_TRY(r->in.reserve(nin));
    BEVW_TRY(r->out.reserve(nout));
    HIP_TRY(hipMemcpyAsync(r->in.p, src, nin, hipMemcpyHostToDevice, r->stream));
    BEVW_TRY(bevw_remap_device(r, r->in.p, batch, r->out.p));
    HIP_TRY(hipMemcpyAsync(dst, r->out.p, nout, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return BEVW_OK;
}

int bevw_remapper_sync(bevw_remapper *r)
{
    if (!r) return fail(BEVW_E_INVALID, "null remapper");
    BEVW_TRY(use_device(r->device));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return BEVW_OK;
}
int bevw_remapper_timer_start(bevw_remapper *r)
{
    if (!r) return fail(BEVW_E_INVALID, "null remapper");
    BEVW_TRY(use_device(r->device));
    HIP_TRY(hipEventRecord(r->ev0, r->stream));
    return BEVW_OK;
}
int bevw_remapper_timer_mark(bevw_remapper *r, int slot)
{
    if (!r) return fail(BEVW_E_INVALID, "null remapper");
    BEVW_TRY(use_device(r->device));
    return r->laps.mark(slot, r->stream);
}
int bevw_remapper_timer_between(bevw_remapper *r, int slot_a, int slot_b, float *elapsed_ms)
{
    if (!r) return fail(BEVW_E_INVALID, "null remapper");
    BEVW_TRY(use_device(r->device));
    return r->laps.between(slot_a, slot_b, elapsed_ms);
}
int bevw_remapper_timer_stop(bevw_remapper *r, float *elapsed_ms)
{
    if (!r || !elapsed_ms) return fail(BEVW_E_INVALID, "null argument");
    BEVW_TRY(use_device(r->device));
    HIP_TRY(hipEventRecord(r->ev1, r->stream));
    HIP_TRY(hipEventSynchronize(r->ev1));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, r->ev0, r->ev1));
    return BEVW_OK;
}

void bevw_remapper_destroy(bevw_remapper *r)
{
    if (!r) return;
    if (hipSetDevice(r->device) == hipSuccess) {
        if (r->stream) (void)hipStreamSynchronize(r->stream);
        r->map1.release(); r->map2.release(); r->in.release(); r->out.release(); r->ones.release();
        plan_release(r->plan);
        r->laps.release();
        if (r->ev0) (void)hipEventDestroy(r->ev0);
        if (r->ev1) (void)hipEventDestroy(r->ev1);
        if (r->stream) (void)hipStreamDestroy(r->stream);
    }
    delete r;
}

int bevw_warp_perspective_u8c3(int device, const uint8_t *src, int src_w, int src_h, const double H[9], int dst_w,
                               int dst_h, int batch, uint8_t *dst)
{
    if (!src || !dst || !H || src_w <= 0 || src_h <= 0 || dst_w <= 0 || dst_h <= 0 || batch < 0)
        return fail(BEVW_E_INVALID, "bad argument");
    if (batch == 0) return BEVW_OK;
    BEVW_TRY(use_device(device));
    Mat3 Minv;
    invert3x3(H, Minv.m);
    DevBuf in, out;
    const size_t nin = (size_t)batch * src_w * src_h * 3, nout = (size_t)batch * dst_w * dst_h * 3;
    int s = in.reserve(nin);
    if (s == BEVW_OK) s = out.reserve(nout);
    if (s == BEVW_OK && hipMemcpy(in.p, src, nin, hipMemcpyHostToDevice) != hipSuccess) s = fail(BEVW_E_HIP, "H2D failed");
    if (s == BEVW_OK) {
        for (int b0 = 0; b0 < batch; b0 += 65535) {
            const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
            dim3 grid((dst_w + 255) / 256, dst_h, nb);
            hipLaunchKernelGGL(k_warp_perspective, grid, dim3(256), 0, 0, in.as<uint8_t>() + (size_t)b0 * src_w * src_h * 3,
                               src_w, src_h, Minv, persp_block_width(dst_w, dst_h), dst_w, dst_h,
                               out.as<uint8_t>() + (size_t)b0 * dst_w * dst_h * 3, g_compat[BEVW_COMPAT_WARP].load());
        }
        s = launch_check("k_warp_perspective");
    }
    if (s == BEVW_OK && hipMemcpy(dst, out.p, nout, hipMemcpyDeviceToHost) != hipSuccess) s = fail(BEVW_E_HIP, "D2H failed");
    in.release();
    out.release();
    return s;
}

}  // extern "C"

// CenterImage.translate (extrinsicCalib.py:54-59)
int bevw_translate_u8c3(int device, const uint8_t *src, int width, int height, int shift_x, int shift_y, int batch, uint8_t *dst)
{
    if (!src || !dst || width <= 0 || height <= 0 || batch < 0) return fail(BEVW_E_INVALID, "bad argument");
    if (batch == 0) return BEVW_OK;
    if (batch > 65535) return fail(BEVW_E_INVALID, "batch > 65535");
    BEVW_TRY(use_device(device));
    const size_t n = (size_t)batch * width * height * 3;
    DevBuf d_src, d_dst;
    int s = d_src.reserve(n);
    if (s == BEVW_OK) s = d_dst.reserve(n);
    if (s == BEVW_OK && hipMemcpy(d_src.p, src, n, hipMemcpyHostToDevice) != hipSuccess) s = fail(BEVW_E_HIP, "H2D copy failed");
    if (s == BEVW_OK) {
        hipLaunchKernelGGL(k_translate, dim3((width + 255) / 256, height, batch), dim3(256), 0, nullptr, d_src.as<uint8_t>(), width,
                           height, shift_x, shift_y, d_dst.as<uint8_t>());
        s = launch_check("k_translate");
    }
    if (s == BEVW_OK && hipMemcpy(dst, d_dst.p, n, hipMemcpyDeviceToHost) != hipSuccess) s = fail(BEVW_E_HIP, "D2H copy failed");
    d_src.release(); d_dst.release();
    return s;
}

// cv2.resize's output size for dsize = (0, 0): (cvRound(w * fx), cvRound(h * fy))
int bevw_resize_dsize(int src_w, int src_h, double fx, double fy, int32_t dsize[2])
{
    if (!dsize || src_w <= 0 || src_h <= 0 || !(fx > 0) || !(fy > 0)) return fail(BEVW_E_INVALID, "bad argument");
    dsize[0] = host_rne((double)src_w * fx);
    dsize[1] = host_rne((double)src_h * fy);
    if (dsize[0] <= 0 || dsize[1] <= 0) return fail(BEVW_E_INVALID, "empty destination");
    return BEVW_OK;
}

// ScaleImage.__call__'s cv2.resize (extrinsicCalib.py:125)
int bevw_resize_linear_u8c3(int device, const uint8_t *src, int src_w, int src_h, double fx, double fy, int batch, uint8_t *dst)
{
    int32_t ds[2];
    BEVW_TRY(bevw_resize_dsize(src_w, src_h, fx, fy, ds));
    if (!src || !dst || batch < 0) return fail(BEVW_E_INVALID, "bad argument");
    if (batch == 0) return BEVW_OK;
    if (batch > 65535 || ds[1] > 65535) return fail(BEVW_E_INVALID, "batch or destination height > 65535");
    BEVW_TRY(use_device(device));
    const size_t nin = (size_t)batch * src_w * src_h * 3, nout = (size_t)batch * ds[0] * ds[1] * 3;
    DevBuf d_src, d_dst;
    int s = d_src.reserve(nin);
    if (s == BEVW_OK) s = d_dst.reserve(nout);
    if (s == BEVW_OK && hipMemcpy(d_src.p, src, nin, hipMemcpyHostToDevice) != hipSuccess) s = fail(BEVW_E_HIP, "H2D copy failed");
    if (s == BEVW_OK) {
        hipLaunchKernelGGL(k_resize_linear, dim3((ds[0] + 255) / 256, ds[1], batch), dim3(256), 0, nullptr, d_src.as<uint8_t>(), src_w,
                           src_h, 1.0 / fx, 1.0 / fy, d_dst.as<uint8_t>(), ds[0], ds[1]);
        s = launch_check("k_resize_linear");
    }
    if (s == BEVW_OK && hipMemcpy(dst, d_dst.p, nout, hipMemcpyDeviceToHost) != hipSuccess) s = fail(BEVW_E_HIP, "D2H copy failed");
    d_src.release(); d_dst.release();
    return s;
}

// ---------------------------------------------------------------------------------------------------------------
// bevw_handle
// ---------------------------------------------------------------------------------------------------------------
struct bevw_handle {
    bevw_config cfg;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;    // balance: the odd slices of a batch (balance_plan_run)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_skew = nullptr, ev_join = nullptr;
    LapTimer laps;
    bool cam_set[4] = {false, false, false, false};
    double K[4][9], D[4][4], H[4][9];
    bool built = false;
    int uw = 0, uh = 0;
    DevBuf und1[4], und2[4], lut1[4], lut2[4], mask[4];
    DevBuf hsv, vsums, deltas, chsums;
    DevBuf in, out, car, tmp;
    DevBuf pre;           // balance: the pre-gain BEV (the gain pass runs out of place)
    Plan plan;
    int schedule_in_use = BEVW_SCHED_PER_PIXEL;
    int projection = BEVW_PROJ_LUT;   // bevw_set_projection
    int compat[BEVW_COMPAT_KEYS] = {1, 1, 0, 0};   // bevw_set_compat values at bevw_build: a handle keeps the arithmetic it was built with
    int pitch_request = BEVW_PITCH_DENSE;   // bevw_set_output_pitch
    int pitch_px = 0;                 // pixels per row of the device-side BEV images (== bev_width unless a pitch was requested)
    DevBuf car_pitched;               // the car sprite with rows of pitch_px pixels (gain pass of a pitched handle)
    AnalyticRig arig;                 // filled by bevw_build
    Plan aplan;                       // analytic modes: the WIDE unit schedule compiled from the projection (analytic_units_build)
    int aplan_mode = -1;              // the projection mode aplan was compiled for (-1: none yet)
    // camera-per-GPU mode: the cameras this handle owns (shard_n == 0: all four, the ordinary BevGenerator)
    int shard_n = 0;
    int shard_cams[4] = {0, 1, 2, 3};
    int shard_box[4] = {0, 0, 0, 0};   // x0, y0, x1, y1: bounding box of the owned masks
    DevBuf sdeltas;
    DevBuf xchg;          // RCCL staging (rank-major V sums)
    bool owns(int cam) const
    {
        if (shard_n == 0) return true;
        for (int k = 0; k < shard_n; ++k) if (shard_cams[k] == cam) return true;
        return false;
    }
};

hipStream_t bevw_internal_handle_stream(bevw_handle *h) { return h->stream; }
int bevw_internal_handle_device(bevw_handle *h) { return h->cfg.device; }

static int fill_poly_device(hipStream_t st, const MaskGeometry &g, int cam, bool blend, uint8_t *d_mask, bool modern)
{
    int pts[8][2];
    const int n = g.polygon(cam, blend, pts);
    PolyJob job = make_poly_job(pts, n, g.bw, g.bh, modern);
    HIP_TRY(hipMemsetAsync(d_mask, 0, (size_t)g.bw * g.bh, st));
    hipLaunchKernelGGL(k_poly_outline, dim3(1), dim3(64), 0, st, job, d_mask, g.bw, g.bh, (uint8_t)255);
    hipLaunchKernelGGL(k_poly_fill, dim3((g.bh + 63) / 64), dim3(64), 0, st, job, d_mask, g.bw, g.bh, (uint8_t)255);
    return launch_check("k_poly_fill");
}

constexpr int kVsumParts = 256;   // blocks per frame of k_vsum at most (luminance_stats): partial sums per frame in bevw_handle::vsums

static int ensure_stats(bevw_handle *h, int batch)
{
    BEVW_TRY(h->vsums.reserve(sizeof(unsigned long long) * 4 * (size_t)batch * kVsumParts));   // per frame: the partial V sums of k_vsum's blocks
    BEVW_TRY(h->deltas.reserve(sizeof(int) * 4 * (size_t)batch));
    BEVW_TRY(h->chsums.reserve(sizeof(unsigned long long) * 3 * (size_t)batch));
    return BEVW_OK;
}

static int stitch_per_pixel(bevw_handle *h, const uint8_t *d_frames, int batch, const uint8_t *d_car, uint8_t *d_out)
{
    const bevw_config &c = h->cfg;
    StitchTables T;
    for (int i = 0; i < 4; ++i) {
        T.lut1[i] = h->lut1[i].as<int16_t>();
        T.lut2[i] = h->lut2[i].as<uint16_t>();
        T.mask[i] = h->mask[i].as<uint8_t>();
    }
    const int *deltas = h->deltas.as<int>();
    const HsvTables *tab = h->hsv.as<HsvTables>();
    unsigned long long *chs = h->chsums.as<unsigned long long>();
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        dim3 grid((c.bev_width + 255) / 256, c.bev_height, nb), block(256);
        const uint8_t *fr = d_frames + (size_t)b0 * 4 * c.frame_width * c.frame_height * 3;
        uint8_t *o = d_out + (size_t)b0 * c.bev_width * c.bev_height * 3;
#define LAUNCH_PP(BL, BA)                                                                                         \
        hipLaunchKernelGGL((k_stitch_pp<BL, BA>), grid, block, 0, h->stream, fr, c.frame_width, c.frame_height, T, \
                           c.bev_width, c.bev_height, deltas ? deltas + b0 * 4 : nullptr, tab, d_car,              \
                           chs ? chs + b0 * 3 : nullptr, o, h->compat[BEVW_COMPAT_REMAP])
        if (c.blend && c.balance) LAUNCH_PP(true, true);
        else if (c.blend) LAUNCH_PP(true, false);
        else if (c.balance) LAUNCH_PP(false, true);
        else LAUNCH_PP(false, false);
#undef LAUNCH_PP
    }
    return launch_check("k_stitch_pp");
}

// Analytic modes on the unit schedule: the projection of every BEV pixel is evaluated ONCE per handle and mode on
// the GPU (k_analytic_map, fp64 or fp32), the host compiles a WIDE unit plan from it (bevw_unit.h: 21-bit fractions, fp32 interpolation
// from the LDS patch), and the map is dropped.  Base tiles the units do not take (frame-border footprints) stay on k_stitch_analytic.
// Leaves h->aplan without units when the geometry does not allow them (the caller then runs the per-pixel kernel on everything).
static int analytic_units_build(bevw_handle *h)
{
    const bevw_config &c = h->cfg;
    plan_release(h->aplan);
    h->aplan_mode = h->projection;
    static const int units_env = [] { const char *s = getenv("BEVW_ANALYTIC_UNITS"); return s ? atoi(s) : 1; }();
    const int fw = c.frame_width, fh = c.frame_height, bw = c.bev_width, bh = c.bev_height;
    if (!units_env || bw % 4 != 0 || fw % 4 != 0 || (size_t)fw * fh * 12 >= (1ull << 31) || fw > 32767 || fh > 32767) return BEVW_OK;
    const size_t bpx = (size_t)bw * bh;
    DevBuf d_sxy, d_frac;
    BEVW_TRY(d_sxy.reserve(bpx * 4));
    BEVW_TRY(d_frac.reserve(bpx * 8));
    std::vector<int16_t> h1[4];
    std::vector<uint8_t> hm[4];
    std::vector<uint32_t> hf[4];
    for (int cam = 0; cam < 4; ++cam) {
        const dim3 grid((bw + 255) / 256, bh), block(256);
        if (h->projection == BEVW_PROJ_ANALYTIC_F32)
            hipLaunchKernelGGL((k_analytic_map<float>), grid, block, 0, h->stream, h->arig, cam, fw, fh, bw, bh, d_sxy.as<int16_t>(), d_frac.as<uint32_t>());
        else
            hipLaunchKernelGGL((k_analytic_map<double>), grid, block, 0, h->stream, h->arig, cam, fw, fh, bw, bh, d_sxy.as<int16_t>(), d_frac.as<uint32_t>());
        BEVW_TRY(launch_check("k_analytic_map"));
        h1[cam].resize(bpx * 2); hf[cam].resize(bpx * 2); hm[cam].resize(bpx);
        HIP_TRY(hipMemcpyAsync(h1[cam].data(), d_sxy.p, bpx * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(hf[cam].data(), d_frac.p, bpx * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(hm[cam].data(), h->mask[cam].p, bpx, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return plan_build_wide(h->aplan, h1, hf, hm, fw, fh, bw, bh, c.blend != 0);
}

// BEVW_PROJ_ANALYTIC(_F32): the unit schedule compiled from the projection (analytic_units_build), or -- balance handles, odd geometry,
// unaligned buffers -- the per-pixel stitch with the projection evaluated in the kernel (k_stitch_analytic)
static int stitch_analytic(bevw_handle *h, const uint8_t *d_frames, int batch, const uint8_t *d_car, uint8_t *d_out)
{
    const bevw_config &c = h->cfg;
    const uint32_t *left_tiles = nullptr;
    int n_left = 0;
    if (!c.balance && ((((uintptr_t)d_out | (uintptr_t)d_car | (uintptr_t)d_frames) & 3u) == 0)) {
        if (h->aplan_mode != h->projection) BEVW_TRY(analytic_units_build(h));
        if (h->aplan.n_un_all) {
            BEVW_TRY(plan_stitch_wide(h->aplan, h->stream, d_frames, batch, c.blend != 0, d_car, d_out));
            if (h->aplan.n_slow == 0) return BEVW_OK;
            left_tiles = static_cast<const uint32_t *>(h->aplan.list_slow);
            n_left = h->aplan.n_slow;
        }
    }
    StitchTables T;
    for (int i = 0; i < 4; ++i) {
        T.lut1[i] = h->lut1[i].as<int16_t>();
        T.lut2[i] = h->lut2[i].as<uint16_t>();
        T.mask[i] = h->mask[i].as<uint8_t>();
    }
    const int *deltas = h->deltas.as<int>();
    const HsvTables *tab = h->hsv.as<HsvTables>();
    unsigned long long *chs = h->chsums.as<unsigned long long>();
    // frames a thread samples with one evaluation of the projection (BEVW_ANALYTIC_FRAMES, read once per process; 1 = the projection per
    // output pixel AND frame: bench.py's direct_stitch_analytic_perpixel_b64)
    static const int fpt_env = [] { const char *s = getenv("BEVW_ANALYTIC_FRAMES"); return s ? atoi(s) : 0; }();
    const int fpt = fpt_env >= 1 && fpt_env <= 1024 ? fpt_env : kAnalyticFrames;
    const int per_launch = 65535 * fpt;
    for (int b0 = 0; b0 < batch; b0 += per_launch) {
        const int nb = batch - b0 < per_launch ? batch - b0 : per_launch;
        dim3 grid((c.bev_width + 255) / 256, c.bev_height, (nb + fpt - 1) / fpt), block(256);
        if (left_tiles) grid = dim3((unsigned)n_left, 1, (nb + fpt - 1) / fpt);
        const int tiles_x = h->aplan.tiles_x;
        const uint8_t *fr = d_frames + (size_t)b0 * 4 * c.frame_width * c.frame_height * 3;
        uint8_t *o = d_out + (size_t)b0 * c.bev_width * c.bev_height * 3;
        if (fpt == 1 && !c.balance && left_tiles == nullptr) {
            // one thread per pixel AND frame: the fused per-output-pixel kernel of its own (bevw_kernels.h: k_stitch_perpixel)
            const dim3 g1((c.bev_width + 255) / 256, c.bev_height, nb);
            if (h->projection == BEVW_PROJ_ANALYTIC_F32) {
                if (c.blend) hipLaunchKernelGGL((k_stitch_perpixel<true, float>), g1, block, 0, h->stream, fr, c.frame_width, c.frame_height, h->arig, T, c.bev_width, c.bev_height, d_car, o);
                else hipLaunchKernelGGL((k_stitch_perpixel<false, float>), g1, block, 0, h->stream, fr, c.frame_width, c.frame_height, h->arig, T, c.bev_width, c.bev_height, d_car, o);
            } else {
                if (c.blend) hipLaunchKernelGGL((k_stitch_perpixel<true, double>), g1, block, 0, h->stream, fr, c.frame_width, c.frame_height, h->arig, T, c.bev_width, c.bev_height, d_car, o);
                else hipLaunchKernelGGL((k_stitch_perpixel<false, double>), g1, block, 0, h->stream, fr, c.frame_width, c.frame_height, h->arig, T, c.bev_width, c.bev_height, d_car, o);
            }
            continue;
        }
#define LAUNCH_AN(BL, BA)                                                                                                   \
        do {                                                                                                                 \
            if (h->projection == BEVW_PROJ_ANALYTIC_F32)                                                                     \
                hipLaunchKernelGGL((k_stitch_analytic<BL, BA, float>), grid, block, 0, h->stream, fr, c.frame_width, c.frame_height, h->arig, T, \
                                   c.bev_width, c.bev_height, nb, deltas ? deltas + b0 * 4 : nullptr, tab, d_car, chs ? chs + b0 * 3 : nullptr, o, left_tiles, tiles_x, fpt); \
            else                                                                                                             \
                hipLaunchKernelGGL((k_stitch_analytic<BL, BA, double>), grid, block, 0, h->stream, fr, c.frame_width, c.frame_height, h->arig, T, \
                                   c.bev_width, c.bev_height, nb, deltas ? deltas + b0 * 4 : nullptr, tab, d_car, chs ? chs + b0 * 3 : nullptr, o, left_tiles, tiles_x, fpt); \
        } while (0)
        if (c.blend && c.balance) LAUNCH_AN(true, true);
        else if (c.blend) LAUNCH_AN(true, false);
        else if (c.balance) LAUNCH_AN(false, true);
        else LAUNCH_AN(false, false);
#undef LAUNCH_AN
    }
    return launch_check("k_stitch_analytic");
}

// luminance statistics of a batch of 4-camera sets -> deltas[batch][4].  d_vsums: kVsumParts entries per frame (ensure_stats): every block
// of k_vsum stores its partial sum, k_lum_delta adds them -- no atomics and no zeroing pass per step (round 5: the 4 KB hipMemsetAsync in
// front of every slice's k_vsum cost 20 us of stream time, twice per config-4 step).
static int luminance_stats(hipStream_t st, const uint8_t *d_frames, int nsets, int fw, int fh, unsigned long long *d_vsums, int *d_deltas)
{
    const size_t frame_bytes = (size_t)fw * fh * 3;
    const int nframes = nsets * 4;
    const int vec_ok = (frame_bytes % 4 == 0 && ((uintptr_t)d_frames & 3u) == 0) ? 1 : 0;   // k_vsum's 12-byte loads
    int bpf = 2048 / (nframes > 0 ? nframes : 1);
    if (bpf < 8) bpf = 8;
    if (bpf > 256) bpf = 256;
    for (int f0 = 0; f0 < nframes; f0 += 65535) {
        const int nf = nframes - f0 < 65535 ? nframes - f0 : 65535;
        hipLaunchKernelGGL(k_vsum, dim3(bpf, nf), dim3(256), 0, st, d_frames + (size_t)f0 * frame_bytes, frame_bytes, vec_ok,
                           d_vsums + (size_t)f0 * kVsumParts, kVsumParts);
    }
    hipLaunchKernelGGL(k_lum_delta, dim3((nsets + 63) / 64), dim3(64), 0, st, d_vsums, (double)fw * (double)fh, nsets,
                       d_deltas, bpf, kVsumParts);
    return launch_check("k_vsum/k_lum_delta");
}

// The gain pass of frame sets [b0, b0 + n) (color_balance + car, surroundBEV.py:43-55, 323-324) on `st`; gain_in = the pre-gain image of
// frame set b0 (the slice's own buffer, or frame b0's place in a whole-batch one: the caller decides).
// from_plan: the channel sums are the partial sums the tile plan's stitch of these frames left behind (k_gain_lut adds them itself)
static int gain_pass(bevw_handle *h, hipStream_t st, const uint8_t *gain_in, const uint8_t *gain_car, const uint8_t *d_car, uint8_t *d_out, int b0, int n,
                     bool lut_ok, bool from_plan = false)
{
    const bevw_config &c = h->cfg;
    const size_t npx_true = (size_t)c.bev_width * c.bev_height, npx = (size_t)h->pitch_px * c.bev_height;
    for (int k0 = b0; k0 < b0 + n; k0 += 65535) {
        const int nb = b0 + n - k0 < 65535 ? b0 + n - k0 : 65535;
        int nsum = 0;
        const uint32_t *ps = (from_plan && lut_ok) ? plan_sum_entries(h->plan, k0, nsum) : nullptr;
        if (lut_ok)
            hipLaunchKernelGGL(k_gain_lut, dim3(xcd_frame_grid(32, (unsigned)nb)), dim3(256), 0, st, gain_in + (size_t)(k0 - b0) * npx * 3, npx,
                               h->chsums.as<unsigned long long>() + (size_t)k0 * 3, gain_car, d_out + (size_t)k0 * npx * 3, 32u,
                               (uint32_t)nb, h->compat[BEVW_COMPAT_ADDWEIGHTED] ? 0 : 1, npx_true, ps, nsum);
        else
            hipLaunchKernelGGL(k_gain, dim3(64, nb), dim3(256), 0, st, d_out + (size_t)k0 * npx * 3, npx,
                               h->chsums.as<unsigned long long>() + (size_t)k0 * 3, d_car, d_out + (size_t)k0 * npx * 3,
                               h->compat[BEVW_COMPAT_ADDWEIGHTED] ? 0 : 1);
    }
    return launch_check("k_gain");
}

// blend + balance on the tile plan (BASELINE config 4), `parts` slices of the batch alternating over the handle's two streams.
// Per slice: V sums of the raw frames (k_vsum: HBM-bound, reads every byte of the four frames) -> deltas -> luminance round trip of the
// sampled texel groups into the compact scratch (k_lum_groups: balanced between its 1.3 GB and its arithmetic) -> the unit stitch with per-unit channel sums
// -> the gain pass (copy rate).  In one stream these run strictly one after the other, a memory-bound kernel while the VALUs idle and a
// VALU-bound one while the memory idles; every quantity is per frame set, so slices are independent, and a slice's kernels overlap
// the neighbouring slice's kernels of the OTHER kind (round 4: profiles/r04/ab_config4_slices.log; same idea as the two slices of a JPEG
// decode batch, bevwarp_jpeg.hip).  BEVW_BAL_SKEW=1 starts the second stream one V-sum pass late (measured: no better).
// (Round 2 measured sub-batches of 16 ... 128 frame sets run ONE AFTER THE OTHER for Infinity-Cache residency: slower, the small grids
// cost more than the cache returns; profiles/r02/sweeps.log.)
static int balance_plan_run(bevw_handle *h, const uint8_t *d_frames, int batch, const uint8_t *d_car, uint8_t *d_out)
{
    const bevw_config &c = h->cfg;
    const bool pitched = h->pitch_px != c.bev_width;
    const size_t npx = (size_t)h->pitch_px * c.bev_height;
    const size_t set_bytes = (size_t)c.frame_width * c.frame_height * 12;
    BEVW_TRY(ensure_stats(h, batch));
    const size_t cstride = h->plan.compact_stride;   // the compact scratch: only the sampled texel groups of a frame set (bevw_unit.h: unit_gsrc_compact)
    static const int parts_env = [] { const char *s = getenv("BEVW_BAL_PARTS"); return s ? atoi(s) : 0; }();
    static const int skew_env = [] { const char *s = getenv("BEVW_BAL_SKEW"); return s ? atoi(s) : 0; }();
    // slices share the plan's padded scratch image when the BEV width is not a multiple of 4 pixels: one slice then
    const bool scratch = h->plan.pitch != h->plan.bw && !h->plan.out_pitched;
    // measured (profiles/r04/ab_config4_slices.log, batch 256): 1 slice 2.172 ms, 2 slices 2.142 (2.171 with the second stream one V-sum pass late),
    // 4 slices 2.22 - 2.24, 8 slices 2.30: three of the four kernels are HBM-bound and the VALU-bound one still moves 1.35 GB, so running
    // them side by side shares the memory instead of filling idle time -- the overlap is worth 1.4 %, more slices cost it again in small grids
    int parts = parts_env > 0 ? parts_env : (batch >= 32 ? 2 : 1);
    if (scratch) parts = 1;
    if (parts > batch) parts = batch;
    if (parts > 1 && !h->stream2 && hipStreamCreate(&h->stream2) != hipSuccess) { (void)hipGetLastError(); h->stream2 = nullptr; parts = 1; }
    // BEVW_BAL_RING=1 (A/B of round 6, profiles/r06/README.md): the two intermediate buffers of a slice -- the compact scratch of shifted texel
    // groups and the pre-gain BEV -- belong to the STREAM, not to the frame sets: slice k reuses the addresses of slice k - 2, so that with
    // slices small enough both stay in the 256 MB Infinity Cache between their writer and their reader and are overwritten there
    static const int ring_env = [] { const char *s = getenv("BEVW_BAL_RING"); return s ? atoi(s) : 0; }();
    const bool ring = ring_env != 0 && parts > 2;
    const int slice_max = (batch + parts - 1) / parts;
    const size_t slots = ring ? (size_t)2 * slice_max : (size_t)batch;
    BEVW_TRY(h->tmp.reserve(cstride * slots));
    // the gain pass reads the pre-gain BEV from a buffer of its own instead of rewriting the output in place: a read stream
    // and a write stream instead of one read-modify-write stream (config 4 2.108 -> 2.052 ms, profiles/r02/sweeps.log); costs
    // one more BEV batch of HBM (0.9 GB at batch 256).  No room for it: the gain pass runs in place -- never an error
    static const int oop = [] { const char *s = getenv("BEVW_GAIN_OOP"); return s ? atoi(s) : 1; }();
    uint8_t *gain_in = d_out;
    if (oop && npx % 4 == 0 && h->pre.reserve(npx * 3 * slots) == BEVW_OK) gain_in = h->pre.as<uint8_t>();
    const bool pre_ring = ring && gain_in != d_out;
    const uint8_t *gain_car = d_car;
    if (pitched && d_car) {   // the gain pass walks the image as a flat array: the sprite needs the same row pitch
        BEVW_TRY(h->car_pitched.reserve(npx * 3));
        BEVW_TRY(plan_pad_image(h->stream, d_car, c.bev_width, h->pitch_px, c.bev_height, h->car_pitched.as<uint8_t>()));
        gain_car = h->car_pitched.as<uint8_t>();
    }
    if (parts > 1) {
        HIP_TRY(hipEventRecord(h->ev_fork, h->stream));          // the caller's uploads (and the padded sprite) were enqueued on stream
        HIP_TRY(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
    }
    for (int part = 0; part < parts; ++part) {
        const int b0 = (int)((long long)batch * part / parts), n = (int)((long long)batch * (part + 1) / parts) - b0;
        if (!n) continue;
        hipStream_t st = (part & 1) ? h->stream2 : h->stream;
        const uint8_t *fr = d_frames + (size_t)b0 * set_bytes;
        // (Deriving the deltas inside k_lum_groups instead of by k_lum_delta, a kernel of its own in between, measured SLOWER: 1.669 against
        // 1.660 ms, profiles/r05/ab_call17...: 22 k blocks repeat four fp64 divisions.  The switch is gone.)
        BEVW_TRY(luminance_stats(st, fr, n, c.frame_width, c.frame_height, h->vsums.as<unsigned long long>() + (size_t)b0 * 4 * kVsumParts,
                                 h->deltas.as<int>() + (size_t)b0 * 4));
        if (part == 0 && parts > 1 && skew_env) {   // the other stream's first slice starts when this one's V sums are done
            HIP_TRY(hipEventRecord(h->ev_skew, h->stream));
            HIP_TRY(hipStreamWaitEvent(h->stream2, h->ev_skew, 0));
        }
        const size_t slot0 = ring ? (size_t)(part & 1) * slice_max : (size_t)b0;   // the slice's first frame-set slot in the intermediate buffers
        uint8_t *shifted = h->tmp.as<uint8_t>() + slot0 * cstride;
        uint8_t *pre = gain_in + (pre_ring ? slot0 : (size_t)b0) * npx * 3;
        BEVW_TRY(plan_lum_groups(h->plan, st, fr, shifted, n, h->deltas.as<int>() + (size_t)b0 * 4, h->hsv.as<HsvTables>()));
        const bool lut_ok = npx % 4 == 0;   // (odd image sizes: the byte-wise gain kernel, in place, from k_reduce_psums' sums)
        BEVW_TRY(plan_stitch(h->plan, st, fr, n, c.blend != 0, false, h->deltas.as<int>() + (size_t)b0 * 4, h->hsv.as<HsvTables>(), nullptr,
                             lut_ok ? nullptr : h->chsums.as<unsigned long long>() + (size_t)b0 * 3, pre, true, batch, b0, shifted));
        BEVW_TRY(gain_pass(h, st, pre, gain_car, d_car, d_out, b0, n, lut_ok, true));
    }
    if (parts > 1) {   // everything the caller enqueues on the handle's stream afterwards sees the whole batch
        HIP_TRY(hipEventRecord(h->ev_join, h->stream2));
        HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    }
    return BEVW_OK;
}

static int run_device(bevw_handle *h, const uint8_t *d_frames, int batch, const uint8_t *d_car, uint8_t *d_out)
{
    const bevw_config &c = h->cfg;
    const bool pitched = h->pitch_px != c.bev_width;
    const size_t npx = (size_t)h->pitch_px * c.bev_height;   // pixels per device image, padding columns included
    if (pitched && (h->projection != BEVW_PROJ_LUT || h->schedule_in_use != BEVW_SCHED_TILE_PLAN ||
                    ((((uintptr_t)d_out | (uintptr_t)d_car | (uintptr_t)d_frames) & 3u) != 0)))
        return fail(BEVW_E_INVALID, "an output pitch needs the tile-plan schedule, the table projection and 4-byte aligned buffers");
    const bool aligned4 = (((uintptr_t)d_out | (uintptr_t)d_car | (uintptr_t)d_frames) & 3u) == 0;
    // balance schedule of the tile plan: 1 = shift the sampled texel groups of the raw frames once (k_lum_groups), then the units;
    // 0 = luminance round trip per fetched texel inside the per-tap kernel
    static const int bal_mode = [] { const char *s = getenv("BEVW_BAL_MODE"); return s ? atoi(s) : 1; }();
    if (h->projection == BEVW_PROJ_LUT && h->schedule_in_use == BEVW_SCHED_TILE_PLAN && aligned4 && c.balance && bal_mode == 1 && h->plan.compact_stride != 0)
        return balance_plan_run(h, d_frames, batch, d_car, d_out);
    if (c.balance) {
        BEVW_TRY(ensure_stats(h, batch));
        BEVW_TRY(luminance_stats(h->stream, d_frames, batch, c.frame_width, c.frame_height,
                                 h->vsums.as<unsigned long long>(), h->deltas.as<int>()));
        HIP_TRY(hipMemsetAsync(h->chsums.p, 0, sizeof(unsigned long long) * 3 * (size_t)batch, h->stream));
    }
    if (h->projection != BEVW_PROJ_LUT) {
        BEVW_TRY(stitch_analytic(h, d_frames, batch, d_car, d_out));
    } else if (h->schedule_in_use == BEVW_SCHED_TILE_PLAN && aligned4) {
        BEVW_TRY(plan_stitch(h->plan, h->stream, d_frames, batch, c.blend != 0, c.balance != 0, h->deltas.as<int>(),
                             h->hsv.as<HsvTables>(), d_car, h->chsums.as<unsigned long long>(), d_out));
    } else {
        BEVW_TRY(stitch_per_pixel(h, d_frames, batch, d_car, d_out));
    }
    if (c.balance) {
        const uint8_t *gain_car = d_car;
        if (pitched && d_car) {   // the gain pass walks the image as a flat array: the sprite needs the same row pitch
            BEVW_TRY(h->car_pitched.reserve(npx * 3));
            BEVW_TRY(plan_pad_image(h->stream, d_car, c.bev_width, h->pitch_px, c.bev_height, h->car_pitched.as<uint8_t>()));
            gain_car = h->car_pitched.as<uint8_t>();
        }
        BEVW_TRY(gain_pass(h, h->stream, d_out, gain_car, d_car, d_out, 0, batch, npx % 4 == 0 && aligned4));
    }
    return BEVW_OK;
}

extern "C" {

int bevw_create(const bevw_config *cfg, bevw_handle **out)
{
    if (!cfg || !out) return fail(BEVW_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->frame_width <= 0 || cfg->frame_height <= 0 || cfg->bev_width <= 0 || cfg->bev_height <= 0)
        return fail(BEVW_E_INVALID, "non-positive frame/BEV size");
    if (cfg->car_width < 0 || cfg->car_height < 0) return fail(BEVW_E_INVALID, "negative car size");
    if (cfg->bev_height > 65535 || cfg->frame_height * cfg->size_scale > 65535.0)
        return fail(BEVW_E_INVALID, "BEV / undistort grid height > 65535 (rows ride in grid.y)");
    if (!(cfg->size_scale > 0) || !(cfg->focal_scale > 0)) return fail(BEVW_E_INVALID, "scales must be positive");
    if ((int)(cfg->frame_width * cfg->size_scale) <= 0 || (int)(cfg->frame_height * cfg->size_scale) <= 0)
        return fail(BEVW_E_INVALID, "empty undistort grid");
    if ((long long)cfg->frame_width * cfg->frame_height * 3 * 4 >= (1ll << 31))
        return fail(BEVW_E_INVALID, "frame too large for 32-bit texel offsets");
    if (cfg->schedule < BEVW_SCHED_AUTO || cfg->schedule > BEVW_SCHED_TILE_PLAN) return fail(BEVW_E_INVALID, "bad schedule");
    BEVW_TRY(use_device(cfg->device));
    bevw_handle *h = new (std::nothrow) bevw_handle();
    if (!h) return fail(BEVW_E_NOMEM, "out of host memory");
    h->cfg = *cfg;
    // (stream2 is created by the first balance step that wants it: HIP multiplexes a process's streams onto a handful of hardware queues --
    // 4 by default -- and every stream that merely exists makes it likelier that two streams meant to overlap share one)
    if (hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess ||
        hipEventCreate(&h->ev1) != hipSuccess || hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_skew, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
        bevw_destroy(h);
        return fail(BEVW_E_HIP, "stream/event creation failed");
    }
    *out = h;
    return BEVW_OK;
}

int bevw_set_camera(bevw_handle *h, int cam, const double K[9], const double D[4], const double H[9])
{
    if (!h || !K || !D || !H) return fail(BEVW_E_INVALID, "null argument");
    if (cam < 0 || cam > 3) return fail(BEVW_E_INVALID, "name should be front/back/left/right");
    memcpy(h->K[cam], K, sizeof h->K[cam]);
    memcpy(h->D[cam], D, sizeof h->D[cam]);
    memcpy(h->H[cam], H, sizeof h->H[cam]);
    h->cam_set[cam] = true;
    h->built = false;
    return BEVW_OK;
}

int bevw_build(bevw_handle *h)
{
    if (h) h->aplan_mode = -1;   // the analytic unit plan follows the masks and calibrations built here

    if (!h) return fail(BEVW_E_INVALID, "null handle");
    for (int c = 0; c < 4; ++c)
        if (h->owns(c) && !h->cam_set[c]) return fail(BEVW_E_INVALID, "camera %d has no K/D/H (bevw_set_camera)", c);
    const bevw_config &cfg = h->cfg;
    BEVW_TRY(use_device(cfg.device));
    for (int k = 0; k < BEVW_COMPAT_KEYS; ++k) h->compat[k] = g_compat[k].load();   // later bevw_set_compat calls do not reach this handle
    hipStream_t st = h->stream;
    const int uw = (int)(cfg.frame_width * cfg.size_scale), uh = (int)(cfg.frame_height * cfg.size_scale);
    const int bw = cfg.bev_width, bh = cfg.bev_height;
    h->uw = uw; h->uh = uh;
    const size_t upx = (size_t)uw * uh, bpx = (size_t)bw * bh;

    // Camera.__init__ (surroundBEV.py:82-88): undistort maps, then the BEV look-up table
    for (int c = 0; c < 4; ++c) {
        BEVW_TRY(h->mask[c].reserve(bpx));   // masks depend on the BEV geometry only: all four, every handle
        if (!h->owns(c)) continue;
        BEVW_TRY(h->und1[c].reserve(upx * 4));
        BEVW_TRY(h->und2[c].reserve(upx * 2));
        BEVW_TRY(h->lut1[c].reserve(bpx * 4));
        BEVW_TRY(h->lut2[c].reserve(bpx * 2));
        double Kd[9];
        camera_mat_dst(h->K[c], cfg.frame_width, cfg.frame_height, cfg.focal_scale, cfg.size_scale, 0.0, 0.0, Kd);
        BEVW_TRY(build_fisheye_maps(st, h->K[c], h->D[c], Kd, uw, uh, h->und1[c].as<int16_t>(), h->und2[c].as<uint16_t>()));
        Mat3 Minv;
        invert3x3(h->H[c], Minv.m);
        for (int k = 0; k < 9; ++k) h->arig.Minv[c][k] = Minv.m[k];
        h->arig.fx[c] = h->K[c][0]; h->arig.fy[c] = h->K[c][4]; h->arig.cx[c] = h->K[c][2]; h->arig.cy[c] = h->K[c][5];
        for (int k = 0; k < 4; ++k) h->arig.d[c][k] = h->D[c][k];
        h->arig.nfx[c] = Kd[0]; h->arig.nfy[c] = Kd[4]; h->arig.ncx[c] = Kd[2]; h->arig.ncy[c] = Kd[5];
        h->arig.uw = uw; h->arig.uh = uh;
        for (int k = 0; k < 9; ++k) h->arig.fMinv[c][k] = (float)Minv.m[k];
        h->arig.ffx[c] = (float)h->K[c][0]; h->arig.ffy[c] = (float)h->K[c][4]; h->arig.fcx[c] = (float)h->K[c][2]; h->arig.fcy[c] = (float)h->K[c][5];
        for (int k = 0; k < 4; ++k) h->arig.fd[c][k] = (float)h->D[c][k];
        h->arig.finv_nfx[c] = (float)(1.0 / Kd[0]); h->arig.finv_nfy[c] = (float)(1.0 / Kd[4]); h->arig.fncx[c] = (float)Kd[2]; h->arig.fncy[c] = (float)Kd[5];
        hipLaunchKernelGGL(k_bev_lut, dim3((bw + 255) / 256, bh), dim3(256), 0, st, Minv, h->und1[c].as<int16_t>(),
                           h->und2[c].as<uint16_t>(), uw, uh, bw, bh, persp_block_width(bw, bh), h->lut1[c].as<int16_t>(),
                           h->lut2[c].as<uint16_t>(), h->compat[BEVW_COMPAT_WARP]);
        BEVW_TRY(launch_check("k_bev_lut"));
    }

    // Mask / BlendMask (surroundBEV.py:119-280)
    MaskGeometry g{bw, bh, cfg.car_width, cfg.car_height};
    if (!cfg.blend) {
        for (int c = 0; c < 4; ++c) BEVW_TRY(fill_poly_device(st, g, c, false, h->mask[c].as<uint8_t>(), h->compat[BEVW_COMPAT_FILLPOLY] != 0));
    } else {
        DevBuf fresh[4];
        int s = BEVW_OK;
        for (int c = 0; c < 4 && s == BEVW_OK; ++c) {
            s = fresh[c].reserve(bpx);
            if (s == BEVW_OK) s = fill_poly_device(st, g, c, true, fresh[c].as<uint8_t>(), h->compat[BEVW_COMPAT_FILLPOLY] != 0);
        }
        // BlendMask.__init__ (:165-186): (own mask, other mask, own seam, other seam), two steps per camera
        struct Step { int other; const char *a0, *a1, *b0, *b1; };
        const Step steps[4][2] = {
            {{BEVW_LEFT, "lT", "cTL", "tL", "cTL"}, {BEVW_RIGHT, "rT", "cTR", "tR", "cTR"}},    // front: FL/LF, FR/RF
            {{BEVW_LEFT, "lB", "cBL", "bL", "cBL"}, {BEVW_RIGHT, "rB", "cBR", "bR", "cBR"}},    // back:  BL/LB, BR/RB
            {{BEVW_FRONT, "tL", "cTL", "lT", "cTL"}, {BEVW_BACK, "bL", "cBL", "lB", "cBL"}},    // left:  LF/FL, LB/BL
            {{BEVW_FRONT, "tR", "cTR", "rT", "cTR"}, {BEVW_BACK, "bR", "cBR", "rB", "cBR"}},    // right: RF/FR, RB/BR
        };
        for (int c = 0; c < 4 && s == BEVW_OK; ++c) {
            if (hipMemcpyAsync(h->mask[c].p, fresh[c].p, bpx, hipMemcpyDeviceToDevice, st) != hipSuccess)
                s = fail(BEVW_E_HIP, "mask copy failed");
            for (int k = 0; k < 2 && s == BEVW_OK; ++k) {
                const Step &sp = steps[c][k];
                hipLaunchKernelGGL(k_blend_weights, dim3((bw + 255) / 256, bh), dim3(256), 0, st, h->mask[c].as<uint8_t>(),
                                   fresh[sp.other].as<uint8_t>(), bw, bh, g.seam(sp.a0, sp.a1), g.seam(sp.b0, sp.b1));
                s = launch_check("k_blend_weights");
            }
        }
        if (s == BEVW_OK && hipStreamSynchronize(st) != hipSuccess) s = fail(BEVW_E_HIP, "mask build failed");
        for (int c = 0; c < 4; ++c) fresh[c].release();
        BEVW_TRY(s);
    }

    // HSV divisor tables
    BEVW_TRY(h->hsv.reserve(sizeof(HsvTables)));
    HsvTables tab = make_hsv_tables();
    HIP_TRY(hipMemcpyAsync(h->hsv.p, &tab, sizeof tab, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));

    // contributor plan for the tile schedule
    StitchTables T;
    const int ncams = h->shard_n ? h->shard_n : 4;
    for (int i = 0; i < 4; ++i) {
        const int c = h->shard_cams[i < ncams ? i : 0];
        T.lut1[i] = h->lut1[c].as<int16_t>();
        T.lut2[i] = h->lut2[c].as<uint16_t>();
        T.mask[i] = h->mask[c].as<uint8_t>();
    }
    // (seam block tiles: measured +0.7 % slower under the per-tile channel sums of the balance path, -1.4 .. -1.8 % without: sweeps.log)
    h->pitch_px = h->pitch_request == BEVW_PITCH_DENSE ? bw : (h->pitch_request == BEVW_PITCH_ALIGNED ? (bw + 63) / 64 * 64 : h->pitch_request);
    BEVW_TRY(plan_build(h->plan, st, T, cfg.frame_width, cfg.frame_height, bw, bh, ncams, h->pitch_px != bw ? h->pitch_px : 0, cfg.blend != 0));
    if (h->shard_n) {
        if (!h->plan.usable) return fail(BEVW_E_INVALID, "camera shard needs the tile plan: %d contributors on some pixel", h->plan.max_contrib);
        // bounding box of the owned masks, widened to multiples of 4 pixels in x so that packed rows stay dword aligned
        std::vector<uint8_t> m(bpx);
        int x0 = bw, y0 = bh, x1 = 0, y1 = 0;
        for (int k = 0; k < h->shard_n; ++k) {
            HIP_TRY(hipMemcpy(m.data(), h->mask[h->shard_cams[k]].p, bpx, hipMemcpyDeviceToHost));
            for (int y = 0; y < bh; ++y) {
                const uint8_t *row = m.data() + (size_t)y * bw;
                int a = 0, b = bw - 1;
                while (a < bw && row[a] == 0) ++a;
                if (a == bw) continue;
                while (row[b] == 0) --b;
                if (a < x0) x0 = a;
                if (b + 1 > x1) x1 = b + 1;
                if (y < y0) y0 = y;
                if (y + 1 > y1) y1 = y + 1;
            }
        }
        if (x1 <= x0 || y1 <= y0) { x0 = y0 = 0; x1 = bw < 4 ? bw : 4; y1 = 1; }
        x0 &= ~3;
        x1 = (x1 + 3) & ~3;
        if (x1 > bw) x1 = bw;
        h->shard_box[0] = x0; h->shard_box[1] = y0; h->shard_box[2] = x1; h->shard_box[3] = y1;
    }
    if (h->pitch_px != bw && (cfg.schedule == BEVW_SCHED_PER_PIXEL || !h->plan.usable))
        return fail(BEVW_E_INVALID, "an output pitch needs the tile-plan schedule");
    h->schedule_in_use = BEVW_SCHED_PER_PIXEL;
    if (h->compat[BEVW_COMPAT_REMAP] && h->shard_n)
        return fail(BEVW_E_INVALID, "BEVW_COMPAT_REMAP 1 is not available on camera-shard handles (they run the tile plan)");
    if (h->compat[BEVW_COMPAT_REMAP]) {
        // half-to-even ties (the candidate rule of OpenCV >= 4.11's float kernels) exist in the per-pixel kernels only: the tile plan's
        // dot-product arithmetic rounds half up
        if (cfg.schedule == BEVW_SCHED_TILE_PLAN || h->pitch_px != bw)
            return fail(BEVW_E_INVALID, "BEVW_COMPAT_REMAP 1 runs the per-pixel schedule (no tile plan, no output pitch)");
    } else if (cfg.schedule == BEVW_SCHED_TILE_PLAN) {
        if (!h->plan.usable) return fail(BEVW_E_INVALID, "tile plan unusable: %d contributors on some pixel", h->plan.max_contrib);
        h->schedule_in_use = BEVW_SCHED_TILE_PLAN;
    } else if (cfg.schedule == BEVW_SCHED_AUTO && h->plan.usable) {
        h->schedule_in_use = BEVW_SCHED_TILE_PLAN;
    }
    HIP_TRY(hipStreamSynchronize(st));
    h->built = true;
    return BEVW_OK;
}

void bevw_destroy(bevw_handle *h)
{
    if (!h) return;
    if (hipSetDevice(h->cfg.device) == hipSuccess) {
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        for (int c = 0; c < 4; ++c) {
            h->und1[c].release(); h->und2[c].release(); h->lut1[c].release(); h->lut2[c].release(); h->mask[c].release();
        }
        h->hsv.release(); h->vsums.release(); h->deltas.release(); h->chsums.release(); h->sdeltas.release();
        h->in.release(); h->out.release(); h->car.release(); h->tmp.release(); h->pre.release(); h->car_pitched.release();
        plan_release(h->plan);
        plan_release(h->aplan);
        h->laps.release();
        if (h->ev0) (void)hipEventDestroy(h->ev0);
        if (h->ev1) (void)hipEventDestroy(h->ev1);
        for (hipEvent_t e : {h->ev_fork, h->ev_skew, h->ev_join}) if (e) (void)hipEventDestroy(e);
        if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
        if (h->stream) (void)hipStreamDestroy(h->stream);
    }
    delete h;
}

static int need_built(bevw_handle *h)
{
    if (!h) return fail(BEVW_E_INVALID, "null handle");
    if (!h->built) return fail(BEVW_E_INVALID, "bevw_build has not been called");
    return use_device(h->cfg.device);
}

int bevw_get_undistort_map(bevw_handle *h, int cam, int16_t *map1, uint16_t *map2)
{
    BEVW_TRY(need_built(h));
    if (cam < 0 || cam > 3) return fail(BEVW_E_INVALID, "name should be front/back/left/right");
    if (!h->owns(cam)) return fail(BEVW_E_INVALID, "camera %d is not owned by this shard", cam);
    const size_t n = (size_t)h->uw * h->uh;
    if (map1) HIP_TRY(hipMemcpy(map1, h->und1[cam].p, n * 4, hipMemcpyDeviceToHost));
    if (map2) HIP_TRY(hipMemcpy(map2, h->und2[cam].p, n * 2, hipMemcpyDeviceToHost));
    return BEVW_OK;
}
int bevw_get_lut(bevw_handle *h, int cam, int16_t *map1, uint16_t *map2)
{
    BEVW_TRY(need_built(h));
    if (cam < 0 || cam > 3) return fail(BEVW_E_INVALID, "name should be front/back/left/right");
    if (!h->owns(cam)) return fail(BEVW_E_INVALID, "camera %d is not owned by this shard", cam);
    const size_t n = (size_t)h->cfg.bev_width * h->cfg.bev_height;
    if (map1) HIP_TRY(hipMemcpy(map1, h->lut1[cam].p, n * 4, hipMemcpyDeviceToHost));
    if (map2) HIP_TRY(hipMemcpy(map2, h->lut2[cam].p, n * 2, hipMemcpyDeviceToHost));
    return BEVW_OK;
}
int bevw_get_mask(bevw_handle *h, int cam, uint8_t *mask)
{
    BEVW_TRY(need_built(h));
    if (cam < 0 || cam > 3 || !mask) return fail(BEVW_E_INVALID, "name should be front/back/left/right");
    HIP_TRY(hipMemcpy(mask, h->mask[cam].p, (size_t)h->cfg.bev_width * h->cfg.bev_height, hipMemcpyDeviceToHost));
    return BEVW_OK;
}
int bevw_set_projection(bevw_handle *h, int mode)
{
    if (!h) return fail(BEVW_E_INVALID, "null handle");
    if (mode != BEVW_PROJ_LUT && mode != BEVW_PROJ_ANALYTIC && mode != BEVW_PROJ_ANALYTIC_F32) return fail(BEVW_E_INVALID, "unknown projection mode %d", mode);
    if (mode != BEVW_PROJ_LUT && h->shard_n) return fail(BEVW_E_INVALID, "analytic projection is not available on camera-shard handles");
    h->projection = mode;
    return BEVW_OK;
}

int bevw_plan_info(bevw_handle *h, int32_t info[8])
{
    BEVW_TRY(need_built(h));
    if (!info) return fail(BEVW_E_INVALID, "null argument");
    memset(info, 0, 8 * sizeof(int32_t));
    info[0] = h->plan.max_contrib;
    info[1] = h->plan.usable ? 1 : 0;
    info[2] = h->schedule_in_use;
    info[3] = h->plan.tiles_x;
    info[4] = h->plan.tiles_y;
    info[5] = h->plan.n_unit_tiles;   // base tiles on the unit schedule (bevw_unit.h)
    info[6] = 0;                      // (rounds 1 - 3: tiles on the L1-gather kernels; retired)
    info[7] = h->plan.n_slow;
    return BEVW_OK;
}

int bevw_set_output_pitch(bevw_handle *h, int pitch_pixels)
{
    if (!h) return fail(BEVW_E_INVALID, "null handle");
    if (h->shard_n) return fail(BEVW_E_INVALID, "an output pitch is not available on camera-shard handles");
    if (pitch_pixels != BEVW_PITCH_DENSE && pitch_pixels != BEVW_PITCH_ALIGNED &&
        (pitch_pixels < h->cfg.bev_width || pitch_pixels % 4 != 0 || pitch_pixels > 65532))
        return fail(BEVW_E_INVALID, "output pitch must be BEVW_PITCH_DENSE, BEVW_PITCH_ALIGNED or a multiple of 4 >= BEV_WIDTH, got %d", pitch_pixels);
    h->pitch_request = pitch_pixels;
    h->built = false;
    return BEVW_OK;
}

int bevw_output_pitch(bevw_handle *h)
{
    if (!h || !h->built) { fail(BEVW_E_INVALID, "bevw_build has not been called"); return BEVW_E_INVALID; }
    return h->pitch_px;
}

// [batch * rows][pitch_px][3] on the device -> dense [batch * rows][bw][3] on the host (rows compacted inside the copy)
static int download_images(bevw_handle *h, uint8_t *out, const void *d_src, size_t rows)
{
    const size_t row_bytes = (size_t)h->cfg.bev_width * 3, src_pitch = (size_t)h->pitch_px * 3;
    if (src_pitch == row_bytes) HIP_TRY(hipMemcpyAsync(out, d_src, row_bytes * rows, hipMemcpyDeviceToHost, h->stream));
    else HIP_TRY(hipMemcpy2DAsync(out, row_bytes, d_src, src_pitch, row_bytes, rows, hipMemcpyDeviceToHost, h->stream));
    return BEVW_OK;
}

int bevw_run_device(bevw_handle *h, const void *d_frames, int batch, const void *d_car, void *d_out)
{
    BEVW_TRY(need_built(h));
    if (h->shard_n) return fail(BEVW_E_INVALID, "handle is a camera shard: use bevw_shard_run_device + bevw_combine_device");
    if (!d_frames || !d_out || batch < 0) return fail(BEVW_E_INVALID, "bad argument");
    if (batch == 0) return BEVW_OK;
    return run_device(h, (const uint8_t *)d_frames, batch, (const uint8_t *)d_car, (uint8_t *)d_out);
}

int bevw_run(bevw_handle *h, const uint8_t *frames, int batch, const uint8_t *car, uint8_t *out)
{
    BEVW_TRY(need_built(h));
    if (h->shard_n) return fail(BEVW_E_INVALID, "handle is a camera shard: use bevw_shard_run_device + bevw_combine_device");
    if (!frames || !out || batch < 0) return fail(BEVW_E_INVALID, "bad argument");
    if (batch == 0) return BEVW_OK;
    const bevw_config &c = h->cfg;
    const size_t nin = (size_t)batch * 4 * c.frame_width * c.frame_height * 3;
    const size_t bev = (size_t)c.bev_width * c.bev_height * 3, dbev = (size_t)h->pitch_px * c.bev_height * 3;
    BEVW_TRY(h->in.reserve(nin));
    BEVW_TRY(h->out.reserve(dbev * batch));
    HIP_TRY(hipMemcpyAsync(h->in.p, frames, nin, hipMemcpyHostToDevice, h->stream));
    const uint8_t *d_car = nullptr;
    if (car) {
        BEVW_TRY(h->car.reserve(bev));
        HIP_TRY(hipMemcpyAsync(h->car.p, car, bev, hipMemcpyHostToDevice, h->stream));
        d_car = h->car.as<uint8_t>();
    }
    BEVW_TRY(run_device(h, h->in.as<uint8_t>(), batch, d_car, h->out.as<uint8_t>()));
    BEVW_TRY(download_images(h, out, h->out.p, (size_t)batch * c.bev_height));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return BEVW_OK;
}

// BevGenerator.__call__ as the reference calls it: four separate camera arrays, one frame set (surroundBEV.py:312-325)
int bevw_run_cameras(bevw_handle *h, const uint8_t *front, const uint8_t *back, const uint8_t *left, const uint8_t *right,
                     const uint8_t *car, uint8_t *out)
{
    BEVW_TRY(need_built(h));
    if (h->shard_n) return fail(BEVW_E_INVALID, "handle is a camera shard: use bevw_shard_run_device + bevw_combine_device");
    if (!front || !back || !left || !right || !out) return fail(BEVW_E_INVALID, "bad argument");
    const bevw_config &c = h->cfg;
    const size_t frame = (size_t)c.frame_width * c.frame_height * 3, bev = (size_t)c.bev_width * c.bev_height * 3;
    BEVW_TRY(h->in.reserve(frame * 4));
    BEVW_TRY(h->out.reserve((size_t)h->pitch_px * c.bev_height * 3));
    const uint8_t *src[4] = {front, back, left, right};
    for (int i = 0; i < 4; ++i)
        HIP_TRY(hipMemcpyAsync(h->in.as<uint8_t>() + frame * i, src[i], frame, hipMemcpyHostToDevice, h->stream));
    const uint8_t *d_car = nullptr;
    if (car) {
        BEVW_TRY(h->car.reserve(bev));
        HIP_TRY(hipMemcpyAsync(h->car.p, car, bev, hipMemcpyHostToDevice, h->stream));
        d_car = h->car.as<uint8_t>();
    }
    BEVW_TRY(run_device(h, h->in.as<uint8_t>(), 1, d_car, h->out.as<uint8_t>()));
    BEVW_TRY(download_images(h, out, h->out.p, (size_t)c.bev_height));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return BEVW_OK;
}

static int camera_remap(bevw_handle *h, const uint8_t *src, int sw, int sh, const int16_t *m1, const uint16_t *m2, int dw,
                        int dh, int batch, uint8_t *dst)
{
    const size_t nin = (size_t)batch * sw * sh * 3, nout = (size_t)batch * dw * dh * 3;
    BEVW_TRY(h->in.reserve(nin));
    BEVW_TRY(h->out.reserve(nout));
    HIP_TRY(hipMemcpyAsync(h->in.p, src, nin, hipMemcpyHostToDevice, h->stream));
    BEVW_TRY(remap_launch(h->stream, h->in.as<uint8_t>(), sw, sh, m1, m2, dw, dh, batch, h->out.as<uint8_t>(), h->compat[BEVW_COMPAT_REMAP]));
    HIP_TRY(hipMemcpyAsync(dst, h->out.p, nout, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return BEVW_OK;
}

int bevw_camera_undistort(bevw_handle *h, int cam, const uint8_t *src, int batch, uint8_t *dst)
{
    BEVW_TRY(need_built(h));
    if (cam < 0 || cam > 3 || !src || !dst || batch < 0) return fail(BEVW_E_INVALID, "bad argument");
    if (!h->owns(cam)) return fail(BEVW_E_INVALID, "camera %d is not owned by this shard", cam);
    if (batch == 0) return BEVW_OK;
    return camera_remap(h, src, h->cfg.frame_width, h->cfg.frame_height, h->und1[cam].as<int16_t>(),
                        h->und2[cam].as<uint16_t>(), h->uw, h->uh, batch, dst);
}

int bevw_camera_raw2bev(bevw_handle *h, int cam, const uint8_t *src, int batch, uint8_t *dst)
{
    BEVW_TRY(need_built(h));
    if (cam < 0 || cam > 3 || !src || !dst || batch < 0) return fail(BEVW_E_INVALID, "bad argument");
    if (!h->owns(cam)) return fail(BEVW_E_INVALID, "camera %d is not owned by this shard", cam);
    if (batch == 0) return BEVW_OK;
    return camera_remap(h, src, h->cfg.frame_width, h->cfg.frame_height, h->lut1[cam].as<int16_t>(),
                        h->lut2[cam].as<uint16_t>(), h->cfg.bev_width, h->cfg.bev_height, batch, dst);
}

int bevw_camera_warp_homography(bevw_handle *h, int cam, const uint8_t *src, int src_w, int src_h, int batch, uint8_t *dst)
{
    BEVW_TRY(need_built(h));
    if (cam < 0 || cam > 3) return fail(BEVW_E_INVALID, "name should be front/back/left/right");
    if (!h->owns(cam)) return fail(BEVW_E_INVALID, "camera %d is not owned by this shard", cam);
    return bevw_warp_perspective_u8c3(h->cfg.device, src, src_w, src_h, h->H[cam], h->cfg.bev_width, h->cfg.bev_height,
                                      batch, dst);
}

// ---------------------------------------------------------------------------------------------------------------
// camera-per-GPU mode
// ---------------------------------------------------------------------------------------------------------------
int bevw_set_camera_shard(bevw_handle *h, const int32_t *cams, int ncams)
{
    if (!h || !cams) return fail(BEVW_E_INVALID, "null argument");
    if (ncams < 1 || ncams > 4) return fail(BEVW_E_INVALID, "a shard owns 1..4 cameras");
    for (int k = 0; k < ncams; ++k) {
        if (cams[k] < 0 || cams[k] > 3) return fail(BEVW_E_INVALID, "name should be front/back/left/right");
        if (k && cams[k] <= cams[k - 1]) return fail(BEVW_E_INVALID, "shard cameras must be distinct and ascending");
    }
    h->shard_n = ncams;
    for (int k = 0; k < 4; ++k) h->shard_cams[k] = k < ncams ? cams[k] : cams[0];
    h->built = false;
    return BEVW_OK;
}

static int need_shard(bevw_handle *h)
{
    BEVW_TRY(need_built(h));
    if (!h->shard_n) return fail(BEVW_E_INVALID, "handle is not a camera shard (bevw_set_camera_shard before bevw_build)");
    return BEVW_OK;
}

int bevw_shard_box(bevw_handle *h, int32_t box[4])
{
    BEVW_TRY(need_shard(h));
    if (!box) return fail(BEVW_E_INVALID, "null argument");
    for (int i = 0; i < 4; ++i) box[i] = h->shard_box[i];
    return BEVW_OK;
}

int bevw_shard_vsums_device(bevw_handle *h, const void *d_frames, int batch, void *d_vsums)
{
    BEVW_TRY(need_shard(h));
    if (!d_frames || !d_vsums || batch < 0) return fail(BEVW_E_INVALID, "bad argument");
    if (batch == 0) return BEVW_OK;
    const bevw_config &c = h->cfg;
    const size_t frame_bytes = (size_t)c.frame_width * c.frame_height * 3;
    const int nframes = batch * h->shard_n;
    HIP_TRY(hipMemsetAsync(d_vsums, 0, sizeof(unsigned long long) * (size_t)nframes, h->stream));
    const int vec_ok = (frame_bytes % 4 == 0 && ((uintptr_t)d_frames & 3u) == 0) ? 1 : 0;
    int bpf = 2048 / nframes;
    if (bpf < 8) bpf = 8;
    if (bpf > 256) bpf = 256;
    for (int f0 = 0; f0 < nframes; f0 += 65535) {
        const int nf = nframes - f0 < 65535 ? nframes - f0 : 65535;
        hipLaunchKernelGGL(k_vsum, dim3(bpf, nf), dim3(256), 0, h->stream, (const uint8_t *)d_frames + (size_t)f0 * frame_bytes,
                           frame_bytes, vec_ok, (unsigned long long *)d_vsums + f0);
    }
    return launch_check("k_vsum");
}

int bevw_shard_run_device(bevw_handle *h, const void *d_frames, int batch, const void *d_all_vsums, void *d_out)
{
    BEVW_TRY(need_shard(h));
    if (!d_frames || !d_out || batch < 0) return fail(BEVW_E_INVALID, "bad argument");
    const bevw_config &c = h->cfg;
    if (c.balance && !d_all_vsums) return fail(BEVW_E_INVALID, "balance needs the V sums of all four cameras");
    if ((((uintptr_t)d_out | (uintptr_t)d_frames) & 3u) != 0) return fail(BEVW_E_INVALID, "device buffers must be 4-byte aligned");
    if (batch == 0) return BEVW_OK;
    const uint8_t *frames = (const uint8_t *)d_frames;
    if (!c.balance) return plan_stitch(h->plan, h->stream, frames, batch, c.blend != 0, false, nullptr, nullptr, nullptr, nullptr,
                                       (uint8_t *)d_out);
    // luminance_balance (surroundBEV.py:57-79) with the means of ALL four cameras, applied to the owned ones
    BEVW_TRY(ensure_stats(h, batch));
    BEVW_TRY(h->sdeltas.reserve(sizeof(int) * 4 * (size_t)batch));
    hipLaunchKernelGGL(k_lum_delta, dim3((batch + 63) / 64), dim3(64), 0, h->stream, (const unsigned long long *)d_all_vsums,
                       (double)c.frame_width * (double)c.frame_height, batch, h->deltas.as<int>());
    ShardCams sc;
    sc.n = h->shard_n;
    for (int k = 0; k < 4; ++k) sc.cam[k] = h->shard_cams[k];
    hipLaunchKernelGGL(k_delta_select, dim3((batch * 4 + 255) / 256), dim3(256), 0, h->stream, h->deltas.as<int>(), sc, batch,
                       h->sdeltas.as<int>());
    BEVW_TRY(launch_check("k_lum_delta/k_delta_select"));
    static const int bal_mode = [] { const char *s = getenv("BEVW_BAL_MODE"); return s ? atoi(s) : 1; }();
    if (bal_mode == 1 && h->plan.compact_stride != 0) {
        BEVW_TRY(h->tmp.reserve(h->plan.compact_stride * (size_t)batch));
        BEVW_TRY(plan_lum_groups(h->plan, h->stream, frames, h->tmp.as<uint8_t>(), batch, h->sdeltas.as<int>(), h->hsv.as<HsvTables>()));
        return plan_stitch(h->plan, h->stream, frames, batch, c.blend != 0, false, h->sdeltas.as<int>(), h->hsv.as<HsvTables>(), nullptr, nullptr,
                           (uint8_t *)d_out, false, 0, 0, h->tmp.as<uint8_t>());
    }
    HIP_TRY(hipMemsetAsync(h->chsums.p, 0, sizeof(unsigned long long) * 3 * (size_t)batch, h->stream));
    return plan_stitch(h->plan, h->stream, frames, batch, c.blend != 0, true, h->sdeltas.as<int>(), h->hsv.as<HsvTables>(), nullptr,
                       h->chsums.as<unsigned long long>(), (uint8_t *)d_out);
}

// ---- the exchange step over RCCL (bevw_comm.h) -------------------------------------------------------------------
struct bevw_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

#define RCCL_TRY(expr)                                                                                        \
    do {                                                                                                      \
        ncclResult_t _r = (expr);                                                                             \
        if (_r != ncclSuccess) return fail(BEVW_E_HIP, "%s failed: %s", #expr, rccl().GetErrorString(_r));     \
    } while (0)

int bevw_comm_available(void) { return rccl().ok ? 1 : 0; }

int bevw_comm_unique_id(uint8_t id[128])
{
    if (!id) return fail(BEVW_E_INVALID, "null argument");
    if (!rccl().ok) return fail(BEVW_E_NO_DEVICE, "RCCL is not available: %s", rccl().why);
    ncclUniqueId u;
    RCCL_TRY(rccl().GetUniqueId(&u));
    static_assert(sizeof(u) == 128, "ncclUniqueId");
    memcpy(id, &u, 128);
    return BEVW_OK;
}

int bevw_comm_create(int device, int rank, int world, const uint8_t id[128], bevw_comm **out)
{
    if (!out || !id) return fail(BEVW_E_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(BEVW_E_INVALID, "rank %d of %d", rank, world);
    BEVW_TRY(use_device(device));
    if (!rccl().ok) return fail(BEVW_E_NO_DEVICE, "RCCL is not available: %s", rccl().why);
    ncclUniqueId u;
    memcpy(&u, id, 128);
    bevw_comm *c = new (std::nothrow) bevw_comm;
    if (!c) return fail(BEVW_E_NOMEM, "out of host memory");
    c->rank = rank; c->world = world; c->device = device;
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return fail(BEVW_E_HIP, "ncclCommInitRank failed: %s", rccl().GetErrorString(r)); }
    *out = c;
    return BEVW_OK;
}

void bevw_comm_destroy(bevw_comm *c)
{
    if (!c) return;
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    delete c;
}

// luminance_balance needs the V sums of all four cameras (surroundBEV.py:60-66): all-gather this rank's [batch][ncams]
// sums inside the camera group and lay them out as [batch][4].  Ranks of a group own equally many cameras.
int bevw_shard_allgather_vsums(bevw_handle *h, bevw_comm *c, const void *d_vsums, int batch, void *d_all_vsums)
{
    BEVW_TRY(need_shard(h));
    if (!c || !d_vsums || !d_all_vsums || batch < 0) return fail(BEVW_E_INVALID, "bad argument");
    if (h->shard_n * c->world != 4) return fail(BEVW_E_INVALID, "%d ranks x %d cameras do not make the 4-camera rig", c->world, h->shard_n);
    if (batch == 0) return BEVW_OK;
    const size_t mine = sizeof(unsigned long long) * (size_t)batch * h->shard_n;
    BEVW_TRY(h->xchg.reserve(mine * (size_t)c->world));
    RCCL_TRY(rccl().AllGather(d_vsums, h->xchg.p, mine, ncclUint8, c->comm, h->stream));
    hipLaunchKernelGGL(k_vsums_interleave, dim3((batch * 4 + 255) / 256), dim3(256), 0, h->stream, h->xchg.as<unsigned long long>(),
                       c->world, h->shard_n, batch, (unsigned long long *)d_all_vsums);
    return launch_check("k_vsums_interleave");
}

// The parts travel to the stitch rank: every other rank sends its packed mask box, the stitch rank receives one box per peer
// (d_recv[r], bytes[r] for r != root; its own entry is ignored).  One grouped call on the handle's stream, no host sync.
int bevw_shard_gather_parts(bevw_handle *h, bevw_comm *c, const void *d_packed, size_t my_bytes, int root, void *const *d_recv,
                            const size_t *bytes)
{
    BEVW_TRY(need_built(h));
    if (!c || root < 0 || root >= c->world) return fail(BEVW_E_INVALID, "bad argument");
    if (c->world == 1) return BEVW_OK;
    if (c->rank != root) {
        if (!d_packed) return fail(BEVW_E_INVALID, "null part");
        RCCL_TRY(rccl().Send(d_packed, my_bytes, ncclUint8, root, c->comm, h->stream));
        return BEVW_OK;
    }
    if (!d_recv || !bytes) return fail(BEVW_E_INVALID, "null receive list");
    RCCL_TRY(rccl().GroupStart());
    for (int r = 0; r < c->world; ++r) {
        if (r == root) continue;
        if (!d_recv[r]) { (void)rccl().GroupEnd(); return fail(BEVW_E_INVALID, "no receive buffer for rank %d", r); }
        ncclResult_t e = rccl().Recv(d_recv[r], bytes[r], ncclUint8, r, c->comm, h->stream);
        if (e != ncclSuccess) { (void)rccl().GroupEnd(); return fail(BEVW_E_HIP, "ncclRecv failed: %s", rccl().GetErrorString(e)); }
    }
    RCCL_TRY(rccl().GroupEnd());
    return BEVW_OK;
}

// Self-test of the transport on ONE rank (a 1-GPU box): all-gather, and a grouped send + receive of `nbytes` to itself,
// on the handle's stream.  d_dst receives a copy of d_src.
int bevw_comm_selftest(bevw_handle *h, bevw_comm *c, const void *d_src, void *d_dst, size_t nbytes)
{
    BEVW_TRY(need_built(h));
    if (!c || !d_src || !d_dst) return fail(BEVW_E_INVALID, "null argument");
    if (c->world != 1) return fail(BEVW_E_INVALID, "the self-test runs on a 1-rank communicator");
    BEVW_TRY(h->xchg.reserve(nbytes));
    RCCL_TRY(rccl().AllGather(d_src, h->xchg.p, nbytes, ncclUint8, c->comm, h->stream));
    RCCL_TRY(rccl().GroupStart());
    ncclResult_t e1 = rccl().Send(h->xchg.p, nbytes, ncclUint8, 0, c->comm, h->stream);
    ncclResult_t e2 = rccl().Recv(d_dst, nbytes, ncclUint8, 0, c->comm, h->stream);
    RCCL_TRY(rccl().GroupEnd());
    if (e1 != ncclSuccess || e2 != ncclSuccess) return fail(BEVW_E_HIP, "ncclSend/ncclRecv to self failed");
    return BEVW_OK;
}

int bevw_shard_pack_device(bevw_handle *h, const void *d_full, int batch, void *d_packed)
{
    BEVW_TRY(need_shard(h));
    if (!d_full || !d_packed || batch < 0) return fail(BEVW_E_INVALID, "bad argument");
    if (batch == 0) return BEVW_OK;
    const bevw_config &c = h->cfg;
    const int *bx = h->shard_box;
    const int rows = bx[3] - bx[1], row_bytes = (bx[2] - bx[0]) * 3;
    const bool dwords = c.bev_width % 4 == 0 && bx[0] % 4 == 0 && (bx[2] - bx[0]) % 4 == 0 &&
                        (((uintptr_t)d_full | (uintptr_t)d_packed) & 3u) == 0;
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        const uint8_t *src = (const uint8_t *)d_full + (size_t)b0 * c.bev_width * c.bev_height * 3;
        uint8_t *dst = (uint8_t *)d_packed + (size_t)b0 * rows * row_bytes;
        if (dwords)
            hipLaunchKernelGGL((k_pack_box<uint32_t>), dim3((row_bytes / 4 + 255) / 256, rows, nb), dim3(256), 0, h->stream, src,
                               c.bev_width, c.bev_height, bx[0], bx[1], bx[2], bx[3], dst);
        else
            hipLaunchKernelGGL((k_pack_box<uint8_t>), dim3((row_bytes + 255) / 256, rows, nb), dim3(256), 0, h->stream, src,
                               c.bev_width, c.bev_height, bx[0], bx[1], bx[2], bx[3], dst);
    }
    return launch_check("k_pack_box");
}

int bevw_combine_device(bevw_handle *h, const void *const *d_parts, const int32_t *boxes, int nparts, int batch, const void *d_car,
                        void *d_out)
{
    BEVW_TRY(need_built(h));
    if (!d_parts || !boxes || !d_out || batch < 0) return fail(BEVW_E_INVALID, "bad argument");
    if (nparts < 1 || nparts > 8) return fail(BEVW_E_INVALID, "1..8 parts");
    const bevw_config &c = h->cfg;
    const int bw = c.bev_width, bh = c.bev_height;
    CombineParts parts;
    memset(&parts, 0, sizeof parts);
    parts.n = nparts;
    bool dwords = bw % 4 == 0 && (((uintptr_t)d_out | (uintptr_t)d_car) & 3u) == 0;
    for (int k = 0; k < nparts; ++k) {
        const int32_t *bx = boxes + k * 4;
        if (!d_parts[k] || bx[0] < 0 || bx[1] < 0 || bx[2] > bw || bx[3] > bh || bx[2] <= bx[0] || bx[3] <= bx[1])
            return fail(BEVW_E_INVALID, "part %d: bad pointer or box", k);
        parts.p[k] = (const uint8_t *)d_parts[k];
        for (int i = 0; i < 4; ++i) parts.box[k][i] = bx[i];
        if (bx[0] % 4 != 0 || (bx[2] - bx[0]) % 4 != 0 || ((uintptr_t)d_parts[k] & 3u) != 0) dwords = false;
    }
    if (batch == 0) return BEVW_OK;
    const size_t npx = (size_t)bw * bh;
    // balance: the car is added after the white balance (surroundBEV.py:321-324), so the sum goes out bare first
    const uint8_t *car_now = c.balance ? nullptr : (const uint8_t *)d_car;
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        CombineParts pb = parts;
        for (int k = 0; k < nparts; ++k)
            pb.p[k] += (size_t)b0 * (size_t)(pb.box[k][2] - pb.box[k][0]) * (size_t)(pb.box[k][3] - pb.box[k][1]) * 3;
        uint8_t *o = (uint8_t *)d_out + (size_t)b0 * npx * 3;
        if (dwords) hipLaunchKernelGGL((k_combine<4>), dim3((bw / 4 + 255) / 256, bh, nb), dim3(256), 0, h->stream, pb, bw, bh, car_now, o);
        else hipLaunchKernelGGL((k_combine<1>), dim3((bw + 255) / 256, bh, nb), dim3(256), 0, h->stream, pb, bw, bh, car_now, o);
    }
    BEVW_TRY(launch_check("k_combine"));
    if (c.balance) {
        BEVW_TRY(ensure_stats(h, batch));
        HIP_TRY(hipMemsetAsync(h->chsums.p, 0, sizeof(unsigned long long) * 3 * (size_t)batch, h->stream));
        for (int b0 = 0; b0 < batch; b0 += 65535) {
            const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
            uint8_t *o = (uint8_t *)d_out + (size_t)b0 * npx * 3;
            unsigned long long *chs = h->chsums.as<unsigned long long>() + (size_t)b0 * 3;
            hipLaunchKernelGGL(k_channel_sums, dim3(64, nb), dim3(256), 0, h->stream, o, npx, chs);
            if (npx % 4 == 0 && dwords)
                hipLaunchKernelGGL(k_gain_lut, dim3(xcd_frame_grid(32, (unsigned)nb)), dim3(256), 0, h->stream, o, npx, chs, (const uint8_t *)d_car, o,
                                   32u, (uint32_t)nb, h->compat[BEVW_COMPAT_ADDWEIGHTED] ? 0 : 1);
            else hipLaunchKernelGGL(k_gain, dim3(64, nb), dim3(256), 0, h->stream, o, npx, chs, (const uint8_t *)d_car, o,
                                    h->compat[BEVW_COMPAT_ADDWEIGHTED] ? 0 : 1);
        }
        BEVW_TRY(launch_check("k_channel_sums/k_gain"));
    }
    return BEVW_OK;
}

int bevw_apply_mask(bevw_handle *h, int cam, const uint8_t *img, int batch, uint8_t *out)
{
    BEVW_TRY(need_built(h));
    if (cam < 0 || cam > 3) return fail(BEVW_E_INVALID, "name should be front/back/left/right");
    if (!img || !out || batch < 0) return fail(BEVW_E_INVALID, "bad argument");
    if (batch == 0) return BEVW_OK;
    const size_t npx = (size_t)h->cfg.bev_width * h->cfg.bev_height, n = npx * 3 * (size_t)batch;
    BEVW_TRY(h->in.reserve(n));
    BEVW_TRY(h->out.reserve(n));
    HIP_TRY(hipMemcpyAsync(h->in.p, img, n, hipMemcpyHostToDevice, h->stream));
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        hipLaunchKernelGGL(k_apply_mask, dim3(256, nb), dim3(256), 0, h->stream, h->in.as<uint8_t>() + (size_t)b0 * npx * 3,
                           h->mask[cam].as<uint8_t>(), npx, h->cfg.blend, h->out.as<uint8_t>() + (size_t)b0 * npx * 3);
    }
    BEVW_TRY(launch_check("k_apply_mask"));
    HIP_TRY(hipMemcpyAsync(out, h->out.p, n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return BEVW_OK;
}

int bevw_luminance_balance(int device, const uint8_t *frames, int batch, int width, int height, uint8_t *out)
{
    if (!frames || !out || batch < 0 || width <= 0 || height <= 0) return fail(BEVW_E_INVALID, "bad argument");
    if (batch == 0) return BEVW_OK;
    BEVW_TRY(use_device(device));
    const size_t fpx = (size_t)width * height, n = (size_t)batch * 4 * fpx * 3;
    DevBuf in, o, vs, dl, tb;
    int s = in.reserve(n);
    if (s == BEVW_OK) s = o.reserve(n);
    if (s == BEVW_OK) s = vs.reserve(sizeof(unsigned long long) * 4 * (size_t)batch * kVsumParts);
    if (s == BEVW_OK) s = dl.reserve(sizeof(int) * 4 * (size_t)batch);
    if (s == BEVW_OK) s = tb.reserve(sizeof(HsvTables));
    HsvTables tab = make_hsv_tables();
    if (s == BEVW_OK && (hipMemcpy(in.p, frames, n, hipMemcpyHostToDevice) != hipSuccess ||
                         hipMemcpy(tb.p, &tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess))
        s = fail(BEVW_E_HIP, "H2D failed");
    if (s == BEVW_OK) s = luminance_stats(0, in.as<uint8_t>(), batch, width, height, vs.as<unsigned long long>(), dl.as<int>());
    if (s == BEVW_OK) {
        const int nframes = batch * 4;
        for (int f0 = 0; f0 < nframes; f0 += 65535) {
            const int nf = nframes - f0 < 65535 ? nframes - f0 : 65535;
            hipLaunchKernelGGL(k_lum_shift, dim3(64, nf), dim3(256), 0, 0, in.as<uint8_t>() + (size_t)f0 * fpx * 3, fpx,
                               dl.as<int>() + f0, tb.as<HsvTables>(), o.as<uint8_t>() + (size_t)f0 * fpx * 3);
        }
        s = launch_check("k_lum_shift");
    }
    if (s == BEVW_OK && hipMemcpy(out, o.p, n, hipMemcpyDeviceToHost) != hipSuccess) s = fail(BEVW_E_HIP, "D2H failed");
    in.release(); o.release(); vs.release(); dl.release(); tb.release();
    return s;
}

int bevw_color_balance(int device, const uint8_t *images, int batch, int width, int height, uint8_t *out)
{
    if (!images || !out || batch < 0 || width <= 0 || height <= 0) return fail(BEVW_E_INVALID, "bad argument");
    if (batch == 0) return BEVW_OK;
    BEVW_TRY(use_device(device));
    const size_t npx = (size_t)width * height, n = (size_t)batch * npx * 3;
    DevBuf in, cs;
    int s = in.reserve(n);
    if (s == BEVW_OK) s = cs.reserve(sizeof(unsigned long long) * 3 * (size_t)batch);
    if (s == BEVW_OK && hipMemcpy(in.p, images, n, hipMemcpyHostToDevice) != hipSuccess) s = fail(BEVW_E_HIP, "H2D failed");
    if (s == BEVW_OK && hipMemset(cs.p, 0, sizeof(unsigned long long) * 3 * (size_t)batch) != hipSuccess)
        s = fail(BEVW_E_HIP, "memset failed");
    if (s == BEVW_OK) {
        for (int b0 = 0; b0 < batch; b0 += 65535) {
            const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
            hipLaunchKernelGGL(k_channel_sums, dim3(64, nb), dim3(256), 0, 0, in.as<uint8_t>() + (size_t)b0 * npx * 3, npx,
                               cs.as<unsigned long long>() + (size_t)b0 * 3);
            hipLaunchKernelGGL(k_gain, dim3(64, nb), dim3(256), 0, 0, in.as<uint8_t>() + (size_t)b0 * npx * 3, npx,
                               cs.as<unsigned long long>() + (size_t)b0 * 3, (const uint8_t *)nullptr,
                               in.as<uint8_t>() + (size_t)b0 * npx * 3, g_compat[BEVW_COMPAT_ADDWEIGHTED].load() ? 0 : 1);
        }
        s = launch_check("k_channel_sums/k_gain");
    }
    if (s == BEVW_OK && hipMemcpy(out, in.p, n, hipMemcpyDeviceToHost) != hipSuccess) s = fail(BEVW_E_HIP, "D2H failed");
    in.release(); cs.release();
    return s;
}

int bevw_sync(bevw_handle *h)
{
    if (!h) return fail(BEVW_E_INVALID, "null handle");
    BEVW_TRY(use_device(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return BEVW_OK;
}
int bevw_timer_start(bevw_handle *h)
{
    if (!h) return fail(BEVW_E_INVALID, "null handle");
    BEVW_TRY(use_device(h->cfg.device));
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    return BEVW_OK;
}
int bevw_timer_mark(bevw_handle *h, int slot)
{
    if (!h) return fail(BEVW_E_INVALID, "null handle");
    BEVW_TRY(use_device(h->cfg.device));
    return h->laps.mark(slot, h->stream);
}
int bevw_timer_between(bevw_handle *h, int slot_a, int slot_b, float *elapsed_ms)
{
    if (!h) return fail(BEVW_E_INVALID, "null handle");
    BEVW_TRY(use_device(h->cfg.device));
    return h->laps.between(slot_a, slot_b, elapsed_ms);
}
int bevw_timer_stop(bevw_handle *h, float *elapsed_ms)
{
    if (!h || !elapsed_ms) return fail(BEVW_E_INVALID, "null argument");
    BEVW_TRY(use_device(h->cfg.device));
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    return BEVW_OK;
}

}  // extern "C"
