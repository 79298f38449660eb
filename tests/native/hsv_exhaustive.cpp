// Host-only exhaustive check of luminance_shift_bgr (cameracalibration_amd/csrc/bevw_device.h) -- runs without a GPU.
//
// The kernels' own function (its __host__ half: v_perm_b32 restated by px_perm, float32 arithmetic without contraction) is run over
// ALL 2^24 BGR colours for a set of V shifts and compared with a plain restatement of OpenCV's BGR2HSV (8 bit) -> cv2.add on V ->
// HSV2BGR (8 bit, float path) written here independently of the device code (the same statement as oracle/bevoracle.c bgr2hsv_px /
// hsv2bgr_px, reference call site surroundBEV.py:57-79).  Also checks the claims the fast path rests on: the hue quotient stays in
// [-30, 150], S in [0, 255], and cvRound(float(V) / 255 * 255) == V.
//   hsv_exhaustive [delta ...]      default deltas: -255 -128 -37 -6 -1 0 1 6 23 128 255
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>

#include "../../cameracalibration_amd/csrc/bevw_device.h"

using namespace bevw;

static int rne_host(double v) { return (int)nearbyint(v); }
static int sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

static int g_sdiv[256], g_hdiv[256];

static void reference(const uint8_t s[3], int delta, uint8_t d[3], int &qmin, int &qmax)
{
    int b = s[0], g = s[1], r = s[2];
    int v = b > g ? b : g; v = v > r ? v : r;
    int vmin = b < g ? b : g; vmin = vmin < r ? vmin : r;
    int diff = v - vmin;
    int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    int sat = (diff * g_sdiv[v] + (1 << 11)) >> 12;
    int hue = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    hue = (hue * g_hdiv[diff] + (1 << 11)) >> 12;
    if (hue < qmin) qmin = hue;
    if (hue > qmax) qmax = hue;
    hue += hue < 0 ? 180 : 0;
    const uint8_t H = (uint8_t)sat8(hue), S = (uint8_t)sat, V = (uint8_t)sat8(v + delta);
    static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    const float hscale = 6.f / 180.f;
    float h = (float)H, fs = S * (1.f / 255.f), fv = V * (1.f / 255.f);
    float ob, og, orr;
    if (fs == 0) ob = og = orr = fv;
    else {
        float tab[4];
        h *= hscale;
        h = fmodf(h, 6.f);
        int sector = (int)floorf(h);
        h -= sector;
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        tab[0] = fv;
        tab[1] = fv * (1.f - fs);
        tab[2] = fv * (1.f - fs * h);
        tab[3] = fv * (1.f - fs * (1.f - h));
        ob = tab[sector_data[sector][0]]; og = tab[sector_data[sector][1]]; orr = tab[sector_data[sector][2]];
    }
    d[0] = (uint8_t)sat8(rne_host(ob * 255.f)); d[1] = (uint8_t)sat8(rne_host(og * 255.f)); d[2] = (uint8_t)sat8(rne_host(orr * 255.f));
}

int main(int argc, char **argv)
{
    if (argc >= 2 && strcmp(argv[1], "--bevw-selfcheck-noop") == 0) return 0;
    std::vector<int> deltas;
    for (int i = 1; i < argc; ++i) deltas.push_back(atoi(argv[i]));
    if (deltas.empty()) deltas = {-255, -128, -37, -6, -1, 0, 1, 6, 23, 128, 255};
    static HsvTables T;
    T.sdiv[0] = T.hdiv[0] = g_sdiv[0] = g_hdiv[0] = 0;
    for (int i = 1; i < 256; ++i) {
        T.sdiv[i] = g_sdiv[i] = rne_host((255 << 12) / (1. * i));
        T.hdiv[i] = g_hdiv[i] = rne_host((180 << 12) / (6. * i));
    }
    for (int i = 0; i < 256; ++i) hsv_hue_entry(i, T.hue[i].x, T.hue[i].y);
    for (int v = 0; v < 256; ++v) {
        const float fv = (float)v * (1.f / 255.f);
        if (rne_host(fv * 255.f) != v) { fprintf(stderr, "FAIL: cvRound(float(%d) / 255 * 255) = %d\n", v, rne_host(fv * 255.f)); return 1; }
    }
    int qmin = 0, qmax = 0;
    size_t bad = 0, total = 0;
    for (int delta : deltas) {
        for (uint32_t c = 0; c < (1u << 24); ++c) {
            const uint8_t s[3] = {(uint8_t)c, (uint8_t)(c >> 8), (uint8_t)(c >> 16)};
            uint8_t want[3];
            reference(s, delta, want, qmin, qmax);
            const uint32_t got = luminance_shift_bgr(c | 0xa5000000u, delta, T);   // byte 3 of the input must be ignored
            const uint32_t w = (uint32_t)want[0] | ((uint32_t)want[1] << 8) | ((uint32_t)want[2] << 16);
            ++total;
            if (got != w && bad++ < 8) fprintf(stderr, "colour %06x delta %d: got %06x, want %06x\n", c, delta, got, w);
        }
    }
    if (qmin < -30 || qmax > 150) { fprintf(stderr, "FAIL: hue quotient in [%d, %d], the table assumes [-30, 150]\n", qmin, qmax); return 1; }
    if (bad) { fprintf(stderr, "FAIL: %zu of %zu texels differ\n", bad, total); return 1; }
    printf("hsv round trip ok: %zu texels (%zu deltas x 2^24 colours), hue quotient in [%d, %d]\n", total, deltas.size(), qmin, qmax);
    return 0;
}
