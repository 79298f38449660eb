// Host-only check of the unit schedule (cameracalibration_amd/csrc/bevw_unit.h) -- runs without a GPU.
//
// The plan compiler (unit_compile) and a CPU emulation of the kernel body (unit_emulate: the per-lane steps of plan_unit_body with
// the kernels' own pair / dot-product helpers, whose host versions restate the gfx950 instructions) are run on
//   * synthetic LUTs (no arguments): a smooth map with a two-contributor stripe, a hole, a border stripe and a sparse corner, or
//   * real tables and frames handed over in a file (tests/test_unit_schedule.py: the oracle's LUTs of a bench rig),
// and every pixel a unit stores is compared with the fixed-point bilinear formula of cv2.remap evaluated directly from the LUT
// (surroundBEV.py:116-117; INTER_LINEAR, 5-bit x 5-bit weights, (sum + 512) >> 10).  Also checked: every quad of a claimed base tile
// is stored exactly once, no quad of an unclaimed base tile is touched, group lists are ascending and inside the frame set.
//
// file mode: unit_emulate <in> <out>
//   in : int32 fw fh bw bh ncams nframes has_car blend | per camera: int16 lut1[bh][bw][2], uint16 lut2[bh][bw], uint8 mask[bh][bw]
//        | uint8 frames[nframes][ncams][fh][fw][3] | uint8 car[bh][bw][3] if has_car
//   out: int32 nunits claimed_tiles lines sectors | uint8 written[bh][bw] | uint8 image[nframes][bh][bw][3] (unwritten pixels 0)
#include <cmath>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

#include "../../cameracalibration_amd/csrc/bevw_plan.h"

using namespace bevw;

#define CHECK(c, ...) do { if (!(c)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)

struct Rig {
    int fw, fh, bw, bh, ncams, nframes, blend = 0, wide = 0;
    std::vector<uint32_t> fr[4];   // wide: 21-bit fractions, [bh][bw][2]
    std::vector<int16_t> l1[4];
    std::vector<uint16_t> l2[4];
    std::vector<uint8_t> mk[4];
    std::vector<uint8_t> frames, car;
};

// the reference's arithmetic for one BEV pixel: per camera remap (fixed-point bilinear), mask (direct: select, blend: trunc(f32(v) * f32(m / 255.0)),
// surroundBEV.py:161-162 / 279-280), saturating adds in camera order (surroundBEV.py:318-320)
static int expected_px(const Rig &r, int b, int x, int y, int out[3])
{
    out[0] = out[1] = out[2] = 0;
    const size_t o = (size_t)y * r.bw + x;
    int n = 0;
    for (int c = 0; c < r.ncams; ++c) {
        const int m = r.mk[c][o];
        if (m == 0) continue;
        const int sx = r.l1[c][o * 2], sy = r.l1[c][o * 2 + 1];
        if (sx >= r.fw || sx + 1 < 0 || sy >= r.fh || sy + 1 < 0) continue;
        const int fx = r.l2[c][o] & 31, fy = (r.l2[c][o] >> 5) & 31;
        const uint8_t *f = r.frames.data() + ((size_t)b * r.ncams + c) * r.fw * r.fh * 3;
        for (int k = 0; k < 3; ++k) {
            const int p00 = f[((size_t)sy * r.fw + sx) * 3 + k], p01 = f[((size_t)sy * r.fw + sx + 1) * 3 + k];
            const int p10 = f[((size_t)(sy + 1) * r.fw + sx) * 3 + k], p11 = f[((size_t)(sy + 1) * r.fw + sx + 1) * 3 + k];
            int v = ((p00 * (32 - fx) + p01 * fx) * (32 - fy) + (p10 * (32 - fx) + p11 * fx) * fy + 512) >> 10;
            if (r.wide) {   // the analytic mode's specification (oracle/np_analytic.py: sample): fp64 bilinear, round half to even
                const double ax = (double)r.fr[c][o * 2] / 2097152.0, ay = (double)r.fr[c][o * 2 + 1] / 2097152.0;
                const double top = (1.0 - ax) * p00 + ax * p01, bot = (1.0 - ax) * p10 + ax * p11;
                v = (int)nearbyint((1.0 - ay) * top + ay * bot);
            }
            if (r.blend) v = (int)((float)v * (float)((double)m / 255.0));
            out[k] = std::min(255, out[k] + v);
        }
        ++n;
    }
    return n;
}

static bool g_allow_no_units = false;   // fuzz mode: a rig whose every base tile has a border footprint compiles nothing -- fine

static int run(const Rig &r, const char *out_path)
{
    CHECK(r.bw % 4 == 0, "this check handles BEV widths that are a multiple of 4");
    const int tiles_x = (r.bw + 31) / 32, tiles_y = (r.bh + 7) / 8;
    // BEVW_EMU_PITCH: pixels per output row (bevw_set_output_pitch); the padding columns inside the last base tile are written as zeros
    // ("aligned": rows of whole 64-byte sectors, as BEVW_PITCH_ALIGNED)
    const char *pe = getenv("BEVW_EMU_PITCH");
    const int pitch = !pe ? r.bw : (strcmp(pe, "aligned") == 0 ? (r.bw + 63) / 64 * 64 : atoi(pe));
    CHECK(pitch >= r.bw && pitch % 4 == 0, "pitch %d", pitch);
    const int bw_own = std::min(pitch, tiles_x * 32);
    std::vector<uint8_t> car_p;   // the sprite with rows of `pitch` pixels
    if (!r.car.empty()) {
        car_p.assign((size_t)pitch * r.bh * 3, 0);
        for (int y = 0; y < r.bh; ++y) memcpy(car_p.data() + (size_t)y * pitch * 3, r.car.data() + (size_t)y * r.bw * 3, (size_t)r.bw * 3);
    }
    std::vector<uint32_t> hdr0 = unit_host_headers(r.l1, r.mk, r.ncams, r.fw, r.fh, r.bw, r.bh, tiles_x, tiles_y), hdr = hdr0;
    UnitPlanHost up;
    UnitTuning tune;   // the library's knobs (csrc/bevwarp.hip: plan_build), so that partitions can be explored without a GPU
    if (const char *e = getenv("BEVW_UNIT_GROUPS")) tune.max_groups = atoi(e);
    if (const char *e = getenv("BEVW_UNIT_ROOT_W")) tune.root_w = atoi(e);
    if (const char *e = getenv("BEVW_UNIT_ROOT_H")) tune.root_h = atoi(e);
    if (const char *e = getenv("BEVW_UNIT_MIN_W")) tune.min_w = atoi(e);
    if (const char *e = getenv("BEVW_UNIT_LINE_COST")) tune.line_cost = atoi(e);
    if (const char *e = getenv("BEVW_UNIT_SECTOR_COST")) tune.sector_cost = atoi(e);
    if (const char *e = getenv("BEVW_UNIT_ALIGN_LINES")) tune.align_lines = atoi(e);
    if (const char *e = getenv("BEVW_UNIT_OWN_EMPTY")) tune.own_empty = atoi(e);
    if (const char *e = getenv("BEVW_UNIT_OWN_DOUBLE")) tune.own_double = atoi(e);
    tune.wide_double = 1;                 // as plan_build (rounds 3 - 5: 0 for blend handles)
    if (const char *e = getenv("BEVW_UNIT_WIDE_DOUBLE")) tune.wide_double = atoi(e);
    if (const char *e = getenv("BEVW_UNIT_SKEW")) tune.skew = atoi(e);
    if (const char *e = getenv("BEVW_UNIT_STAGGER")) tune.stagger = atoi(e);
    if (const char *e = getenv("BEVW_UNIT_RUN_COST")) tune.run_cost = atoi(e);
    if (const char *e = getenv("BEVW_UNIT_ROW_ORDER")) tune.row_order = atoi(e);
    if (const char *e = getenv("BEVW_UNIT_BIG")) tune.big_class = atoi(e);
    if (r.wide && tune.max_groups > kUnitMaxGroups - 1) tune.max_groups = kUnitMaxGroups - 1;   // as analytic_units_build (csrc/bevwarp.hip)
    const double t_compile = [&] {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        unit_compile(r.l1, r.l2, r.mk, r.ncams, r.fw, r.fh, r.bw, r.bh, pitch, tiles_x, tiles_y, hdr, up, tune, r.wide ? r.fr : nullptr);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        return (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    }();
    if (getenv("BEVW_EMU_TIME")) printf("unit_compile: %.1f ms (one host thread)\n", t_compile);
    if (up.desc.empty() && g_allow_no_units) { printf("unit schedule ok: no unit (every base tile left to the other classes)\n"); return 0; }
    CHECK(!up.desc.empty(), "no unit compiled");
    if (getenv("BEVW_EMU_HIST")) {   // owned pixels by unit width (diagnostics: narrow units write short row runs)
        size_t hist[9] = {}, tot = 0;
        for (const UnitDesc &d : up.desc) {
            const int w = (int)(d.shape & 0xffffu), h = (int)(d.shape >> 16);
            int b = 0;
            while ((16 << b) < w) ++b;
            hist[b] += (size_t)w * h; tot += (size_t)w * h;
        }
        for (int b = 0; b < 6; ++b) printf("width <= %3d: %5.1f %% of the unit area\n", 16 << b, 100.0 * hist[b] / tot);
    }
    const size_t set_bytes = (size_t)r.fw * r.fh * 3 * r.ncams;
    size_t claimed = 0;
    for (size_t t = 0; t < hdr.size(); ++t) {
        if (hdr[t] & kHdrBlock) {
            ++claimed;
            CHECK(!(hdr0[t] & kHdrSlow), "base tile %zu claimed although it has border footprints", t);
        }
    }
    CHECK(claimed == up.claimed_tiles, "claimed tiles %zu != %zu", claimed, up.claimed_tiles);
    // group lists
    std::vector<int> cls_of(up.desc.size(), -1);
    for (int c = 0; c < kUnitClasses; ++c)
        for (uint32_t u : up.list[c]) { CHECK(cls_of[u] < 0, "unit %u in two classes", u); cls_of[u] = c; }
    for (size_t u = 0; u < up.desc.size(); ++u) {
        CHECK(cls_of[u] >= 0, "unit %zu in no class", u);
        const UnitDesc &d = up.desc[u];
        const int GR = kUnitClassGR[cls_of[u]], NQ = kUnitClassNQ[cls_of[u]];
        const uint32_t *gs = up.gsrc.data() + (size_t)d.gs_off * kUnitThreads;
        CHECK((int)d.groups + (r.wide ? 1 : 0) <= GR * kUnitThreads, "unit %zu: %u groups in a class of %d", u, d.groups, GR * kUnitThreads);
        uint32_t last = 0, seen = 0;
        for (int s = 0; s < GR * kUnitThreads; ++s) {
            if (gs[s] == kPairNoGroup) continue;   // padding: the groups of one source line stay inside one 64-lane instruction
            CHECK(gs[s] % 12 == 0 && (size_t)gs[s] + 16 <= set_bytes, "unit %zu: group %d out of the frame set", u, s);
            CHECK(seen == 0 || gs[s] > last, "unit %zu: groups not ascending at %d", u, s);
            // every group that starts in the same 128-byte line sits in the same 64-lane instruction
            if (seen) { const int prev = s - 1; (void)prev; }
            last = gs[s]; ++seen;
        }
        CHECK(seen == d.groups, "unit %zu: %u groups listed, %u expected", u, seen, d.groups);
        for (int s = 1; s < GR * kUnitThreads; ++s)
            if (gs[s] != kPairNoGroup && gs[s - 1] != kPairNoGroup && (gs[s] >> 7) == (gs[s - 1] >> 7))
                CHECK((s >> 6) == ((s - 1) >> 6), "unit %zu: the groups of one source line straddle two load instructions at slot %d", u, s);
        const int w = (int)(d.shape & 0xffffu), h = (int)(d.shape >> 16);
        CHECK(w % 4 == 0 && w <= kUnitMaxWidth && (4 << d.lq) >= w && ((h + (64 >> d.lq) - 1) / (64 >> d.lq)) <= NQ * kUnitWaves, "unit %zu: %d x %d does not fit its class", u, w, h);
    }
    // emulate every unit on every frame
    std::vector<uint8_t> img((size_t)r.nframes * pitch * r.bh * 3, 0), written((size_t)pitch * r.bh, 0);
    for (int b = 0; b < r.nframes; ++b) {
        std::vector<uint8_t> wr((size_t)pitch * r.bh, 0);
        for (size_t u = 0; u < up.desc.size(); ++u)
            unit_emulate(up, (uint32_t)u, cls_of[u], r.frames.data() + (size_t)b * set_bytes, set_bytes, r.blend != 0, r.car.empty() ? nullptr : car_p.data(), pitch,
                         img.data() + (size_t)b * pitch * r.bh * 3, nullptr, &wr);
        if (b == 0)
            for (size_t i = 0; i < wr.size(); ++i) written[i] = (uint8_t)(wr[i] / 3);
        size_t bad = 0, off_by_one = 0, total = 0;
        for (int y = 0; y < r.bh; ++y)
            for (int x = 0; x < pitch; ++x) {
                const bool cl = x < bw_own && (hdr[(size_t)(y / 8) * tiles_x + x / 32] & kHdrBlock) != 0;
                CHECK(wr[(size_t)y * pitch + x] == (cl ? 3 : 0), "pixel (%d, %d): %d bytes stored (claimed %d; 3 = each byte once)", x, y, wr[(size_t)y * pitch + x], (int)cl);
                if (x >= r.bw) {   // padding: zeros
                    for (int k = 0; k < 3; ++k) CHECK(img[(((size_t)b * r.bh + y) * pitch + x) * 3 + k] == 0, "padding pixel (%d, %d) not zero", x, y);
                    continue;
                }
                if (!cl) continue;
                int e[3];
                expected_px(r, b, x, y, e);
                for (int k = 0; k < 3; ++k) {
                    int v = e[k];
                    if (!r.car.empty()) v = std::min(255, v + r.car[((size_t)y * r.bw + x) * 3 + k]);
                    const int got = img[(((size_t)b * r.bh + y) * pitch + x) * 3 + k];
                    ++total;
                    // wide: fp32 interpolation with 21-bit fractions against the fp64 specification -- a rounding flip here and there
                    if (r.wide && (got == v + 1 || got == v - 1) && !r.blend) { ++off_by_one; continue; }
                    if (r.wide && r.blend && got != v && std::abs(got - v) <= 1) { ++off_by_one; continue; }
                    if (got != v && bad++ < 5) fprintf(stderr, "frame %d pixel (%d, %d) channel %d: %d, expected %d\n", b, x, y, k, got, v);
                }
            }
        CHECK(bad == 0, "%zu wrong bytes in frame %d", bad, b);
        CHECK(off_by_one * 1000 <= total, "%zu of %zu bytes one LSB off the fp64 specification in frame %d", off_by_one, total, b);
        if (r.wide && b == 0) printf("wide plan: %zu of %zu bytes one LSB off the fp64 specification\n", off_by_one, total);
    }
    // sums of the balance variant against the image
    if (!r.wide) {
        uint32_t sums[3] = {0, 0, 0};
        std::vector<uint8_t> tmp((size_t)pitch * r.bh * 3, 0);
        for (size_t u = 0; u < up.desc.size(); ++u) unit_emulate(up, (uint32_t)u, cls_of[u], r.frames.data(), set_bytes, r.blend != 0, nullptr, pitch, tmp.data(), sums);
        unsigned long long want[3] = {0, 0, 0};
        for (int y = 0; y < r.bh; ++y)
            for (int x = 0; x < r.bw; ++x)
                if (hdr[(size_t)(y / 8) * tiles_x + x / 32] & kHdrBlock) {
                    int e[3];
                    expected_px(r, 0, x, y, e);
                    for (int k = 0; k < 3; ++k) want[k] += e[k];
                }
        for (int k = 0; k < 3; ++k) CHECK(sums[k] == want[k], "channel sum %d: %u, expected %llu", k, sums[k], want[k]);
    }
    // the COMPACT scratch of the balance schedule (bevw_unit.h: unit_gsrc_compact): only the sampled 4-texel groups of a frame set, 12 bytes
    // each in ascending order; the units read it through group lists of ranks.  Emulated here with the group list k_plan_touch produces on the
    // GPU (every texel of every contributor's 2 x 2 footprint), restated on the host: the image must not change by a byte.
    if (!r.wide && r.fw % 4 == 0) {
        const uint32_t gw = (uint32_t)r.fw / 4, row_bytes = (uint32_t)r.fw * 3;
        std::vector<uint8_t> touched((size_t)r.ncams * r.fh * gw, 0);
        for (int y = 0; y < r.bh; ++y)
            for (int x = 0; x < r.bw; ++x) {
                const size_t o = (size_t)y * r.bw + x;
                int count = 0;
                for (int c = 0; c < r.ncams; ++c) {
                    if (r.mk[c][o] == 0) continue;
                    const int sx = r.l1[c][o * 2], sy = r.l1[c][o * 2 + 1];
                    if (sx >= r.fw || sx + 1 < 0 || sy >= r.fh || sy + 1 < 0) continue;
                    if (count++ >= 2) continue;
                    for (int dy = 0; dy < 2; ++dy)
                        for (int dx = 0; dx < 2; ++dx) {
                            const int tx = sx + dx, ty = sy + dy;
                            if ((unsigned)tx < (unsigned)r.fw && (unsigned)ty < (unsigned)r.fh) touched[((size_t)c * r.fh + ty) * gw + tx / 4] = 1;
                        }
                }
            }
        std::vector<uint32_t> groups;
        for (size_t bit = 0; bit < touched.size(); ++bit)
            if (touched[bit]) groups.push_back((uint32_t)(bit / gw) * row_bytes + (uint32_t)(bit % gw) * 12u);
        UnitPlanHost upc = up;
        CHECK(unit_gsrc_compact(up.gsrc, groups, upc.gsrc), "a unit loads a group the sampled-group list does not hold");
        const size_t stride = unit_compact_stride(groups.size());
        std::vector<uint8_t> compact(stride, 0x5a), img2((size_t)pitch * r.bh * 3, 0);
        for (size_t i = 0; i < groups.size(); ++i) memcpy(compact.data() + i * 12, r.frames.data() + groups[i], 12);
        for (size_t u = 0; u < up.desc.size(); ++u)
            unit_emulate(upc, (uint32_t)u, cls_of[u], compact.data(), stride, r.blend != 0, r.car.empty() ? nullptr : car_p.data(), pitch, img2.data());
        CHECK(memcmp(img2.data(), img.data(), img2.size()) == 0, "the image from the compact scratch differs from the image from the frames");
        printf("compact scratch ok: %zu groups (%.1f %% of the frame set's bytes)\n", groups.size(), 100.0 * groups.size() * 12 / set_bytes);
    }
    printf("unit schedule ok: %zu units (classes", up.desc.size());
    for (int c = 0; c < kUnitClasses; ++c)
        printf(" %dx%d%s:%zu[px %zu groups %zu lines %zu sectors %zu]", kUnitClassNQ[c], kUnitClassGR[c], kUnitClassCON[c] == 2 ? "d" : "", up.list[c].size(), up.cls_pixels[c], up.cls_groups[c],
               up.cls_lines[c], up.cls_sectors[c]);
    printf("), %zu of %zu base tiles claimed, %zu source lines + %zu write sectors per frame\n", claimed, hdr.size(), up.lines, up.sectors);
    if (out_path) {
        FILE *f = fopen(out_path, "wb");
        CHECK(f, "cannot write %s", out_path);
        const int32_t head[4] = {(int32_t)up.desc.size(), (int32_t)claimed, (int32_t)up.lines, (int32_t)up.sectors};
        fwrite(head, 4, 4, f);
        for (int y = 0; y < r.bh; ++y) fwrite(written.data() + (size_t)y * pitch, 1, (size_t)r.bw, f);     // (dense rows, whatever the pitch)
        for (size_t row = 0; row < (size_t)r.nframes * r.bh; ++row) fwrite(img.data() + row * pitch * 3, 1, (size_t)r.bw * 3, f);
        fclose(f);
    }
    return 0;
}

int main(int argc, char **argv)
{
    Rig r;
    if (argc >= 2) {
        FILE *f = fopen(argv[1], "rb");
        CHECK(f, "cannot read %s", argv[1]);
        int32_t head[8];
        CHECK(fread(head, 4, 8, f) == 8, "short header");
        r.fw = head[0]; r.fh = head[1]; r.bw = head[2]; r.bh = head[3]; r.ncams = head[4]; r.nframes = head[5]; r.blend = head[7] & 1; r.wide = (head[7] >> 1) & 1;
        const size_t npx = (size_t)r.bw * r.bh;
        for (int c = 0; c < r.ncams; ++c) {
            r.l1[c].resize(npx * 2); r.l2[c].resize(npx); r.mk[c].resize(npx);
            CHECK(fread(r.l1[c].data(), 2, npx * 2, f) == npx * 2 && fread(r.l2[c].data(), 2, npx, f) == npx && fread(r.mk[c].data(), 1, npx, f) == npx, "short tables");
            if (r.wide) { r.fr[c].resize(npx * 2); CHECK(fread(r.fr[c].data(), 4, npx * 2, f) == npx * 2, "short fractions"); }
        }
        r.frames.resize((size_t)r.nframes * r.ncams * r.fw * r.fh * 3);
        CHECK(fread(r.frames.data(), 1, r.frames.size(), f) == r.frames.size(), "short frames");
        if (head[6]) { r.car.resize(npx * 3); CHECK(fread(r.car.data(), 1, r.car.size(), f) == r.car.size(), "short car"); }
        fclose(f);
        return run(r, argc >= 3 ? argv[2] : nullptr);
    }
    if (argc >= 1 && getenv("BEVW_EMU_FUZZ")) {
        // BEVW_EMU_FUZZ="seed count": random small rigs -- sizes, 1..4 cameras with smooth maps of random scale / orientation that leave the frame
        // in places, band masks with overlaps (blend weights when blend), holes, random fractions, wide plans with "no sample" patches, car
        // sprites -- each checked pixel by pixel against the formula as above
        unsigned seed = 1, count = 8;
        sscanf(getenv("BEVW_EMU_FUZZ"), "%u %u", &seed, &count);
        g_allow_no_units = true;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
        auto uni = [&](int lo, int hi) { return lo + (int)(rnd() % (unsigned)(hi - lo + 1)); };
        for (unsigned it = 0; it < count; ++it) {
            Rig q;
            q.fw = 4 * uni(16, 110); q.fh = uni(40, 260); q.bw = 4 * uni(9, 80); q.bh = uni(9, 150); q.ncams = uni(1, 4); q.nframes = 2;
            q.blend = uni(0, 1); q.wide = uni(0, 2) == 0;
            const size_t n = (size_t)q.bw * q.bh;
            for (int c = 0; c < q.ncams; ++c) {
                q.l1[c].resize(n * 2); q.l2[c].resize(n); q.mk[c].assign(n, 0);
                if (q.wide) q.fr[c].resize(n * 2);
                const double sc = 0.3 + (rnd() % 1000) / 1000.0 * 2.7, th = (rnd() % 4) * 1.5707963267948966 + ((int)(rnd() % 200) - 100) / 400.0;
                const double a = sc * cos(th), b = -sc * sin(th), d = sc * sin(th), e = sc * cos(th);
                const double cx = q.fw / 2.0 + uni(-30, 30), cy = q.fh / 2.0 + uni(-20, 20);
                const int band0 = uni(0, q.bw / 2), band1 = uni(q.bw / 2, q.bw), vertical = uni(0, 1);
                const int hx0 = uni(0, q.bw - 1), hy0 = uni(0, q.bh - 1), hw = uni(0, 70), hh = uni(0, 40);
                for (int y = 0; y < q.bh; ++y)
                    for (int x = 0; x < q.bw; ++x) {
                        const size_t o = (size_t)y * q.bw + x;
                        const double u = x - q.bw / 2.0, v = y - q.bh / 2.0;
                        const double px = a * u + b * v + cx + 0.002 * u * v, py = d * u + e * v + cy + 0.001 * u * u;
                        const double fx = floor(px), fy = floor(py);
                        q.l1[c][o * 2] = (int16_t)std::max(-3000.0, std::min(3000.0, fx));
                        q.l1[c][o * 2 + 1] = (int16_t)std::max(-3000.0, std::min(3000.0, fy));
                        q.l2[c][o] = (uint16_t)(rnd() & 1023u);
                        if (q.wide) {
                            q.fr[c][o * 2] = ((uint32_t)(q.l2[c][o] & 31) << 16) | (rnd() & 0xffffu);
                            q.fr[c][o * 2 + 1] = ((uint32_t)((q.l2[c][o] >> 5) & 31) << 16) | (rnd() & 0xffffu);
                            if (x >= hx0 / 2 && x < hx0 / 2 + hw / 2 && y >= hy0 / 2 && y < hy0 / 2 + hh) q.l1[c][o * 2] = q.l1[c][o * 2 + 1] = (int16_t)-32768;
                        }
                        const int t = vertical ? y * q.bw / std::max(1, q.bh) : x;
                        int m = (t >= band0 && t < band1) ? 255 : 0;
                        if (m && q.blend && (t - band0 < 12 || band1 - t <= 12)) m = 20 * (1 + (t - band0 < 12 ? t - band0 : band1 - 1 - t));   // a weight ramp at the band edges
                        if (x >= hx0 && x < hx0 + hw && y >= hy0 && y < hy0 + hh) m = 0;      // a hole
                        q.mk[c][o] = (uint8_t)m;
                    }
            }
            q.frames.resize((size_t)q.nframes * q.ncams * q.fw * q.fh * 3);
            for (uint8_t &v : q.frames) v = (uint8_t)(rnd() >> 4);
            if (uni(0, 1)) {
                q.car.assign(n * 3, 0);
                const int x0 = uni(0, q.bw - 1), y0 = uni(0, q.bh - 1);
                for (int y = y0; y < std::min(q.bh, y0 + 40); ++y) for (int x = x0; x < std::min(q.bw, x0 + 60); ++x) for (int k = 0; k < 3; ++k) q.car[((size_t)y * q.bw + x) * 3 + k] = (uint8_t)(rnd() >> 3);
            }
            printf("fuzz %u: %d x %d frames, %d x %d BEV, %d cameras, blend %d, wide %d, car %d: ", it, q.fw, q.fh, q.bw, q.bh, q.ncams, q.blend, q.wide, (int)!q.car.empty());
            fflush(stdout);
            if (run(q, nullptr)) return 1;
        }
        return 0;
    }
    // synthetic rig: two cameras, 520 x 300 frames, 328 x 150 BEV (partial tiles at the right / bottom edge)
    r.fw = 520; r.fh = 300; r.bw = 328; r.bh = 150; r.ncams = 2; r.nframes = 2;
    const size_t npx = (size_t)r.bw * r.bh;
    for (int c = 0; c < 2; ++c) { r.l1[c].resize(npx * 2); r.l2[c].resize(npx); r.mk[c].assign(npx, 0); }
    for (int y = 0; y < r.bh; ++y)
        for (int x = 0; x < r.bw; ++x) {
            const size_t o = (size_t)y * r.bw + x;
            // camera 0: dense on the left (3/4 texel per pixel), sparse towards the right (up to 3 texels per pixel, rows 2 apart)
            const int sx0 = x < 160 ? x * 3 / 4 + 3 : 123 + (x - 160) * 2 + (y & 1), sy0 = x < 160 ? y * 3 / 4 + 5 : y * 2 - 3;
            r.l1[0][o * 2] = (int16_t)sx0; r.l1[0][o * 2 + 1] = (int16_t)sy0;
            r.l2[0][o] = (uint16_t)(((x * 7 + y) & 31) | (((y * 5 + x) & 31) << 5));
            r.mk[0][o] = 255;
            // camera 1: a rotated view (BEV y runs along source x) that takes over in a stripe of columns 96 .. 135 with an overlap at 96 .. 103
            r.l1[1][o * 2] = (int16_t)(400 - y); r.l1[1][o * 2 + 1] = (int16_t)(x / 2 + 20);
            r.l2[1][o] = (uint16_t)(((x * 3) & 31) | (((y * 11) & 31) << 5));
            if (x >= 96 && x < 136) { r.mk[1][o] = 255; if (x >= 104) r.mk[0][o] = 0; }
        }
    for (int y = 40; y < 56; ++y) for (int x = 192; x < 256; ++x) r.mk[0][(size_t)y * r.bw + x] = 0;    // a hole (empty base tiles)
    for (int y = 100; y < 108; ++y) for (int x = 0; x < 32; ++x) r.mk[0][(size_t)y * r.bw + x] = 200;   // a blend weight: not a unit's tile
    r.frames.resize((size_t)r.nframes * r.ncams * r.fw * r.fh * 3);
    uint32_t seed = 12345u;
    for (uint8_t &v : r.frames) { seed = seed * 1664525u + 1013904223u; v = (uint8_t)(seed >> 24); }
    if (run(r, nullptr)) return 1;
    r.blend = 1;            // the same tables read as blend weights (the 200 stripe is a real weight now, the overlap columns add two weighted terms)
    if (run(r, nullptr)) return 1;
    r.blend = 0;
    // once more with a car sprite over part of the image
    r.car.assign(npx * 3, 0);
    for (int y = 30; y < 90; ++y) for (int x = 50; x < 210; ++x) for (int k = 0; k < 3; ++k) r.car[((size_t)y * r.bw + x) * 3 + k] = (uint8_t)(x + 2 * y + 40 * k);
    if (run(r, nullptr)) return 1;
    // wide plans (analytic mode): the same geometry with 21-bit fractions, direct and blend; some pixels sample nothing (INT16_MIN mark)
    r.wide = 1;
    for (int c = 0; c < 2; ++c) {
        r.fr[c].resize(npx * 2);
        for (size_t o = 0; o < npx; ++o) {
            seed = seed * 1664525u + 1013904223u;
            r.fr[c][o * 2] = ((uint32_t)(r.l2[c][o] & 31) << 16) | (seed >> 16);
            seed = seed * 1664525u + 1013904223u;
            r.fr[c][o * 2 + 1] = ((uint32_t)((r.l2[c][o] >> 5) & 31) << 16) | (seed >> 16);
        }
    }
    for (int y = 60; y < 75; ++y) for (int x = 10; x < 75; ++x) r.l1[0][((size_t)y * r.bw + x) * 2] = r.l1[0][((size_t)y * r.bw + x) * 2 + 1] = (int16_t)-32768;
    if (run(r, nullptr)) return 1;
    r.blend = 1;
    return run(r, nullptr);
}
