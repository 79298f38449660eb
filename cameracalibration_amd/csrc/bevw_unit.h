// bevw_unit.h -- the UNIT schedule of the tile plan (included by bevw_plan.h after bevw_block.h; round 3).
//
// Why.  The stitch is bound by the NUMBER of vector-L1 -> L2 requests.  Round 2's schedule paid, per
// frame of BASELINE config 3, ~61 k source-line requests (13 k for the dense 64 x 32 block tiles, 48 k for the per-wave
// 32 x 8 classes that hold the 29 % of the tiles no block tile could take: every one of those waves fetches the lines of
// its own small footprint) and 75 k write requests (rows of 96 / 192 bytes straddle 64-byte sectors).  The floor of the
// geometry is 22 k + 55 k.  A UNIT is what the plan compiler makes of that arithmetic:
//
//   * the BEV is cut by a k-d partition into rectangles of <= 4096 pixels (up to 256 pixels wide) whose distinct source
//     texel groups fit ONE LDS patch of <= 1024 groups (32 KB of pair entries) -- wide and flat where the BEV x axis runs
//     along source rows (front / back cameras), narrow and tall where the BEV y axis does (left / right), small where every
//     pixel samples its own texels (near the car).  A split is made only when a rectangle does not fit, and in the direction
//     that costs fewer source lines + write sectors (counted on the CPU from the LUTs: unit_compile below);
//   * one block of 4 waves owns a unit for the frames of its batch chunk: per frame every lane loads up to GR texel groups
//     (buffer_load_dwordx4, ascending list, masked lanes touch no memory), converts them ONCE into pair entries
//     (bevw_pair.h) in the block's patch, and then interpolates up to NQ pixel quads from it;
//   * a wave-store writes the longest row runs the unit's width allows: one run of 768 contiguous bytes for a 256-pixel
//     wide unit, two of 384 for 128 (13.0 / 13.9 write requests per 256 pixels instead of 15.5 / 19.1).
//
// Classes (NQ quads per lane x GR rounds of 256 groups): dense far field (4, 1) and (4, 2) -- double-buffered patch, ONE
// barrier per frame, as the block tiles of round 2; mid and near field (2, 4) and (1, 4) -- the patch takes the whole 32 KB,
// two barriers per frame.  (A (4, 4) class needs 163 VGPRs = 3 waves per SIMD for every class of the merged launch; without
// it the partition pays 1.5 % more requests and every class stays below 128.)  Units are classes of the merged launch
// (k_plan_units: 256 threads, 32 KB).
// Base tiles with a two-contributor pixel (seams, blend overlaps), a blend weight below 255 or a frame-border footprint keep
// their round-2 classes: a unit never stores a quad of such a base tile (kUnitSkip).  Every unit pixel therefore has weight
// 255 = 1.0f exactly, and the kernel needs no blend variant: trunc(f32(v) * 1.0f) == v.  Base tiles without any contributor
// (under the car) ARE unit tiles: a unit without groups only writes (car sprite or zeros) in whole row runs.
//
// Plan format.  4 bytes per pixel: pair k (2 bits) | group slot of footprint row 0 (10) | of row 1 (10) | fx (5) | fy (5); equal
// slots = no contributor.  The patch holds the four pair entries of group slot s at s * 32 bytes, so the low 12 bits of an
// entry ARE the qword index of its row-0 pair.  Group slots are dealt so that the groups that start in one 128-byte source
// line never straddle two 64-lane load instructions (each line is requested once: round 3's first version, with plain
// ascending slots, requested every sixth line twice).
//
// WIDE plans (the analytic projection mode, bevw_set_projection: positions from the camera model instead of the reference's
// tables, row n1 of DESIGN.md section 0) carry 21-bit fractions: a second dword per pixel holds the low 16 bits of fx and fy,
// the 5-bit fields of the entry the high bits.  The interpolation then runs in fp32 (bilinear_pairs_f32) on the same patch.
// A pixel without a contributor points both rows at a group slot no group was dealt to (the masked lane's load returns zeros,
// which the conversion keeps): it interpolates to 0 whatever its fractions.
//
// The plan is compiled on the HOST (unit_compile); tests/native/unit_emulate.cpp runs the same compiler and the per-lane
// arithmetic below (unit_emulate) on a CPU, so the indexing of this file is checked without a GPU.
#pragma once
#include <algorithm>
#include <cstring>
#include <vector>

// format of the wave-store (unit_store_quad16 below): 0 = 12 bytes per lane; 1 = lane quads repacked into 3 x 16 bytes where the output rows
// are whole 64-byte sectors (the streaming layout); 2 = in every layout.  Round 6's A/B (profiles/r06/README.md section 1): the format is NOT
// what limits the store stream -- the vector L1 merges either form into one write request per 64-byte sector (TCP_TCC_WRITE_REQ is the same),
// the microbenchmark's rates agree to +-2 %, and the repack's 11 VALU instructions per quad slot cost config 3 4 % -> 0 stays the default.
#ifndef BEVW_UNIT_STORE16
#define BEVW_UNIT_STORE16 0
#endif

namespace bevw {

constexpr int kUnitWaves = 4;
constexpr int kUnitThreads = kUnitWaves * 64;
constexpr int kUnitMaxNQ = 4;                // pixel quads per lane
constexpr int kUnitMaxGR = 4;                // rounds of kUnitThreads groups
constexpr int kUnitMaxGroups = kUnitMaxGR * kUnitThreads;        // 1024 group slots = 32 KB of pair entries
constexpr int kUnitMaxWidth = 256;           // pixels: 64 quads = one wave-store per row
constexpr int kUnitClasses = 8;               // five single-contributor classes (7 = the big one), three for units with up to two contributors per pixel
constexpr int kUnitClassNQ[kUnitClasses] = {4, 4, 2, 1, 2, 1, 1, 4};
constexpr int kUnitClassGR[kUnitClasses] = {1, 2, 4, 4, 1, 4, 1, 4};
constexpr int kUnitClassCON[kUnitClasses] = {1, 1, 1, 1, 2, 2, 2, 1};
constexpr int kUnitClassBig = 7;              // 4096 pixels AND 1024 groups (round 4: rounds 2 - 3 could not afford its ~160 VGPRs beside the retired schedules)
constexpr uint32_t kUnitSkip = 1u << 22;     // entry of pixel 0 of a quad, with equal slots: the quad's base tile belongs to another class -> not stored

// unit descriptor: 8 dwords, read with scalar loads
struct UnitDesc {
    uint32_t pos;        // x0 (int16: the unskewed left edge, may be negative) | y0 << 16 (pixels)
    uint32_t shape;      // w | h << 16 (pixels; w a multiple of 4)
    uint32_t ent_off;    // entries of the unit start at un_entries[ent_off * 64]: per quad slot one uint4 per lane (4 pixels); two-contributor
                         // classes three (first entries, second entries, the two u8 blend weights of every pixel as w0 | w1 << 8)
    uint32_t gs_off;     // group offsets of the unit start at un_gsrc[gs_off * kUnitThreads]
    uint32_t lq;         // log2 of the lanes per row (quads per row rounded up to a power of two, >= 4)
    uint32_t sum_tile;   // a base tile owned by this unit (diagnostics; the channel sums of the balance path are indexed by the unit itself)
    uint32_t groups;     // distinct groups; 0 = the unit only writes (no contributor anywhere)
    uint32_t pixels;     // contributing pixels (diagnostics)
};

// lane -> quad of the unit: slot `sidx` (0 .. 4 NQ - 1) covers 64 >> lq consecutive rows of (1 << lq) quads
__host__ __device__ __forceinline__ void unit_quad(uint32_t lq, int sidx, int lane, int &qx, int &row)
{
    qx = lane & ((1 << lq) - 1);
    row = sidx * (64 >> lq) + (lane >> lq);
}
// Row shift of the unit grid: row y of every unit is shifted right by unit_skew(c, y) pixels, c chosen so that the byte offset of pixel
// (64 k + shift, y) inside the image is a multiple of 64 for every k: interior unit boundaries then fall on 64-byte sector boundaries in
// EVERY row, and no sector is written in two pieces by two blocks (a partially written sector costs the memory system far more than
// a request: DESIGN.md section 4).  c = 0: plain rectangles.
// The constant carries the multiplier in bits 0..7 and the period mask (15 for 64-byte, 7 for 32-byte alignment) in bits 8..15.
__host__ __device__ __forceinline__ int unit_skew(uint32_t c, int y) { return 4 * (int)(((c & 255u) * (uint32_t)y) & (c >> 8)); }
// slot index of (wave, quad slot j): rows are dealt to the waves round-robin, so the waves of a block work on neighbouring rows
__host__ __device__ __forceinline__ int unit_slot(int wave, int j) { return j * kUnitWaves + wave; }
// plan entry of one pixel
__host__ __device__ __forceinline__ uint32_t unit_entry(uint32_t slot0, uint32_t slot1, uint32_t k, uint32_t code)
{
    return k | (slot0 << 2) | (slot1 << 12) | ((code & 1023u) << 22);
}
// decoded: qword indices of the two pair entries inside the frame's patch, x weights {32 - fx, fx, 0, 0}, y weights x 64
__host__ __device__ __forceinline__ void unit_decode(uint32_t e, uint32_t &i0, uint32_t &i1, uint32_t &wxa, uint32_t &wy)
{
    const uint32_t fx = (e >> 22) & 31u, fy = e >> 27;
    i0 = e & 0xfffu;
    i1 = ((e >> 10) & 0xffcu) | (e & 3u);
    wxa = (i0 >> 2) != (i1 >> 2) ? ((32u - fx) | (fx << 8)) : 0u;   // equal slots: no contributor -> zero x weights add exactly 0
    wy = ((32u - fy) << 6) | (fy << 22);
}

constexpr uint32_t kUnitFracBits = 21;        // wide plans: fraction = value / 2^21
// wide plans: i0, i1 as unit_decode; the fractions as floats
__host__ __device__ __forceinline__ void unit_decode_wide(uint32_t e, uint32_t f, uint32_t &i0, uint32_t &i1, float &fx, float &fy)
{
    i0 = e & 0xfffu;
    i1 = ((e >> 10) & 0xffcu) | (e & 3u);
    fx = (float)((((e >> 22) & 31u) << 16) | (f & 0xffffu)) * (1.0f / (float)(1u << kUnitFracBits));
    fy = (float)(((e >> 27) << 16) | (f >> 16)) * (1.0f / (float)(1u << kUnitFracBits));
}
// 16-byte store format: dwords q .. q + 3 of S = {d0, d1, d2, n0, n1, n2} -- lane q (0 .. 2) of a lane quad holds d, its right neighbour n
__host__ __device__ __forceinline__ void unit_repack16(int q, const uint32_t d[3], const uint32_t n[3], uint32_t o[4])
{
    o[0] = q == 0 ? d[0] : (q == 1 ? d[1] : d[2]);
    o[1] = q == 0 ? d[1] : (q == 1 ? d[2] : n[0]);
    o[2] = q == 0 ? d[2] : (q == 1 ? n[0] : n[1]);
    o[3] = q == 0 ? n[0] : (q == 1 ? n[1] : n[2]);
}
__host__ __device__ __forceinline__ bool unit_no_contributor(uint32_t e) { return ((e >> 2) & 1023u) == ((e >> 12) & 1023u); }
// one pixel from its two pair entries in fp32: b0 b1 g0 g1 | r0 r1 of both footprint rows -> B | G << 8 | R << 16, round half to even.
// (fmaf on both sides: the CPU emulation computes the same bits as the kernel.)
__host__ __device__ __forceinline__ uint32_t bilinear_pairs_f32(uint2 q0, uint2 q1, float fx, float fy)
{
    auto lerp = [](float a, float b, float t) { return fmaf(t, b - a, a); };
    auto ch = [&](uint32_t w0, uint32_t w1, int s) {
        const float t = lerp((float)((w0 >> s) & 255u), (float)((w0 >> (s + 8)) & 255u), fx);
        const float b = lerp((float)((w1 >> s) & 255u), (float)((w1 >> (s + 8)) & 255u), fx);
        return (uint32_t)rintf(lerp(t, b, fy));     // a convex combination of bytes: 0 .. 255
    };
    return ch(q0.x, q1.x, 0) | (ch(q0.x, q1.x, 16) << 8) | (ch(q0.y, q1.y, 0) << 16);
}

// base-tile headers of a set of tables as k_plan_build leaves them (bevw_plan.h): second contributor / border footprint / empty.
// (Host twin of that kernel's flags, for plans that are compiled from tables the GPU plan builder never saw: the analytic mode, and
// tests/native/unit_emulate.cpp.)  A footprint with sx + 1 < 0 (e.g. the analytic map's "no sample" mark, INT16_MIN) contributes nothing.
static inline std::vector<uint32_t> unit_host_headers(const std::vector<int16_t> lut1[4], const std::vector<uint8_t> mask[4], int ncams, int fw, int fh,
                                                      int bw, int bh, int tiles_x, int tiles_y);

struct UnitPlanHost {
    std::vector<UnitDesc> desc;
    std::vector<uint32_t> entries;             // per unit: [4 NQ slots][64 lanes][4 pixels]
    std::vector<uint32_t> gsrc;                // per unit: [GR rounds][256 lanes]
    std::vector<uint32_t> list[kUnitClasses];  // unit ids by class
    std::vector<uint32_t> all;                 // every unit in partition order: id | class << 28
    size_t claimed_tiles = 0;
    uint32_t skew = 0;                         // unit_skew constant of this plan
    bool wide = false;                         // 21-bit fractions: one more uint4 per lane, quad slot and contributor
    // request arithmetic of the compiled partition (per frame): distinct 128-byte source lines, 64-byte write sectors
    size_t lines = 0, sectors = 0;
    size_t cls_lines[kUnitClasses] = {}, cls_sectors[kUnitClasses] = {}, cls_pixels[kUnitClasses] = {}, cls_groups[kUnitClasses] = {};
};

struct UnitTuning {
    int max_groups = kUnitMaxGroups;
    int root_w = 256, root_h = 64;   // the k-d partition starts from cells of this size
    int min_w = 16;                  // narrowest unit (pixels)
    // cost of a cut = line_cost x source lines + sector_cost x write sectors: a read request holds its place in the L1's miss queue about
    // twice as long as a write request (~1200 against ~520 clk), but a sector that two units share is written in two
    // pieces, and a partially written sector that reaches the memory costs about as much as a read
    int line_cost = 2, sector_cost = 3;
    int align_lines = 1;             // group slots: the groups of one source line stay inside one 64-lane load instruction
    int own_empty = 1;               // base tiles without a contributor are unit tiles
    int own_double = 1;              // base tiles with a second contributor or a blend weight are unit tiles (classes with two entries per pixel)
    int wide_double = 1;             // the two-quad class of those units (BEVW_UNIT_WIDE_DOUBLE=0: as rounds 3 - 5 compiled blend handles)
    int skew = 0;                    // unit boundaries aligned to this many bytes in every row (unit_skew): 0 (off), 32 or 64
    int row_order = 0;               // launch order of the rows of root cells (see the end of unit_compile)
    int own_padding = 1;             // the padding columns of a pitched output are written (zeros) by the units at the right edge
    int run_cost = 0;                // cost of one more row run (experiment: wider units write longer runs; profiles/r03/sweeps.log)
    int stagger = 0;                 // pixels the column grid of the root cells shifts per row of cells (0: the same columns in every row)
    int big_class = 1;               // the (4 quads, 4 rounds) class: fewer cuts where a cell has many pixels AND many groups
};

// Host-side plan compiler of the units.  tables: host copies of the LUTs of every camera.  hdr: base-tile headers (32 x 8 tiles,
// tiles_x per row) as k_plan_build / k_plan_pair_build left them; base tiles claimed by a unit get kHdrBlock.  `pitch` = pixels per
// output row (bw rounded up to 4).  The contributor rule is k_plan_build's.
static inline void unit_compile(const std::vector<int16_t> lut1[4], const std::vector<uint16_t> lut2[4], const std::vector<uint8_t> mask[4],
                                int ncams, int fw, int fh, int bw, int bh, int pitch, int tiles_x, int tiles_y, std::vector<uint32_t> &hdr,
                                UnitPlanHost &out, const UnitTuning &tune = UnitTuning(), const std::vector<uint32_t> *frac = nullptr)
{
    // frac != nullptr: a WIDE plan -- frac[c][2 o], frac[c][2 o + 1] = the 21-bit fractions of pixel o (lut2 is not read)
    const bool wide = frac != nullptr;
    out.wide = wide;
    const uint32_t frame_bytes = (uint32_t)fw * fh * 3, gpr = (uint32_t)fw / 4;
    const size_t set_bytes = (size_t)frame_bytes * ncams;
    constexpr uint32_t kNone = 0xffffffffu;
    // ---- per pixel: footprint offset, fraction code and weight of up to two contributors; base tiles a unit may own ------------
    // own: 0 = not a unit tile (border footprints, claimed already), 1 = single-contributor unit tile (every weight 255),
    //      2 = unit tile with a second contributor or a blend weight somewhere (seams, blend overlaps)
    std::vector<uint8_t> own((size_t)tiles_x * tiles_y, 0);
    for (size_t t = 0; t < own.size(); ++t) {
        own[t] = (hdr[t] & (kHdrSlow | kHdrBlock)) ? 0 : ((hdr[t] & kHdrSecond) ? (tune.own_double ? 2 : 0) : 1);
        if ((hdr[t] & kHdrEmpty) && !tune.own_empty) own[t] = 0;
    }
    std::vector<uint32_t> poff[2], pcode[2], pfrac[2];
    std::vector<uint8_t> pmask[2];
    for (int k = 0; k < 2; ++k) {
        poff[k].assign((size_t)pitch * bh, kNone); pcode[k].assign((size_t)pitch * bh, 0u); pmask[k].assign((size_t)pitch * bh, 0);
        if (wide) pfrac[k].assign((size_t)pitch * bh, 0u);
    }
    for (int y = 0; y < bh; ++y)
        for (int x = 0; x < bw; ++x) {
            const size_t t = (size_t)(y / 8) * tiles_x + x / 32;
            if (!own[t]) continue;
            const size_t o = (size_t)y * bw + x;
            int count = 0;
            for (int c = 0; c < ncams; ++c) {
                const uint32_t m = mask[c][o];
                if (m == 0) continue;
                const int sx = lut1[c][o * 2], sy = lut1[c][o * 2 + 1];
                if (sx >= fw || sx + 1 < 0 || sy >= fh || sy + 1 < 0) continue;   // whole footprint outside: adds 0
                const uint32_t off = (uint32_t)c * frame_bytes + ((uint32_t)sy * fw + sx) * 3;
                if ((size_t)(off / 12u + gpr) * 12u + 16u > set_bytes || count >= 2) { own[t] = 0; break; }   // the last group's window would overrun
                if (m != 255u && own[t] == 1) own[t] = tune.own_double ? 2 : 0;                            // a blend weight
                poff[count][(size_t)y * pitch + x] = off;
                if (wide) {
                    const uint32_t fx = frac[c][o * 2] & ((1u << kUnitFracBits) - 1u), fy = frac[c][o * 2 + 1] & ((1u << kUnitFracBits) - 1u);
                    pcode[count][(size_t)y * pitch + x] = (fx >> 16) | ((fy >> 16) << 5);
                    pfrac[count][(size_t)y * pitch + x] = (fx & 0xffffu) | (fy << 16);
                } else {
                    pcode[count][(size_t)y * pitch + x] = lut2[c][o] & (kQTab2 - 1);
                }
                pmask[count][(size_t)y * pitch + x] = (uint8_t)m;
                ++count;
            }
        }
    for (int y = 0; y < bh; ++y)      // base tiles dropped above: their pixels leave the units
        for (int x = 0; x < bw; ++x)
            if (own[(size_t)(y / 8) * tiles_x + x / 32] == 0) poff[0][(size_t)y * pitch + x] = poff[1][(size_t)y * pitch + x] = kNone;
    int pass = 1;                     // the kind of base tile the partition pass at hand owns (1: single, 2: double)
    // Rows wider than the image (an output pitch of whole sectors, bevw_set_output_pitch): the quads of the padding columns belong to the
    // unit of the base tile they lie in -- they have no contributor and are written as zeros, so that the LAST sector of every row reaches
    // the memory whole too (a partially written sector costs about as much as ten whole ones: tools/store_pattern.hip,
    // profiles/r03/store_pattern_units.log: 0.239 -> 0.200 ms for the store stream of a 256-frame batch)
    const int bw_own = tune.own_padding ? std::min(pitch, tiles_x * 32) : bw;
    auto owned = [&](int x, int y) { return x < bw_own && y < bh && own[(size_t)(y / 8) * tiles_x + x / 32] == pass; };
    // rows of P = 3 pitch bytes: byte P y + 12 q is a multiple of 64 <=> q = -(P / 4) * 11 * y (mod 16)   (3 * 11 = 1 mod 16).  Only when
    // every image of a batch starts on a sector boundary (P * bh a multiple of 64; the batch base is assumed 64-byte aligned -- with any
    // other base the plan is still correct, just not sector-aligned).
    const uint32_t P = (uint32_t)pitch * 3u;
    uint32_t skew = 0;
    int col_step = 16;
    if ((tune.skew == 64 || tune.skew == 32) && ((size_t)P * bh) % (size_t)tune.skew == 0) {
        const uint32_t m = (uint32_t)tune.skew / 4u, inv = tune.skew == 64 ? 11u : 3u;   // 3 * 11 = 1 (mod 16), 3 * 3 = 1 (mod 8)
        const uint32_t c = (m - ((P / 4u) * inv) % m) % m;
        if (c != 0) { skew = c | ((m - 1u) << 8); col_step = tune.skew; }    // c == 0: the rows are aligned already
        else col_step = tune.skew;
    }
    out.skew = skew;
    const int min_w = std::max(col_step, tune.min_w);
    auto px_x = [&](int u, int y) { return u + unit_skew(skew, y); };   // unskewed column -> image column (may lie outside [0, pitch))

    // ---- request arithmetic of a rectangle: distinct groups, distinct 128-byte lines, contributing pixels -------------------
    std::vector<uint32_t> gstamp(set_bytes / 12 + 2, 0u), lstamp(set_bytes / 128 + 2, 0u);
    uint32_t stamp = 0;
    struct Stats { int groups, lines, pixels, quads; };
    auto cell_stats = [&](int x0, int y0, int w, int h) {
        Stats s = {0, 0, 0, 0};
        ++stamp;
        for (int y = y0; y < y0 + h; ++y)
            for (int u = x0; u < x0 + w; ++u) {
                const int x = px_x(u, y);
                if (x < 0 || x >= pitch) continue;
                if (!owned(x & ~3, y)) continue;
                if ((x & 3) == 0) ++s.quads;
                for (int con = 0; con < 2; ++con) {
                    const uint32_t off = poff[con][(size_t)y * pitch + x];
                    if (off == kNone) continue;
                    ++s.pixels;
                    for (uint32_t r = 0; r < 2; ++r) {
                        const uint32_t k = off / 12u + r * gpr;
                        if (gstamp[k] == stamp) continue;
                        gstamp[k] = stamp;
                        ++s.groups;
                        const uint32_t a = k * 12u;
                        for (uint32_t l = a >> 7; l <= (a + 15u) >> 7; ++l)
                            if (lstamp[l] != stamp) { lstamp[l] = stamp; ++s.lines; }
                    }
                }
            }
        return s;
    };
    auto write_sectors = [&](int x0, int y0, int w, int h) {   // 64-byte sectors the row runs of the rectangle's OWNED quads touch
        long n = 0;
        for (int y = y0; y < y0 + h; ++y) {
            const int xa = std::max(0, px_x(x0, y)), xb = std::min(pitch, px_x(x0 + w, y));
            int run0 = -1;
            for (int x = xa; x <= xb; x += 4) {
                const bool in = x < xb && owned(x, y);
                if (in && run0 < 0) run0 = x;
                if (!in && run0 >= 0) {
                    const long a0 = ((long)y * pitch + run0) * 3, a1 = ((long)y * pitch + x) * 3;
                    n += (a1 - 1) / 64 - a0 / 64 + 1;
                    run0 = -1;
                }
            }
        }
        return n;
    };
    // the rectangle shrunk to the bounding box of its owned quads (columns stay on the column grid)
    auto shrink = [&](int &x0, int &y0, int &w, int &h) {
        int ya = y0 + h, yb = y0, ua = x0 + w, ub = x0;
        for (int y = y0; y < y0 + h; ++y)
            for (int u = x0; u < x0 + w; u += 4) {
                const int x = px_x(u, y);
                if (x < 0 || x >= pitch || !owned(x, y)) continue;
                ya = std::min(ya, y); yb = std::max(yb, y + 1); ua = std::min(ua, u); ub = std::max(ub, u + 4);
            }
        if (yb <= ya) return;
        const int xa = x0 + (ua - x0) / col_step * col_step, xb = std::min(x0 + w, x0 + (ub - x0 + col_step - 1) / col_step * col_step);
        x0 = xa; w = xb - xa; y0 = ya; h = yb - ya;
    };
    auto lanes_log2 = [](int w) { int lq = 2; while ((4 << lq) < w) ++lq; return lq; };   // quads per row rounded up to 4 .. 64 lanes
    auto slots_needed = [&](int w, int h) { const int rps = 64 >> lanes_log2(w); return (h + rps - 1) / rps; };

    // ---- emit one unit ---------------------------------------------------------------------------------------------------------
    const int root_w = std::min(tune.root_w, kUnitMaxWidth);
    struct Order { uint64_t key; uint32_t entry; uint32_t row; uint64_t cost; };
    // left edge of the root cells of the row of cells that holds y0.  tune.stagger: the column grid shifts from one row of cells to the next
    // (experiment: blocks that cover the same columns in every row write slower than blocks whose phase cycles, tools/store_pattern.hip)
    auto row_start = [&](int y0) {
        int xs = skew ? -col_step : 0;
        if (tune.stagger > 0) xs -= (((y0 / tune.root_h) * tune.stagger) % root_w) / col_step * col_step;
        return xs;
    };
    std::vector<Order> order;       // every unit with its place in the launch order: root cell by root cell, both passes interleaved
    std::vector<uint32_t> keys, slot;
    auto emit = [&](int x0, int y0, int w, int h, const Stats &st) {
        keys.clear();
        for (int y = y0; y < y0 + h; ++y)
            for (int u = x0; u < x0 + w; ++u) {
                const int x = px_x(u, y);
                if (x < 0 || x >= pitch || !owned(x & ~3, y)) continue;
                for (int con = 0; con < 2; ++con) {
                    const uint32_t off = poff[con][(size_t)y * pitch + x];
                    if (off == kNone) continue;
                    keys.push_back(off / 12u);
                    keys.push_back(off / 12u + gpr);
                }
            }
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        const int count = (int)keys.size();
        // group slots in ascending address order; the groups that START in one 128-byte line are kept inside one 64-lane instruction
        slot.assign((size_t)count, 0u);
        uint32_t used = 0;
        for (int i = 0; i < count;) {
            int j = i;
            while (j < count && (keys[(size_t)j] * 12u) >> 7 == (keys[(size_t)i] * 12u) >> 7) ++j;
            if (tune.align_lines && (used & 63u) + (uint32_t)(j - i) > 64u) used = (used + 63u) & ~63u;
            for (; i < j; ++i) slot[(size_t)i] = used++;
        }
        const int lq = lanes_log2(w), slots = slots_needed(w, h);
        // class: the cheapest (NQ, GR) that holds the unit (cost ~ 4 pixels x 14 VALU per quad slot, 8 v_perm + 2 ds_write per round)
        int cls = -1, best = 1 << 30;
        for (int c = 0; c < kUnitClasses; ++c) {
            if (kUnitClassCON[c] != pass || kUnitClassNQ[c] * kUnitWaves < slots || kUnitClassGR[c] * kUnitThreads < (int)used + (wide ? 1 : 0)) continue;
            if (c == 4 && !tune.wide_double) continue;
            if (c == kUnitClassBig && !tune.big_class) continue;
            const int cost = kUnitClassNQ[c] * 64 + kUnitClassGR[c] * 14;
            if (cost < best) { best = cost; cls = c; }
        }
        if (cls < 0) return false;
        const int NQ = kUnitClassNQ[cls], GR = kUnitClassGR[cls], fpart = pass == 2 ? 3 : 1, parts = fpart + (wide ? pass : 0);
        // wide: a pixel without a contributor reads the zero slot (the first one no group was dealt to)
        const uint32_t zent = wide ? unit_entry(used, used, 0u, 0u) : 0u;
        UnitDesc d;
        d.pos = (uint32_t)(uint16_t)(int16_t)x0 | ((uint32_t)y0 << 16);
        d.shape = (uint32_t)w | ((uint32_t)h << 16);
        d.ent_off = (uint32_t)(out.entries.size() / 256);
        d.gs_off = (uint32_t)(out.gsrc.size() / kUnitThreads);
        d.lq = (uint32_t)lq;
        d.sum_tile = 0;
        d.groups = (uint32_t)count;
        d.pixels = (uint32_t)st.pixels;
        auto slot_of = [&](uint32_t key) { return slot[(size_t)(std::lower_bound(keys.begin(), keys.end(), key) - keys.begin())]; };
        const size_t e0 = out.entries.size();
        out.entries.resize(e0 + (size_t)NQ * kUnitWaves * parts * 64 * 4, 0u);
        bool have_sum_tile = false;
        for (int sidx = 0; sidx < NQ * kUnitWaves; ++sidx)
            for (int lane = 0; lane < 64; ++lane) {
                int qx, row;
                unit_quad((uint32_t)lq, sidx, lane, qx, row);
                const int y = y0 + row, x = px_x(x0 + 4 * qx, y);
                uint32_t *e = out.entries.data() + e0 + (((size_t)sidx * parts) * 64 + lane) * 4;   // the lane's 4 pixels; part k at e + k * 256
                if (wide)
                    for (int p = 0; p < 4; ++p)
                        for (int con = 0; con < pass; ++con) e[con * 256 + p] = zent;
                if (4 * qx >= w || row >= h || x < 0 || x >= pitch) continue;   // lane without a quad: zero entries, masked in the kernel
                if (!owned(x, y)) { e[0] = zent | kUnitSkip; continue; }
                if (!have_sum_tile) { d.sum_tile = (uint32_t)((y / 8) * tiles_x + x / 32); have_sum_tile = true; }
                for (int p = 0; p < 4; ++p)
                    for (int con = 0; con < pass; ++con) {
                        const uint32_t off = poff[con][(size_t)y * pitch + x + p];
                        if (off == kNone) continue;
                        const uint32_t key = off / 12u, pk = (off - key * 12u) / 3u;
                        e[con * 256 + p] = unit_entry(slot_of(key), slot_of(key + gpr), pk, pcode[con][(size_t)y * pitch + x + p]);
                        if (pass == 2) e[2 * 256 + p] |= (uint32_t)pmask[con][(size_t)y * pitch + x + p] << (8 * con);
                        if (wide) e[(fpart + con) * 256 + p] = pfrac[con][(size_t)y * pitch + x + p];
                    }
            }
        const size_t g0 = out.gsrc.size();
        out.gsrc.resize(g0 + (size_t)GR * kUnitThreads, kPairNoGroup);
        for (int i = 0; i < count; ++i) out.gsrc[g0 + slot[(size_t)i]] = keys[(size_t)i] * 12u;
        out.list[cls].push_back((uint32_t)out.desc.size());
        const size_t ws = (size_t)write_sectors(x0, y0, w, h);
        order.push_back({((uint64_t)(y0 / tune.root_h) << 48) | ((uint64_t)((x0 - row_start(y0)) / root_w) << 32) | (uint64_t)order.size(),
                         (uint32_t)out.desc.size() | ((uint32_t)cls << 28), (uint32_t)(y0 / tune.root_h),
                         (uint64_t)tune.line_cost * (uint64_t)st.lines + (uint64_t)ws});   // ~ the block's running time per frame
        out.desc.push_back(d);
        out.lines += (size_t)st.lines;
        out.sectors += ws;
        out.cls_lines[cls] += (size_t)st.lines; out.cls_sectors[cls] += ws; out.cls_pixels[cls] += (size_t)st.pixels; out.cls_groups[cls] += (size_t)count;
        return true;
    };

    // ---- k-d partition -----------------------------------------------------------------------------------------------------------
    struct Cell { int x0, y0, w, h; };
    std::vector<Cell> stack;
    for (pass = 1; pass <= 2; ++pass) {
    for (int y0 = 0; y0 < bh; y0 += tune.root_h)
        for (int x0 = row_start(y0); x0 < pitch; x0 += root_w) stack.push_back({x0, y0, std::min(root_w, pitch - x0), std::min(tune.root_h, bh - y0)});
    std::reverse(stack.begin(), stack.end());   // pop in row-major order: neighbouring units are neighbours in the class lists
    while (!stack.empty()) {
        Cell c = stack.back();
        stack.pop_back();
        shrink(c.x0, c.y0, c.w, c.h);
        const Stats st = cell_stats(c.x0, c.y0, c.w, c.h);
        if (st.quads == 0) continue;              // nothing a unit owns in here
        bool fits = false;   // some class holds the rectangle (emit decides finally: line-aligned slots may need a few more)
        for (int k = 0; k < kUnitClasses; ++k)
            fits = fits || (kUnitClassCON[k] == pass && !(k == 4 && !tune.wide_double) && !(k == kUnitClassBig && !tune.big_class) && st.groups <= std::min(tune.max_groups, kUnitClassGR[k] * kUnitThreads - (wide ? 1 : 0)) && slots_needed(c.w, c.h) <= kUnitClassNQ[k] * kUnitWaves);
        // a rectangle that is mostly other classes' quads (a seam crossing it diagonally) idles most of its lanes: cut it further
        if (fits && (long)st.quads * 8 < (long)(c.w / 4) * c.h * 3 && (long)c.w * c.h > 1024) fits = false;
        if (fits && emit(c.x0, c.y0, c.w, c.h, st)) continue;
        // split: rows at a multiple of the rows per wave-slot, columns at a multiple of 16 pixels; the cheaper cut wins
        long best = -1;
        Cell a = c, b = c;
        const int rps = 64 >> lanes_log2(c.w);
        if (c.h > 1) {
            int h2 = (c.h + 1) / 2;
            h2 = std::min(c.h - 1, (h2 + rps - 1) / rps * rps);
            const Stats s0 = cell_stats(c.x0, c.y0, c.w, h2), s1 = cell_stats(c.x0, c.y0 + h2, c.w, c.h - h2);
            best = (long)tune.line_cost * (s0.lines + s1.lines);
            a = {c.x0, c.y0, c.w, h2};
            b = {c.x0, c.y0 + h2, c.w, c.h - h2};
        }
        if (c.w > min_w) {
            const int w2 = std::min(c.w - 4, (c.w / 2 + col_step - 1) / col_step * col_step);
            const Stats s0 = cell_stats(c.x0, c.y0, w2, c.h), s1 = cell_stats(c.x0 + w2, c.y0, c.w - w2, c.h);
            const long cost = (long)tune.line_cost * (s0.lines + s1.lines) +
                              (long)tune.sector_cost * (write_sectors(c.x0, c.y0, w2, c.h) + write_sectors(c.x0 + w2, c.y0, c.w - w2, c.h) -
                                                        write_sectors(c.x0, c.y0, c.w, c.h)) +
                              (long)tune.run_cost * c.h;      // a column cut doubles the row runs the two units write
            if (best < 0 || cost < best) {
                best = cost;
                a = {c.x0, c.y0, w2, c.h};
                b = {c.x0 + w2, c.y0, c.w - w2, c.h};
            }
        }
        if (best < 0) {
            // one row of the narrowest width that still does not fit (cannot happen: 4 quads x 4 pixels x 2 groups): the partition is
            // unusable -> compile nothing (callers fall back to round 2's classes)
            out = UnitPlanHost();
            return;
        }
        stack.push_back(b);
        stack.push_back(a);
    }
    }
    // Rows of root cells share no sector with each other (only x-neighbours do), so their order is free: tune.row_order
    //   0 top to bottom; 1 the rows with the longest-running unit first; 2 the rows with the most work first; 3 bottom to top
    if (tune.row_order >= 1 && tune.row_order <= 3 && !order.empty()) {
        uint32_t nrows = 0;
        for (const Order &o : order) nrows = std::max(nrows, o.row + 1);
        std::vector<uint64_t> rmax(nrows, 0), rsum(nrows, 0);
        for (const Order &o : order) { rmax[o.row] = std::max(rmax[o.row], o.cost); rsum[o.row] += o.cost; }
        std::vector<uint32_t> rows(nrows), rank(nrows);
        for (uint32_t r = 0; r < nrows; ++r) rows[r] = r;
        if (tune.row_order == 1) std::stable_sort(rows.begin(), rows.end(), [&](uint32_t a, uint32_t b) { return rmax[a] > rmax[b]; });
        else if (tune.row_order == 2) std::stable_sort(rows.begin(), rows.end(), [&](uint32_t a, uint32_t b) { return rsum[a] > rsum[b]; });
        else std::reverse(rows.begin(), rows.end());
        for (uint32_t i = 0; i < nrows; ++i) rank[rows[i]] = i;
        for (Order &o : order) o.key = (o.key & 0x0000ffffffffffffull) | ((uint64_t)rank[o.row] << 48);
    }
    std::sort(order.begin(), order.end(), [](const Order &l, const Order &r) { return l.key < r.key; });
    // tune.row_order 4: longest unit first whatever its place (pure LPT); 5 / 6: the spatial order twice -- first the units above the
    // median / upper-quartile cost, then the rest -- so that the blocks that start last are short ones (experiments: profiles/r03/sweeps.log)
    if (tune.row_order == 4) std::stable_sort(order.begin(), order.end(), [](const Order &l, const Order &r) { return l.cost > r.cost; });
    if ((tune.row_order == 5 || tune.row_order == 6) && !order.empty()) {
        std::vector<uint64_t> cs;
        for (const Order &o : order) cs.push_back(o.cost);
        std::sort(cs.begin(), cs.end());
        const uint64_t cut = cs[tune.row_order == 5 ? cs.size() / 2 : cs.size() / 4];
        std::stable_partition(order.begin(), order.end(), [cut](const Order &o) { return o.cost >= cut; });
    }
    for (const Order &o : order) out.all.push_back(o.entry);
    for (size_t t = 0; t < own.size(); ++t)
        if (own[t] != 0) { hdr[t] |= kHdrBlock; ++out.claimed_tiles; }
}

// The COMPACT scratch of the balance schedule (bevw_plan.h: k_lum_groups): the luminance-shifted copy of a frame set holds ONLY the
// sampled 4-texel groups, 12 bytes each, in ascending order of their offsets in the frame set (Plan::groups) -- slot i of the scratch is group
// groups[i].  A unit's group list for that layout is the rank of every group times 12.  The 16-byte load of slot i reaches 4 bytes into
// slot i + 1; they matter only to a pixel that samples texel pair 3 of the group (texels 4g + 3, 4g + 4), and then texel 4g + 4 lies in the
// pixel's footprint, i.e. group g + 1 (same row: sx + 1 < fw) is sampled as well and IS slot i + 1.
// Returns false when a unit loads a group the list does not hold (the compact layout is then not used).
static inline bool unit_gsrc_compact(const std::vector<uint32_t> &gsrc, const std::vector<uint32_t> &groups, std::vector<uint32_t> &out)
{
    out.assign(gsrc.size(), kPairNoGroup);
    for (size_t i = 0; i < gsrc.size(); ++i) {
        if (gsrc[i] == kPairNoGroup) continue;
        const auto it = std::lower_bound(groups.begin(), groups.end(), gsrc[i]);
        if (it == groups.end() || *it != gsrc[i]) return false;
        out[i] = (uint32_t)(it - groups.begin()) * 12u;
    }
    return true;
}
// bytes between the compact scratch copies of consecutive frame sets: the groups + the 4 bytes the last slot's load reaches beyond them,
// rounded to whole 64-byte sectors
static inline size_t unit_compact_stride(size_t ngroups) { return (ngroups * 12 + 16 + 63) / 64 * 64; }

static inline std::vector<uint32_t> unit_host_headers(const std::vector<int16_t> lut1[4], const std::vector<uint8_t> mask[4], int ncams, int fw, int fh,
                                                      int bw, int bh, int tiles_x, int tiles_y)
{
    std::vector<uint32_t> hdr((size_t)tiles_x * tiles_y, 0u);
    std::vector<uint8_t> any(hdr.size(), 0);
    const uint32_t frame_bytes = (uint32_t)fw * fh * 3;
    for (int y = 0; y < bh; ++y)
        for (int x = 0; x < bw; ++x) {
            const size_t o = (size_t)y * bw + x, t = (size_t)(y / 8) * tiles_x + x / 32;
            int count = 0;
            for (int c = 0; c < ncams; ++c) {
                if (mask[c][o] == 0) continue;
                const int sx = lut1[c][o * 2], sy = lut1[c][o * 2 + 1];
                if (sx >= fw || sx + 1 < 0 || sy >= fh || sy + 1 < 0) continue;
                const bool interior = (unsigned)sx < (unsigned)(fw - 1) && (unsigned)sy < (unsigned)(fh - 1);
                const uint32_t toff = ((uint32_t)sy * fw + sx) * 3;
                if (!(interior && (toff & ~3u) + (uint32_t)fw * 3 + 12 <= frame_bytes)) hdr[t] |= kHdrSlow;
                ++count;
            }
            if (count > 1) hdr[t] |= kHdrSecond;
            if (count > 0) any[t] = 1;
        }
    for (size_t t = 0; t < hdr.size(); ++t)
        if (!any[t]) hdr[t] |= kHdrEmpty;
    return hdr;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// CPU emulation of one unit and one frame: the per-lane steps of plan_unit_body in program order (group loads -> pair conversion
// -> patch -> pixel interpolation -> 12-byte stores), with the same helpers.  Test infrastructure (tests/native/unit_emulate.cpp).
// ---------------------------------------------------------------------------------------------------------------------------------
static inline void unit_emulate(const UnitPlanHost &up, uint32_t unit, int cls, const uint8_t *frame_set, size_t set_bytes, bool blend,
                                const uint8_t *car, int pitch, uint8_t *out_img, uint32_t sums[3] = nullptr, std::vector<uint8_t> *written = nullptr)
    // written: per PIXEL, the number of its BYTES stored (3 = every byte exactly once)
{
    const UnitDesc &d = up.desc[unit];
    const int NQ = kUnitClassNQ[cls], GR = kUnitClassGR[cls], NCON = kUnitClassCON[cls], fpart = NCON == 2 ? 3 : 1, parts = fpart + (up.wide ? NCON : 0);
    const bool store16 = BEVW_UNIT_STORE16 == 2 || (BEVW_UNIT_STORE16 == 1 && ((uint32_t)pitch * 3u) % 64u == 0u);   // as plan_unit_run
    std::vector<uint8_t> patch((size_t)kUnitMaxGroups * 32, 0xcd);
    if (d.groups != 0)
        for (int r = 0; r < GR; ++r)
            for (int tid = 0; tid < kUnitThreads; ++tid) {
                const uint32_t off = up.gsrc[((size_t)d.gs_off + r) * kUnitThreads + tid];
                uint32_t w[4] = {0, 0, 0, 0};   // a masked lane's buffer load returns zeros
                if (off != kPairNoGroup && (size_t)off + 16 <= set_bytes) memcpy(w, frame_set + off, 16);
                uint4 A, B;
                pair_convert(w[0], w[1], w[2], w[3], A, B);
                uint8_t *sp = patch.data() + (size_t)(r * kUnitThreads + tid) * 32;
                memcpy(sp, &A, 16);
                memcpy(sp + 16, &B, 16);
            }
    const int ux = (int)(int16_t)(d.pos & 0xffffu), uy = (int)(d.pos >> 16), uw = (int)(d.shape & 0xffffu), uh = (int)(d.shape >> 16);
    for (int wave = 0; wave < kUnitWaves; ++wave)
        for (int j = 0; j < NQ; ++j) {
            uint32_t od[64][3];     // the wave's packed quads of this slot, the lanes' store masks and offsets: the store format needs the neighbours
            bool st[64];
            size_t so[64];
            for (int lane = 0; lane < 64; ++lane) {
                const int sidx = unit_slot(wave, j);
                int qx, row;
                unit_quad(d.lq, sidx, lane, qx, row);
                const uint32_t *e = up.entries.data() + (((size_t)d.ent_off + (size_t)sidx * parts) * 64 + lane) * 4;   // part k at e + k * 256
                uint32_t P[4] = {0, 0, 0, 0};
                if (d.groups != 0)
                    for (int p = 0; p < 4; ++p) {
                        int px[3] = {0, 0, 0};
                        for (int con = 0; con < NCON; ++con) {
                            uint32_t i0, i1, wxa, wy, acc[3];
                            uint2 q0, q1;
                            if (up.wide) {
                                float fx, fy;
                                unit_decode_wide(e[con * 256 + p], e[(fpart + con) * 256 + p], i0, i1, fx, fy);
                                memcpy(&q0, patch.data() + (size_t)i0 * 8, 8);
                                memcpy(&q1, patch.data() + (size_t)i1 * 8, 8);
                                const uint32_t P3 = bilinear_pairs_f32(q0, q1, fx, fy);
                                for (int k = 0; k < 3; ++k) acc[k] = ((P3 >> (8 * k)) & 255u) << 16;
                            } else {
                                unit_decode(e[con * 256 + p], i0, i1, wxa, wy);
                                memcpy(&q0, patch.data() + (size_t)i0 * 8, 8);
                                memcpy(&q1, patch.data() + (size_t)i1 * 8, 8);
                                bilinear_pairs(q0, q1, wxa, wxa << 16, wy, acc);
                            }
                            const uint32_t wm = NCON == 2 ? blend_weight_q23((e[2 * 256 + p] >> (8 * con)) & 255u) : 0u;
                            for (int k = 0; k < 3; ++k) {
                                const int v = (int)((acc[k] >> 16) & 255u), c = (blend && NCON == 2) ? (int)blend_apply_q23((uint32_t)v, wm) : v;
                                px[k] = con == 0 ? c : (px[k] + c < 255 ? px[k] + c : 255);
                            }
                        }
                        P[p] = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
                        if (sums)
                            for (int k = 0; k < 3; ++k) sums[k] += (uint32_t)px[k];
                    }
                const int x = ux + 4 * qx + unit_skew(up.skew, uy + row);
                st[lane] = !(4 * qx >= uw || row >= uh || x < 0 || x >= pitch || (unit_no_contributor(e[0]) && (e[0] & kUnitSkip)));   // else: the lane's store is masked
                so[lane] = ((size_t)(uy + row) * pitch + x) * 3;
                if (st[lane] && car) {
                    uint32_t c[3];
                    memcpy(c, car + so[lane], 12);
                    add_car(P, c[0], c[1], c[2]);
                }
                pack_pixels(P, od[lane][0], od[lane][1], od[lane][2]);
            }
            // the wave-store (unit_store_quad16 / unit_store_quad): whole lane quads write 3 x 16 bytes, the other lanes 12 bytes
            for (int lane = 0; lane < 64; ++lane) {
                const int q = lane & 3, l0 = lane & ~3;
                const bool whole = store16 && st[l0] && st[l0 + 1] && st[l0 + 2] && st[l0 + 3];
                if (whole) {
                    if (q == 3) continue;
                    uint32_t o[4];
                    unit_repack16(q, od[lane], od[lane + 1], o);   // (lane + 1 <= l0 + 3: the next lane of the quad)
                    memcpy(out_img + so[lane] + 4 * q, o, 16);
                    if (written) for (int k = 0; k < 16; ++k) ++(*written)[(so[lane] + 4 * q + k) / 3];
                } else if (st[lane]) {
                    memcpy(out_img + so[lane], od[lane], 12);
                    if (written) for (int k = 0; k < 12; ++k) ++(*written)[(so[lane] + k) / 3];
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------------------------------------------
// one block of 4 waves: unit from the class list, frames of the chunk.  lds: 32 KB.
// GR <= 2: the frame's patch (GR x 8 KB) is double-buffered -- frame b+1 is converted into the other half while frame b is
// interpolated, one s_barrier per frame.  GR == 4: one patch; conversion and interpolation are separated by two barriers per frame.
// The groups of the next TWO frames are in flight in registers in both cases.  Register budget: 16 registers per quad slot
// (LDS indices and weights of 4 pixels) + 8 per round of groups in flight; nothing else lives across the frame loop -- the car
// sprite is re-read per frame by the few units that lie under it.
// NCON == 2: two plan entries and two blend weights per pixel (seams, blend overlaps): second contribution added with saturation
// (cv2.add, surroundBEV.py:318-320), weights applied as trunc(f32(v) * f32(m / 255.0)) when BLEND (surroundBEV.py:279-280) -- in its exact
// integer form (v * (m * 32897)) >> 23 (blend_apply_q23, bevw_device.h: one 24-bit multiply and a shift instead of cvt / mul / cvt).
// WIDE: the plan carries 21-bit fractions and the pixels are interpolated in fp32 (the analytic projection mode); wxa / wy then hold the
// bit patterns of fx / fy.
#ifndef BEVW_UNIT_DEPTH
#define BEVW_UNIT_DEPTH 2
#endif
#ifndef BEVW_UNIT_NO_BIG
#define BEVW_UNIT_NO_BIG 0
#endif
// Timing experiments that produce wrong pixels (the memory-only replay of the request stream) live in bevw_unit_experiments.h and are compiled
// only into tagged variant builds (-DBEVW_EXPERIMENT=<n>, cameracalibration_amd/build.py refuses flags without a tag).
#ifdef BEVW_EXPERIMENT
#include "bevw_unit_experiments.h"
#endif
// sum of v over the 64 lanes of the wave, uniform result: four DPP adds inside the rows of 16 lanes (quad swaps, half-row and row mirrors),
// two row broadcasts (gfx9 wave64: lane 15 of every row into the next row, lane 31 into rows 2 and 3), then lane 63 holds the total
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);     // quad_perm [1, 0, 3, 2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);     // quad_perm [2, 3, 0, 1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);    // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);    // row_mirror: every lane of a row has the row's sum
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// A quad's 12 bytes to the output image.  The output is written once and never read: as a STREAMING store (nt | sc0) it leaves the L2 to the
// texel groups and the plan (config 3 0.439 -> 0.410 ms, undistort 0.10 -> 0.079; profiles/r04/ab_store_policy.log) -- when its rows are
// whole 64-byte sectors.  In the dense layout (rows of 3240 bytes) neighbouring quads of two rows share sectors, streaming stores send
// them to memory in pieces (0.46 -> 0.51 ms): those keep the default policy.  `streaming` is uniform over the launch.
// a plan entry of the unit (BEVW_PLAN_NT, bevw_device.h: read once per block and chunk of frames)
__device__ __forceinline__ uint4 unit_plan_load(const uint4 *p)
{
    if (!BEVW_PLAN_NT) return *p;
    const pair_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const pair_u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void unit_store_quad(pair_u32x3 v, __amdgpu_buffer_rsrc_t ro, int off, bool streaming)
{
    if (streaming) __builtin_amdgcn_raw_buffer_store_b96(v, ro, off, 0, kPairStreamAux);
    else __builtin_amdgcn_raw_buffer_store_b96(v, ro, off, 0, kPairStoreAux);
}
// Round 6: the FORMAT of the wave-store.  64 lanes x 12 bytes put a lane across every 64-byte sector boundary of the row run (64 / 12 = 5.33 lanes
// per sector); the 4 x 12 bytes of a LANE QUAD (16 pixels of one row: lanes 4k .. 4k+3 always lie in one row, lq >= 2) repacked into 3 x 16 bytes
// and stored as buffer_store_dwordx4 from 3 of the 4 lanes cover the same 48 bytes with every 16-byte piece inside one sector.  The repack is
// the funnel S[q .. q + 3] of S = {d0, d1, d2, n0, n1, n2}, n = the next lane's dwords (3 v_mov_b32 quad_perm [1, 2, 3, 3], 8 v_cndmask).
// Lane quads in which some lane does not store (a skipped base tile never cuts a lane quad, but a unit's last columns or the image's right edge
// may) keep the 12-byte store: `off12` is in range only for their lanes, and the second store is skipped wave-uniformly (`part_any`).
// tools/store_pattern.hip "format" is the microbenchmark of exactly this pair; profiles/r06/README.md has the A/B.
__device__ __forceinline__ void unit_store_quad16(uint32_t d0, uint32_t d1, uint32_t d2, __amdgpu_buffer_rsrc_t ro, int off16, int q, bool streaming)
{
    const uint32_t d[3] = {d0, d1, d2};
    const uint32_t n[3] = {(uint32_t)__builtin_amdgcn_update_dpp(0, (int)d0, 0xF9, 0xf, 0xf, false),    // quad_perm [1, 2, 3, 3]
                           (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d1, 0xF9, 0xf, 0xf, false),
                           (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d2, 0xF9, 0xf, 0xf, false)};
    uint32_t o[4];
    unit_repack16(q, d, n, o);
    const pair_u32x4 v = {o[0], o[1], o[2], o[3]};
    if (streaming) __builtin_amdgcn_raw_buffer_store_b128(v, ro, off16, 0, kPairStreamAux);
    else __builtin_amdgcn_raw_buffer_store_b128(v, ro, off16, 0, kPairStoreAux);
}
template <bool BLEND, bool SUMS, int NQ, int GR, int NCON, bool WIDE = false>
__device__ __forceinline__ void plan_unit_run(const PlanArgs &a, uint32_t chunk, uint32_t unit, uint8_t *lds, uint4 *wave_sums = nullptr)
{
    static_assert(NQ >= 1 && NQ <= kUnitMaxNQ && GR >= 1 && GR <= kUnitMaxGR && (NCON == 1 || NCON == 2), "unit class");
    static_assert(!(WIDE && SUMS), "wide plans carry no channel sums");
    constexpr int kFPart = NCON == 2 ? 3 : 1;          // narrow part: the entries of each contributor (+ the blend weights)
    constexpr int kParts = kFPart + (WIDE ? NCON : 0); // uint4 per lane and quad slot in the plan
    constexpr bool kWeights = BLEND && NCON == 2;
    const uint32_t *dp = reinterpret_cast<const uint32_t *>(a.un_desc + unit);
    const uint32_t pos = __builtin_amdgcn_readfirstlane(dp[0]), shape = __builtin_amdgcn_readfirstlane(dp[1]);
    const uint32_t ent_off = __builtin_amdgcn_readfirstlane(dp[2]), gs_off = __builtin_amdgcn_readfirstlane(dp[3]);
    const uint32_t lq = __builtin_amdgcn_readfirstlane(dp[4]);
    const uint32_t ngroups = __builtin_amdgcn_readfirstlane(dp[6]);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ux = (int)(int16_t)(pos & 0xffffu), uy = (int)(pos >> 16), uw = (int)(shape & 0xffffu), uh = (int)(shape >> 16);
    // bytes between consecutive frame sets as this kernel reads them: whole frames, or the compact scratch of the balance schedule (unit_gsrc_compact)
    const size_t set_bytes = a.set_stride ? (size_t)a.set_stride : (size_t)a.fw * a.fh * 3 * a.ncams, img_bytes = (size_t)a.pitch * a.bh * 3;
    const bool streaming = ((uint32_t)a.pitch * 3u) % 64u == 0u;   // rows of whole sectors (see unit_store_quad)
    constexpr int kPatch = GR * kUnitThreads * 32;              // one frame's pair entries
    constexpr bool DB = 2 * kPatch <= kUnitMaxGroups * 32;      // both halves fit the block's 32 KB

    uint32_t i0[NQ][NCON][4], i1[NQ][NCON][4], wxa[NQ][NCON][4], wy[NQ][NCON][4], gs[GR], ooff_masked[NQ];
    // 16-byte store format (unit_store_quad16): the lane's offset of its 16-byte piece (lanes q < 3 of whole lane quads; out of range otherwise);
    // bit j of part_bits = slot j of this lane stores 12 bytes after all (its lane quad is not whole); part_any: some lane of the wave does
    const bool store16 = BEVW_UNIT_STORE16 == 2 || (BEVW_UNIT_STORE16 == 1 && streaming);
    uint32_t ooff16[NQ], part_bits = 0;
    bool part_any[NQ];
    uint32_t wq[NQ][NCON][4];    // blend weights as 24-bit integer factors (blend_weight_q23)
    const bool with_car = !SUMS && a.car != nullptr;
    const __amdgpu_buffer_rsrc_t rcar = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(with_car ? a.car : a.out), 0,
                                                                          with_car ? (uint32_t)img_bytes : 0u, kBufferWord3);
    uint32_t car_or = 0;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int sidx = unit_slot(wave, j);
        int qx, row;
        unit_quad(lq, sidx, lane, qx, row);
        const int x = ux + 4 * qx + unit_skew((uint32_t)a.un_skew, uy + row);
        const uint32_t ooff = ((uint32_t)(uy + row) * a.pitch + (uint32_t)x) * 3;
        uint32_t e0 = 0;
#pragma unroll
        for (int con = 0; con < NCON; ++con) {
            const uint4 e4 = unit_plan_load(a.un_entries + ((size_t)ent_off + sidx * kParts + con) * 64 + lane);   // the lane's 4 pixels of this slot
            const uint32_t e[4] = {e4.x, e4.y, e4.z, e4.w};
            if (con == 0) e0 = e[0];
            if (WIDE) {
                const uint4 f4 = unit_plan_load(a.un_entries + ((size_t)ent_off + sidx * kParts + kFPart + con) * 64 + lane);
                const uint32_t f[4] = {f4.x, f4.y, f4.z, f4.w};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float fx, fy;
                    unit_decode_wide(e[p], f[p], i0[j][con][p], i1[j][con][p], fx, fy);
                    wxa[j][con][p] = __float_as_uint(fx); wy[j][con][p] = __float_as_uint(fy);
                }
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) unit_decode(e[p], i0[j][con][p], i1[j][con][p], wxa[j][con][p], wy[j][con][p]);
            }
        }
        if (kWeights) {
            const uint4 w4 = unit_plan_load(a.un_entries + ((size_t)ent_off + sidx * kParts + 2) * 64 + lane);
            const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int p = 0; p < 4; ++p) { wq[j][0][p] = blend_weight_q23(w[p] & 255u); wq[j][NCON - 1][p] = blend_weight_q23((w[p] >> 8) & 255u); }
        }
        const bool store = 4 * qx < uw && row < uh && x >= 0 && x < a.pitch && !(unit_no_contributor(e0) && (e0 & kUnitSkip));
        ooff_masked[j] = store ? ooff : kPairNoGroup;   // out of range of the image's buffer descriptor: neither read (car) nor written
        {
            const bool whole = ((__builtin_amdgcn_ballot_w64(store) >> (lane & ~3)) & 0xfull) == 0xfull;
            ooff16[j] = (store16 && whole && (lane & 3) < 3) ? ooff + 4u * (uint32_t)(lane & 3) : kPairNoGroup;
            const bool part = store && !(store16 && whole);
            part_bits |= (part ? 1u : 0u) << j;
            part_any[j] = __builtin_amdgcn_ballot_w64(part) != 0;
        }
        const pair_u32x3 c = __builtin_amdgcn_raw_buffer_load_b96(rcar, (int)ooff_masked[j], 0, 0);
        car_or |= c.x | c.y | c.z;
    }
    const bool car_any = __builtin_amdgcn_ballot_w64(car_or != 0) != 0;
    // the 12 output bytes of quad slot j to image `ro` (unit_store_quad16 / unit_store_quad)
    auto store_slot = [&](int j, uint32_t d0, uint32_t d1, uint32_t d2, const __amdgpu_buffer_rsrc_t ro) {
        if (BEVW_UNIT_STORE16 == 0) {
            unit_store_quad(pair_u32x3{d0, d1, d2}, ro, (int)ooff_masked[j], streaming);
            return;
        }
        if (store16) unit_store_quad16(d0, d1, d2, ro, (int)ooff16[j], lane & 3, streaming);
        if (part_any[j]) unit_store_quad(pair_u32x3{d0, d1, d2}, ro, ((part_bits >> j) & 1u) ? (int)ooff_masked[j] : (int)kPairNoGroup, streaming);
    };
    const int b_begin = (int)chunk * a.nb, b_end = min(a.batch, b_begin + a.nb);

    if (ngroups == 0) {
        // no contributor anywhere (under the car): every frame of the chunk gets the sprite or zeros, in whole row runs
        // (balance: zeros -- the sprite is added after the gains, as in the reference)
#pragma unroll 1
        for (int b = b_begin; b < b_end; ++b) {
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b * img_bytes, 0, (uint32_t)img_bytes, kBufferWord3);
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const pair_u32x3 c = __builtin_amdgcn_raw_buffer_load_b96(rcar, (int)ooff_masked[j], 0, 0);   // zeros without a sprite
                store_slot(j, c.x, c.y, c.z, ro);
            }
            // its channel sums are zero, and it says so for every frame: no entry depends on what the buffer held before (another layout's sums)
            if (SUMS && wave == 0 && lane < 3) a.psums[((size_t)b * a.nsum + unit) * 3 + lane] = 0u;
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < GR; ++r) gs[r] = once_load<BEVW_PLAN_NT>(a.un_gsrc + ((size_t)gs_off + r) * kUnitThreads + threadIdx.x);

    // (Dealing the frames of a chunk strided over the batch, or rotating the class lists per XCD, changes nothing: profiles/r03/placement.md)
    auto frame_of = [&](int b) { return min(b, b_end - 1); };   // past the chunk: the last frame once more
    // frames whose groups are in flight in registers.  BEVW_UNIT_DEPTH (even) deepens it for the classes with <= 2 rounds of groups (8 registers per
    // round and frame); measured: profiles/r04/README.md
    constexpr int D = (GR <= 2) ? BEVW_UNIT_DEPTH : 2;
    static_assert(D >= 2 && D % 2 == 0, "the patch halves alternate with the ring");
    pair_u32x4 pf[D][GR];
    auto issue = [&](int b, int ring) {
        const uint8_t *src = a.frames + (size_t)frame_of(b) * set_bytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, (uint32_t)set_bytes, kBufferWord3);
#pragma unroll
        for (int r = 0; r < GR; ++r) pf[ring][r] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)gs[r], 0, kPairLoadAux);
    };
    auto land = [&](int ring) {   // the groups of ring slot `ring` -> the patch half of that frame: 32 bytes per group slot
#pragma unroll
        for (int r = 0; r < GR; ++r) {
            uint4 A, B;
            pair_convert(pf[ring][r].x, pf[ring][r].y, pf[ring][r].z, pf[ring][r].w, A, B);
            uint4 *sp = reinterpret_cast<uint4 *>(lds + (DB ? (ring & 1) * kPatch : 0)) + (r * kUnitThreads + (int)threadIdx.x) * 2;
            sp[0] = A;
            sp[1] = B;
        }
    };
    auto acc_to_px = [](const uint32_t acc[3]) {
        return __builtin_amdgcn_perm(acc[2], __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u), 0x0c060100u);
    };
#ifdef BEVW_EXPERIMENT_FRAME
    auto frame = [&](int b, int ring) { BEVW_EXPERIMENT_FRAME(b, ring) };
#else
    auto frame = [&](int b, int ring) {
        if (!DB) {
            land(ring);            // every wave finished reading the previous frame: barrier at the end of its step
            block_lds_barrier();
        }
        const uint2 *const pw = reinterpret_cast<const uint2 *>(lds + (DB ? (ring & 1) * kPatch : 0));
        issue(b + D, ring);        // the ring slot of frame b has been converted
        uint32_t d[NQ][3];
        uint32_t tb = 0, tg = 0, tr = 0;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (WIDE) {
                uint32_t P[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    uint32_t v = bilinear_pairs_f32(pw[i0[j][0][p]], pw[i1[j][0][p]], __uint_as_float(wxa[j][0][p]), __uint_as_float(wy[j][0][p]));
                    if (NCON == 2) {
                        const uint32_t v1 = bilinear_pairs_f32(pw[i0[j][NCON - 1][p]], pw[i1[j][NCON - 1][p]], __uint_as_float(wxa[j][NCON - 1][p]),
                                                               __uint_as_float(wy[j][NCON - 1][p]));
                        uint32_t px = 0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const uint32_t a0 = (v >> (8 * k)) & 255u, a1 = (v1 >> (8 * k)) & 255u;
                            const uint32_t c0 = kWeights ? blend_apply_q23(a0, wq[j][0][p]) : a0, c1 = kWeights ? blend_apply_q23(a1, wq[j][NCON - 1][p]) : a1;
                            px |= min(255u, c0 + c1) << (8 * k);
                        }
                        v = px;
                    }
                    P[p] = v;
                }
                if (car_any) {
                    const pair_u32x3 c = __builtin_amdgcn_raw_buffer_load_b96(rcar, (int)ooff_masked[j], 0, 0);
                    add_car(P, c.x, c.y, c.z);
                }
                pack_pixels(P, d[j][0], d[j][1], d[j][2]);
                continue;
            }
            uint32_t acc[4][3];
#pragma unroll
            for (int p = 0; p < 4; ++p) bilinear_pairs(pw[i0[j][0][p]], pw[i1[j][0][p]], wxa[j][0][p], wxa[j][0][p] << 16, wy[j][0][p], acc[p]);
            if (NCON == 1 && !car_any) {
                pack_accs(acc, d[j][0], d[j][1], d[j][2]);
            } else {
                uint32_t P[4];
                if (NCON == 1) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) P[p] = acc_to_px(acc[p]);
                } else {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        uint32_t acc1[3];
                        bilinear_pairs(pw[i0[j][NCON - 1][p]], pw[i1[j][NCON - 1][p]], wxa[j][NCON - 1][p], wxa[j][NCON - 1][p] << 16, wy[j][NCON - 1][p], acc1);
                        uint32_t px = 0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const uint32_t v0 = (acc[p][k] >> 16) & 255u, v1 = (acc1[k] >> 16) & 255u;
                            const uint32_t c0 = kWeights ? blend_apply_q23(v0, wq[j][0][p]) : v0, c1 = kWeights ? blend_apply_q23(v1, wq[j][NCON - 1][p]) : v1;
                            px |= min(255u, c0 + c1) << (8 * k);
                        }
                        P[p] = px;
                    }
                }
                if (car_any) {     // uniform over the wave; the sprite is not kept in registers across the frame loop
                    const pair_u32x3 c = __builtin_amdgcn_raw_buffer_load_b96(rcar, (int)ooff_masked[j], 0, 0);
                    add_car(P, c.x, c.y, c.z);
                }
                pack_pixels(P, d[j][0], d[j][1], d[j][2]);
            }
            if (SUMS) {
                // the lane's own channel sums over its quad slots, from the 12 packed bytes B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3 (no sprite in
                // this mode: the car is added behind the gains); the wave reduction follows the loop
                tb = __builtin_amdgcn_udot4(d[j][2], 0x00000100u, __builtin_amdgcn_udot4(d[j][1], 0x00010000u, __builtin_amdgcn_udot4(d[j][0], 0x01000001u, tb, false), false), false);
                tg = __builtin_amdgcn_udot4(d[j][2], 0x00010000u, __builtin_amdgcn_udot4(d[j][1], 0x01000001u, __builtin_amdgcn_udot4(d[j][0], 0x00000100u, tg, false), false), false);
                tr = __builtin_amdgcn_udot4(d[j][2], 0x01000001u, __builtin_amdgcn_udot4(d[j][1], 0x00000100u, __builtin_amdgcn_udot4(d[j][0], 0x00010000u, tr, false), false), false);
            }
        }
        if (SUMS) {
            // One wave reduction per frame, on the VALU (DPP adds): round 3 reduced every quad slot with __shfl_xor = 12 ds_bpermute_b32 per
            // slot, more LDS-pipe instructions than the slot's pixel reads.  A lane's NQ x 4 pixels sum to <= 4080 per channel, a wave to
            // <= 261120.  Skipped quads and lanes without a quad have zero entries: they add 0.  The four waves' sums meet in LDS behind the
            // frame's barrier (two slots, alternating with the frame), and ONE lane triple of the block stores the unit's entry: no atomics
            const uint32_t wb = wave_sum_dpp(tb), wg = wave_sum_dpp(tg), wr = wave_sum_dpp(tr);
            if (lane == 0) wave_sums[(ring & 1) * kUnitWaves + wave] = make_uint4(wb, wg, wr, 0u);
        }
        if (DB) land((ring + 1) % D);    // frame b+1 into the other half: nobody reads it before the barrier
        {
            uint8_t *img = a.out + (size_t)frame_of(b) * img_bytes;   // past the chunk: re-writes the last frame with the same bytes
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(img, 0, (uint32_t)img_bytes, kBufferWord3);
#pragma unroll
            for (int j = 0; j < NQ; ++j) store_slot(j, d[j][0], d[j][1], d[j][2], ro);
        }
        block_lds_barrier();       // DB: half[ring ^ 1] complete for everybody, half[ring] free for frame b+2; else: the patch is free
        if (SUMS && wave == 0 && lane < 3 && b < b_end) {
            // (slot ring & 1 is written again in frame b + 2, behind the barrier of frame b + 1, which this wave passes after these reads)
            const uint32_t *ws = reinterpret_cast<const uint32_t *>(wave_sums + (ring & 1) * kUnitWaves) + lane;
            a.psums[((size_t)b * a.nsum + unit) * 3 + lane] = ws[0] + ws[4] + ws[8] + ws[12];
        }
    };
#endif
#pragma unroll
    for (int u = 0; u < D; ++u) issue(b_begin + u, u);
#ifndef BEVW_EXPERIMENT_FRAME
    if (DB) {
        land(0);
        block_lds_barrier();
    }
#endif
#pragma unroll 1
    for (int b = b_begin; b < b_end; b += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) frame(b + u, u);
    }
}

// block -> (chunk, unit) of the list of ALL units in the partition's own (spatial) order, class in bits 28..31: neighbouring units run
// at the same time on the same XCD, whatever their class, so the two halves of a sector that two units share meet in the L2
template <bool BLEND, bool SUMS>
__device__ __forceinline__ void plan_unit_any(const PlanArgs &a, uint32_t block_id, uint8_t *lds, uint4 *wave_sums)
{
    uint32_t chunk, group;
    if (!plan_block_map(a, block_id, chunk, group)) return;
    if ((int)group >= a.nlist) return;
    const uint32_t e = __builtin_amdgcn_readfirstlane(a.tile_list[group]), unit = e & 0x0fffffffu;
#define BEVW_UNIT_CASE(C) case C: plan_unit_run<BLEND, SUMS, kUnitClassNQ[C], kUnitClassGR[C], kUnitClassCON[C]>(a, chunk, unit, lds, wave_sums); break;
    switch (e >> 28) {
        BEVW_UNIT_CASE(0) BEVW_UNIT_CASE(1) BEVW_UNIT_CASE(2) BEVW_UNIT_CASE(3)
        // class 4 (two quads per lane, two contributors): with float blend weights (rounds 3 - 5) its blend variant needed 177 .. 197 VGPRs and
        // blend handles went without it; with the integer weights of round 6 (blend_apply_q23) every variant stays below 168
        BEVW_UNIT_CASE(4)
        BEVW_UNIT_CASE(5)
#if !BEVW_UNIT_NO_BIG   // (experiment builds without the (4, 4) class: every other class fits 128 VGPRs = 4 waves per SIMD; plans then need BEVW_UNIT_BIG=0)
        BEVW_UNIT_CASE(7)
#endif
        default: plan_unit_run<BLEND, SUMS, kUnitClassNQ[6], kUnitClassGR[6], kUnitClassCON[6]>(a, chunk, unit, lds, wave_sums); break;
    }
#undef BEVW_UNIT_CASE
}

// THE per-frame kernel of the tile plan: every unit of every class, one launch per step.  3 blocks of 32 KB per CU.
#ifndef BEVW_PLAN_ALL_WAVES
#define BEVW_PLAN_ALL_WAVES 3   // waves per SIMD the kernel is compiled for (<= 168 VGPRs; 4 measured no faster: profiles/r03/sweeps.log)
#endif
template <bool BLEND, bool SUMS>
__global__ void __launch_bounds__(kUnitThreads) __attribute__((amdgpu_waves_per_eu(BEVW_PLAN_ALL_WAVES, BEVW_PLAN_ALL_WAVES))) k_plan_units(PlanArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t patch[kUnitMaxGroups * 32];
    __shared__ uint4 wave_sums[SUMS ? 2 * kUnitWaves : 1];   // balance: the waves' channel sums of a frame (plan_unit_run)
#ifdef BEVW_EXPERIMENT_TRACE_BEGIN
    BEVW_EXPERIMENT_TRACE_BEGIN();
#endif
    plan_unit_any<BLEND, SUMS>(a, blockIdx.x, patch, wave_sums);
#ifdef BEVW_EXPERIMENT_TRACE_END
    BEVW_EXPERIMENT_TRACE_END(a)
#endif
}

// wide plans (analytic projection): every unit class in one launch, partition order, as plan_unit_any
template <bool BLEND>
__global__ void __launch_bounds__(kUnitThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) k_plan_unit_wide(PlanArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t patch[kUnitMaxGroups * 32];
    uint32_t chunk, group;
    if (!plan_block_map(a, blockIdx.x, chunk, group)) return;
    if ((int)group >= a.nlist) return;
    const uint32_t e = __builtin_amdgcn_readfirstlane(a.tile_list[group]), unit = e & 0x0fffffffu;
#define BEVW_UNIT_CASE(C) case C: plan_unit_run<BLEND, false, kUnitClassNQ[C], kUnitClassGR[C], kUnitClassCON[C], true>(a, chunk, unit, patch); break;
    switch (e >> 28) {
        BEVW_UNIT_CASE(0) BEVW_UNIT_CASE(1) BEVW_UNIT_CASE(2) BEVW_UNIT_CASE(3)
        BEVW_UNIT_CASE(4) BEVW_UNIT_CASE(5) BEVW_UNIT_CASE(7)
        default: plan_unit_run<BLEND, false, kUnitClassNQ[6], kUnitClassGR[6], kUnitClassCON[6], true>(a, chunk, unit, patch); break;
    }
#undef BEVW_UNIT_CASE
}

}  // namespace bevw
