// Host-only run of the JPEG kernels' per-lane code (cameracalibration_amd/csrc/bevw_jpeg.h) -- needs no GPU.
//
// Every lane function of the JPEG kernels is __host__ __device__; this program drives them in plain loops in the order the kernels
// do (one loop iteration = one lane), including the synchronisation fixed point of the parallel Huffman decoder, and writes what
// the device would write.  The walkers the kernels actually run (bevw_jpeg_walk.h: the straight-line lane walker of the synchronisation
// passes, the scalar walker of the serial tail, the storing walker of the final pass) run NEXT to decode_sub, the plain statement of the
// algorithm, on every subsequence and every entry state the fixed point goes through: any difference in an exit state, a block count, a DC
// sum or a coefficient ends the program with an error.  tests/test_jpeg_emulate.py compares the results with the oracle (oracle/jpegoracle.c) and with Pillow's
// libjpeg-turbo.  Compiled with hipcc (only the host part runs).
//
//   jpeg_emulate decode <in.jpg> <out.bin>       out: int32 w h rounds nsub | uint8 bgr[h][w][3]
//   jpeg_emulate encode <in.bin> <out.jpg>       in : int32 w h quality sampling | uint8 bgr[h][w][3]
//   jpeg_emulate unstuff <in.bin>                arbitrary bytes taken as entropy-coded data: the lanes' un-stuffing against the sequential one
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

#include "../../cameracalibration_amd/csrc/bevw_jpeg.h"
#include "../../cameracalibration_amd/csrc/bevw_jpeg_walk.h"

using namespace bevw::jpg;

static std::vector<uint8_t> read_file(const char *p)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(p, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize((size_t)n);
    if (n && fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}

// k_jpeg_find_end / k_jpeg_count_raw / k_jpeg_unstuff: lanes of 16 bytes over the file's entropy-coded bytes as they are, checked against the
// sequential statement of the same thing (unstuff_scan)
static bool lane_unstuff(const std::vector<uint8_t> &raw, size_t scan_off, std::vector<uint8_t> &stream, std::vector<uint32_t> &seg_byte)
{
    const uint32_t raw_bytes = (uint32_t)(raw.size() - scan_off);
    std::vector<uint8_t> slot((size_t)raw_bytes + 64, 0);
    memcpy(slot.data(), raw.data() + scan_off, raw_bytes);
    auto load16 = [&](uint32_t pos) {
        RawBytes R;
        memcpy(R.w, slot.data() + pos, 16);
        R.prev = pos ? slot[pos - 1] : 0u;
        R.next = pos + 16u < raw_bytes ? slot[pos + 16u] : 0xD9u;
        return R;
    };
    uint32_t term = raw_bytes;
    for (uint32_t pos = 0; pos < raw_bytes; pos += 16) term = std::min(term, raw_first_terminator(load16(pos), pos, raw_bytes));
    stream.assign(raw.size() + 64, 0);
    seg_byte.assign(1, 0);
    uint32_t kept = 0;
    for (uint32_t pos = 0; pos < term; pos += 16) {
        const RawBytes R = load16(pos);
        uint32_t keep, rst;
        raw_classify(R, pos, term, raw_bytes, keep, rst);
        for (int i = 0; i < 16; ++i) {
            if (rst & (1u << i)) seg_byte.push_back(kept);
            if (keep & (1u << i)) stream[kept++] = (uint8_t)R.at(i);
        }
    }
    seg_byte.push_back(kept);
    std::vector<uint8_t> ref(raw.size() + 64, 0);
    std::vector<uint32_t> ref_seg;
    const size_t nb = unstuff_scan(raw.data(), raw.size(), scan_off, ref.data(), ref_seg);
    if (nb != kept || ref_seg != seg_byte || memcmp(ref.data(), stream.data(), kept) != 0) {
        fprintf(stderr, "lane un-stuffing differs from the sequential one (%u vs %zu bytes, %zu vs %zu segments)\n", kept, nb, seg_byte.size() - 1, ref_seg.size() - 1);
        return false;
    }
    return true;
}

static int do_unstuff(const char *in)
{
    const std::vector<uint8_t> raw = read_file(in);
    std::vector<uint8_t> stream;
    std::vector<uint32_t> seg;
    if (!lane_unstuff(raw, 0, stream, seg)) return 3;
    printf("un-stuffed %zu bytes: %zu segments, %u bytes kept\n", raw.size(), seg.size() - 1, seg.back());
    return 0;
}

static bool same_walk(const SubOut &a, const SubOut &b) { return a.exit == b.exit && a.cnt == b.cnt && a.dc0 == b.dc0 && a.dc1 == b.dc1 && a.dc2 == b.dc2; }
static long g_walks = 0;
// decode_sub<false> and the two walkers the synchronisation kernels run, on the same subsequence from the same state
static bool walk3(const WordSource &src, const TableSet &T, const Geom &G, uint64_t in, uint32_t end, SubOut &R, int j)
{
    R = decode_sub<false>(src, T.t, G, in, end, nullptr, 0, 0, 0, 0, 0);
    const SubOut L = decode_sub_lanes(src, T.t, G, in, end), S = decode_sub_scalar(src, T.t, G, in, end);
    ++g_walks;
    if (same_walk(R, L) && same_walk(R, S)) return true;
    fprintf(stderr, "walkers differ on subsequence %d from state %llx: decode_sub exit %llx cnt %d dc %d %d %d | lanes %llx %d %d %d %d | scalar %llx %d %d %d %d\n", j,
            (unsigned long long)in, (unsigned long long)R.exit, R.cnt, R.dc0, R.dc1, R.dc2, (unsigned long long)L.exit, L.cnt, L.dc0, L.dc1, L.dc2,
            (unsigned long long)S.exit, S.cnt, S.dc0, S.dc1, S.dc2);
    return false;
}

static int do_decode(const char *in, const char *out)
{
    const std::vector<uint8_t> raw = read_file(in);
    Parsed P;
    std::string why;
    const int st = parse_header(raw.data(), raw.size(), P, why);
    if (st) { fprintf(stderr, "parse: %d %s\n", st, why.c_str()); return 2; }
    const Geom G = make_geom(P.w, P.h, P.nc, P.hs, P.vs);
    TableSet T;
    for (int c = 0; c < P.nc; ++c)
        if (!make_hufftab(P.dc[P.td[c]], T.t[2 * c]) || !make_hufftab(P.ac[P.ta[c]], T.t[2 * c + 1])) { fprintf(stderr, "bad huffman table\n"); return 2; }
    std::vector<uint8_t> stream;
    std::vector<uint32_t> seg_byte;
    if (!lane_unstuff(raw, P.scan_off, stream, seg_byte)) return 3;
    const uint32_t *words = (const uint32_t *)stream.data();
    const int nseg = (int)seg_byte.size() - 1;
    const uint32_t seg_blocks = P.ri ? (uint32_t)P.ri * (uint32_t)G.bpm : kNoRestart;
    // subsequences per segment
    std::vector<uint32_t> seg_sub(nseg + 1, 0);
    for (int s = 0; s < nseg; ++s) seg_sub[s + 1] = seg_sub[s] + ((seg_byte[s + 1] - seg_byte[s]) * 8 + kSubBits - 1) / kSubBits;
    const int nsub = (int)seg_sub[nseg];
    // k_jpeg_columns: word w of subsequence j at col[w * nsub + j]
    std::vector<uint32_t> col((size_t)kColWords * std::max(nsub, 1)), word0(nsub);
    for (int s = 0; s < nseg; ++s)
        for (uint32_t j = seg_sub[s]; j < seg_sub[s + 1]; ++j) {
            word0[j] = (seg_byte[s] * 8 + (j - seg_sub[s]) * kSubBits) >> 5;
            for (int w = 0; w < kColWords; ++w) col[(size_t)w * nsub + j] = words[word0[j] + w];
        }
    auto source = [&](int j) { return WordSource{words, col.data() + j, (uint32_t)nsub, word0[j]}; };
    std::vector<uint64_t> entry(nsub), exitst(nsub);
    std::vector<SubOut> sums(nsub);
    std::vector<uint32_t> endbit(nsub), segof(nsub);
    std::vector<uint8_t> first(nsub);
    // k_jpeg_sync0: every subsequence from the guessed state
    for (int s = 0; s < nseg; ++s)
        for (uint32_t j = seg_sub[s]; j < seg_sub[s + 1]; ++j) {
            const uint32_t start = seg_byte[s] * 8 + (j - seg_sub[s]) * kSubBits;
            uint32_t end = start + kSubBits;
            if (end > seg_byte[s + 1] * 8) end = seg_byte[s + 1] * 8;
            entry[j] = pack_state(start, 0, 0);
            endbit[j] = end;
            segof[j] = (uint32_t)s;
            first[j] = j == seg_sub[s];
            if (!walk3(source((int)j), T, G, entry[j], end, sums[j], (int)j)) return 4;
            exitst[j] = sums[j].exit;
        }
    // k_jpeg_sync: rounds until no exit state changes (lanes of a round read the exit states of the previous round or newer)
    int rounds = 0;
    for (;;) {
        bool changed = false;
        ++rounds;
        for (int j = nsub - 1; j >= 1; --j) {   // descending: every lane sees the PREVIOUS round's states (the slowest legal schedule)
            if (first[j]) continue;
            const uint64_t in = exitst[j - 1];
            if (in == entry[j]) continue;
            entry[j] = in;
            if (!walk3(source(j), T, G, in, endbit[j], sums[j], j)) return 4;
            if (sums[j].exit != exitst[j]) { exitst[j] = sums[j].exit; changed = true; }
        }
        if (!changed) break;
        if (rounds > nsub + 2) { fprintf(stderr, "no fixed point\n"); return 3; }
    }
    // prefix sums with a reset at every restart segment
    std::vector<uint32_t> base_blk(nsub);
    std::vector<int32_t> base_dc(3 * (size_t)nsub);
    {
        uint32_t cb = 0;
        int32_t d0 = 0, d1 = 0, d2 = 0;
        for (int j = 0; j < nsub; ++j) {
            if (first[j]) { cb = 0; d0 = d1 = d2 = 0; }
            base_blk[j] = (seg_blocks == kNoRestart ? 0u : segof[j] * seg_blocks) + cb;
            base_dc[3 * j] = d0; base_dc[3 * j + 1] = d1; base_dc[3 * j + 2] = d2;
            cb += (uint32_t)sums[j].cnt; d0 += sums[j].dc0; d1 += sums[j].dc1; d2 += sums[j].dc2;
        }
    }
    // k_jpeg_coef: every block assembled and stored by the lane in whose range it starts
    std::vector<int16_t> coef((size_t)G.nblk * 64, 0x7777);   // no zero fill: every block must be stored whole by its owner
    uint8_t nat[64];
    for (int k = 0; k < 64; ++k) nat[k] = (uint8_t)natural_of(k);
    for (int j = 0; j < nsub; ++j) {
        uint32_t cap = (uint32_t)G.nblk;
        if (seg_blocks != kNoRestart) { const uint64_t c2 = (uint64_t)(segof[j] + 1) * seg_blocks; if (c2 < cap) cap = (uint32_t)c2; }
        int16_t lbuf[64] = {0};
        decode_sub<true>(source(j), T.t, G, entry[j], endbit[j], coef.data(), base_blk[j], cap, base_dc[3 * j], base_dc[3 * j + 1], base_dc[3 * j + 2], nat, lbuf);
    }
    {   // k_jpeg_coef's own walker (decode_sub_store) into a second buffer: the same coefficients, every block stored whole
        std::vector<int16_t> coef2((size_t)G.nblk * 64, 0x7777);
        for (int j = 0; j < nsub; ++j) {
            uint32_t cap = (uint32_t)G.nblk;
            if (seg_blocks != kNoRestart) { const uint64_t c2 = (uint64_t)(segof[j] + 1) * seg_blocks; if (c2 < cap) cap = (uint32_t)c2; }
            int16_t lbuf[64] = {0};
            decode_sub_store(source(j), T.t, G, entry[j], endbit[j], coef2.data(), base_blk[j], cap, base_dc[3 * j], base_dc[3 * j + 1], base_dc[3 * j + 2], nat, lbuf,
                             true, nullptr);
        }
        if (coef2 != coef) {
            size_t d = 0;
            while (coef2[d] == coef[d]) ++d;
            fprintf(stderr, "decode_sub_store differs from decode_sub<true>: first at coefficient %zu of block %zu (%d against %d)\n", d % 64, d / 64, coef2[d], coef[d]);
            return 4;
        }
    }
    // k_jpeg_idct: lane = (block, column), then (block, row)
    std::vector<uint8_t> planes((size_t)G.plane_bytes);
    for (int c = 0; c < G.nc; ++c) {
        const uint16_t *q = P.q[P.tq[c]];
        const int pw = G.wb[c] * 8;
        for (int b = 0; b < G.wb[c] * G.hb[c]; ++b) {
            const int16_t *cf = coef.data() + ((size_t)G.blk_off[c] + b) * 64;
            int32_t ws[64];
            for (int col = 0; col < 8; ++col) {
                int32_t in[8], o[8];
                for (int r = 0; r < 8; ++r) in[r] = (int32_t)cf[r * 8 + col] * (int32_t)q[r * 8 + col];
                idct_1d(in, o, 11);
                for (int r = 0; r < 8; ++r) ws[r * 8 + col] = o[r];
            }
            const int bx = b % G.wb[c], by = b / G.wb[c];
            for (int r = 0; r < 8; ++r) {
                int32_t o[8];
                idct_1d(ws + r * 8, o, 18);
                for (int x = 0; x < 8; ++x) planes[(size_t)G.plane_off[c] + (size_t)(by * 8 + r) * pw + bx * 8 + x] = (uint8_t)range_limit(o[x]);
            }
        }
    }
    // k_jpeg_color
    std::vector<uint8_t> bgr((size_t)G.w * G.h * 3);
    for (int y = 0; y < G.h; ++y)
        for (int x = 0; x < G.w; ++x) {
            const int Yv = planes[(size_t)G.plane_off[0] + (size_t)y * G.wb[0] * 8 + x];
            uint32_t px;
            if (G.nc == 1) px = (uint32_t)Yv * 0x010101u;
            else {
                const int cb = upsample_at(planes.data() + G.plane_off[1], G.wb[1] * 8, G.dw, G.dh, G.hs, G.vs, x, y);
                const int cr = upsample_at(planes.data() + G.plane_off[2], G.wb[2] * 8, G.dw, G.dh, G.hs, G.vs, x, y);
                px = ycc_to_bgr(Yv, cb, cr);
            }
            uint8_t *o = bgr.data() + ((size_t)y * G.w + x) * 3;
            o[0] = (uint8_t)px; o[1] = (uint8_t)(px >> 8); o[2] = (uint8_t)(px >> 16);
        }
    FILE *f = fopen(out, "wb");
    if (!f) return 2;
    const int32_t hdr[4] = {G.w, G.h, rounds, nsub};
    fwrite(hdr, 4, 4, f);
    fwrite(bgr.data(), 1, bgr.size(), f);
    fclose(f);
    printf("decoded %dx%d nc=%d %dx%d: %d segments, %d subsequences, %d rounds, %ld walks by three walkers each\n", G.w, G.h, G.nc, G.hs, G.vs, nseg, nsub, rounds, g_walks);
    return 0;
}

static int do_encode(const char *in, const char *out)
{
    const std::vector<uint8_t> raw = read_file(in);
    if (raw.size() < 16) return 2;
    const int32_t *hd = (const int32_t *)raw.data();
    const int w = hd[0], h = hd[1], quality = hd[2], sampling = hd[3];
    const uint8_t *bgr = raw.data() + 16;
    const Geom G = make_geom(w, h, 3, sampling >> 4, sampling & 15);
    EncTables T;
    make_enc_tables(quality, T);
    // k_jenc_ycc
    std::vector<uint8_t> planes((size_t)G.plane_bytes);
    for (int yo = 0; yo < G.hb[1] * 8; ++yo)
        for (int xc = 0; xc < G.wb[1] * 8; ++xc)
            enc_ycc_at(bgr, (size_t)w * 3, G, xc, yo, planes.data() + G.plane_off[0], planes.data() + G.plane_off[1], planes.data() + G.plane_off[2]);
    if (G.hs == 2 && G.vs == 2) {   // k_jenc_ycc_h2v2: the same planes, 8 x 2 luma pixels per lane (misaligned rows included: w * 3 need not be a multiple of 4)
        std::vector<uint8_t> planes2((size_t)G.plane_bytes, 0xA5);
        std::vector<uint8_t> fill((size_t)G.plane_bytes, 0xA5);
        for (int yo = 0; yo < G.hb[1] * 8; ++yo)
            for (int t = 0; 4 * t < G.wb[1] * 8; ++t)
                enc_ycc_h2v2_tile(bgr, (size_t)w * 3, G, t, yo, planes2.data() + G.plane_off[0], planes2.data() + G.plane_off[1], planes2.data() + G.plane_off[2]);
        if (planes2 != planes) { fprintf(stderr, "enc_ycc_h2v2_tile differs from enc_ycc_at\n"); return 4; }
    }
    // k_jenc_fdct: lane = (block, row), then (block, column)
    std::vector<int16_t> zz((size_t)G.nblk * 64);
    for (int g = 0; g < G.nblk; ++g) {
        int comp, rx, ry;
        bool dc_only;
        enc_block_root(G, g, comp, rx, ry, dc_only);
        const int pw = G.wb[comp] * 8;
        const uint8_t *Pl = planes.data() + G.plane_off[comp];
        int32_t ws[64];
        for (int r = 0; r < 8; ++r) {
            int32_t in8[8], o[8];
            for (int x = 0; x < 8; ++x) in8[x] = (int32_t)Pl[(size_t)(ry * 8 + r) * pw + rx * 8 + x] - 128;
            fdct_1d(in8, o, 0);
            for (int x = 0; x < 8; ++x) ws[r * 8 + x] = o[x];
        }
        for (int c = 0; c < 8; ++c) {
            int32_t in8[8], o[8];
            for (int r = 0; r < 8; ++r) in8[r] = ws[r * 8 + c];
            fdct_1d(in8, o, 1);
            for (int r = 0; r < 8; ++r) {
                int32_t v = quantize(o[r], T.q[comp ? 1 : 0][r * 8 + c], T.recip[comp ? 1 : 0][r * 8 + c]);
                if (dc_only && (r | c)) v = 0;
                zz[(size_t)g * 64 + zigzag_of(r * 8 + c)] = (int16_t)v;
            }
        }
    }
    // k_jenc_fdct's last step (AC code bits per block, the quantised DC aside) + k_jenc_scan (DC code bits, prefix)
    std::vector<uint16_t> acbits(G.nblk);
    std::vector<int16_t> dcq(G.nblk);
    for (int g = 0; g < G.nblk; ++g) {
        const int t = (g % G.bpm) < G.nY ? 0 : 1;
        acbits[g] = (uint16_t)ac_code_bits(zz.data() + (size_t)g * 64, T.ac[t].len);
        {   // k_jenc_fdct prices the block with eight lanes (ac_code_bits_octet): the same number
            const int16_t *z = zz.data() + (size_t)g * 64;
            const uint64_t nonzero = nonzero_mask(z);
            uint32_t sum = 0;
            for (int r8 = 0; r8 < 8; ++r8) sum += ac_code_bits_octet(z + 8 * r8, r8, nonzero, T.ac[t].len);
            if (sum != acbits[g]) { fprintf(stderr, "ac_code_bits_octet: block %d: %u against %u\n", g, sum, (unsigned)acbits[g]); return 4; }
        }
        dcq[g] = zz[(size_t)g * 64];
    }
    std::vector<uint32_t> pos(G.nblk + 1, 0);
    for (int g = 0; g < G.nblk; ++g) {
        const int z = g % G.bpm, pr = dc_predecessor(g, G);
        const int last = pr < 0 ? 0 : dcq[pr];
        const int t = z < G.nY ? 0 : 1;
        const uint32_t len = (uint32_t)acbits[g] + dc_code_bits((int)dcq[g] - last, T.dc[t].len);
        if (len != encode_block<false>(zz.data() + (size_t)g * 64, last, T.dc[t], T.ac[t], nullptr, 0, nonzero_mask(zz.data() + (size_t)g * 64))) { fprintf(stderr, "split length differs at block %d\n", g); return 3; }
        pos[g + 1] = pos[g] + len;
    }
    const uint32_t total_bits = pos[G.nblk];
    const uint32_t nbytes = (total_bits + 7) / 8;
    std::vector<uint32_t> words(nbytes / 4 + 2, 0);
    // k_jenc_bits
    for (int g = 0; g < G.nblk; ++g) {
        const int z = g % G.bpm, pr = dc_predecessor(g, G);
        const int last = pr < 0 ? 0 : zz[(size_t)pr * 64];
        const int t = z < G.nY ? 0 : 1;
        const uint32_t n = encode_block<true>(zz.data() + (size_t)g * 64, last, T.dc[t], T.ac[t], words.data(), pos[g], nonzero_mask(zz.data() + (size_t)g * 64));
        if (n != pos[g + 1] - pos[g]) { fprintf(stderr, "length mismatch at block %d\n", g); return 3; }
    }
    // k_jenc_stuff
    std::vector<uint8_t> file = make_file_header(w, h, G.hs, G.vs, T);
    const int pad = (int)(nbytes * 8 - total_bits);
    for (uint32_t i = 0; i < nbytes; ++i) {
        uint8_t b = (uint8_t)(words[i >> 2] >> (24 - 8 * (i & 3)));
        if (i == nbytes - 1) b |= (uint8_t)((1u << pad) - 1u);
        file.push_back(b);
        if (b == 0xFF) file.push_back(0);
    }
    file.push_back(0xFF);
    file.push_back(0xD9);
    FILE *f = fopen(out, "wb");
    if (!f) return 2;
    fwrite(file.data(), 1, file.size(), f);
    fclose(f);
    printf("encoded %dx%d q%d %dx%d: %u bits, %zu bytes\n", w, h, quality, G.hs, G.vs, total_bits, file.size());
    return 0;
}


// jpeg_emulate stats <in.jpg>: how the synchronisation converges for a few guessing strategies (offline study, not a test)
static int do_stats(const char *in)
{
    const std::vector<uint8_t> raw = read_file(in);
    Parsed P;
    std::string why;
    if (parse_header(raw.data(), raw.size(), P, why)) return 2;
    const Geom G = make_geom(P.w, P.h, P.nc, P.hs, P.vs);
    TableSet T;
    for (int c = 0; c < P.nc; ++c) { make_hufftab(P.dc[P.td[c]], T.t[2 * c]); make_hufftab(P.ac[P.ta[c]], T.t[2 * c + 1]); }
    std::vector<uint8_t> stream;
    std::vector<uint32_t> seg_byte;
    if (!lane_unstuff(raw, P.scan_off, stream, seg_byte)) return 3;
    const uint32_t *words = (const uint32_t *)stream.data();
    const uint32_t total_bits = seg_byte.back() * 8;
    const int nsub = (int)((total_bits + kSubBits - 1) / kSubBits);   // (files without restart markers)
    const WordSource src{words, nullptr, 0, 0};
    // the true states at every subsequence start: one sequential walk
    std::vector<uint64_t> truth(nsub + 1);
    {
        uint64_t st = pack_state(0, 0, 0);
        for (int j = 0; j < nsub; ++j) {
            truth[j] = st;
            const uint32_t end = std::min<uint32_t>((uint32_t)(j + 1) * kSubBits, total_bits);
            st = decode_sub<false>(src, T.t, G, st, end, nullptr, 0, 0, 0, 0, 0).exit;
        }
        truth[nsub] = st;
    }
    for (int warm : {0, 128, 256, 512, 1024}) {
        std::vector<uint64_t> entry(nsub), exitst(nsub);
        for (int j = 0; j < nsub; ++j) {
            const uint32_t start = (uint32_t)j * kSubBits, end = std::min<uint32_t>(start + kSubBits, total_bits);
            uint64_t e = pack_state(start, 0, 0);
            if (j > 0 && warm) {
                const uint32_t w0 = start > (uint32_t)warm ? start - (uint32_t)warm : 0;
                e = decode_sub<false>(src, T.t, G, pack_state(w0, 0, 0), start, nullptr, 0, 0, 0, 0, 0).exit;
            }
            entry[j] = j == 0 ? truth[0] : e;
            exitst[j] = decode_sub<false>(src, T.t, G, entry[j], end, nullptr, 0, 0, 0, 0, 0).exit;
        }
        int right_entry = 0, right_exit = 0;
        for (int j = 0; j < nsub; ++j) { right_entry += entry[j] == truth[j]; right_exit += exitst[j] == truth[j + 1]; }
        printf("warm-up %4d bits: after the first pass %5.1f %% of the entry guesses and %5.1f %% of the exit states are the true ones; re-decodes per round:", warm,
               100.0 * right_entry / nsub, 100.0 * right_exit / nsub);
        int rounds = 0;
        for (;;) {
            std::vector<uint64_t> prev = exitst;
            int redo = 0;
            for (int j = 1; j < nsub; ++j) {
                if (prev[j - 1] == entry[j]) continue;
                entry[j] = prev[j - 1];
                const uint32_t end = std::min<uint32_t>((uint32_t)(j + 1) * kSubBits, total_bits);
                exitst[j] = decode_sub<false>(src, T.t, G, entry[j], end, nullptr, 0, 0, 0, 0, 0).exit;
                ++redo;
            }
            if (!redo) break;
            ++rounds;
            printf(" %d", redo);
            if (warm == 0 && redo <= 2 && getenv("BEVW_STATS_VERBOSE"))
                for (int j = 1; j < nsub; ++j)
                    if (exitst[j] != prev[j])
                        printf("\n    round %d sub %d: exit was p=%u z=%u k=%u, is p=%u z=%u k=%u (truth p=%u z=%u k=%u)", rounds, j, (uint32_t)prev[j], (uint32_t)(prev[j] >> 32) & 255u,
                               (uint32_t)(prev[j] >> 40) & 255u, (uint32_t)exitst[j], (uint32_t)(exitst[j] >> 32) & 255u, (uint32_t)(exitst[j] >> 40) & 255u, (uint32_t)truth[j + 1],
                               (uint32_t)(truth[j + 1] >> 32) & 255u, (uint32_t)(truth[j + 1] >> 40) & 255u);
        }
        printf("  (%d rounds, %d subsequences)\n", rounds, nsub);
    }
    // the lever DESIGN.md section 9 names: from round 3 on, a lane that has to walk again also walks the `window` subsequences behind it under all
    // bpm block phases (same bit / zigzag position as their current entry), and the chain is resolved through those tables as far as the
    // positions agree -- rounds and walks this would take
    for (int window : {0, 4, 8, 16}) {
        std::vector<uint64_t> entry(nsub), exitst(nsub);
        for (int j = 0; j < nsub; ++j) {
            const uint32_t start = (uint32_t)j * kSubBits, end = std::min<uint32_t>(start + kSubBits, total_bits);
            entry[j] = j == 0 ? truth[0] : pack_state(start, 0, 0);
            exitst[j] = decode_sub<false>(src, T.t, G, entry[j], end, nullptr, 0, 0, 0, 0, 0).exit;
        }
        int rounds = 0;
        long walks = nsub, ahead = 0, broke = 0;
        auto pk = [](uint64_t st) { return st & ~((uint64_t)255 << 32); };   // the state without the block index
        for (;;) {
            std::vector<uint64_t> prev = exitst;
            int redo = 0;
            for (int j = 1; j < nsub; ++j) {
                if (prev[j - 1] == entry[j]) continue;
                ++redo;
                entry[j] = prev[j - 1];
                const uint32_t end = std::min<uint32_t>((uint32_t)(j + 1) * kSubBits, total_bits);
                exitst[j] = decode_sub<false>(src, T.t, G, entry[j], end, nullptr, 0, 0, 0, 0, 0).exit;
                ++walks;
                if (window && rounds >= 2) {   // resolve ahead through all-phase tables
                    uint64_t carry = exitst[j];
                    for (int t = j + 1; t < std::min(nsub, j + 1 + window); ++t) {
                        if (pk(carry) != pk(entry[t])) { ++broke; break; }   // the position changed too: that link needs a real round
                        walks += G.bpm;                          // the table of subsequence t: one walk per phase
                        if (carry == entry[t]) break;            // the chain ends here
                        entry[t] = carry;
                        const uint32_t e2 = std::min<uint32_t>((uint32_t)(t + 1) * kSubBits, total_bits);
                        exitst[t] = decode_sub<false>(src, T.t, G, carry, e2, nullptr, 0, 0, 0, 0, 0).exit;
                        prev[t] = exitst[t];                     // later lanes of this round see the resolved state
                        carry = exitst[t];
                        ++ahead;
                    }
                }
            }
            if (!redo) break;
            ++rounds;
        }
        printf("all-phase tables over %2d subsequences ahead (from round 3): %d rounds, %ld walks, %ld links resolved ahead, %ld chains stopped by a position change (%d subsequences)\n", window, rounds, walks, ahead, broke, nsub);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 4 && !strcmp(argv[1], "decode")) return do_decode(argv[2], argv[3]);
    if (argc == 4 && !strcmp(argv[1], "encode")) return do_encode(argv[2], argv[3]);
    if (argc == 3 && !strcmp(argv[1], "unstuff")) return do_unstuff(argv[2]);
    if (argc == 3 && !strcmp(argv[1], "stats")) return do_stats(argv[2]);
    fprintf(stderr, "usage: jpeg_emulate decode in.jpg out.bin | encode in.bin out.jpg\n");
    return 1;
}
