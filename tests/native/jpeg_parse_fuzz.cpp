// Host-only mutation fuzz of the JPEG marker parser (cameracalibration_amd/csrc/bevw_jpeg.h: parse_header, exif_orientation, make_hufftab,
// unstuff_scan) -- the one part of the codec that reads attacker-controlled lengths on the HOST.  Meant to be built with
// -fsanitize=address,undefined (tests/test_sanitizers.py): every out-of-bounds read, overflow or invalid shift aborts the run.
//
//   jpeg_parse_fuzz <seed> <mutations per file> file.jpg [file.jpg ...]
//
// Per mutated file the harness does what bevw_jpeg_probe / bevw_jpeg_decode_stage do with the bytes on the host (csrc/bevwarp_jpeg.hip):
// parse the header, derive the Huffman tables the scan refers to, size the staging slot from scan_off, un-stuff the entropy-coded bytes
// (the host statement of the un-stuffing kernels).  The mutated bytes live in an exactly-sized heap block so that ASan sees every overrun.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <hip/hip_runtime.h>

#include "../../cameracalibration_amd/csrc/bevw_jpeg.h"

using namespace bevw;

static uint32_t g_seed = 1;
static uint32_t rnd() { g_seed = g_seed * 1664525u + 1013904223u; return g_seed >> 8; }

static int exercise(const uint8_t *d, size_t n, size_t &accepted)
{
    jpg::Parsed P;
    std::string why;
    const int st = jpg::parse_header(d, n, P, why);
    if (st != jpg::kParseOk) return st;
    ++accepted;
    if (P.scan_off > n) { fprintf(stderr, "FAIL: scan_off %zu beyond the file (%zu bytes)\n", P.scan_off, n); exit(1); }
    if (P.w <= 0 || P.h <= 0 || P.w > 65535 || P.h > 65535 || (P.nc != 1 && P.nc != 3)) { fprintf(stderr, "FAIL: accepted geometry %d x %d x %d\n", P.w, P.h, P.nc); exit(1); }
    for (int c = 0; c < P.nc; ++c) {
        jpg::HuffTab T;
        (void)jpg::make_hufftab(P.dc[P.td[c]], T);
        (void)jpg::make_hufftab(P.ac[P.ta[c]], T);
    }
    std::vector<uint8_t> dst(n - P.scan_off + 16);
    std::vector<uint32_t> seg;
    const size_t o = jpg::unstuff_scan(d, n, P.scan_off, dst.data(), seg);
    if (o + 16 > dst.size() || seg.size() < 2) { fprintf(stderr, "FAIL: un-stuffed %zu bytes into %zu\n", o, dst.size()); exit(1); }
    return st;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && strcmp(argv[1], "--bevw-selfcheck-noop") == 0) return 0;
    if (argc < 4) { fprintf(stderr, "usage: jpeg_parse_fuzz <seed> <mutations per file> file.jpg ...\n"); return 2; }
    g_seed = (uint32_t)atoi(argv[1]);
    const int per_file = atoi(argv[2]);
    size_t runs = 0, accepted = 0, pristine_ok = 0;
    for (int a = 3; a < argc; ++a) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) { fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
        std::vector<uint8_t> seed;
        uint8_t buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) seed.insert(seed.end(), buf, buf + got);
        fclose(f);
        size_t acc0 = 0;
        if (exercise(seed.data(), seed.size(), acc0) == jpg::kParseOk) ++pristine_ok;
        // the header ends at the first SOS payload: mutations concentrate there, the entropy-coded bytes get a few as well
        size_t hdr = seed.size();
        for (size_t i = 0; i + 1 < seed.size(); ++i)
            if (seed[i] == 0xFF && seed[i + 1] == 0xDA) { hdr = i + 16 < seed.size() ? i + 16 : seed.size(); break; }
        for (int it = 0; it < per_file; ++it) {
            std::vector<uint8_t> m = seed;
            const unsigned kind = rnd() % 7;
            if (kind == 0) {                                   // a few byte flips in the header
                for (unsigned k = 0, nk = 1 + rnd() % 4; k < nk; ++k) m[rnd() % hdr] ^= (uint8_t)(1u << (rnd() % 8));
            } else if (kind == 1) {                            // random bytes in the header
                for (unsigned k = 0, nk = 1 + rnd() % 8; k < nk; ++k) m[rnd() % hdr] = (uint8_t)rnd();
            } else if (kind == 2) {                            // truncation (anywhere, mostly inside the header)
                m.resize(rnd() % 3 ? rnd() % (hdr + 1) : rnd() % (m.size() + 1));
            } else if (kind == 3) {                            // a segment length field set to an extreme or random value
                std::vector<size_t> segs;
                for (size_t i = 2; i + 3 < hdr;) {
                    if (m[i] != 0xFF) break;
                    segs.push_back(i + 2);
                    i += 2 + (((size_t)m[i + 2] << 8) | m[i + 3]);
                }
                if (!segs.empty()) {
                    const size_t p = segs[rnd() % segs.size()];
                    static const uint16_t ext[] = {0, 1, 2, 3, 0xffff, 0x7fff, 17, 18, 19, 64, 65, 66, 67};
                    const uint16_t v = rnd() % 2 ? ext[rnd() % (sizeof ext / sizeof ext[0])] : (uint16_t)rnd();
                    m[p] = (uint8_t)(v >> 8); m[p + 1] = (uint8_t)v;
                }
            } else if (kind == 4) {                            // marker bytes sprinkled over the entropy-coded data + truncation
                for (unsigned k = 0, nk = 1 + rnd() % 6; k < nk && m.size() > hdr + 2; ++k) {
                    const size_t p = hdr + rnd() % (m.size() - hdr - 1);
                    m[p] = 0xFF; m[p + 1] = (uint8_t)(rnd() % 3 ? 0xD0 + rnd() % 10 : rnd());
                }
                if (rnd() % 2) m.resize(hdr + rnd() % (m.size() - hdr + 1));
            } else if (kind == 5) {                            // DHT: code counts moved between lengths (the segment keeps its size; over-subscribed codes)
                std::vector<size_t> tabs;
                for (size_t i = 2; i + 3 < hdr;) {
                    if (m[i] != 0xFF) break;
                    const size_t L = ((size_t)m[i + 2] << 8) | m[i + 3];
                    if (m[i + 1] == 0xC4)
                        for (size_t o = i + 4; o + 17 <= i + 2 + L;) {
                            tabs.push_back(o);
                            size_t cnt = 0;
                            for (int l = 1; l <= 16; ++l) cnt += m[o + l];
                            o += 17 + cnt;
                        }
                    i += 2 + L;
                }
                if (!tabs.empty()) {
                    const size_t o = tabs[rnd() % tabs.size()];
                    for (unsigned k = 0, nk = 1 + rnd() % 3; k < nk; ++k) {
                        const int a = 1 + rnd() % 16, b = 1 + rnd() % 16;
                        const int mv = m[o + a] ? 1 + rnd() % m[o + a] : 0;
                        if (m[o + b] + mv <= 255) { m[o + a] = (uint8_t)(m[o + a] - mv); m[o + b] = (uint8_t)(m[o + b] + mv); }
                    }
                }
            } else {                                           // a header segment duplicated / moved
                const size_t from = rnd() % hdr, len = rnd() % 64, to = rnd() % hdr;
                for (size_t k = 0; k < len && from + k < m.size() && to + k < m.size(); ++k) m[to + k] = seed[from + k];
            }
            // exactly-sized heap copy: reads one byte beyond the file are ASan errors
            uint8_t *exact = (uint8_t *)malloc(m.size() ? m.size() : 1);
            memcpy(exact, m.data(), m.size());
            exercise(m.size() ? exact : nullptr, m.size(), accepted);
            free(exact);
            ++runs;
        }
    }
    printf("jpeg parser fuzz ok: %zu mutated files (%zu still accepted), %zu of %d pristine files accepted\n", runs, accepted, pristine_ok, argc - 3);
    return pristine_ok == (size_t)(argc - 3) ? 0 : 1;
}
