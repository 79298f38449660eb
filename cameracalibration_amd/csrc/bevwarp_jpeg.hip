// bevwarp_jpeg.hip -- the JPEG translation unit of libbevwarp.so: cv2.imread (main.py:74-77) / cv2.imwrite (surroundBEV.py:340) either
// side of the path, on the GPU.  Kernels: bevw_jpeg_codec.h; lane code shared with the CPU emulator:
// bevw_jpeg.h.  C-ABI: the bevw_jpeg_* entry points of include/bevwarp.h.
#include "bevw_host.h"

#include <algorithm>
#include <new>
#include <string>
#include <thread>

#include "bevw_jpeg_codec.h"

using namespace bevw;

struct bevw_jpeg {
    int device = 0;
    hipStream_t st = nullptr, st2 = nullptr;   // st2: the odd slices of a decode batch
    hipStream_t st_more[2] = {nullptr, nullptr};   // third and fourth slice stream (BEVW_JPEG_STREAMS > 2; created on first use)
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_more[2] = {nullptr, nullptr};
    LapTimer timer;
    // decode: what bevw_jpeg_decode_stage left on the device
    jpg::Geom G{};
    int n = 0;
    bool staged = false, decoded = false;
    int orientation = 1;                       // EXIF orientation of the staged batch (cv2.imread applies it)
    bool luma_planes = true;                   // the last decode wrote the luma sample planes (false: fused luma IDCT + colour conversion)
    size_t total_sub = 0;
    uint32_t max_sub = 0, max_chunk = 0;
    hipEvent_t ev_x = nullptr;                 // ordering against an engine's stream (bevw_jpeg_wait_engine / bevw_wait_jpeg)
    PinnedBuf h_stream;
    std::vector<jpg::ImageDesc> h_desc;
    std::vector<uint32_t> h_term;
    std::vector<jpg::TableSet> h_tabs;
    std::vector<uint16_t> h_quant;
    DevBuf d_raw, d_stream, d_desc, d_seg_byte, d_seg_sub, d_term, d_nrst, d_chunk_keep, d_chunk_rst, d_tabs, d_quant;
    DevBuf d_turn;            // files with an EXIF orientation: the images as stored, before k_jpeg_orient
    DevBuf d_list, d_count;   // round 2: per image, the subsequences that walk again
    DevBuf d_entry, d_exit, d_exit2, d_sums, d_base, d_endbit, d_meta, d_word0, d_cols, d_rounds, d_coef, d_planes, d_img;
    // encode
    jpg::Geom EG{};
    int en = 0, e_quality = -1, e_sampling = -1;
    jpg::EncTables etabs;
    std::vector<uint8_t> header;
    DevBuf d_etabs, d_header, d_eplanes, d_zz, d_acbits, d_dcq, d_nzmask, d_bitlen, d_bitbuf, d_totals, d_chunk_ff, d_files, d_sizes, d_src;
    DevBuf d_packed, d_offsets;                // bevw_jpeg_encoded_fetch: the files of a batch back to back
    size_t buf_words = 0, file_cap = 0;
    std::vector<uint32_t> sizes;
    bool encoded = false, sizes_valid = false;
};

static int jpeg_parse_fail(int st, int index, const std::string &why)
{
    return fail(BEVW_E_INVALID, "JPEG %d: %s%s", index, st == jpg::kParseUnsupported ? "outside the supported subset: " : "", why.c_str());
}

int bevw_jpeg_probe(const uint8_t *data, size_t len, int32_t info[8])
{
    if (!data || !info) return fail(BEVW_E_INVALID, "bevw_jpeg_probe: null argument");
    jpg::Parsed P;
    std::string why;
    const int st = jpg::parse_header(data, len, P, why);
    if (st) return jpeg_parse_fail(st, 0, why);
    info[0] = P.w; info[1] = P.h; info[2] = P.nc; info[3] = P.hs; info[4] = P.vs; info[5] = P.ri; info[6] = P.orientation; info[7] = 0;
    return BEVW_OK;
}

int bevw_jpeg_create(int device, bevw_jpeg **out)
{
    if (!out) return fail(BEVW_E_INVALID, "bevw_jpeg_create: null out");
    *out = nullptr;
    BEVW_TRY(use_device(device));
    bevw_jpeg *j = new (std::nothrow) bevw_jpeg();
    if (!j) return fail(BEVW_E_NOMEM, "out of host memory");
    j->device = device;
    hipError_t e = hipStreamCreateWithFlags(&j->st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&j->st2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&j->ev_a, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&j->ev_b, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&j->ev_x, hipEventDisableTiming);
    if (e != hipSuccess) { bevw_jpeg_destroy(j); return fail(BEVW_E_HIP, "stream / event creation failed: %s", hipGetErrorString(e)); }
    *out = j;
    return BEVW_OK;
}

void bevw_jpeg_destroy(bevw_jpeg *j)
{
    if (!j) return;
    (void)hipSetDevice(j->device);
    if (j->st2) { (void)hipStreamSynchronize(j->st2); (void)hipStreamDestroy(j->st2); }
    for (int i = 0; i < 2; ++i) {
        if (j->st_more[i]) { (void)hipStreamSynchronize(j->st_more[i]); (void)hipStreamDestroy(j->st_more[i]); }
        if (j->ev_more[i]) (void)hipEventDestroy(j->ev_more[i]);
    }
    if (j->st) { (void)hipStreamSynchronize(j->st); (void)hipStreamDestroy(j->st); }
    if (j->ev_a) (void)hipEventDestroy(j->ev_a);
    if (j->ev_b) (void)hipEventDestroy(j->ev_b);
    if (j->ev_x) (void)hipEventDestroy(j->ev_x);
    j->timer.release();
    delete j;
}

int bevw_jpeg_decode_stage(bevw_jpeg *j, const uint8_t *const *data, const size_t *len, int n)
{
    if (!j || !data || !len || n <= 0 || n > 65535) return fail(BEVW_E_INVALID, "bevw_jpeg_decode_stage: bad argument (1 <= n <= 65535)");
    BEVW_TRY(use_device(j->device));
    HIP_TRY(hipStreamSynchronize(j->st));   // the staging buffers of the previous batch may still be in flight
    j->staged = j->decoded = false;
    std::vector<jpg::Parsed> P((size_t)n);
    std::vector<size_t> slot_off((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) {
        std::string why;
        if (!data[i]) return fail(BEVW_E_INVALID, "JPEG %d: null pointer", i);
        const int st = jpg::parse_header(data[i], len[i], P[i], why);
        if (st) return jpeg_parse_fail(st, i, why);
        if (i && P[i].orientation != P[0].orientation)
            return fail(BEVW_E_INVALID, "JPEG %d has EXIF orientation %d but the batch has %d: one orientation per batch", i, P[i].orientation, P[0].orientation);
        if (i && (P[i].w != P[0].w || P[i].h != P[0].h || P[i].nc != P[0].nc || P[i].hs != P[0].hs || P[i].vs != P[0].vs))
            return fail(BEVW_E_INVALID, "JPEG %d is %dx%d (%d components, luma %dx%d) but the batch is %dx%d (%d, %dx%d): one geometry per batch", i,
                        P[i].w, P[i].h, P[i].nc, P[i].hs, P[i].vs, P[0].w, P[0].h, P[0].nc, P[0].hs, P[0].vs);
        slot_off[i + 1] = slot_off[i] + (((len[i] - P[i].scan_off + 32 + 15) & ~(size_t)15) + 16);
    }
    const size_t bound = slot_off[n];
    if (bound >= ((size_t)1 << 32)) return fail(BEVW_E_INVALID, "batch of %zu entropy-coded bytes: split it (4 GiB per batch)", bound);
    j->G = jpg::make_geom(P[0].w, P[0].h, P[0].nc, P[0].hs, P[0].vs);
    j->orientation = P[0].orientation;
    const jpg::Geom &G = j->G;
    if (G.nblk >= (1 << 26)) return fail(BEVW_E_INVALID, "JPEG of %dx%d: more than 2^26 blocks per image", G.w, G.h);   // (k_jpeg_coef packs block index << 6 | lane)
    BEVW_TRY(j->h_stream.reserve(bound));
    j->h_desc.assign((size_t)n, jpg::ImageDesc());
    j->h_term.assign((size_t)n, 0);
    j->h_tabs.clear();
    j->h_quant.assign((size_t)n * 192, 0);
    std::vector<std::string> keys;
    size_t seg_total = 0, sub_total = 0, chunk_total = 0;
    uint32_t max_sub = 0, max_chunk = 0;
    const uint32_t nmcu = (uint32_t)G.mcux * (uint32_t)G.mcuy;
    for (int i = 0; i < n; ++i) {
        jpg::ImageDesc &D = j->h_desc[i];
        // tables: identical table sets are shared (cameras of one rig write the same ones)
        std::string key;
        for (int c = 0; c < P[i].nc; ++c) {
            key.append((const char *)&P[i].dc[P[i].td[c]], sizeof(jpg::RawHuff));
            key.append((const char *)&P[i].ac[P[i].ta[c]], sizeof(jpg::RawHuff));
        }
        size_t t = 0;
        while (t < keys.size() && keys[t] != key) ++t;
        if (t == keys.size()) {
            jpg::TableSet T;
            memset(&T, 0, sizeof T);
            for (int c = 0; c < P[i].nc; ++c)
                if (!jpg::make_hufftab(P[i].dc[P[i].td[c]], T.t[2 * c]) || !jpg::make_hufftab(P[i].ac[P[i].ta[c]], T.t[2 * c + 1]))
                    return fail(BEVW_E_INVALID, "JPEG %d: over-subscribed Huffman table", i);
            keys.push_back(key);
            j->h_tabs.push_back(T);
        }
        D.tables = (uint32_t)t;
        D.quant = (uint32_t)i;
        for (int c = 0; c < P[i].nc; ++c) memcpy(&j->h_quant[(size_t)i * 192 + c * 64], P[i].q[P[i].tq[c]], 128);
        // what the un-stuffing kernels need: the slot, the raw length, the segments DRI promises, room for the subsequences
        const uint32_t raw = (uint32_t)(len[i] - P[i].scan_off);
        if (raw < 2) return fail(BEVW_E_INVALID, "JPEG %d: no entropy-coded data behind the scan header", i);
        D.stream_word = (uint32_t)(slot_off[i] >> 2);
        D.raw_bytes = raw;
        D.nseg = P[i].ri ? (nmcu + (uint32_t)P[i].ri - 1) / (uint32_t)P[i].ri : 1u;
        D.seg_blocks = P[i].ri ? (uint32_t)P[i].ri * (uint32_t)G.bpm : jpg::kNoRestart;
        D.seg_first = (uint32_t)seg_total;
        D.sub_first = (uint32_t)sub_total;
        D.chunk_first = (uint32_t)chunk_total;
        const uint32_t sub_ub = (raw * 8u + (uint32_t)jpg::kSubBits - 1u) / (uint32_t)jpg::kSubBits + D.nseg;   // every segment rounds up once
        const uint32_t chunks = std::max(1u, (raw + jpg::kRawChunk - 1u) / jpg::kRawChunk);
        seg_total += D.nseg + 1;
        sub_total += sub_ub;
        chunk_total += chunks;
        max_sub = std::max(max_sub, sub_ub);
        max_chunk = std::max(max_chunk, chunks);
        j->h_term[i] = raw;
    }
    if (sub_total >= ((size_t)1 << 31)) return fail(BEVW_E_INVALID, "batch too large");
    // The staging copy is a plain copy of the entropy-coded bytes (dealt over host threads); the GPU removes the stuffing.
    {
        static const int threads_env = [] { const char *e = getenv("BEVW_JPEG_HOST_THREADS"); return e ? atoi(e) : 0; }();
        int nt = threads_env > 0 ? threads_env : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
        nt = std::max(1, std::min(nt, n / 8));   // a thread per >= 8 files, else not worth starting
        auto work = [&](int t) {
            for (int i = t; i < n; i += nt) {
                uint8_t *dst = (uint8_t *)j->h_stream.p + slot_off[i];
                const size_t raw = len[i] - P[i].scan_off;
                memcpy(dst, data[i] + P[i].scan_off, raw);
                memset(dst + raw, 0, slot_off[i + 1] - slot_off[i] - raw);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
        work(0);
        for (std::thread &th : pool) th.join();
    }
    j->n = n;
    j->total_sub = sub_total;
    j->max_sub = max_sub;
    BEVW_TRY(j->d_raw.reserve(bound + 64));
    BEVW_TRY(j->d_stream.reserve(bound + 1024));   // k_jpeg_columns copies a fixed kColWords words per subsequence, and a lane finishing a block of corrupt data can run ~250 bytes past the end
    BEVW_TRY(j->d_desc.reserve(j->h_desc.size() * sizeof(jpg::ImageDesc)));
    BEVW_TRY(j->d_seg_byte.reserve(seg_total * 4));
    BEVW_TRY(j->d_seg_sub.reserve(seg_total * 4));
    BEVW_TRY(j->d_term.reserve((size_t)n * 4));
    BEVW_TRY(j->d_nrst.reserve((size_t)n * 4));
    BEVW_TRY(j->d_chunk_keep.reserve(chunk_total * 4));
    BEVW_TRY(j->d_chunk_rst.reserve(chunk_total * 4));
    BEVW_TRY(j->d_tabs.reserve(j->h_tabs.size() * sizeof(jpg::TableSet)));
    BEVW_TRY(j->d_quant.reserve(j->h_quant.size() * 2));
    HIP_TRY(hipMemcpyAsync(j->d_raw.p, j->h_stream.p, bound, hipMemcpyHostToDevice, j->st));
    HIP_TRY(hipMemcpyAsync(j->d_desc.p, j->h_desc.data(), j->h_desc.size() * sizeof(jpg::ImageDesc), hipMemcpyHostToDevice, j->st));
    HIP_TRY(hipMemcpyAsync(j->d_term.p, j->h_term.data(), (size_t)n * 4, hipMemcpyHostToDevice, j->st));
    HIP_TRY(hipMemcpyAsync(j->d_tabs.p, j->h_tabs.data(), j->h_tabs.size() * sizeof(jpg::TableSet), hipMemcpyHostToDevice, j->st));
    HIP_TRY(hipMemcpyAsync(j->d_quant.p, j->h_quant.data(), j->h_quant.size() * 2, hipMemcpyHostToDevice, j->st));
    j->max_chunk = max_chunk;
    j->staged = true;
    return BEVW_OK;
}

int bevw_jpeg_decode_run_device(bevw_jpeg *j, void *d_out, size_t image_stride_bytes, size_t row_pitch_bytes)
{
    if (!j || !d_out) return fail(BEVW_E_INVALID, "bevw_jpeg_decode_run_device: null argument");
    if (!j->staged) return fail(BEVW_E_INVALID, "bevw_jpeg_decode_run_device before bevw_jpeg_decode_stage");
    const jpg::Geom &G = j->G;
    const bool oriented = j->orientation != 1;
    const int ow = j->orientation >= 5 ? G.h : G.w, oh = j->orientation >= 5 ? G.w : G.h;   // what cv2.imread returns
    if (row_pitch_bytes < (size_t)ow * 3 || image_stride_bytes < row_pitch_bytes * (size_t)oh)
        return fail(BEVW_E_INVALID, "output layout (pitch %zu, stride %zu) too small for %dx%d BGR", row_pitch_bytes, image_stride_bytes, ow, oh);
    BEVW_TRY(use_device(j->device));
    // files with an EXIF orientation: decoded as stored into a scratch batch, then turned into the caller's layout (k_jpeg_orient)
    void *const d_final = d_out;
    const size_t final_stride = image_stride_bytes, final_pitch = row_pitch_bytes;
    if (oriented) {
        BEVW_TRY(j->d_turn.reserve((size_t)j->n * G.w * G.h * 3));
        d_out = j->d_turn.p;
        image_stride_bytes = (size_t)G.w * G.h * 3;
        row_pitch_bytes = (size_t)G.w * 3;
    }
    const size_t ns = j->total_sub ? j->total_sub : 1, n = (size_t)j->n;
    BEVW_TRY(j->d_entry.reserve(ns * 8));
    BEVW_TRY(j->d_exit.reserve(ns * 8));
    BEVW_TRY(j->d_exit2.reserve(ns * 8));
    BEVW_TRY(j->d_sums.reserve(ns * 16));
    BEVW_TRY(j->d_base.reserve(ns * 16));
    BEVW_TRY(j->d_endbit.reserve(ns * 4));
    BEVW_TRY(j->d_meta.reserve(ns * 4));
    BEVW_TRY(j->d_word0.reserve(ns * 4));
    BEVW_TRY(j->d_cols.reserve(ns * 4 * (size_t)jpg::kColWords));
    BEVW_TRY(j->d_rounds.reserve(n * 4));
    BEVW_TRY(j->d_list.reserve(ns * 4));
    BEVW_TRY(j->d_count.reserve(n * 4));
    BEVW_TRY(j->d_coef.reserve(n * (size_t)G.nblk * 128));
    BEVW_TRY(j->d_planes.reserve(n * (size_t)G.plane_bytes));
    jpg::SubArrays A{j->d_entry.as<uint64_t>(), j->d_exit.as<uint64_t>(), j->d_sums.as<int4>(), j->d_base.as<int4>(), j->d_endbit.as<uint32_t>(),
                     j->d_meta.as<uint32_t>(), j->d_word0.as<uint32_t>(), j->d_cols.as<uint32_t>()};
    const jpg::ImageDesc *img = j->d_desc.as<jpg::ImageDesc>();
    const uint32_t *stream = j->d_stream.as<uint32_t>();
    const jpg::TableSet *tabs = j->d_tabs.as<jpg::TableSet>();
    // Un-stuffing on the device, part of every decode (round 4: inside the run, so that a timed decode counts it -- what is RESIDENT after
    // bevw_jpeg_decode_stage is the files' entropy-coded bytes as they are in the files): where the data ends, what stays, where the restart
    // segments start, the subsequences
    HIP_TRY(hipMemcpyAsync(j->d_term.p, j->h_term.data(), (size_t)j->n * 4, hipMemcpyHostToDevice, j->st));   // (k_jpeg_find_end lowers it: a fresh copy per run)
    HIP_TRY(hipMemsetAsync(j->d_nrst.p, 0xFF, (size_t)j->n * 4, j->st));   // an image no kernel closes can never pass k_jpeg_subs' check
    jpg::ImageDesc *imgw = j->d_desc.as<jpg::ImageDesc>();
    const uint8_t *raw = j->d_raw.as<uint8_t>();
    // (these four kernels run per slice, below: a slice's un-stuffing overlaps the other slice's first passes)
    // (no zero fill of the coefficient buffer: k_jpeg_coef stores every block whole)
    // The batch runs as `parts` independent slices alternating over two streams: the tail of the synchronisation (a few lanes per image
    // walking their subsequences again, the rest of the chip idle) of one slice overlaps the throughput-bound kernels of the other.
    static const int parts_env = [] { const char *e = getenv("BEVW_JPEG_PARTS"); return e ? atoi(e) : 0; }();
    const size_t parts = parts_env > 0 ? std::min<size_t>((size_t)parts_env, n) : (n >= 32 ? 2 : 1);   // measured: 2 slices -6 %, 4 and more lose (launches too small)
    const bool aligned = (uintptr_t)d_out % 4 == 0 && image_stride_bytes % 4 == 0 && row_pitch_bytes % 4 == 0;
    static const int streams_env = [] { const char *e = getenv("BEVW_JPEG_STREAMS"); return e ? atoi(e) : 0; }();
    const size_t nstreams = std::min<size_t>(parts, streams_env > 0 ? (size_t)std::min(streams_env, 4) : 2);
    hipStream_t lanes[4] = {j->st, j->st2, nullptr, nullptr};
    for (size_t i = 2; i < nstreams; ++i) {
        if (!j->st_more[i - 2]) {
            HIP_TRY(hipStreamCreateWithFlags(&j->st_more[i - 2], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&j->ev_more[i - 2], hipEventDisableTiming));
        }
        lanes[i] = j->st_more[i - 2];
    }
    if (parts > 1) {
        HIP_TRY(hipEventRecord(j->ev_a, j->st));          // the staging copies were enqueued on st
        for (size_t i = 1; i < nstreams; ++i) HIP_TRY(hipStreamWaitEvent(lanes[i], j->ev_a, 0));
    }
    for (size_t part = 0; part < parts; ++part) {
        const size_t first = n * part / parts, m = n * (part + 1) / parts - first;
        if (!m) continue;
        hipStream_t st = lanes[part % nstreams];
        const jpg::ImageDesc *im = img + first;
        {
            const dim3 gc(j->max_chunk, (unsigned)m);
            uint32_t *term = j->d_term.as<uint32_t>() + first, *nrst = j->d_nrst.as<uint32_t>() + first;
            jpg::k_jpeg_find_end<<<gc, 256, 0, st>>>(imgw + first, raw, term);
            BEVW_TRY(launch_check("k_jpeg_find_end"));
            jpg::k_jpeg_count_raw<<<gc, 256, 0, st>>>(imgw + first, raw, term, j->d_chunk_keep.as<uint32_t>(), j->d_chunk_rst.as<uint32_t>());
            BEVW_TRY(launch_check("k_jpeg_count_raw"));
            jpg::k_jpeg_unstuff<<<gc, 256, 0, st>>>(imgw + first, raw, term, j->d_chunk_keep.as<uint32_t>(), j->d_chunk_rst.as<uint32_t>(),
                                                     j->d_stream.as<uint8_t>(), j->d_seg_byte.as<uint32_t>(), nrst);
            BEVW_TRY(launch_check("k_jpeg_unstuff"));
            jpg::k_jpeg_subs<<<(unsigned)m, 256, 0, st>>>(imgw + first, nrst, j->d_seg_byte.as<uint32_t>(), j->d_seg_sub.as<uint32_t>());
            BEVW_TRY(launch_check("k_jpeg_subs"));
        }
        int16_t *coef = j->d_coef.as<int16_t>() + first * (size_t)G.nblk * 64;
        uint8_t *planes = j->d_planes.as<uint8_t>() + first * (size_t)G.plane_bytes;
        if (j->max_sub) {
            const dim3 gs((j->max_sub + 255) / 256, (unsigned)m);
            jpg::k_jpeg_columns<<<gs, 256, 0, st>>>(im, stream, j->d_seg_byte.as<uint32_t>(), j->d_seg_sub.as<uint32_t>(), A);
            BEVW_TRY(launch_check("k_jpeg_columns"));
            jpg::k_jpeg_sync0<<<gs, 256, 0, st>>>(im, stream, tabs, G, A);
            BEVW_TRY(launch_check("k_jpeg_sync0"));
            // two whole-batch rounds (ping-pong of the exit states, back in d_exit afterwards), then the per-image fixed point
            jpg::k_jpeg_sync_round<<<gs, 256, 0, st>>>(im, stream, tabs, G, A, j->d_exit.as<uint64_t>(), j->d_exit2.as<uint64_t>());
            BEVW_TRY(launch_check("k_jpeg_sync_round"));
            // round 2 walks a third of the subsequences: listed per image and walked densely packed (bevw_jpeg_codec.h: k_jpeg_mark)
            uint32_t *cnt = j->d_count.as<uint32_t>() + first;
            HIP_TRY(hipMemsetAsync(cnt, 0, m * 4, st));
            jpg::k_jpeg_mark<<<gs, 256, 0, st>>>(im, A, j->d_exit2.as<uint64_t>(), j->d_exit.as<uint64_t>(), j->d_list.as<uint32_t>(), cnt);
            BEVW_TRY(launch_check("k_jpeg_mark"));
            jpg::k_jpeg_sync_list<<<gs, 256, 0, st>>>(im, stream, tabs, G, A, j->d_exit2.as<uint64_t>(), j->d_exit.as<uint64_t>(), j->d_list.as<uint32_t>(), cnt);
            BEVW_TRY(launch_check("k_jpeg_sync_list"));
            jpg::k_jpeg_sync<<<(unsigned)m, jpg::kSyncThreads, 0, st>>>(im, stream, tabs, G, A, j->d_rounds.as<uint32_t>() + first);
            BEVW_TRY(launch_check("k_jpeg_sync"));
            jpg::k_jpeg_coef<<<gs, 256, 0, st>>>(im, stream, tabs, G, A, coef);
            BEVW_TRY(launch_check("k_jpeg_coef"));
        }
        uint8_t *dst = (uint8_t *)d_out + first * image_stride_bytes;
        static const int fuse_env = [] { const char *e = getenv("BEVW_JPEG_FUSE"); return e ? atoi(e) : 1; }();
        if (fuse_env && aligned && G.nc == 3 && G.hs == 2 && G.vs == 2 && G.dw > 2 && G.w % 8 == 0) {
            // the camera case: chroma blocks -> sample planes, then luma inverse DCT + colour conversion in one kernel (no luma plane in memory)
            const int nchroma = G.nblk - G.blk_off[1];
            jpg::k_jpeg_idct<<<dim3((nchroma + 31) / 32, (unsigned)m), 256, 0, st>>>(im, G, coef, j->d_quant.as<uint16_t>(), planes, G.blk_off[1]);
            BEVW_TRY(launch_check("k_jpeg_idct"));
            jpg::k_jpeg_idct_color_h2v2<<<dim3((G.mcux + 7) / 8, G.mcuy, (unsigned)m), 256, 0, st>>>(im, G, coef, j->d_quant.as<uint16_t>(), planes, dst,
                                                                                                     image_stride_bytes, row_pitch_bytes);
            BEVW_TRY(launch_check("k_jpeg_idct_color_h2v2"));
            j->luma_planes = false;
            continue;
        }
        jpg::k_jpeg_idct<<<dim3((G.nblk + 31) / 32, (unsigned)m), 256, 0, st>>>(im, G, coef, j->d_quant.as<uint16_t>(), planes, 0);
        BEVW_TRY(launch_check("k_jpeg_idct"));
        j->luma_planes = true;
        if (aligned && G.nc == 3 && G.hs == 2 && G.vs == 2 && G.dw > 2) {
            jpg::k_jpeg_color_h2v2<<<dim3(((G.w + 7) / 8 + 63) / 64, (G.h + 3) / 4, (unsigned)m), dim3(64, 4), 0, st>>>(G, planes, dst, image_stride_bytes,
                                                                                                                           row_pitch_bytes);
            BEVW_TRY(launch_check("k_jpeg_color_h2v2"));
        } else {
            jpg::k_jpeg_color<<<dim3(((G.w + 3) / 4 + 63) / 64, (G.h + 3) / 4, (unsigned)m), dim3(64, 4), 0, st>>>(G, planes, dst, image_stride_bytes,
                                                                                                                      row_pitch_bytes);
            BEVW_TRY(launch_check("k_jpeg_color"));
        }
    }
    if (parts > 1) {   // everything the caller enqueues on st afterwards (and bevw_jpeg_sync) sees the whole batch
        HIP_TRY(hipEventRecord(j->ev_b, j->st2));
        HIP_TRY(hipStreamWaitEvent(j->st, j->ev_b, 0));
        for (size_t i = 2; i < nstreams; ++i) {
            HIP_TRY(hipEventRecord(j->ev_more[i - 2], lanes[i]));
            HIP_TRY(hipStreamWaitEvent(j->st, j->ev_more[i - 2], 0));
        }
    }
    if (oriented) {
        jpg::k_jpeg_orient<<<dim3((ow + 255) / 256, (unsigned)oh, (unsigned)j->n), 256, 0, j->st>>>(j->d_turn.as<uint8_t>(), G.w, G.h, j->orientation,
                                                                                                     (uint8_t *)d_final, final_stride, final_pitch);
        BEVW_TRY(launch_check("k_jpeg_orient"));
    }
    j->decoded = true;
    return BEVW_OK;
}

int bevw_jpeg_decode(bevw_jpeg *j, const uint8_t *const *data, const size_t *len, int n, uint8_t *out)
{
    if (!out) return fail(BEVW_E_INVALID, "bevw_jpeg_decode: null out");
    BEVW_TRY(bevw_jpeg_decode_stage(j, data, len, n));
    const size_t image = (size_t)j->G.w * j->G.h * 3;
    BEVW_TRY(j->d_img.reserve(image * (size_t)n));
    BEVW_TRY(bevw_jpeg_decode_run_device(j, j->d_img.p, image, (size_t)(j->orientation >= 5 ? j->G.h : j->G.w) * 3));
    HIP_TRY(hipMemcpyAsync(out, j->d_img.p, image * (size_t)n, hipMemcpyDeviceToHost, j->st));
    HIP_TRY(hipStreamSynchronize(j->st));
    int64_t info[8];
    BEVW_TRY(bevw_jpeg_decode_info(j, info));
    if (info[6])
        return fail(BEVW_E_INVALID, "%lld of the %d files end before their image is complete (truncated / corrupt entropy-coded data)", (long long)info[6], n);
    return BEVW_OK;
}

int bevw_jpeg_decode_info(bevw_jpeg *j, int64_t info[8])
{
    if (!j || !info) return fail(BEVW_E_INVALID, "bevw_jpeg_decode_info: null argument");
    if (!j->staged) return fail(BEVW_E_INVALID, "nothing staged");
    BEVW_TRY(use_device(j->device));
    int64_t rounds = 0, short_images = 0;
    if (j->decoded && j->max_sub) {
        std::vector<uint32_t> r((size_t)j->n);
        HIP_TRY(hipMemcpyAsync(r.data(), j->d_rounds.p, r.size() * 4, hipMemcpyDeviceToHost, j->st));
        HIP_TRY(hipStreamSynchronize(j->st));
        for (uint32_t v : r) {
            rounds = std::max<int64_t>(rounds, v & 0x7fffffffu);
            short_images += v >> 31;
        }
    }
    std::vector<jpg::ImageDesc> desc((size_t)j->n);   // stream_bytes / nsub are written by the un-stuffing kernels
    HIP_TRY(hipMemcpyAsync(desc.data(), j->d_desc.p, desc.size() * sizeof(jpg::ImageDesc), hipMemcpyDeviceToHost, j->st));
    HIP_TRY(hipStreamSynchronize(j->st));
    size_t stream_bytes = 0, subs = 0;
    for (const jpg::ImageDesc &D : desc) { stream_bytes += D.stream_bytes; subs += D.nsub; if (D.error && !(j->decoded && j->max_sub)) ++short_images; }
    info[0] = j->n; info[1] = j->orientation >= 5 ? j->G.h : j->G.w; info[2] = j->orientation >= 5 ? j->G.w : j->G.h;   // (the size cv2.imread returns)
    info[3] = (int64_t)subs; info[4] = rounds; info[5] = (int64_t)stream_bytes;
    info[6] = short_images; info[7] = (int64_t)j->h_tabs.size();
    return BEVW_OK;
}

int bevw_jpeg_get_planes(bevw_jpeg *j, int index, uint8_t *planes)
{
    if (!j || !planes || !j->decoded || index < 0 || index >= j->n) return fail(BEVW_E_INVALID, "bevw_jpeg_get_planes: nothing decoded / bad index");
    BEVW_TRY(use_device(j->device));
    if (!j->luma_planes) {   // the fused path keeps no luma plane: transform this image's blocks once more (the coefficients are still there)
        jpg::k_jpeg_idct<<<dim3((j->G.nblk + 31) / 32, 1), 256, 0, j->st>>>(j->d_desc.as<jpg::ImageDesc>() + index, j->G,
                                                                              j->d_coef.as<int16_t>() + (size_t)index * j->G.nblk * 64, j->d_quant.as<uint16_t>(),
                                                                              j->d_planes.as<uint8_t>() + (size_t)index * j->G.plane_bytes, 0);
        BEVW_TRY(launch_check("k_jpeg_idct"));
    }
    HIP_TRY(hipMemcpyAsync(planes, j->d_planes.as<uint8_t>() + (size_t)index * j->G.plane_bytes, (size_t)j->G.plane_bytes, hipMemcpyDeviceToHost, j->st));
    HIP_TRY(hipStreamSynchronize(j->st));
    return BEVW_OK;
}

int bevw_jpeg_encode_bound(int width, int height, int sampling, size_t *bound)
{
    const int hs = sampling >> 4, vs = sampling & 15;
    if (!bound || width <= 0 || height <= 0 || width > 65500 || height > 65500 || !((hs == 1 && vs == 1) || (hs == 2 && vs == 1) || (hs == 2 && vs == 2)))
        return fail(BEVW_E_INVALID, "bevw_jpeg_encode_bound: bad size / sampling (0x11, 0x21, 0x22)");
    const jpg::Geom G = jpg::make_geom(width, height, 3, hs, vs);
    *bound = 1024 + (size_t)G.nblk * 209 * 2;   // header + every block at its longest, every byte stuffed
    return BEVW_OK;
}

int bevw_jpeg_encode_run_device(bevw_jpeg *j, const void *d_bgr, int n, int width, int height, size_t image_stride_bytes, size_t row_pitch_bytes,
                                int quality, int sampling)
{
    if (!j || !d_bgr || n <= 0 || n > 65535) return fail(BEVW_E_INVALID, "bevw_jpeg_encode_run_device: bad argument (1 <= n <= 65535)");
    size_t bound = 0;
    BEVW_TRY(bevw_jpeg_encode_bound(width, height, sampling, &bound));
    if (quality < 1 || quality > 100) return fail(BEVW_E_INVALID, "JPEG quality %d outside 1..100", quality);
    if (row_pitch_bytes < (size_t)width * 3 || image_stride_bytes < row_pitch_bytes * (size_t)height)
        return fail(BEVW_E_INVALID, "input layout (pitch %zu, stride %zu) too small for %dx%d BGR", row_pitch_bytes, image_stride_bytes, width, height);
    BEVW_TRY(use_device(j->device));
    j->encoded = j->sizes_valid = false;
    const jpg::Geom G = jpg::make_geom(width, height, 3, sampling >> 4, sampling & 15);
    if (quality != j->e_quality || sampling != j->e_sampling || width != j->EG.w || height != j->EG.h) {
        HIP_TRY(hipStreamSynchronize(j->st));
        jpg::make_enc_tables(quality, j->etabs);
        j->header = jpg::make_file_header(width, height, G.hs, G.vs, j->etabs);
        BEVW_TRY(j->d_etabs.reserve(sizeof(jpg::EncTables)));
        BEVW_TRY(j->d_header.reserve(j->header.size()));
        HIP_TRY(hipMemcpyAsync(j->d_etabs.p, &j->etabs, sizeof(jpg::EncTables), hipMemcpyHostToDevice, j->st));
        HIP_TRY(hipMemcpyAsync(j->d_header.p, j->header.data(), j->header.size(), hipMemcpyHostToDevice, j->st));
        j->e_quality = quality;
        j->e_sampling = sampling;
    }
    j->EG = G;
    j->en = n;
    const size_t N = (size_t)n;
    j->buf_words = ((size_t)G.nblk * 209 + 3) / 4 + 4;
    j->file_cap = (j->header.size() + j->buf_words * 8 + 16 + 15) & ~(size_t)15;   // every byte stuffed + EOI: cannot overflow
    BEVW_TRY(j->d_eplanes.reserve(N * (size_t)G.plane_bytes));
    BEVW_TRY(j->d_zz.reserve(N * (size_t)G.nblk * 128));
    BEVW_TRY(j->d_bitlen.reserve(N * (size_t)G.nblk * 4));
    BEVW_TRY(j->d_bitbuf.reserve(N * j->buf_words * 4));
    BEVW_TRY(j->d_totals.reserve(N * 8));
    BEVW_TRY(j->d_files.reserve(N * j->file_cap));
    BEVW_TRY(j->d_sizes.reserve(N * 4));
    const jpg::EncTables *tabs = j->d_etabs.as<jpg::EncTables>();
    if (G.hs == 2 && G.vs == 2 && image_stride_bytes % 4 == 0)   // the camera case: 8 x 2 pixels per lane, dword loads and stores
        jpg::k_jenc_ycc_h2v2<<<dim3((G.wb[1] * 2 + 63) / 64, (G.hb[1] * 8 + 3) / 4, (unsigned)n), dim3(64, 4), 0, j->st>>>(
            G, (const uint8_t *)d_bgr, image_stride_bytes, row_pitch_bytes, j->d_eplanes.as<uint8_t>());
    else
        jpg::k_jenc_ycc<<<dim3((G.wb[1] * 8 + 63) / 64, (G.hb[1] * 8 + 3) / 4, (unsigned)n), dim3(64, 4), 0, j->st>>>(
            G, (const uint8_t *)d_bgr, image_stride_bytes, row_pitch_bytes, j->d_eplanes.as<uint8_t>());
    BEVW_TRY(launch_check("k_jenc_ycc"));
    BEVW_TRY(j->d_acbits.reserve(N * (size_t)G.nblk * 2));
    BEVW_TRY(j->d_dcq.reserve(N * (size_t)G.nblk * 2));
    BEVW_TRY(j->d_nzmask.reserve(N * (size_t)G.nblk * 8));
    const uint32_t nchunk = (uint32_t)((j->buf_words * 4 + jpg::kStuffChunk - 1) / jpg::kStuffChunk);
    BEVW_TRY(j->d_chunk_ff.reserve(N * nchunk * 4));
    jpg::k_jenc_fdct<<<dim3((G.nblk + 31) / 32, (unsigned)n), 256, 0, j->st>>>(G, j->d_eplanes.as<uint8_t>(), tabs, j->d_zz.as<int16_t>(),
                                                                                 j->d_acbits.as<uint16_t>(), j->d_dcq.as<int16_t>(), j->d_nzmask.as<uint64_t>());
    BEVW_TRY(launch_check("k_jenc_fdct"));
    jpg::k_jenc_scan<<<(unsigned)n, jpg::kSyncThreads, 0, j->st>>>(G, j->d_acbits.as<uint16_t>(), j->d_dcq.as<int16_t>(), tabs, j->d_bitlen.as<uint32_t>(),
                                                                    j->d_bitbuf.as<uint32_t>(), j->buf_words, j->d_totals.as<uint32_t>());
    BEVW_TRY(launch_check("k_jenc_scan"));
    jpg::k_jenc_bits<<<dim3((G.nblk + 255) / 256, (unsigned)n), 256, 0, j->st>>>(G, j->d_zz.as<int16_t>(), j->d_dcq.as<int16_t>(), tabs,
                                                                                   j->d_bitlen.as<uint32_t>(), j->d_bitbuf.as<uint32_t>(), j->buf_words, j->d_nzmask.as<uint64_t>());
    BEVW_TRY(launch_check("k_jenc_bits"));
    jpg::k_jenc_ffcount<<<dim3(nchunk, (unsigned)n), 256, 0, j->st>>>(j->d_bitbuf.as<uint32_t>(), j->buf_words, j->d_totals.as<uint32_t>(),
                                                                        j->d_chunk_ff.as<uint32_t>(), nchunk);
    BEVW_TRY(launch_check("k_jenc_ffcount"));
    jpg::k_jenc_stuff<<<dim3(nchunk, (unsigned)n), 256, 0, j->st>>>(j->d_bitbuf.as<uint32_t>(), j->buf_words, j->d_totals.as<uint32_t>(),
                                                                      j->d_chunk_ff.as<uint32_t>(), nchunk, j->d_header.as<uint8_t>(),
                                                                      (uint32_t)j->header.size(), j->d_files.as<uint8_t>(), j->file_cap,
                                                                      j->d_sizes.as<uint32_t>());
    BEVW_TRY(launch_check("k_jenc_stuff"));
    j->encoded = true;
    return BEVW_OK;
}

int bevw_jpeg_encoded_sizes(bevw_jpeg *j, size_t *sizes)
{
    if (!j || !sizes || !j->encoded) return fail(BEVW_E_INVALID, "bevw_jpeg_encoded_sizes: nothing encoded");
    BEVW_TRY(use_device(j->device));
    if (!j->sizes_valid) {
        j->sizes.assign((size_t)j->en, 0);
        HIP_TRY(hipMemcpyAsync(j->sizes.data(), j->d_sizes.p, (size_t)j->en * 4, hipMemcpyDeviceToHost, j->st));
        HIP_TRY(hipStreamSynchronize(j->st));
        j->sizes_valid = true;
    }
    for (int i = 0; i < j->en; ++i) {
        if (!j->sizes[i]) return fail(BEVW_E_HIP, "image %d overflowed its file buffer (internal error)", i);
        sizes[i] = j->sizes[i];
    }
    return BEVW_OK;
}

int bevw_jpeg_encoded_copy(bevw_jpeg *j, int index, uint8_t *dst, size_t cap)
{
    if (!j || !dst || !j->encoded || index < 0 || index >= j->en) return fail(BEVW_E_INVALID, "bevw_jpeg_encoded_copy: nothing encoded / bad index");
    if (!j->sizes_valid) {
        std::vector<size_t> tmp((size_t)j->en);
        BEVW_TRY(bevw_jpeg_encoded_sizes(j, tmp.data()));
    }
    if (cap < j->sizes[index]) return fail(BEVW_E_INVALID, "file %d needs %u bytes, %zu given", index, j->sizes[index], cap);
    BEVW_TRY(use_device(j->device));
    HIP_TRY(hipMemcpyAsync(dst, j->d_files.as<uint8_t>() + (size_t)index * j->file_cap, j->sizes[index], hipMemcpyDeviceToHost, j->st));
    HIP_TRY(hipStreamSynchronize(j->st));
    return BEVW_OK;
}

int bevw_jpeg_encode(bevw_jpeg *j, const uint8_t *bgr, int n, int width, int height, int quality, int sampling, uint8_t *out, size_t cap_each,
                     size_t *sizes)
{
    if (!j || !bgr || !out || !sizes || n <= 0) return fail(BEVW_E_INVALID, "bevw_jpeg_encode: bad argument");
    BEVW_TRY(use_device(j->device));
    const size_t image = (size_t)width * height * 3;
    BEVW_TRY(j->d_src.reserve(image * (size_t)n));
    HIP_TRY(hipMemcpyAsync(j->d_src.p, bgr, image * (size_t)n, hipMemcpyHostToDevice, j->st));
    BEVW_TRY(bevw_jpeg_encode_run_device(j, j->d_src.p, n, width, height, image, (size_t)width * 3, quality, sampling));
    BEVW_TRY(bevw_jpeg_encoded_sizes(j, sizes));
    for (int i = 0; i < n; ++i) BEVW_TRY(bevw_jpeg_encoded_copy(j, i, out + (size_t)i * cap_each, cap_each));
    return BEVW_OK;
}

// every file of the last encode, back to back: a device-side gather + ONE device-to-host copy (bevw_jpeg_encoded_copy costs a
// synchronising copy per file -- 64 round trips per batch of the files-in / file-out pipeline)
namespace bevw { namespace jpg {
static __global__ void k_jenc_pack(const uint8_t *__restrict__ files, size_t file_cap, const uint32_t *__restrict__ sizes,
                                   const unsigned long long *__restrict__ offsets, uint8_t *__restrict__ packed)
{
    const uint32_t size = sizes[blockIdx.y];
    const uint8_t *src = files + (size_t)blockIdx.y * file_cap;
    uint8_t *dst = packed + offsets[blockIdx.y];
    for (uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 16u; i < size; i += gridDim.x * 256u * 16u) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + i);   // file slots are 16-byte aligned and padded
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        const uint32_t n = size - i < 16u ? size - i : 16u;
        for (uint32_t k = 0; k < n; ++k) dst[i + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
    }
}
} }

int bevw_jpeg_encoded_fetch(bevw_jpeg *j, uint8_t *dst, size_t cap, size_t *offsets)
{
    if (!j || !dst || !offsets || !j->encoded) return fail(BEVW_E_INVALID, "bevw_jpeg_encoded_fetch: nothing encoded / null argument");
    std::vector<size_t> sz((size_t)j->en);
    BEVW_TRY(bevw_jpeg_encoded_sizes(j, sz.data()));
    std::vector<unsigned long long> off((size_t)j->en + 1, 0);
    for (int i = 0; i < j->en; ++i) off[(size_t)i + 1] = off[(size_t)i] + sz[(size_t)i];
    for (int i = 0; i <= j->en; ++i) offsets[i] = (size_t)off[(size_t)i];
    const size_t total = (size_t)off[(size_t)j->en];
    if (cap < total) return fail(BEVW_E_INVALID, "the %d files need %zu bytes, %zu given", j->en, total, cap);
    BEVW_TRY(use_device(j->device));
    BEVW_TRY(j->d_packed.reserve(total + 16));
    BEVW_TRY(j->d_offsets.reserve(off.size() * 8));
    HIP_TRY(hipMemcpyAsync(j->d_offsets.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, j->st));
    jpg::k_jenc_pack<<<dim3(16, (unsigned)j->en), 256, 0, j->st>>>(j->d_files.as<uint8_t>(), j->file_cap, j->d_sizes.as<uint32_t>(),
                                                                    j->d_offsets.as<unsigned long long>(), j->d_packed.as<uint8_t>());
    BEVW_TRY(launch_check("k_jenc_pack"));
    HIP_TRY(hipMemcpyAsync(dst, j->d_packed.p, total, hipMemcpyDeviceToHost, j->st));
    HIP_TRY(hipStreamSynchronize(j->st));
    return BEVW_OK;
}

// Ordering between a codec context's stream and an engine's stream WITHOUT the host: decode -> stitch -> encode as one chain.
//   bevw_jpeg_wait_engine: what is enqueued on the codec afterwards starts when everything enqueued on the engine so far is done;
//   bevw_wait_jpeg:        the engine waits for the codec (e.g. bevw_run_device behind bevw_jpeg_decode_run_device).
int bevw_jpeg_wait_engine(bevw_jpeg *j, bevw_handle *h)
{
    if (!j || !h) return fail(BEVW_E_INVALID, "bevw_jpeg_wait_engine: null argument");
    if (bevw_internal_handle_device(h) != j->device) return fail(BEVW_E_INVALID, "codec context and engine live on different devices");
    BEVW_TRY(use_device(j->device));
    HIP_TRY(hipEventRecord(j->ev_x, bevw_internal_handle_stream(h)));
    HIP_TRY(hipStreamWaitEvent(j->st, j->ev_x, 0));
    return BEVW_OK;
}

int bevw_wait_jpeg(bevw_handle *h, bevw_jpeg *j)
{
    if (!j || !h) return fail(BEVW_E_INVALID, "bevw_wait_jpeg: null argument");
    if (bevw_internal_handle_device(h) != j->device) return fail(BEVW_E_INVALID, "codec context and engine live on different devices");
    BEVW_TRY(use_device(j->device));
    HIP_TRY(hipEventRecord(j->ev_x, j->st));
    HIP_TRY(hipStreamWaitEvent(bevw_internal_handle_stream(h), j->ev_x, 0));
    return BEVW_OK;
}

int bevw_jpeg_sync(bevw_jpeg *j)
{
    if (!j) return fail(BEVW_E_INVALID, "null jpeg context");
    BEVW_TRY(use_device(j->device));
    HIP_TRY(hipStreamSynchronize(j->st));
    return BEVW_OK;
}

int bevw_jpeg_timer_mark(bevw_jpeg *j, int slot)
{
    if (!j) return fail(BEVW_E_INVALID, "null jpeg context");
    BEVW_TRY(use_device(j->device));
    return j->timer.mark(slot, j->st);
}

int bevw_jpeg_timer_between(bevw_jpeg *j, int slot_a, int slot_b, float *elapsed_ms)
{
    if (!j) return fail(BEVW_E_INVALID, "null jpeg context");
    BEVW_TRY(use_device(j->device));
    return j->timer.between(slot_a, slot_b, elapsed_ms);
}
