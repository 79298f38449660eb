// bevw_jpeg_codec.h -- kernels and host orchestration of the JPEG decode / encode stages (row f4); the per-lane arithmetic is in
// bevw_jpeg.h.  Included by bevwarp.hip after its error / buffer helpers (fail, HIP_TRY, BEVW_TRY, DevBuf, LapTimer, launch_check).
#pragma once
#include "bevw_jpeg.h"
#include "bevw_jpeg_walk.h"

namespace bevw {
namespace jpg {

#ifndef BEVW_JPEG_TAIL
#define BEVW_JPEG_TAIL 16
#endif
constexpr int kSyncTail = BEVW_JPEG_TAIL;   // k_jpeg_sync: rounds that list at most this many subsequences (<= 64) are walked on the scalar unit
static_assert(kSyncTail <= 64, "k_jpeg_sync keeps the stretch starts of a round one per lane");
constexpr int kSyncThreads = 1024;   // one block per image in the per-image kernels (fixed point, prefix sums, stuffing)

template <typename T> __device__ __forceinline__ void lds_copy(T *dst, const T *__restrict__ src)
{
    static_assert(sizeof(T) % 16 == 0, "16-byte pieces");
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    for (unsigned i = threadIdx.x + threadIdx.y * blockDim.x; i < sizeof(T) / 16; i += blockDim.x * blockDim.y) d[i] = s[i];
}

// Exclusive prefix over one value per thread of a 1-D block of whole waves, done the plain way in two levels: partials to LDS, lane 0 of every
// wave walks its wave's 64, thread 0 walks the waves' totals.  (One thread walking all 1024 took ~40 us of the one-block-per-image kernels:
// a lone lane gets one LDS round trip per step.)  lds: blockDim.x + blockDim.x / 64 + 1 values.
template <typename V> __device__ __forceinline__ V block_exscan_serial(V v, V *lds, V &total)
{
    const unsigned T = blockDim.x, W = T >> 6, w = threadIdx.x >> 6;
    lds[threadIdx.x] = v;
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) {
        V run = V();
        for (unsigned i = w * 64; i < w * 64 + 64; ++i) { const V t = lds[i]; lds[i] = run; run = run + t; }
        lds[T + w] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        V run = V();
        for (unsigned i = 0; i < W; ++i) { const V t = lds[T + i]; lds[T + i] = run; run = run + t; }
        lds[T + W] = run;
    }
    __syncthreads();
    const V r = lds[threadIdx.x] + lds[T + w];
    total = lds[T + W];
    __syncthreads();
    return r;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}
// sum over a 256-thread block (every thread gets it); lds: 4 words
__device__ __forceinline__ uint32_t block256_sum(uint32_t v, uint32_t *lds)
{
    const uint32_t w = wave_sum(v);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = w;
    __syncthreads();
    const uint32_t t = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return t;
}

// ---- un-stuffing ------------------------------------------------------------------------------------------------------------------
// The host copies the entropy-coded bytes of every file into its slot as they are.  Here: find where the data ends (the first 0xFF that is
// followed by anything but 0x00 / 0xFF / RSTn), drop the stuffed zeros, the RSTn markers and fill bytes, record where every restart
// segment starts, cut the segments into subsequences.  Chunks of kRawChunk bytes, 16 bytes per lane; counts per chunk, then a prefix.
constexpr uint32_t kRawChunk = 4096;

__device__ __forceinline__ RawBytes raw_load16(const uint8_t *__restrict__ raw, uint32_t pos, uint32_t raw_bytes)
{
    RawBytes R;
    const uint4 v = *reinterpret_cast<const uint4 *>(raw + pos);   // slots are 16-byte aligned and padded
    R.w[0] = v.x; R.w[1] = v.y; R.w[2] = v.z; R.w[3] = v.w;
    R.prev = pos ? raw[pos - 1] : 0u;
    R.next = pos + 16u < raw_bytes ? raw[pos + 16u] : 0xD9u;      // the end of the buffer closes the data like a marker would
    return R;
}

__global__ __launch_bounds__(256) void k_jpeg_find_end(const ImageDesc *__restrict__ img, const uint8_t *__restrict__ raw, uint32_t *__restrict__ term)
{
    const ImageDesc D = img[blockIdx.y];
    const uint32_t pos = blockIdx.x * kRawChunk + threadIdx.x * 16u;
    if (pos >= D.raw_bytes) return;
    const uint32_t t = raw_first_terminator(raw_load16(raw + (size_t)D.stream_word * 4, pos, D.raw_bytes), pos, D.raw_bytes);
    if (t != 0xffffffffu) atomicMin(&term[blockIdx.y], t);
}

__global__ __launch_bounds__(256) void k_jpeg_count_raw(const ImageDesc *__restrict__ img, const uint8_t *__restrict__ raw, const uint32_t *__restrict__ term,
                                                        uint32_t *__restrict__ chunk_keep, uint32_t *__restrict__ chunk_rst)
{
    __shared__ uint32_t lds[4];
    const ImageDesc D = img[blockIdx.y];
    const uint32_t end = term[blockIdx.y];
    if (blockIdx.x * kRawChunk >= D.raw_bytes) return;
    const uint32_t pos = blockIdx.x * kRawChunk + threadIdx.x * 16u;
    uint32_t keep = 0, rst = 0;
    if (pos < end) raw_classify(raw_load16(raw + (size_t)D.stream_word * 4, pos, D.raw_bytes), pos, end, D.raw_bytes, keep, rst);
    const uint32_t nk = block256_sum(__popc(keep), lds), nr = block256_sum(__popc(rst), lds);
    if (threadIdx.x == 0) { chunk_keep[D.chunk_first + blockIdx.x] = nk; chunk_rst[D.chunk_first + blockIdx.x] = nr; }
}

__global__ __launch_bounds__(256) void k_jpeg_unstuff(ImageDesc *__restrict__ img, const uint8_t *__restrict__ raw, const uint32_t *__restrict__ term,
                                                      const uint32_t *__restrict__ chunk_keep, const uint32_t *__restrict__ chunk_rst,
                                                      uint8_t *__restrict__ stream, uint32_t *__restrict__ seg_byte, uint32_t *__restrict__ nrst_out)
{
    __shared__ uint32_t lds[4], lds2[4];
    const ImageDesc D = img[blockIdx.y];
    if (blockIdx.x * kRawChunk >= D.raw_bytes) return;
    const uint32_t end = term[blockIdx.y];
    uint32_t kb = 0, rb = 0;
    for (uint32_t i = threadIdx.x; i < blockIdx.x; i += 256u) { kb += chunk_keep[D.chunk_first + i]; rb += chunk_rst[D.chunk_first + i]; }
    kb = block256_sum(kb, lds);
    rb = block256_sum(rb, lds);
    const uint32_t pos = blockIdx.x * kRawChunk + threadIdx.x * 16u;
    uint32_t keep = 0, rst = 0;
    RawBytes R;
    R.w[0] = R.w[1] = R.w[2] = R.w[3] = R.prev = R.next = 0;
    if (pos < end) { R = raw_load16(raw + (size_t)D.stream_word * 4, pos, D.raw_bytes); raw_classify(R, pos, end, D.raw_bytes, keep, rst); }
    const uint32_t nk = __popc(keep), nr = __popc(rst);
    const uint32_t ik = wave_incl_scan(nk), ir = wave_incl_scan(nr);
    if ((threadIdx.x & 63) == 63) { lds[threadIdx.x >> 6] = ik; lds2[threadIdx.x >> 6] = ir; }
    __syncthreads();
    uint32_t ok = kb + ik - nk, orr = rb + ir - nr;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) { ok += lds[w]; orr += lds2[w]; }
    const uint32_t chunk_k = lds[0] + lds[1] + lds[2] + lds[3], chunk_r = lds2[0] + lds2[1] + lds2[2] + lds2[3];
    uint8_t *S = stream + (size_t)D.stream_word * 4;
    uint32_t *SB = seg_byte + D.seg_first;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (rst & (1u << i)) {   // the segment that starts behind this marker
            ++orr;
            if (orr < D.nseg) SB[orr] = ok;   // more markers than DRI allows: counted, not stored (k_jpeg_subs flags the image)
        }
        if (keep & (1u << i)) S[ok++] = (uint8_t)R.at(i);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) SB[0] = 0;
    if (threadIdx.x == 0 && (blockIdx.x + 1) * kRawChunk >= D.raw_bytes) {   // the last chunk closes the image
        const uint32_t total = kb + chunk_k, nrst = rb + chunk_r;
        nrst_out[blockIdx.y] = nrst;
        img[blockIdx.y].stream_bytes = total;
        if (nrst + 1u <= D.nseg) SB[nrst + 1u] = total;
        for (int i = 0; i < 16; ++i) S[total + i] = 0;   // the bit readers look a few bytes past the end
    }
}

// One block per image: the restart markers found against DRI, the segments cut into subsequences.
__global__ __launch_bounds__(256) void k_jpeg_subs(ImageDesc *__restrict__ img, const uint32_t *__restrict__ nrst, const uint32_t *__restrict__ seg_byte,
                                                   uint32_t *__restrict__ seg_sub)
{
    if (threadIdx.x) return;
    ImageDesc &D = img[blockIdx.x];
    const uint32_t *SB = seg_byte + D.seg_first;
    uint32_t *SS = seg_sub + D.seg_first;
    if (nrst[blockIdx.x] + 1u != D.nseg) { D.error = 1; D.nsub = 0; return; }
    uint32_t subs = 0, empty = 0;
    for (uint32_t s = 0; s < D.nseg; ++s) {
        SS[s] = subs;
        // a restart segment holds at least one MCU = at least one byte: a segment without data (two adjacent RSTn markers, an RSTn right in
        // front of EOI, a scan that is only FF D9) owns no subsequence, so nothing downstream would ever look at its blocks -- the
        // image is truncated / corrupt (never stale coefficients of an earlier batch passed off as pixels)
        if (SB[s + 1] <= SB[s]) ++empty;
        subs += ((SB[s + 1] - SB[s]) * 8u + (uint32_t)kSubBits - 1u) / (uint32_t)kSubBits;
    }
    SS[D.nseg] = subs;
    if (empty != 0 || subs == 0) { D.error = 1; D.nsub = 0; return; }
    D.error = 0;
    D.nsub = subs;
}

// ---- decode ----------------------------------------------------------------------------------------------------------------------
struct SubArrays {
    uint64_t *entry;    // state at the first symbol of the subsequence
    uint64_t *exitst;   // state at the first symbol of the next one
    int4 *sums;         // blocks completed, DC difference sums of the three components
    int4 *base;         // exclusive prefix of `sums` inside the restart segment (x = index of the block in progress, scan order)
    uint32_t *endbit;   // last bit (exclusive) the subsequence owns
    uint32_t *meta;     // restart segment | first-of-segment << 31
    uint32_t *word0;    // stream index of the first word of the subsequence's column
    uint32_t *cols;     // the columns: word w of subsequence j of an image at cols[kColWords * sub_first + w * nsub + j]
};

__device__ __forceinline__ WordSource word_source(const SubArrays &A, const ImageDesc &D, const uint32_t *__restrict__ stream, uint32_t j)
{
    return WordSource{stream + D.stream_word, A.cols + (size_t)kColWords * D.sub_first + j, D.nsub, A.word0[(size_t)D.sub_first + j]};
}

// Lays the words of every subsequence out as a column (see WordSource) and records where it starts; also the per-subsequence constants
// the later kernels need (last bit owned, restart segment, first-of-segment flag, the guessed entry state).
__global__ __launch_bounds__(256) void k_jpeg_columns(const ImageDesc *__restrict__ img, const uint32_t *__restrict__ stream,
                                                      const uint32_t *__restrict__ seg_byte, const uint32_t *__restrict__ seg_sub, SubArrays A)
{
    const ImageDesc D = img[blockIdx.y];
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= D.nsub) return;
    const uint32_t *sb = seg_byte + D.seg_first, *ss = seg_sub + D.seg_first;
    uint32_t lo = 0, hi = D.nseg - 1;
    while (lo < hi) {   // the last segment whose first subsequence is <= j (empty segments share their successor's index)
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (ss[mid] <= j) lo = mid; else hi = mid - 1;
    }
    const uint32_t start = sb[lo] * 8u + (j - ss[lo]) * (uint32_t)kSubBits;
    uint32_t end = start + (uint32_t)kSubBits;
    if (end > sb[lo + 1] * 8u) end = sb[lo + 1] * 8u;
    const size_t slot = (size_t)D.sub_first + j;
    A.entry[slot] = pack_state(start, 0, 0);
    A.endbit[slot] = end;
    A.meta[slot] = lo | (j == ss[lo] ? 0x80000000u : 0u);
    A.word0[slot] = start >> 5;
    const uint32_t *src = stream + D.stream_word + (start >> 5);
    uint32_t *dst = A.cols + (size_t)kColWords * D.sub_first + j;
    uint32_t v[kColWords];
#pragma unroll
    for (int w = 0; w < kColWords; ++w) v[w] = src[w];   // (the stream buffer is padded: the last subsequence of a batch may read past its data)
#pragma unroll
    for (int w = 0; w < kColWords; ++w) dst[(size_t)w * D.nsub] = v[w];
}

__global__ __launch_bounds__(256) void k_jpeg_sync0(const ImageDesc *__restrict__ img, const uint32_t *__restrict__ stream,
                                                    const TableSet *__restrict__ tabs, Geom G, SubArrays A)
{
    __shared__ TableSet T;
    const ImageDesc D = img[blockIdx.y];
    lds_copy(&T, tabs + D.tables);
    __syncthreads();
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= D.nsub) return;
    const size_t slot = (size_t)D.sub_first + j;
    const SubOut R = decode_sub_lanes(word_source(A, D, stream, j), T.t, G, A.entry[slot], A.endbit[slot]);
    A.exitst[slot] = R.exit;
    A.sums[slot] = make_int4(R.cnt, R.dc0, R.dc1, R.dc2);
}

// One synchronisation round over the whole batch at full occupancy: exit states are read from `xin` (the previous round) and written to
// `xout`, so no lane sees a state of its own round.  The host runs an even number of these before the per-image kernel below, which
// then only has the stragglers left (the first re-decode touches nearly every subsequence: the guessed state is almost never the true one).
__global__ __launch_bounds__(256) void k_jpeg_sync_round(const ImageDesc *__restrict__ img, const uint32_t *__restrict__ stream,
                                                         const TableSet *__restrict__ tabs, Geom G, SubArrays A, const uint64_t *__restrict__ xin,
                                                         uint64_t *__restrict__ xout)
{
    __shared__ TableSet T;
    const ImageDesc D = img[blockIdx.y];
    lds_copy(&T, tabs + D.tables);
    __syncthreads();
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= D.nsub) return;
    const size_t slot = (size_t)D.sub_first + j;
    uint64_t out = xin[slot];
    if (!(A.meta[slot] & 0x80000000u)) {
        const uint64_t in = xin[slot - 1];
        if (in != A.entry[slot]) {
            A.entry[slot] = in;
            const SubOut R = decode_sub_lanes(word_source(A, D, stream, j), T.t, G, in, A.endbit[slot]);
            A.sums[slot] = make_int4(R.cnt, R.dc0, R.dc1, R.dc2);
            out = R.exit;
        }
    }
    xout[slot] = out;
}

// Round 2 and later touch a shrinking minority of the subsequences (1219, 387, 174, 78 ... of 1224 on the reference's camera files), but a wave
// runs for as long as its slowest lane: scattered over the image, 32 % of the lanes keep every wave busy for a full walk.  k_jpeg_mark lists,
// per image, the subsequences whose entry state is not their predecessor's exit state (and copies every exit state to the other ping-pong
// buffer, so that the listed lanes are the only ones k_jpeg_sync_list has to write); k_jpeg_sync_list walks exactly those, densely packed.
__global__ __launch_bounds__(256) void k_jpeg_mark(const ImageDesc *__restrict__ img, SubArrays A, const uint64_t *__restrict__ xin,
                                                   uint64_t *__restrict__ xout, uint32_t *__restrict__ list, uint32_t *__restrict__ count)
{
    const ImageDesc D = img[blockIdx.y];
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    const bool live = j < D.nsub;
    const size_t slot = (size_t)D.sub_first + (live ? j : 0u);
    bool need = false;
    if (live) {
        xout[slot] = xin[slot];
        need = !(A.meta[slot] & 0x80000000u) && xin[slot - 1] != A.entry[slot];
    }
    const unsigned long long m = __ballot(need);
    if (!m) return;
    const int lane = (int)(threadIdx.x & 63u);
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&count[blockIdx.y], (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, 0, 64);
    if (need) list[(size_t)D.sub_first + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = j;
}

__global__ __launch_bounds__(256) void k_jpeg_sync_list(const ImageDesc *__restrict__ img, const uint32_t *__restrict__ stream,
                                                        const TableSet *__restrict__ tabs, Geom G, SubArrays A, const uint64_t *__restrict__ xin,
                                                        uint64_t *__restrict__ xout, const uint32_t *__restrict__ list, const uint32_t *__restrict__ count)
{
    __shared__ TableSet T;
    const uint32_t n = count[blockIdx.y];
    if (blockIdx.x * 256u >= n) return;   // (uniform over the block)
    const ImageDesc D = img[blockIdx.y];
    lds_copy(&T, tabs + D.tables);
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = list[(size_t)D.sub_first + i];
    const size_t slot = (size_t)D.sub_first + j;
    const uint64_t in = xin[slot - 1];
    A.entry[slot] = in;
    const SubOut R = decode_sub_lanes(word_source(A, D, stream, j), T.t, G, in, A.endbit[slot]);
    A.sums[slot] = make_int4(R.cnt, R.dc0, R.dc1, R.dc2);
    xout[slot] = R.exit;
}

__device__ __forceinline__ int4 add4(int4 a, int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// One block per image: re-decode every subsequence whose entry state is not its predecessor's exit state until nothing changes, then
// the prefix sums.  A round in which no lane changed an exit state leaves entry[j] == exit[j - 1] for every j, and the first subsequence
// of every restart segment starts in a known state, so by induction every entry state is then the sequential decoder's.
__global__ __launch_bounds__(kSyncThreads) void k_jpeg_sync(const ImageDesc *__restrict__ img, const uint32_t *__restrict__ stream,
                                                             const TableSet *__restrict__ tabs, Geom G, SubArrays A, uint32_t *__restrict__ rounds_out)
{
    __shared__ TableSet T;
    __shared__ int4 part[kSyncThreads];
    __shared__ int s_reset[kSyncThreads];
    const ImageDesc D = img[blockIdx.x];
    lds_copy(&T, tabs + D.tables);
    __syncthreads();
    volatile uint64_t *vexit = A.exitst + D.sub_first;
    uint64_t *entry = A.entry + D.sub_first;
    uint32_t rounds = 0;
    // Every round first LISTS the subsequences that have to walk again (their entry state is not their predecessor's exit state) and then
    // walks them densely packed: a round costs as many wave-walks as its list fills waves (174, 78, 37 ... subsequences: 3, 2, 1 waves),
    // not one per wave that holds a straggler (all 16 waves of the block for as long as some lane of each has work).
    __shared__ uint32_t s_list[kSyncThreads];
    __shared__ uint32_t s_n, s_lead[64];
    for (;;) {
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        int changed = 0;
        for (uint32_t j = threadIdx.x; j < D.nsub; j += kSyncThreads) {
            if (j == 0 || (A.meta[D.sub_first + j] & 0x80000000u)) continue;
            if (vexit[j - 1] == entry[j]) continue;
            const uint32_t at = atomicAdd(&s_n, 1u);
            if (at < (uint32_t)kSyncThreads) s_list[at] = j;
            else changed = 1;   // more than one block-load of work: the rest is found again in the next round
        }
        __syncthreads();
        const uint32_t n = min(s_n, (uint32_t)kSyncThreads);
        constexpr uint32_t kWaves = kSyncThreads / 64;
        if (n <= (uint32_t)kSyncTail) {
            // The tail: a wave per listed subsequence (round-robin beyond 16), walked on the scalar unit (decode_sub_scalar) -- and the wave goes on with the
            // successor itself for as long as the successor's recorded entry state is not the exit state just computed.  An unsynchronised
            // run (the reference's right camera: 10 subsequences of ordinary texture) is ONE serial chain; a round per link adds the
            // round's listing and barriers to every link, and walks every link twice (the successor of a listed subsequence is listed too,
            // with a stale entry state).  So a listed subsequence whose predecessor is listed as well is left to the predecessor's wave,
            // and a wave stops in front of a subsequence that another wave of this round walks.
            if (threadIdx.x < n) {   // s_lead: this listed subsequence's predecessor is not listed -- it starts a stretch, a wave walks it
                uint32_t lead = 1u;
                for (uint32_t i = 0; i < n; ++i) lead = s_list[i] + 1u == s_list[threadIdx.x] ? 0u : lead;
                s_lead[threadIdx.x] = lead;
            }
            __syncthreads();
            const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
            const uint32_t lane_lead = lane < n && s_lead[lane] ? s_list[lane] : 0xffffffffu;   // the stretch starts, one per lane (n <= 64)
            for (uint32_t idx = w; idx < n; idx += kWaves) {
                if (!uni(s_lead[idx])) {
                    changed = 1;   // (left to the predecessor's wave; if that wave stops early, the next round lists it again)
                    continue;
                }
                uint32_t j = uni(s_list[idx]);
                const uint64_t first = vexit[j - 1];
                uint32_t in_lo = uni((uint32_t)first), in_hi = uni((uint32_t)(first >> 32));
                for (;;) {
                    const uint64_t in = ((uint64_t)in_hi << 32) | in_lo;
                    const SubOut R = decode_sub_scalar(word_source(A, D, stream, j), (tabs + D.tables)->t, G, in, A.endbit[D.sub_first + j]);
                    const uint32_t x_lo = uni((uint32_t)R.exit), x_hi = uni((uint32_t)(R.exit >> 32));
                    if ((threadIdx.x & 63u) == 0) {
                        entry[j] = in;
                        A.sums[D.sub_first + j] = make_int4(R.cnt, R.dc0, R.dc1, R.dc2);
                        vexit[j] = R.exit;
                    }
                    ++j;
                    if (j >= D.nsub || (uni(A.meta[D.sub_first + j]) & 0x80000000u)) break;   // nothing depends on this exit state
                    // j starts another stretch of this round: its wave read the exit state in front of it when it started -- maybe before the
                    // store above.  Whatever entry[j] says right now, only another round can tell (FIRST this test, then the look at entry[j]:
                    // the other way round a stale walk of j could pass for synchronised and end the fixed point early)
                    if (__any(lane_lead == j)) { changed = 1; break; }
                    const uint64_t next_in = entry[j];
                    if (uni((uint32_t)next_in) == x_lo && uni((uint32_t)(next_in >> 32)) == x_hi) break;   // synchronised again
                    in_lo = x_lo;
                    in_hi = x_hi;
                }
            }
        } else if (threadIdx.x < n) {
            const uint32_t j = s_list[threadIdx.x];
            const uint64_t in = vexit[j - 1];
            entry[j] = in;
            const SubOut R = decode_sub_lanes(word_source(A, D, stream, j), T.t, G, in, A.endbit[D.sub_first + j]);
            A.sums[D.sub_first + j] = make_int4(R.cnt, R.dc0, R.dc1, R.dc2);
            if (R.exit != vexit[j]) { vexit[j] = R.exit; changed = 1; }
        }
        ++rounds;
        __threadfence_block();
        if (!__syncthreads_or(changed)) break;
        if (rounds > D.nsub + 2u) break;   // cannot happen (induction); never spin
    }
    if (threadIdx.x == 0) rounds_out[blockIdx.x] = rounds | (D.error ? 0x80000000u : 0u);
    // exclusive prefix of (blocks, dc0, dc1, dc2) with a reset at the first subsequence of every restart segment:
    // thread t owns the subsequences [t L, (t + 1) L)
    const uint32_t L = (D.nsub + kSyncThreads - 1) / kSyncThreads;
    const uint32_t j0 = threadIdx.x * L, j1 = min(j0 + L, D.nsub);
    int4 run = make_int4(0, 0, 0, 0);
    int reset = 0;
    for (uint32_t j = j0; j < j1; ++j) {
        if (A.meta[D.sub_first + j] & 0x80000000u) { run = make_int4(0, 0, 0, 0); reset = 1; }
        run = add4(run, A.sums[D.sub_first + j]);
    }
    part[threadIdx.x] = run;
    s_reset[threadIdx.x] = reset;
    __syncthreads();
    // two levels (a lone lane walking all 1024 entries took a fifth of the kernel): lane 0 of every wave walks its wave's 64 entries, thread 0 the 16
    // wave totals; an entry behind a reset inside its own wave does not see what came before the wave
    __shared__ int4 wave_part[kSyncThreads / 64];
    __shared__ int wave_reset[kSyncThreads / 64];
    const uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) {
        int4 carry = make_int4(0, 0, 0, 0);
        int seen = 0;
        for (uint32_t i = wv * 64u; i < wv * 64u + 64u; ++i) {
            const int4 t = part[i];
            const int r = s_reset[i];
            part[i] = carry;                     // what thread i starts from inside its wave (void after its first reset)
            s_reset[i] = seen;                   // a reset in front of i inside the wave: part[i] is complete
            carry = r ? t : add4(carry, t);
            seen |= r;
        }
        wave_part[wv] = carry;
        wave_reset[wv] = seen;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int4 carry = make_int4(0, 0, 0, 0);
        for (int i = 0; i < kSyncThreads / 64; ++i) {
            const int4 t = wave_part[i];
            const int r = wave_reset[i];
            wave_part[i] = carry;
            carry = r ? t : add4(carry, t);
        }
    }
    __syncthreads();
    run = part[threadIdx.x];
    if (!s_reset[threadIdx.x]) run = add4(run, wave_part[wv]);
    for (uint32_t j = j0; j < j1; ++j) {
        const uint32_t m = A.meta[D.sub_first + j];
        if (m & 0x80000000u) run = make_int4(0, 0, 0, 0);
        const uint32_t seg = m & 0x7fffffffu;
        const uint32_t blk0 = D.seg_blocks == kNoRestart ? 0u : seg * D.seg_blocks;
        A.base[D.sub_first + j] = make_int4((int)(blk0 + (uint32_t)run.x), run.y, run.z, run.w);
        run = add4(run, A.sums[D.sub_first + j]);
        // a restart segment (or the whole scan) that ends before all of its blocks are there: a truncated or corrupt file.  The image is
        // flagged (bit 31 of rounds_out); the blocks that are missing keep whatever the coefficient buffer held.
        if (j + 1 == D.nsub || (A.meta[D.sub_first + j + 1] & 0x80000000u)) {
            const uint32_t want = D.seg_blocks == kNoRestart ? (uint32_t)G.nblk : min(D.seg_blocks, (uint32_t)G.nblk - blk0);
            if ((uint32_t)run.x < want) atomicOr(&rounds_out[blockIdx.x], 0x80000000u);
        }
    }
}

// The final pass: every subsequence decoded from its true entry state, coefficients written.  A block is assembled in the LDS slot of the
// lane in whose range it starts (kLaneBlock int16 per lane) and stored by the whole wave, eight blocks per pass (decode_sub_store,
// bevw_jpeg_walk.h); the natural-order table sits in LDS too (a global-memory look-up would queue behind the stores: on gfx9 loads and
// stores share one in-order counter).
__global__ __launch_bounds__(256) void k_jpeg_coef(const ImageDesc *__restrict__ img, const uint32_t *__restrict__ stream,
                                                   const TableSet *__restrict__ tabs, Geom G, SubArrays A, int16_t *__restrict__ coef)
{
    __shared__ TableSet T;
    __shared__ __attribute__((aligned(16))) int16_t lbuf[256][kLaneBlock];
    __shared__ uint8_t nat[64];
    __shared__ uint32_t wlist[4][64];   // decode_sub: the blocks a wave completed in one step
    const ImageDesc D = img[blockIdx.y];
    if (blockIdx.x * 256u >= D.nsub) return;
    lds_copy(&T, tabs + D.tables);
    if (threadIdx.x < 64) nat[threadIdx.x] = (uint8_t)natural_of((int)threadIdx.x);
    for (int i = threadIdx.x; i < 256 * kLaneBlock / 2; i += 256) reinterpret_cast<uint32_t *>(&lbuf[0][0])[i] = 0u;
    __syncthreads();
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    const bool alive = j < D.nsub;   // the other lanes of the last wave have no subsequence but take part in the stores
    const uint32_t jj = alive ? j : D.nsub - 1u;
    const size_t slot = (size_t)D.sub_first + jj;
    const int4 b = A.base[slot];
    uint32_t cap = (uint32_t)G.nblk;
    if (D.seg_blocks != kNoRestart) {
        const unsigned long long c2 = (unsigned long long)((A.meta[slot] & 0x7fffffffu) + 1u) * D.seg_blocks;
        if (c2 < cap) cap = (uint32_t)c2;
    }
    decode_sub_store(word_source(A, D, stream, jj), T.t, G, A.entry[slot], A.endbit[slot], coef + (size_t)blockIdx.y * G.nblk * 64, (uint32_t)b.x, cap, b.y,
                     b.z, b.w, nat, lbuf[threadIdx.x], alive, wlist[threadIdx.x >> 6]);
}

// jpeg_idct_islow: a wave transforms 8 blocks; lane = (block, column) for the column pass, (block, row) for the row pass, the 8 x 8
// intermediate goes through LDS (row pitch 8, block pitch 72 words: conflict-free for the 32-bit writes of a half wave).
// first: the first block (scan-order index inside the image's coefficient buffer) of the launch -- 0, or G.blk_off[1] when the luma blocks are
// transformed by k_jpeg_idct_color_h2v2
__global__ __launch_bounds__(256) void k_jpeg_idct(const ImageDesc *__restrict__ img, Geom G, const int16_t *__restrict__ coef,
                                                   const uint16_t *__restrict__ quant, uint8_t *__restrict__ planes, int first)
{
    __shared__ int32_t ws[4][8 * 72];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = lane >> 3, c = lane & 7;
    const int g = first + (blockIdx.x * 4 + wave) * 8 + b;
    const bool valid = g < G.nblk;
    int comp = 0;
    if (G.nc == 3) comp = g >= G.blk_off[2] ? 2 : (g >= G.blk_off[1] ? 1 : 0);
    int32_t in[8], out[8];
    if (valid) {
        const int16_t *cf = coef + ((size_t)blockIdx.y * G.nblk + g) * 64;
        const uint16_t *q = quant + (size_t)img[blockIdx.y].quant * 192 + comp * 64;
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = (int32_t)once_load<BEVW_COEF_NT>(cf + r * 8 + c) * (int32_t)q[r * 8 + c];
        idct_1d(in, out, 11);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[wave][b * 72 + r * 8 + c] = out[r];
    }
    __syncthreads();
    if (!valid) return;
#pragma unroll
    for (int x = 0; x < 8; ++x) in[x] = ws[wave][b * 72 + c * 8 + x];   // row c of block b
    idct_1d(in, out, 18);
    const int bi = g - G.blk_off[comp], bx = bi % G.wb[comp], by = bi / G.wb[comp], pw = G.wb[comp] * 8;
    uint2 v;
    v.x = range_limit(out[0]) | (range_limit(out[1]) << 8) | (range_limit(out[2]) << 16) | (range_limit(out[3]) << 24);
    v.y = range_limit(out[4]) | (range_limit(out[5]) << 8) | (range_limit(out[6]) << 16) | (range_limit(out[7]) << 24);
    uint8_t *P = planes + (size_t)blockIdx.y * G.plane_bytes + G.plane_off[comp];
    *reinterpret_cast<uint2 *>(P + (size_t)(by * 8 + c) * pw + bx * 8) = v;
}

// jdsample.c + jdcolor.c, any sampling: `count` neighbouring BGR pixels of row y from position x0 on, byte stores.
__device__ __forceinline__ void color_pixels_generic(const Geom &G, const uint8_t *__restrict__ P, uint8_t *__restrict__ o, int x0, int count, int y)
{
    const uint8_t *Yp = P + G.plane_off[0] + (size_t)y * (G.wb[0] * 8) + x0;
    for (int i = 0; i < count; ++i) {
        uint32_t px;
        if (G.nc == 1) {
            px = (uint32_t)Yp[i] * 0x010101u;
        } else {
            const int cb = upsample_at(P + G.plane_off[1], G.wb[1] * 8, G.dw, G.dh, G.hs, G.vs, x0 + i, y);
            const int cr = upsample_at(P + G.plane_off[2], G.wb[2] * 8, G.dw, G.dh, G.hs, G.vs, x0 + i, y);
            px = ycc_to_bgr(Yp[i], cb, cr);
        }
        o[3 * i] = (uint8_t)px;
        o[3 * i + 1] = (uint8_t)(px >> 8);
        o[3 * i + 2] = (uint8_t)(px >> 16);
    }
}

// A lane makes 4 neighbouring BGR pixels.  dst image i starts at out + i * image_stride, rows are row_pitch bytes.
__global__ __launch_bounds__(256) void k_jpeg_color(Geom G, const uint8_t *__restrict__ planes, uint8_t *__restrict__ out, size_t image_stride,
                                                    size_t row_pitch)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= G.w || y >= G.h) return;
    color_pixels_generic(G, planes + (size_t)blockIdx.z * G.plane_bytes, out + (size_t)blockIdx.z * image_stride + (size_t)y * row_pitch + (size_t)x0 * 3,
                         x0, min(4, G.w - x0), y);
}

// The camera case -- 4:2:0, dword-aligned destination: a lane makes 8 neighbouring pixels of one row from one 8-byte luma load and, per
// chroma component and source row, one aligned 4-sample load plus the two neighbours (h2v2_fancy_upsample: column sums 3 near + far, then
// (3 this + neighbour + 8 | 7) >> 4), and stores 24 bytes as 6 dwords.  Lanes that straddle the right edge take the generic path.
__global__ __launch_bounds__(256) void k_jpeg_color_h2v2(Geom G, const uint8_t *__restrict__ planes, uint8_t *__restrict__ out, size_t image_stride,
                                                         size_t row_pitch)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 8, y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= G.w || y >= G.h) return;
    const uint8_t *P = planes + (size_t)blockIdx.z * G.plane_bytes;
    uint8_t *o = out + (size_t)blockIdx.z * image_stride + (size_t)y * row_pitch + (size_t)x0 * 3;
    if (x0 + 8 > G.w) {
        color_pixels_generic(G, P, o, x0, G.w - x0, y);
        return;
    }
    const int cy = y >> 1, cx = x0 >> 1, cp = G.wb[1] * 8;
    int ny = (y & 1) ? cy + 1 : cy - 1;
    ny = ny < 0 ? 0 : (ny > G.dh - 1 ? G.dh - 1 : ny);
    const int li = cx > 0 ? cx - 1 : 0, ri = cx + 4 > G.dw - 1 ? G.dw - 1 : cx + 4;
    int up[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const uint8_t *r0 = P + G.plane_off[1 + c] + (size_t)cy * cp, *r1 = P + G.plane_off[1 + c] + (size_t)ny * cp;
        const uint32_t a = *reinterpret_cast<const uint32_t *>(r0 + cx), b = *reinterpret_cast<const uint32_t *>(r1 + cx);
        int cs[6];
        cs[0] = 3 * r0[li] + r1[li];
        cs[5] = 3 * r0[ri] + r1[ri];
#pragma unroll
        for (int i = 0; i < 4; ++i) cs[1 + i] = 3 * (int)((a >> (8 * i)) & 255u) + (int)((b >> (8 * i)) & 255u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            up[c][2 * i] = (3 * cs[1 + i] + cs[i] + 8) >> 4;
            up[c][2 * i + 1] = (3 * cs[1 + i] + cs[2 + i] + 7) >> 4;
        }
    }
    const uint2 yv = *reinterpret_cast<const uint2 *>(P + G.plane_off[0] + (size_t)y * (G.wb[0] * 8) + x0);
    uint32_t px[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) px[i] = ycc_to_bgr((int)(((i < 4 ? yv.x : yv.y) >> (8 * (i & 3))) & 255u), up[0][i], up[1][i]);
    uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        o32[3 * h + 0] = px[4 * h] | (px[4 * h + 1] << 24);
        o32[3 * h + 1] = (px[4 * h + 1] >> 8) | (px[4 * h + 2] << 16);
        o32[3 * h + 2] = (px[4 * h + 2] >> 16) | (px[4 * h + 3] << 8);
    }
}

// The camera case in ONE pass over the luma coefficients (round 4): 4:2:0, dword-aligned destination, width a multiple of 8.  A work-group owns
// 8 MCUs of one MCU row = 128 x 16 pixels = 32 luma blocks, 8 per wave: jpeg_idct_islow exactly as k_jpeg_idct (column pass, transpose through
// LDS, row pass), the samples go to a 16 x 128 tile in LDS instead of a luma plane in memory, and the 256 lanes then convert the tile as
// k_jpeg_color_h2v2 does (8 pixels per lane, chroma from the planes k_jpeg_idct wrote for the chroma blocks -- a sixth of the sample bytes
// each).  Saves the luma plane's round trip through HBM (2/3 of the plane bytes written and read) and a launch per slice.
__global__ __launch_bounds__(256) void k_jpeg_idct_color_h2v2(const ImageDesc *__restrict__ img, Geom G, const int16_t *__restrict__ coef,
                                                              const uint16_t *__restrict__ quant, const uint8_t *__restrict__ planes,
                                                              uint8_t *__restrict__ out, size_t image_stride, size_t row_pitch)
{
    __shared__ int32_t ws[4][8 * 72];
    __shared__ __attribute__((aligned(8))) uint8_t ytile[16][136];   // row pitch 136: the 8-byte row pieces of a wave's 8 blocks fall into different banks
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = lane >> 3, c = lane & 7;
    const int mrow = blockIdx.y, mx0 = blockIdx.x * 8;
    const int q = wave * 8 + b, ty = q >> 4, tx = q & 15;               // the tile is 16 blocks wide, 2 tall
    const int bx = mx0 * 2 + tx, by = mrow * 2 + ty;
    const bool valid = bx < G.wb[0];
    int32_t in[8], o8[8];
    if (valid) {
        const int16_t *cf = coef + ((size_t)blockIdx.z * G.nblk + (size_t)by * G.wb[0] + bx) * 64;   // (luma blocks start at 0)
        const uint16_t *qt = quant + (size_t)img[blockIdx.z].quant * 192;
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = (int32_t)once_load<BEVW_COEF_NT>(cf + r * 8 + c) * (int32_t)qt[r * 8 + c];
        idct_1d(in, o8, 11);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[wave][b * 72 + r * 8 + c] = o8[r];
    }
    __syncthreads();
    if (valid) {
#pragma unroll
        for (int x = 0; x < 8; ++x) in[x] = ws[wave][b * 72 + c * 8 + x];   // row c of block q
        idct_1d(in, o8, 18);
        uint2 v;
        v.x = range_limit(o8[0]) | (range_limit(o8[1]) << 8) | (range_limit(o8[2]) << 16) | (range_limit(o8[3]) << 24);
        v.y = range_limit(o8[4]) | (range_limit(o8[5]) << 8) | (range_limit(o8[6]) << 16) | (range_limit(o8[7]) << 24);
        *reinterpret_cast<uint2 *>(&ytile[ty * 8 + c][tx * 8]) = v;
    }
    __syncthreads();
    // colour: lane t -> 8 pixels of row t / 16 of the tile, from column (t % 16) * 8 on: a wave stores 4 row pieces of 384 contiguous bytes
    const int row = (int)threadIdx.x >> 4, oct = (int)threadIdx.x & 15;
    const int x0 = mx0 * 16 + oct * 8, y = mrow * 16 + row;
    if (x0 >= G.w || y >= G.h) return;   // (G.w % 8 == 0: no lane straddles the right edge)
    const uint8_t *P = planes + (size_t)blockIdx.z * G.plane_bytes;
    const int cy = y >> 1, cx = x0 >> 1, cp = G.wb[1] * 8;
    int ny = (y & 1) ? cy + 1 : cy - 1;
    ny = ny < 0 ? 0 : (ny > G.dh - 1 ? G.dh - 1 : ny);
    const int li = cx > 0 ? cx - 1 : 0, ri = cx + 4 > G.dw - 1 ? G.dw - 1 : cx + 4;
    int up[2][8];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint8_t *r0 = P + G.plane_off[1 + k] + (size_t)cy * cp, *r1 = P + G.plane_off[1 + k] + (size_t)ny * cp;
        const uint32_t a = *reinterpret_cast<const uint32_t *>(r0 + cx), bb = *reinterpret_cast<const uint32_t *>(r1 + cx);
        int cs[6];
        cs[0] = 3 * r0[li] + r1[li];
        cs[5] = 3 * r0[ri] + r1[ri];
#pragma unroll
        for (int i = 0; i < 4; ++i) cs[1 + i] = 3 * (int)((a >> (8 * i)) & 255u) + (int)((bb >> (8 * i)) & 255u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            up[k][2 * i] = (3 * cs[1 + i] + cs[i] + 8) >> 4;
            up[k][2 * i + 1] = (3 * cs[1 + i] + cs[2 + i] + 7) >> 4;
        }
    }
    const uint2 yv = *reinterpret_cast<const uint2 *>(&ytile[row][oct * 8]);
    uint32_t px[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) px[i] = ycc_to_bgr((int)(((i < 4 ? yv.x : yv.y) >> (8 * (i & 3))) & 255u), up[0][i], up[1][i]);
    uint32_t *o32 = reinterpret_cast<uint32_t *>(out + (size_t)blockIdx.z * image_stride + (size_t)y * row_pitch + (size_t)x0 * 3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        o32[3 * h + 0] = px[4 * h] | (px[4 * h + 1] << 24);
        o32[3 * h + 1] = (px[4 * h + 1] >> 8) | (px[4 * h + 2] << 16);
        o32[3 * h + 2] = (px[4 * h + 2] >> 16) | (px[4 * h + 3] << 8);
    }
}

// cv2.imread applies the EXIF orientation tag (OpenCV's ExifTransform: 2 = flip horizontally, 3 = rotate by 180, 4 = flip vertically,
// 5 = transpose, 6 = transpose + horizontal flip (90 degrees clockwise), 7 = transpose + both flips, 8 = transpose + vertical flip).  src: the
// decoded image as stored, dense [h][w][3]; dst: the oriented image (ow x oh = w x h, or h x w for 5 .. 8) in the caller's layout.  One thread
// per output pixel: files with an orientation tag are the exception, not the camera path.
__global__ __launch_bounds__(256) void k_jpeg_orient(const uint8_t *__restrict__ src, int w, int h, int orientation, uint8_t *__restrict__ dst,
                                                     size_t image_stride, size_t row_pitch)
{
    const int ow = orientation >= 5 ? h : w, oh = orientation >= 5 ? w : h;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= ow || y >= oh) return;
    int sx, sy;
    switch (orientation) {
        case 2: sx = w - 1 - x; sy = y; break;
        case 3: sx = w - 1 - x; sy = h - 1 - y; break;
        case 4: sx = x; sy = h - 1 - y; break;
        case 5: sx = y; sy = x; break;
        case 6: sx = y; sy = h - 1 - x; break;
        case 7: sx = w - 1 - y; sy = h - 1 - x; break;
        case 8: sx = w - 1 - y; sy = x; break;
        default: sx = x; sy = y; break;
    }
    const uint8_t *s = src + (size_t)blockIdx.z * w * h * 3 + ((size_t)sy * w + sx) * 3;
    uint8_t *d = dst + (size_t)blockIdx.z * image_stride + (size_t)y * row_pitch + (size_t)x * 3;
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

// ---- encode ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_jenc_ycc(Geom G, const uint8_t *__restrict__ bgr, size_t image_stride, size_t row_pitch,
                                                  uint8_t *__restrict__ planes)
{
    const int xc = blockIdx.x * 64 + threadIdx.x, yo = blockIdx.y * 4 + threadIdx.y;
    if (xc >= G.wb[1] * 8 || yo >= G.hb[1] * 8) return;
    uint8_t *P = planes + (size_t)blockIdx.z * G.plane_bytes;
    enc_ycc_at(bgr + (size_t)blockIdx.z * image_stride, row_pitch, G, xc, yo, P + G.plane_off[0], P + G.plane_off[1], P + G.plane_off[2]);
}

// jpeg_fdct_islow + quantisation: lane = (block, row) for the row pass, (block, column) for the column pass; blocks are numbered and
// stored in scan order (MCU by MCU), coefficients in zigzag order -- what the entropy coder walks.  The quantised block also goes to LDS,
// where one lane per block counts the bits of its AC codes (jchuff.c encode_one_block without the DC part, which needs the neighbour).
// the same for 4:2:0, 8 x 2 luma pixels per lane (enc_ycc_h2v2_tile): grid (ceil(chroma width / 4 / 64), ceil(chroma rows / 4), images), block (64, 4)
__global__ __launch_bounds__(256) void k_jenc_ycc_h2v2(Geom G, const uint8_t *__restrict__ bgr, size_t image_stride, size_t row_pitch,
                                                       uint8_t *__restrict__ planes)
{
    const int t = blockIdx.x * 64 + threadIdx.x, yo = blockIdx.y * 4 + threadIdx.y;
    if (4 * t >= G.wb[1] * 8 || yo >= G.hb[1] * 8) return;
    uint8_t *P = planes + (size_t)blockIdx.z * G.plane_bytes;
    enc_ycc_h2v2_tile(bgr + (size_t)blockIdx.z * image_stride, row_pitch, G, t, yo, P + G.plane_off[0], P + G.plane_off[1], P + G.plane_off[2]);
}
__global__ __launch_bounds__(256) void k_jenc_fdct(Geom G, const uint8_t *__restrict__ planes, const EncTables *__restrict__ tabs,
                                                   int16_t *__restrict__ zz, uint16_t *__restrict__ acbits, int16_t *__restrict__ dcq,
                                                   uint64_t *__restrict__ nzmask)
{
    __shared__ int32_t ws[4][8 * 72];
    __shared__ int16_t zl[4][8][64];
    __shared__ uint8_t alen[2][256];
    for (int i = threadIdx.x; i < 512; i += 256) alen[i >> 8][i & 255] = tabs->ac[i >> 8].len[i & 255];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = lane >> 3, r = lane & 7;
    const int g = (blockIdx.x * 4 + wave) * 8 + b;
    const bool valid = g < G.nblk;
    int comp = 0, rx = 0, ry = 0;
    bool dc_only = false;
    int32_t in[8], out[8];
    if (valid) {
        enc_block_root(G, g, comp, rx, ry, dc_only);
        const int pw = G.wb[comp] * 8;
        const uint8_t *P = planes + (size_t)blockIdx.y * G.plane_bytes + G.plane_off[comp] + (size_t)(ry * 8 + r) * pw + rx * 8;
        const uint2 v = *reinterpret_cast<const uint2 *>(P);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            in[x] = (int32_t)((v.x >> (8 * x)) & 255u) - 128;
            in[4 + x] = (int32_t)((v.y >> (8 * x)) & 255u) - 128;
        }
        fdct_1d(in, out, 0);
#pragma unroll
        for (int x = 0; x < 8; ++x) ws[wave][b * 72 + r * 8 + x] = out[x];
    }
    __syncthreads();
    if (valid) {
        const int c = r;   // the lane now owns column c of its block
#pragma unroll
        for (int y = 0; y < 8; ++y) in[y] = ws[wave][b * 72 + y * 8 + c];
        fdct_1d(in, out, 1);
        const uint16_t *q = tabs->q[comp ? 1 : 0];
        const uint32_t *qr = tabs->recip[comp ? 1 : 0];
        int16_t *o = zz + ((size_t)blockIdx.y * G.nblk + g) * 64;
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            int32_t v = quantize(out[y], (int32_t)q[y * 8 + c], qr[y * 8 + c]);
            if (dc_only && (y | c)) v = 0;
            const int k = zigzag_of(y * 8 + c);
            o[k] = (int16_t)v;
            zl[wave][b][k] = (int16_t)v;
        }
    }
    __syncthreads();
    // the bits of the block's AC codes, eight lanes per block (ac_code_bits_octet; one lane walking all 63 coefficients left 7 of 8 lanes idle
    // for longer than both DCT passes took): nonzero mask through LDS, eight partial sums through LDS
    __shared__ uint8_t nzb[4][8][8];
    __shared__ uint32_t part[4][8][8];
    int16_t z8[8];
    if (valid) {
        const uint4 v = *reinterpret_cast<const uint4 *>(&zl[wave][b][8 * r]);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            z8[i] = (int16_t)(w[i >> 1] >> (16 * (i & 1)));
            m |= z8[i] != 0 ? 1u << i : 0u;
        }
        nzb[wave][b][r] = (uint8_t)(r == 0 ? (m | 1u) : m);
    }
    __syncthreads();
    if (valid) {
        const uint2 mm = *reinterpret_cast<const uint2 *>(&nzb[wave][b][0]);
        part[wave][b][r] = ac_code_bits_octet(z8, r, ((uint64_t)mm.y << 32) | mm.x, alen[comp ? 1 : 0]);
    }
    __syncthreads();
    if (valid && r == 0) {
        const size_t blk = (size_t)blockIdx.y * G.nblk + g;
        uint32_t bits = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) bits += part[wave][b][i];
        acbits[blk] = (uint16_t)bits;
        dcq[blk] = zl[wave][b][0];
        const uint2 mm = *reinterpret_cast<const uint2 *>(&nzb[wave][b][0]);
        nzmask[blk] = ((uint64_t)mm.y << 32) | mm.x;   // for k_jenc_bits (encode_block)
    }
}

// One block per image: the length of every block's code (AC bits from k_jenc_fdct + the DC difference's code), bit offsets (exclusive
// prefix), the byte count, and the bit buffer zeroed up to the last word any block will touch.
__global__ __launch_bounds__(kSyncThreads) void k_jenc_scan(Geom G, const uint16_t *__restrict__ acbits, const int16_t *__restrict__ dcq,
                                                             const EncTables *__restrict__ tabs, uint32_t *__restrict__ bitpos,
                                                             uint32_t *__restrict__ bitbuf, size_t buf_words, uint32_t *__restrict__ totals)
{
    __shared__ uint32_t lds[kSyncThreads + kSyncThreads / 64 + 1];
    __shared__ uint8_t dlen[2][16];
    if (threadIdx.x < 32) dlen[threadIdx.x >> 4][threadIdx.x & 15] = tabs->dc[threadIdx.x >> 4].len[threadIdx.x & 15];
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * G.nblk;
    uint32_t *B = bitpos + base;
    const uint32_t n = (uint32_t)G.nblk, L = (n + kSyncThreads - 1) / kSyncThreads;
    const uint32_t g0 = threadIdx.x * L, g1 = min(g0 + L, n);
    uint32_t run = 0;
    for (uint32_t g = g0; g < g1; ++g) {
        const int pr = dc_predecessor((int)g, G);
        const int diff = (int)dcq[base + g] - (pr < 0 ? 0 : (int)dcq[base + pr]);
        const uint32_t len = (uint32_t)acbits[base + g] + dc_code_bits(diff, dlen[((int)g % G.bpm) < G.nY ? 0 : 1]);
        B[g] = len;
        run += len;
    }
    uint32_t total;
    uint32_t carry = block_exscan_serial<uint32_t>(run, lds, total);
    for (uint32_t g = g0; g < g1; ++g) { const uint32_t t = B[g]; B[g] = carry; carry += t; }
    const uint32_t nbytes = (total + 7u) >> 3;
    if (threadIdx.x == 0) { totals[2 * blockIdx.x] = total; totals[2 * blockIdx.x + 1] = nbytes; }
    uint32_t *W = bitbuf + (size_t)blockIdx.x * buf_words;
    const uint32_t nw = min((uint32_t)buf_words, (nbytes >> 2) + 2u);
    for (uint32_t i = threadIdx.x; i < nw; i += kSyncThreads) W[i] = 0u;
}

// Every block written at its bit offset.  The 256 blocks of a work-group are staged in LDS (row pitch 33 words: a lane walking its own
// block meets no bank conflict), because 64 lanes reading 64 different 128-byte lines of global memory per instruction is what made the
// first version slow.
__global__ __launch_bounds__(256) void k_jenc_bits(Geom G, const int16_t *__restrict__ zz, const int16_t *__restrict__ dcq,
                                                   const EncTables *__restrict__ tabs, const uint32_t *__restrict__ bitpos, uint32_t *__restrict__ bitbuf,
                                                   size_t buf_words, const uint64_t *__restrict__ nzmask)
{
    __shared__ EncTables T;
    __shared__ uint32_t zl[256 * 33];
    lds_copy(&T, tabs);
    const int g0 = blockIdx.x * 256;
    const int count = min(256, G.nblk - g0);
    const size_t base = (size_t)blockIdx.y * G.nblk;
    const uint32_t *Z = reinterpret_cast<const uint32_t *>(zz + (base + g0) * 64);
    for (int i = threadIdx.x; i < count * 32; i += 256) zl[(i >> 5) * 33 + (i & 31)] = Z[i];
    __syncthreads();
    const int g = g0 + threadIdx.x;
    if (g >= G.nblk) return;
    const int pr = dc_predecessor(g, G);
    const int last = pr < 0 ? 0 : dcq[base + pr];
    const int t = (g % G.bpm) < G.nY ? 0 : 1;
    encode_block<true>(reinterpret_cast<const int16_t *>(zl + threadIdx.x * 33), last, T.dc[t], T.ac[t], bitbuf + (size_t)blockIdx.y * buf_words,
                       bitpos[base + g], nzmask[base + g]);
}

// The file = header | entropy-coded bytes with a 0x00 after every 0xFF (jchuff.c emit_bits; the last byte is filled with 1-bits first,
// flush_bits) | EOI.  The bytes are cut into chunks of kStuffChunk; k_jenc_ffcount counts the 0xFF bytes of every chunk, k_jenc_stuff
// adds up the counts of the chunks in front of its own, scans its 256 lanes (16 bytes each) and writes.
constexpr uint32_t kStuffChunk = 4096;

// the 16 bytes [byte0, byte0 + 16) of an image's bit buffer as 4 big-endian words, the last byte of the stream padded with 1-bits;
// returns how many of the bytes that exist are 0xFF
__device__ __forceinline__ uint32_t stuff_load16(const uint32_t *__restrict__ W, uint32_t byte0, uint32_t nbytes, uint32_t pad, uint32_t v[4])
{
    uint32_t nff = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t b = byte0 + 4u * i;
        uint32_t w = b < nbytes ? W[b >> 2] : 0u;
        if (b < nbytes && nbytes - b <= 4u) w |= ((1u << pad) - 1u) << (8u * (3u - (nbytes - 1u - b)));   // the stream's last byte is in this word
        v[i] = w;
#pragma unroll
        for (int k = 0; k < 4; ++k) nff += (b + k < nbytes) && ((w >> (24 - 8 * k)) & 255u) == 255u;
    }
    return nff;
}

__global__ __launch_bounds__(256) void k_jenc_ffcount(const uint32_t *__restrict__ bitbuf, size_t buf_words, const uint32_t *__restrict__ totals,
                                                      uint32_t *__restrict__ chunk_ff, uint32_t nchunk)
{
    __shared__ uint32_t lds[4];
    const uint32_t total_bits = totals[2 * blockIdx.y], nbytes = totals[2 * blockIdx.y + 1];
    if (blockIdx.x * kStuffChunk >= nbytes) return;
    uint32_t v[4];
    const uint32_t nff = stuff_load16(bitbuf + (size_t)blockIdx.y * buf_words, blockIdx.x * kStuffChunk + threadIdx.x * 16u, nbytes, nbytes * 8u - total_bits, v);
    const uint32_t t = block256_sum(nff, lds);
    if (threadIdx.x == 0) chunk_ff[(size_t)blockIdx.y * nchunk + blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void k_jenc_stuff(const uint32_t *__restrict__ bitbuf, size_t buf_words, const uint32_t *__restrict__ totals,
                                                    const uint32_t *__restrict__ chunk_ff, uint32_t nchunk, const uint8_t *__restrict__ header,
                                                    uint32_t header_len, uint8_t *__restrict__ files, size_t file_cap, uint32_t *__restrict__ sizes)
{
    __shared__ uint32_t lds[4];
    const uint32_t total_bits = totals[2 * blockIdx.y], nbytes = totals[2 * blockIdx.y + 1];
    const uint32_t c0 = blockIdx.x * kStuffChunk;
    if (c0 >= nbytes) return;
    uint8_t *F = files + (size_t)blockIdx.y * file_cap;
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < header_len; i += 256u) F[i] = header[i];
    // 0xFF bytes in the chunks before this one
    uint32_t before = 0;
    for (uint32_t i = threadIdx.x; i < blockIdx.x; i += 256u) before += chunk_ff[(size_t)blockIdx.y * nchunk + i];
    before = block256_sum(before, lds);
    uint32_t v[4];
    const uint32_t byte0 = c0 + threadIdx.x * 16u;
    const uint32_t nff = stuff_load16(bitbuf + (size_t)blockIdx.y * buf_words, byte0, nbytes, nbytes * 8u - total_bits, v);
    // exclusive prefix over the 256 lanes
    const uint32_t incl = wave_incl_scan(nff);
    if ((threadIdx.x & 63) == 63) lds[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t excl = incl - nff;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) excl += lds[w];
    const uint32_t chunk_total = lds[0] + lds[1] + lds[2] + lds[3];
    uint8_t *E = F + header_len;
    size_t o = (size_t)byte0 + before + excl;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t b = byte0 + 4u * i + k;
            if (b < nbytes) {
                const uint32_t x = (v[i] >> (24 - 8 * k)) & 255u;
                E[o++] = (uint8_t)x;
                if (x == 255u) E[o++] = 0;
            }
        }
    if (threadIdx.x == 0 && c0 + kStuffChunk >= nbytes) {   // the last chunk closes the file
        const size_t end = (size_t)nbytes + before + chunk_total;
        E[end] = 0xFF;
        E[end + 1] = 0xD9;
        sizes[blockIdx.y] = (uint32_t)(header_len + end + 2);
    }
}

}  // namespace jpg
}  // namespace bevw
