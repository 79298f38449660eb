// bevw_jpeg_codec.h -- kernels and host orchestration of the JPEG decode / encode stages (row f4); the per-lane arithmetic is in
// bevw_jpeg.h.  Included by bevwarp.hip after its error / buffer helpers (fail, HIP_TRY, BEVW_TRY, DevBuf, LapTimer, launch_check).
#pragma once
#include "bevw_jpeg.h"

namespace bevw {
namespace jpg {

constexpr int kSyncThreads = 1024;   // one block per image in the per-image kernels (fixed point, prefix sums, stuffing)

template <typename T> __device__ __forceinline__ void lds_copy(T *dst, const T *__restrict__ src)
{
    static_assert(sizeof(T) % 16 == 0, "16-byte pieces");
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    for (unsigned i = threadIdx.x + threadIdx.y * blockDim.x; i < sizeof(T) / 16; i += blockDim.x * blockDim.y) d[i] = s[i];
}

// Exclusive prefix over one value per thread of a 1-D block, done the plain way: partials to LDS, thread 0 walks them.  The blocks that
// use it run one per image and call it a handful of times; a walk over 1024 LDS words is a few microseconds and cannot be wrong.
template <typename V> __device__ __forceinline__ V block_exscan_serial(V v, V *lds, V &total)
{
    lds[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        V run = V();
        for (unsigned i = 0; i < blockDim.x; ++i) { const V t = lds[i]; lds[i] = run; run = run + t; }
        lds[blockDim.x] = run;
    }
    __syncthreads();
    const V r = lds[threadIdx.x];
    total = lds[blockDim.x];
    __syncthreads();
    return r;
}

// ---- decode ----------------------------------------------------------------------------------------------------------------------
struct SubArrays {
    uint64_t *entry;    // state at the first symbol of the subsequence
    uint64_t *exitst;   // state at the first symbol of the next one
    int4 *sums;         // blocks completed, DC difference sums of the three components
    int4 *base;         // exclusive prefix of `sums` inside the restart segment (x = index of the block in progress, scan order)
    uint32_t *endbit;   // last bit (exclusive) the subsequence owns
    uint32_t *meta;     // restart segment | first-of-segment << 31
};

__global__ __launch_bounds__(256) void k_jpeg_sync0(const ImageDesc *__restrict__ img, const uint32_t *__restrict__ stream,
                                                    const TableSet *__restrict__ tabs, Geom G, const uint32_t *__restrict__ seg_byte,
                                                    const uint32_t *__restrict__ seg_sub, SubArrays A)
{
    __shared__ TableSet T;
    const ImageDesc D = img[blockIdx.y];
    lds_copy(&T, tabs + D.tables);
    __syncthreads();
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
    const uint64_t e = pack_state(start, 0, 0);
    const SubOut R = decode_sub<false>(stream + D.stream_word, T.t, G, e, end, nullptr, 0, 0, 0, 0, 0);
    const size_t slot = (size_t)D.sub_first + j;
    A.entry[slot] = e;
    A.exitst[slot] = R.exit;
    A.sums[slot] = make_int4(R.cnt, R.dc0, R.dc1, R.dc2);
    A.endbit[slot] = end;
    A.meta[slot] = lo | (j == ss[lo] ? 0x80000000u : 0u);
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
    const uint32_t *words = stream + D.stream_word;
    volatile uint64_t *vexit = A.exitst + D.sub_first;
    uint64_t *entry = A.entry + D.sub_first;
    uint32_t rounds = 0;
    for (;;) {
        int changed = 0;
        for (uint32_t j = threadIdx.x; j < D.nsub; j += kSyncThreads) {
            if (j == 0 || (A.meta[D.sub_first + j] & 0x80000000u)) continue;
            const uint64_t in = vexit[j - 1];
            if (in == entry[j]) continue;
            entry[j] = in;
            const SubOut R = decode_sub<false>(words, T.t, G, in, A.endbit[D.sub_first + j], nullptr, 0, 0, 0, 0, 0);
            A.sums[D.sub_first + j] = make_int4(R.cnt, R.dc0, R.dc1, R.dc2);
            if (R.exit != vexit[j]) { vexit[j] = R.exit; changed = 1; }
        }
        ++rounds;
        __threadfence_block();
        if (!__syncthreads_or(changed)) break;
        if (rounds > D.nsub + 2u) break;   // cannot happen (induction); never spin
    }
    if (threadIdx.x == 0) rounds_out[blockIdx.x] = rounds;
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
    if (threadIdx.x == 0) {
        int4 carry = make_int4(0, 0, 0, 0);
        for (int i = 0; i < kSyncThreads; ++i) {
            const int4 t = part[i];
            const int r = s_reset[i];
            part[i] = carry;                     // what thread i starts from (void after its first reset)
            carry = r ? t : add4(carry, t);
        }
    }
    __syncthreads();
    run = part[threadIdx.x];
    for (uint32_t j = j0; j < j1; ++j) {
        const uint32_t m = A.meta[D.sub_first + j];
        if (m & 0x80000000u) run = make_int4(0, 0, 0, 0);
        const uint32_t seg = m & 0x7fffffffu;
        const uint32_t blk0 = D.seg_blocks == kNoRestart ? 0u : seg * D.seg_blocks;
        A.base[D.sub_first + j] = make_int4((int)(blk0 + (uint32_t)run.x), run.y, run.z, run.w);
        run = add4(run, A.sums[D.sub_first + j]);
    }
}

__global__ __launch_bounds__(256) void k_jpeg_coef(const ImageDesc *__restrict__ img, const uint32_t *__restrict__ stream,
                                                   const TableSet *__restrict__ tabs, Geom G, SubArrays A, int16_t *__restrict__ coef)
{
    __shared__ TableSet T;
    const ImageDesc D = img[blockIdx.y];
    lds_copy(&T, tabs + D.tables);
    __syncthreads();
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= D.nsub) return;
    const size_t slot = (size_t)D.sub_first + j;
    const int4 b = A.base[slot];
    uint32_t cap = (uint32_t)G.nblk;
    if (D.seg_blocks != kNoRestart) {
        const unsigned long long c2 = (unsigned long long)((A.meta[slot] & 0x7fffffffu) + 1u) * D.seg_blocks;
        if (c2 < cap) cap = (uint32_t)c2;
    }
    decode_sub<true>(stream + D.stream_word, T.t, G, A.entry[slot], A.endbit[slot], coef + (size_t)blockIdx.y * G.nblk * 64, (uint32_t)b.x, cap, b.y,
                     b.z, b.w);
}

// jpeg_idct_islow: a wave transforms 8 blocks; lane = (block, column) for the column pass, (block, row) for the row pass, the 8 x 8
// intermediate goes through LDS (row pitch 8, block pitch 72 words: conflict-free for the 32-bit writes of a half wave).
__global__ __launch_bounds__(256) void k_jpeg_idct(const ImageDesc *__restrict__ img, Geom G, const int16_t *__restrict__ coef,
                                                   const uint16_t *__restrict__ quant, uint8_t *__restrict__ planes)
{
    __shared__ int32_t ws[4][8 * 72];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = lane >> 3, c = lane & 7;
    const int g = (blockIdx.x * 4 + wave) * 8 + b;
    const bool valid = g < G.nblk;
    int comp = 0;
    if (G.nc == 3) comp = g >= G.blk_off[2] ? 2 : (g >= G.blk_off[1] ? 1 : 0);
    int32_t in[8], out[8];
    if (valid) {
        const int16_t *cf = coef + ((size_t)blockIdx.y * G.nblk + g) * 64;
        const uint16_t *q = quant + (size_t)img[blockIdx.y].quant * 192 + comp * 64;
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = (int32_t)cf[r * 8 + c] * (int32_t)q[r * 8 + c];
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

// jdsample.c + jdcolor.c: a lane makes 4 neighbouring BGR pixels.  dst image i starts at out + i * image_stride, rows are row_pitch bytes.
__global__ __launch_bounds__(256) void k_jpeg_color(Geom G, const uint8_t *__restrict__ planes, uint8_t *__restrict__ out, size_t image_stride,
                                                    size_t row_pitch, int aligned)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= G.w || y >= G.h) return;
    const uint8_t *P = planes + (size_t)blockIdx.z * G.plane_bytes;
    const uint8_t *Yp = P + G.plane_off[0] + (size_t)y * (G.wb[0] * 8) + x0;   // the luma plane is at least 4 samples wider than x0
    uint32_t px[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = x0 + i;
        if (G.nc == 1) {
            px[i] = (uint32_t)Yp[i] * 0x010101u;
        } else {
            const int xs = x < G.w ? x : G.w - 1;   // lanes past the right edge compute a pixel that is never stored
            const int cb = upsample_at(P + G.plane_off[1], G.wb[1] * 8, G.dw, G.dh, G.hs, G.vs, xs, y);
            const int cr = upsample_at(P + G.plane_off[2], G.wb[2] * 8, G.dw, G.dh, G.hs, G.vs, xs, y);
            px[i] = ycc_to_bgr(Yp[i], cb, cr);
        }
    }
    uint8_t *o = out + (size_t)blockIdx.z * image_stride + (size_t)y * row_pitch + (size_t)x0 * 3;
    if (aligned && x0 + 4 <= G.w) {
        uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
        o32[0] = px[0] | (px[1] << 24);
        o32[1] = (px[1] >> 8) | (px[2] << 16);
        o32[2] = (px[2] >> 16) | (px[3] << 8);
    } else {
        for (int i = 0; i < 4 && x0 + i < G.w; ++i) {
            o[3 * i] = (uint8_t)px[i];
            o[3 * i + 1] = (uint8_t)(px[i] >> 8);
            o[3 * i + 2] = (uint8_t)(px[i] >> 16);
        }
    }
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
// stored in scan order (MCU by MCU), coefficients in zigzag order -- what the entropy coder walks.
__global__ __launch_bounds__(256) void k_jenc_fdct(Geom G, const uint8_t *__restrict__ planes, const EncTables *__restrict__ tabs,
                                                   int16_t *__restrict__ zz)
{
    __shared__ int32_t ws[4][8 * 72];
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
    if (!valid) return;
    const int c = r;   // the lane now owns column c of its block
#pragma unroll
    for (int y = 0; y < 8; ++y) in[y] = ws[wave][b * 72 + y * 8 + c];
    fdct_1d(in, out, 1);
    const uint16_t *q = tabs->q[comp ? 1 : 0];
    int16_t *o = zz + ((size_t)blockIdx.y * G.nblk + g) * 64;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        int32_t v = quantize(out[y], (int32_t)q[y * 8 + c]);
        if (dc_only && (y | c)) v = 0;
        o[zigzag_of(y * 8 + c)] = (int16_t)v;
    }
}

__global__ __launch_bounds__(256) void k_jenc_len(Geom G, const int16_t *__restrict__ zz, const EncTables *__restrict__ tabs, uint32_t *__restrict__ bitlen)
{
    __shared__ EncTables T;
    lds_copy(&T, tabs);
    __syncthreads();
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G.nblk) return;
    const int16_t *Z = zz + (size_t)blockIdx.y * G.nblk * 64;
    const int pr = dc_predecessor(g, G);
    const int last = pr < 0 ? 0 : Z[(size_t)pr * 64];
    const int t = (g % G.bpm) < G.nY ? 0 : 1;
    bitlen[(size_t)blockIdx.y * G.nblk + g] = encode_block<false>(Z + (size_t)g * 64, last, T.dc[t], T.ac[t], nullptr, 0);
}

// One block per image: bit offsets of the blocks (exclusive prefix of their lengths, in place), the byte count, and the bit buffer zeroed
// up to the last word any block will touch.
__global__ __launch_bounds__(kSyncThreads) void k_jenc_scan(Geom G, uint32_t *__restrict__ bitlen, uint32_t *__restrict__ bitbuf, size_t buf_words,
                                                             uint32_t *__restrict__ totals)
{
    __shared__ uint32_t lds[kSyncThreads + 1];
    uint32_t *B = bitlen + (size_t)blockIdx.x * G.nblk;
    const uint32_t n = (uint32_t)G.nblk, L = (n + kSyncThreads - 1) / kSyncThreads;
    const uint32_t g0 = threadIdx.x * L, g1 = min(g0 + L, n);
    uint32_t run = 0;
    for (uint32_t g = g0; g < g1; ++g) run += B[g];
    uint32_t total;
    uint32_t carry = block_exscan_serial<uint32_t>(run, lds, total);
    for (uint32_t g = g0; g < g1; ++g) { const uint32_t t = B[g]; B[g] = carry; carry += t; }
    const uint32_t nbytes = (total + 7u) >> 3;
    if (threadIdx.x == 0) { totals[2 * blockIdx.x] = total; totals[2 * blockIdx.x + 1] = nbytes; }
    uint32_t *W = bitbuf + (size_t)blockIdx.x * buf_words;
    const uint32_t nw = min((uint32_t)buf_words, (nbytes >> 2) + 2u);
    for (uint32_t i = threadIdx.x; i < nw; i += kSyncThreads) W[i] = 0u;
}

__global__ __launch_bounds__(256) void k_jenc_bits(Geom G, const int16_t *__restrict__ zz, const EncTables *__restrict__ tabs,
                                                   const uint32_t *__restrict__ bitpos, uint32_t *__restrict__ bitbuf, size_t buf_words)
{
    __shared__ EncTables T;
    lds_copy(&T, tabs);
    __syncthreads();
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G.nblk) return;
    const int16_t *Z = zz + (size_t)blockIdx.y * G.nblk * 64;
    const int pr = dc_predecessor(g, G);
    const int last = pr < 0 ? 0 : Z[(size_t)pr * 64];
    const int t = (g % G.bpm) < G.nY ? 0 : 1;
    encode_block<true>(Z + (size_t)g * 64, last, T.dc[t], T.ac[t], bitbuf + (size_t)blockIdx.y * buf_words, bitpos[(size_t)blockIdx.y * G.nblk + g]);
}

// One block per image: the file = header | entropy-coded bytes with a 0x00 after every 0xFF (jchuff.c emit_bits; the last byte is filled
// with 1-bits first, flush_bits) | EOI.  Thread t owns the bytes [t L, (t + 1) L): count its 0xFF bytes, prefix, write.
__global__ __launch_bounds__(kSyncThreads) void k_jenc_stuff(const uint32_t *__restrict__ bitbuf, size_t buf_words, const uint32_t *__restrict__ totals,
                                                              const uint8_t *__restrict__ header, uint32_t header_len, uint8_t *__restrict__ files,
                                                              size_t file_cap, uint32_t *__restrict__ sizes)
{
    __shared__ uint32_t lds[kSyncThreads + 1];
    const uint32_t *W = bitbuf + (size_t)blockIdx.x * buf_words;
    const uint32_t total_bits = totals[2 * blockIdx.x], nbytes = totals[2 * blockIdx.x + 1];
    const uint32_t pad = nbytes * 8u - total_bits;
    uint8_t *F = files + (size_t)blockIdx.x * file_cap;
    for (uint32_t i = threadIdx.x; i < header_len; i += kSyncThreads) F[i] = header[i];
    const uint32_t L = (((nbytes + kSyncThreads - 1) / kSyncThreads) + 3u) & ~3u;   // whole words per thread
    const uint32_t b0 = min(threadIdx.x * L, nbytes), b1 = min(b0 + L, nbytes);
    auto byte_at = [&](uint32_t i) -> uint32_t {
        uint32_t v = (W[i >> 2] >> (24u - 8u * (i & 3u))) & 255u;
        if (i == nbytes - 1u) v |= (1u << pad) - 1u;
        return v;
    };
    uint32_t nff = 0;
    for (uint32_t i = b0; i < b1; ++i) nff += byte_at(i) == 255u;
    uint32_t total_ff;
    uint32_t o = block_exscan_serial<uint32_t>(nff, lds, total_ff);
    const size_t need = (size_t)header_len + nbytes + total_ff + 2;
    if (need > file_cap) {   // cannot happen with the capacity the host reserves; never write out of bounds
        if (threadIdx.x == 0) sizes[blockIdx.x] = 0;
        return;
    }
    uint8_t *E = F + header_len;
    o += b0;
    for (uint32_t i = b0; i < b1; ++i) {
        const uint32_t v = byte_at(i);
        E[o++] = (uint8_t)v;
        if (v == 255u) E[o++] = 0;
    }
    if (threadIdx.x == 0) {
        E[nbytes + total_ff] = 0xFF;
        E[nbytes + total_ff + 1] = 0xD9;
        sizes[blockIdx.x] = (uint32_t)need;
    }
}

}  // namespace jpg
}  // namespace bevw
