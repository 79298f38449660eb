// valu_rates.hip -- issue cost of the integer / byte VALU instructions the stitch bodies are built from, in shader clocks
// per wave64 instruction per SIMD (scratch tool).  gfx950 issues some VALU ops in 2 clocks per wave (32 lanes / clk) and
// most integer ops in 4: the table decides which formulation of the fixed-point bilinear is cheapest.
// Method: 16 independent chains per lane, 4096 ops per chain, W waves per SIMD resident (grid = 256 CUs x 1 block of
// 256 * W threads ... one block per CU, 4 * W waves), cycles from s_memtime around the loop; reported =
// cycles / (ops per wave x waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define OPS(X) \
    X(0, "v_add_u32 %0, %1, %0", "v_add_u32") \
    X(1, "v_or_b32 %0, %1, %0", "v_or_b32") \
    X(2, "v_and_b32 %0, %1, %0", "v_and_b32") \
    X(3, "v_lshlrev_b32 %0, 3, %0", "v_lshlrev_b32") \
    X(4, "v_mul_u32_u24 %0, %1, %0", "v_mul_u32_u24") \
    X(5, "v_mad_u32_u24 %0, %1, %2, %0", "v_mad_u32_u24") \
    X(6, "v_perm_b32 %0, %1, %0, %2", "v_perm_b32") \
    X(7, "v_dot4_u32_u8 %0, %1, %2, %0", "v_dot4_u32_u8") \
    X(8, "v_dot2_u32_u16 %0, %1, %2, %0", "v_dot2_u32_u16") \
    X(9, "v_lshl_or_b32 %0, %1, 16, %0", "v_lshl_or_b32") \
    X(10, "v_lshl_add_u32 %0, %1, 3, %0", "v_lshl_add_u32") \
    X(11, "v_and_or_b32 %0, %1, %2, %0", "v_and_or_b32") \
    X(12, "v_bfe_u32 %0, %0, 3, 8", "v_bfe_u32") \
    X(13, "v_alignbyte_b32 %0, %1, %0, 1", "v_alignbyte_b32") \
    X(14, "v_cndmask_b32 %0, %0, %1, vcc", "v_cndmask_b32") \
    X(15, "v_mov_b32 %0, %1", "v_mov_b32") \
    X(16, "v_fma_f32 %0, %1, %2, %0", "v_fma_f32") \
    X(17, "v_pk_fma_f32 %0, %1, %2, %0", "v_pk_fma_f32 (64-bit operands)") \
    X(18, "v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2", "v_add_u32_sdwa") \
    X(19, "v_mul_u32_u24_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0", "v_mul_u32_u24_sdwa") \
    X(20, "v_cvt_f32_ubyte0 %0, %1", "v_cvt_f32_ubyte0") \
    X(21, "v_min_u32 %0, %1, %0", "v_min_u32") \
    X(22, "v_or3_b32 %0, %1, %2, %0", "v_or3_b32") \
    X(23, "v_mul_lo_u32 %0, %1, %0", "v_mul_lo_u32") \
    X(24, "v_pk_mad_u16 %0, %1, %2, %0", "v_pk_mad_u16") \
    X(25, "v_pk_add_u16 %0, %1, %0", "v_pk_add_u16") \
    X(26, "v_max_u32 %0, %1, %0", "v_max_u32") \
    X(27, "v_fmac_f32 %0, %1, %2", "v_fmac_f32 (VOP2)") \
    X(28, "v_xor_b32 %0, %1, %0", "v_xor_b32") \
    X(29, "v_sad_u8 %0, %1, %2, %0", "v_sad_u8") \
    X(30, "v_mad_u16 %0, %1, %2, %0", "v_mad_u16") \
    X(31, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp") \
    X(32, "v_mul_f32 %0, %1, %0", "v_mul_f32") \
    X(33, "v_add_f32 %0, %1, %0", "v_add_f32") \
    X(34, "v_cvt_f32_u32 %0, %0", "v_cvt_f32_u32") \
    X(35, "v_cvt_i32_f32 %0, %0", "v_cvt_i32_f32") \
    X(36, "v_floor_f32 %0, %0", "v_floor_f32") \
    X(37, "v_max3_u32 %0, %1, %2, %0", "v_max3_u32") \
    X(38, "v_med3_i32 %0, %1, %2, %0", "v_med3_i32") \
    X(39, "v_cmp_lt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc", "v_cmp_lt_u32 + v_cndmask_b32 (vcc), per PAIR") \
    X(40, "v_cmp_lt_u32 vcc, %1, %0", "v_cmp_lt_u32 (vcc)") \
    X(41, "v_cvt_pk_u8_f32 %0, %1, 1, %0", "v_cvt_pk_u8_f32") \
    X(42, "v_mul_f64 %0, %1, %0", "(placeholder, see 43)") \
    X(44, "v_sub_u32 %0, %1, %0", "v_sub_u32") \
    X(45, "v_ashrrev_i32 %0, 3, %0", "v_ashrrev_i32") \
    X(46, "v_rndne_f32 %0, %0", "v_rndne_f32") \
    X(47, "v_fract_f32 %0, %0", "v_fract_f32")

template <int OP>
__global__ void __launch_bounds__(1024) k_valu(int iters, uint32_t *__restrict__ sink, unsigned long long *__restrict__ cyc)
{
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    uint32_t r[16];
    u2 r2[8];
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 17 + i;
    for (int i = 0; i < 8; ++i) r2[i] = u2{r[i], r[i + 8]};
    const uint32_t w = threadIdx.x | 0x01020304u;
    const u2 w2 = u2{w, w};
    const unsigned long long smask = __builtin_amdgcn_readfirstlane(iters) * 0x9e3779b97f4a7c15ull;   // a wave-uniform 64-bit lane mask in SGPRs
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 17) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r2[i & 7]) : "v"(r2[(i + 1) & 7]), "v"(w2)); continue; }
            if (OP == 42) continue;
            if (OP == 43) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "s"(smask)); continue; }
            if (OP == 48) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r2[i & 7]) : "v"(r[(i + 1) & 15]), "v"(w) : "vcc"); continue; }
            if (OP == 49) { asm volatile("v_mul_f64 %0, %1, %0" : "+v"(r2[i & 7]) : "v"(r2[(i + 1) & 7])); continue; }
#define X(N, ASM, NAME) if (OP == N) asm volatile(ASM : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(w) : "vcc");
            OPS(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
    for (int i = 0; i < 16; ++i) acc ^= r[i];
    for (int i = 0; i < 8; ++i) acc ^= r2[i].x ^ r2[i].y;
    if (acc == 0x12345678u) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);
}

int main()
{
    uint32_t *sink; unsigned long long *cyc;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&cyc, 8));
    const int iters = 256;
    printf("%-34s %s\n", "instruction", "clk per wave64 instruction per SIMD at 1 / 2 / 4 waves per SIMD");
#define X(N, ASM, NAME)                                                                                          \
    {                                                                                                            \
        printf("%-34s", NAME);                                                                                   \
        for (int wps = 1; wps <= 4; wps *= 2) {                                                                  \
            CK(hipMemset(cyc, 0, 8));                                                                            \
            hipLaunchKernelGGL((k_valu<N>), dim3(256), dim3(256 * wps), 0, 0, iters, sink, cyc);                 \
            CK(hipDeviceSynchronize());                                                                          \
            unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));                              \
            printf("  %6.2f", (double)c / ((double)iters * 16 * wps));                                           \
        }                                                                                                        \
        printf("\n");                                                                                            \
    }
    OPS(X)
    X(43, "", "v_cndmask_b32_e64 (SGPR-pair mask)")
    X(48, "", "v_mad_u64_u32")
    X(49, "", "v_mul_f64")
#undef X
    return 0;
}
