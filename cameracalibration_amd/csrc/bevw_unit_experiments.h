// bevw_unit_experiments.h -- TIMING EXPERIMENTS of the unit kernel.  They produce WRONG PIXELS and are never part of libbevwarp.so:
// bevw_unit.h includes this file only under -DBEVW_EXPERIMENT=<n>, which cameracalibration_amd/build.py accepts only for tagged
// variant builds (build_var/libbevwarp_<tag>.so).  Results: profiles/r03 .. r06 README.md.
//
// BEVW_EXPERIMENT = 1 / 2 / 3: the MEMORY-ONLY REPLAY of plan_unit_run -- the frame step keeps exactly the kernel's own loads and stores
// (same addresses, same instructions, same policies) and drops the conversion, the LDS patch, the interpolation and the barriers:
//   1  loads and stores     2  loads only (a store that never fires keeps them alive)     3  stores only
// What the replay takes is what the memory system gives this request stream; what the kernel takes above it is issue time, LDS and
// barriers.
//
// (Retired in round 6, results recorded under profiles/r05: the LDS-conflict ablation BEVW_UNIT_ABL_LDS -3.5 %, the barrier ablation
// BEVW_UNIT_ABL_BARRIER -1.2 %, early stores +1.7 %, wave priorities around the memory instructions 0 %, sums without the wave reduction.)
#pragma once

// the body of plan_unit_run's `frame(b, ring)` lambda; every name it uses is a local of plan_unit_run
#define BEVW_EXPERIMENT_FRAME(b, ring)                                                                                                  \
    uint32_t keep = 0;                                                                                                                  \
    _Pragma("unroll") for (int r = 0; r < GR; ++r) keep ^= pf[ring][r].x ^ pf[ring][r].y ^ pf[ring][r].z ^ pf[ring][r].w;                \
    if (BEVW_EXPERIMENT != 3) issue(b + D, ring);                                                                                       \
    uint8_t *img = a.out + (size_t)frame_of(b) * img_bytes;                                                                             \
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(img, 0, (uint32_t)img_bytes, kBufferWord3);                     \
    _Pragma("unroll") for (int j = 0; j < NQ; ++j) {                                                                                    \
        if (BEVW_EXPERIMENT == 2 && keep != 0x12345679u) continue;                                                                      \
        store_slot(j, keep, keep + i0[j][0][0], keep, ro);                                                                              \
    }
