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
//
// BEVW_EXPERIMENT = 4: the BLOCK TIMELINE of the product kernel (right pixels): every block of k_plan_units records its start and end on the
// constant 100 MHz clock (s_memrealtime) and the hardware ids of the CU it ran on; bevw_experiment_trace() (exported by experiment builds only,
// not in include/bevwarp.h) copies the records of the LAST launch to the host.  tools/block_timeline.py turns them into the idle tail, the
// load imbalance between the XCDs and the duration of a block by unit class.
#pragma once

#if BEVW_EXPERIMENT == 4   // (this file is included inside namespace bevw)
struct UnitTraceRec { unsigned long long t0, t1; uint32_t hw_id, xcc_id, unit_class, pad; };
constexpr uint32_t kUnitTraceCap = 1u << 16;
__device__ UnitTraceRec g_unit_trace[kUnitTraceCap];
#define BEVW_EXPERIMENT_TRACE_BEGIN() const unsigned long long bevw_trace_t0 = wall_clock64()
#define BEVW_EXPERIMENT_TRACE_END(a)                                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x < kUnitTraceCap) {                                                                               \
        UnitTraceRec r;                                                                                                                 \
        r.t0 = bevw_trace_t0; r.t1 = wall_clock64();                                                                                    \
        r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4); r.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);                      \
        uint32_t chunk_, group_;                                                                                                        \
        r.unit_class = plan_block_map(a, blockIdx.x, chunk_, group_) && (int)group_ < a.nlist ? (a.tile_list[group_] >> 28) | (chunk_ << 8) : 0xffu; \
        r.pad = 0;                                                                                                                      \
        g_unit_trace[blockIdx.x] = r;                                                                                                   \
    }
#endif

#if BEVW_EXPERIMENT >= 1 && BEVW_EXPERIMENT <= 3
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
#endif
