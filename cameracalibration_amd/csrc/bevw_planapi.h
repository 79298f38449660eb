// bevw_planapi.h -- what the rest of the library sees of the tile plan (defined in bevwarp_plan.hip, the only translation unit that
// instantiates the plan kernels of bevw_plan.h / bevw_unit.h).  Every function returns a BEVW_* status and leaves its message in
// bevw_last_error().
#pragma once
#include "bevw_host.h"
#include "bevw_plan.h"

namespace bevw {

// compile LUT + masks into a plan (table kernels, unit compiler on the host).  out_pitch: pixels per output row when the caller's images
// are pitched (0: dense); blend: the handle applies blend weights (its units carry no two-quad two-contributor class)
int plan_build(Plan &p, hipStream_t st, const StitchTables &T, int fw, int fh, int bw, int bh, int ncams, int out_pitch, bool blend);

// one step: see plan_stitch_impl (bevw_plan.h)
int plan_stitch(Plan &p, hipStream_t st, const uint8_t *d_frames, int batch, bool blend, bool balance, const int *d_deltas, const HsvTables *d_tab,
                const uint8_t *d_car, unsigned long long *d_chsums, uint8_t *d_out, bool sums = false, int psums_frames = 0, int psums_first = 0,
                const uint8_t *d_scratch = nullptr);

// balance: luminance round trip of the sampled texel groups of the raw frames into the compact scratch (p.compact_stride bytes per frame set)
int plan_lum_groups(const Plan &p, hipStream_t st, const uint8_t *d_frames, uint8_t *d_scratch, int batch, const int *d_deltas, const HsvTables *d_tab);

// rows of bw pixels -> rows of pitch pixels (the car sprite of a pitched handle)
int plan_pad_image(hipStream_t st, const uint8_t *d_src, int bw, int pitch, int bh, uint8_t *d_dst);

// analytic projection modes: the WIDE unit plan compiled from the projection map of every camera (sxy: int16 [bh][bw][2] top-left texels,
// frac: uint32 [bh][bw][2] 21-bit fractions, mask: uint8 [bh][bw]; host copies), and its launch
int plan_build_wide(Plan &p, const std::vector<int16_t> sxy[4], const std::vector<uint32_t> frac[4], const std::vector<uint8_t> mask[4], int fw, int fh,
                    int bw, int bh, bool blend);
int plan_stitch_wide(const Plan &p, hipStream_t st, const uint8_t *d_frames, int batch, bool blend, const uint8_t *d_car, uint8_t *d_out);

bool plan_units_enabled();   // BEVW_PLAN_UNITS (default 1)

}  // namespace bevw
