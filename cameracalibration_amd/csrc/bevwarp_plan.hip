// bevwarp_plan.hip -- the tile plan of libbevwarp.so: the one translation unit that instantiates the per-frame stitch kernels
// (bevw_plan.h: k_stitch_plan; bevw_unit.h: k_plan_units, k_plan_unit_wide) and the plan compiler's kernels.  Interface: bevw_planapi.h.
#include "bevw_planapi.h"

namespace bevw {

static UnitTuning unit_tuning_env()
{
    UnitTuning t;
    if (const char *s = getenv("BEVW_UNIT_GROUPS")) t.max_groups = atoi(s);
    if (const char *s = getenv("BEVW_UNIT_ROOT_W")) t.root_w = atoi(s);
    if (const char *s = getenv("BEVW_UNIT_ROOT_H")) t.root_h = atoi(s);
    if (const char *s = getenv("BEVW_UNIT_MIN_W")) t.min_w = atoi(s);
    if (const char *s = getenv("BEVW_UNIT_LINE_COST")) t.line_cost = atoi(s);
    if (const char *s = getenv("BEVW_UNIT_SECTOR_COST")) t.sector_cost = atoi(s);
    if (const char *s = getenv("BEVW_UNIT_ALIGN_LINES")) t.align_lines = atoi(s);
    if (const char *s = getenv("BEVW_UNIT_OWN_EMPTY")) t.own_empty = atoi(s);
    if (const char *s = getenv("BEVW_UNIT_SKEW")) t.skew = atoi(s);
    if (const char *s = getenv("BEVW_UNIT_ROW_ORDER")) t.row_order = atoi(s);
    if (const char *s = getenv("BEVW_UNIT_OWN_PADDING")) t.own_padding = atoi(s);
    if (const char *s = getenv("BEVW_UNIT_STAGGER")) t.stagger = atoi(s);
    if (const char *s = getenv("BEVW_UNIT_RUN_COST")) t.run_cost = atoi(s);
    if (const char *s = getenv("BEVW_UNIT_BIG")) t.big_class = atoi(s);
    if (const char *s = getenv("BEVW_UNIT_WIDE_DOUBLE")) t.wide_double = atoi(s);
    return t;
}

// tuning knobs for experiments (defaults are the shipped configuration)
static const PlanTuning &plan_tuning()
{
    static const PlanTuning tune = [] {
        PlanTuning t;
        if (const char *s = getenv("BEVW_PLAN_NB")) t.nb = atoi(s);
        if (const char *s = getenv("BEVW_PLAN_XCDMAP")) t.xcd_map = atoi(s);
        if (const char *s = getenv("BEVW_PLAN_UNITS")) t.units = atoi(s);
        return t;
    }();
    return tune;
}

bool plan_units_enabled() { return plan_tuning().units != 0; }

int plan_build(Plan &p, hipStream_t st, const StitchTables &T, int fw, int fh, int bw, int bh, int ncams, int out_pitch, bool blend)
{
    static const UnitTuning unit_tune = unit_tuning_env();
    UnitTuning tune = unit_tune;
    (void)blend;   // (rounds 3 - 5 compiled blend handles without the two-quad two-contributor class: its float blend variant needed 177+ VGPRs;
                   // with round 6's integer weights it fits the kernel's budget, bevw_unit.h: plan_unit_any)
    // rows of whole sectors (an output pitch): column cuts on sector boundaries are free, all others split a sector for good -> a higher
    // price per write sector (3 -> 8: -0.4 ... -2 % on config 3, -2 % on the 4K rig, nothing slower; profiles/r03/sweeps.log).  The dense
    // layout keeps 3: there every cut shares sectors and the price only drives the source lines up (40 k -> 50 k per frame)
    if (out_pitch > 0 && !getenv("BEVW_UNIT_SECTOR_COST")) tune.sector_cost = 8;
    // One-camera plans (the fisheye remapper, BASELINE config 2: 378 units x 8 chunks = 3.9 rounds of blocks over the chip's 768 slots) are
    // launched longest unit first: with so few rounds the tail of the spatial order costs more than the neighbours' shared lines return --
    // batch 16 / 32 / 64 / 128: -4 ... -5.5 % (0.0804 -> 0.0765 ms at 64), batch 256 +1.8 %.  The 4-camera plans keep the spatial order
    // (config 3 +4 % with it, the 4K rig +5.6 %).  profiles/r06/call9..11
    if (ncams == 1 && !getenv("BEVW_UNIT_ROW_ORDER")) tune.row_order = 4;
    hipError_t e = plan_build_impl(p, st, T, fw, fh, bw, bh, ncams, plan_tuning().units != 0, tune, out_pitch);
    if (e != hipSuccess) return fail(BEVW_E_HIP, "tile-plan build failed: %s", hipGetErrorString(e));
    return BEVW_OK;
}

int plan_stitch(Plan &p, hipStream_t st, const uint8_t *d_frames, int batch, bool blend, bool balance, const int *d_deltas, const HsvTables *d_tab,
                const uint8_t *d_car, unsigned long long *d_chsums, uint8_t *d_out, bool sums, int psums_frames, int psums_first, const uint8_t *d_scratch)
{
    hipError_t e = plan_stitch_impl(p, st, d_frames, batch, blend, balance, d_deltas, d_tab, d_car, d_chsums, d_out, plan_tuning(), sums, psums_frames,
                                    psums_first, d_scratch);
    if (e != hipSuccess) return fail(BEVW_E_HIP, "tile-plan stitch launch failed: %s", hipGetErrorString(e));
    return BEVW_OK;
}

int plan_lum_groups(const Plan &p, hipStream_t st, const uint8_t *d_frames, uint8_t *d_scratch, int batch, const int *d_deltas, const HsvTables *d_tab)
{
    hipError_t e = plan_lum_band(p, st, d_frames, d_scratch, batch, d_deltas, d_tab);
    if (e != hipSuccess) return fail(BEVW_E_HIP, "k_lum_groups launch failed: %s", hipGetErrorString(e));
    return BEVW_OK;
}

int plan_pad_image(hipStream_t st, const uint8_t *d_src, int bw, int pitch, int bh, uint8_t *d_dst)
{
    const size_t n = (size_t)pitch * bh * 3;
    hipLaunchKernelGGL(k_plan_pad, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_src, bw, pitch, bh, d_dst);
    return launch_check("k_plan_pad");
}

int plan_build_wide(Plan &p, const std::vector<int16_t> sxy[4], const std::vector<uint32_t> frac[4], const std::vector<uint8_t> mask[4], int fw, int fh,
                    int bw, int bh, bool blend)
{
    plan_release(p);
    p.ncams = 4;
    p.fw = fw; p.fh = fh; p.bw = bw; p.bh = bh; p.pitch = bw;
    p.tiles_x = (bw + 4 * kPlanLX - 1) / (4 * kPlanLX); p.tiles_y = (bh + kPlanLY - 1) / kPlanLY; p.ntiles = p.tiles_x * p.tiles_y;
    std::vector<uint32_t> hdr = unit_host_headers(sxy, mask, 4, fw, fh, bw, bh, p.tiles_x, p.tiles_y);
    UnitTuning tune = unit_tuning_env();
    if (tune.max_groups > kUnitMaxGroups - 1) tune.max_groups = kUnitMaxGroups - 1;   // one group slot stays free: the zeros of pixels without a contributor
    tune.skew = 0;
    (void)blend;
    std::vector<uint16_t> no_codes[4];
    UnitPlanHost up;
    unit_compile(sxy, no_codes, mask, 4, fw, fh, bw, bh, bw, p.tiles_x, p.tiles_y, hdr, up, tune, frac);
    if (up.desc.empty()) return BEVW_OK;
    hipError_t e = plan_upload_units(p, up);
    if (e != hipSuccess) return fail(BEVW_E_HIP, "wide unit plan upload failed: %s", hipGetErrorString(e));
    p.un_skew = 0;
    std::vector<uint32_t> left;
    for (size_t t = 0; t < hdr.size(); ++t)
        if (!(hdr[t] & kHdrBlock)) left.push_back((uint32_t)t);
    e = plan_upload_list(left, &p.list_slow);
    if (e != hipSuccess) return fail(BEVW_E_HIP, "wide unit plan upload failed: %s", hipGetErrorString(e));
    p.n_slow = (int)left.size();
    return BEVW_OK;
}

int plan_stitch_wide(const Plan &p, hipStream_t st, const uint8_t *d_frames, int batch, bool blend, const uint8_t *d_car, uint8_t *d_out)
{
    hipError_t e = plan_unit_wide_launch(p, st, d_frames, batch, blend, d_car, d_out, plan_tuning());
    if (e != hipSuccess) return fail(BEVW_E_HIP, "k_plan_unit_wide launch failed: %s", hipGetErrorString(e));
    return BEVW_OK;
}

}  // namespace bevw

#ifdef BEVW_EXPERIMENT_TRACE_END
// experiment builds only (bevw_unit_experiments.h, BEVW_EXPERIMENT=4): the block records of the last k_plan_units launch
extern "C" int bevw_experiment_trace(void *out, size_t nbytes)
{
    if (nbytes > sizeof(bevw::UnitTraceRec) * bevw::kUnitTraceCap) nbytes = sizeof(bevw::UnitTraceRec) * bevw::kUnitTraceCap;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(bevw::g_unit_trace), nbytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
