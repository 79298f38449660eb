/*
 * bevwarp.h -- C-ABI of libbevwarp.so, the MI355X (gfx950) surround-BEV warping engine.
 *
 * The reference (dyfcalid/CameraCalibration) is pure Python on top of cv2; it has no FFI of its own, so this header
 * DEFINES the boundary a maintainer binds with ctypes (INTEGRATION.md shows the stub).  Every entry point names
 * the reference call it replaces (file:line under the reference tree).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no exceptions cross the boundary.
 *   - every function returns 0 on success or a negative bevw_status; bevw_last_error() returns a thread-local text.
 *   - images are uint8, HWC interleaved BGR, C-contiguous (what cv2.imread hands the reference).
 *   - camera order is front, back, left, right (surroundBEV.py:285-286).
 *   - an opaque handle owns one HIP stream and all device tables; calls on one handle are serialised, distinct
 *     handles may be used from distinct threads.
 *   - "_device" variants take device pointers (frames already resident in HBM); the others copy host buffers.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails with BEVW_E_NO_DEVICE.
 */
#ifndef BEVWARP_H
#define BEVWARP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BEVW_ABI_VERSION 5

typedef enum bevw_status {
    BEVW_OK = 0,
    BEVW_E_INVALID = -1,    /* bad argument / bad state (e.g. run before build) */
    BEVW_E_NO_DEVICE = -2,  /* no HIP device, or the requested ordinal does not exist */
    BEVW_E_HIP = -3,        /* a HIP runtime call failed; text in bevw_last_error() */
    BEVW_E_NOMEM = -4
} bevw_status;

typedef enum bevw_camera_id { BEVW_FRONT = 0, BEVW_BACK = 1, BEVW_LEFT = 2, BEVW_RIGHT = 3 } bevw_camera_id;

/* Which device schedule bevw_run* uses.  All schedules produce identical bytes. */
typedef enum bevw_schedule {
    BEVW_SCHED_AUTO = 0,     /* tile-plan kernel when the plan fits (<= 2 contributing cameras per BEV pixel) */
    BEVW_SCHED_PER_PIXEL = 1,/* one thread per BEV pixel, loops the 4 cameras through LUT + mask (always valid) */
    BEVW_SCHED_TILE_PLAN = 2 /* register-resident contributor plan, batch loop inside the block */
} bevw_schedule;

/* The ten fields of the reference's argparse Namespace (surroundBEV.py:6-17) that shape the tables, plus placement. */
typedef struct bevw_config {
    int32_t frame_width;   /* FRAME_WIDTH  */
    int32_t frame_height;  /* FRAME_HEIGHT */
    int32_t bev_width;     /* BEV_WIDTH    */
    int32_t bev_height;    /* BEV_HEIGHT   */
    int32_t car_width;     /* CAR_WIDTH    */
    int32_t car_height;    /* CAR_HEIGHT   */
    double focal_scale;    /* FOCAL_SCALE  */
    double size_scale;     /* SIZE_SCALE   */
    int32_t blend;         /* BevGenerator(blend=...)   surroundBEV.py:283 */
    int32_t balance;       /* BevGenerator(balance=...) surroundBEV.py:283 */
    int32_t device;        /* HIP device ordinal */
    int32_t schedule;      /* bevw_schedule */
} bevw_config;

typedef struct bevw_handle bevw_handle;   /* a BevGenerator: 4 cameras + masks        (surroundBEV.py:282-325) */
typedef struct bevw_remapper bevw_remapper; /* one fixed-point remap table on the device (cv2.remap call sites)   */
typedef struct bevw_comm bevw_comm;       /* an RCCL communicator of one camera group (camera-per-GPU mode)        */
typedef struct bevw_jpeg bevw_jpeg;       /* a JPEG decode / encode context: one HIP stream + scratch (cv2.imread / cv2.imwrite) */

/* ---- library / device ------------------------------------------------------------------------------------ */
int bevw_abi_version(void);

/* OpenCV-version-sensitive arithmetic behind the reference's cv2 calls.  The reference pins no OpenCV version
 * ("opencv(>=3.4.2)", README.md:14); several of the primitives it calls changed their results across releases, so the choice
 * is a switch instead of a constant.  bevw_set_compat sets the process-wide DEFAULT; a handle takes a snapshot of the values in
 * bevw_build and keeps it (later calls, or calls from other threads, never change the results of a handle that exists); the stand-alone
 * entry points without a handle read the default at CALL time: bevw_color_balance (BEVW_COMPAT_ADDWEIGHTED) and
 * bevw_warp_perspective_u8c3 (BEVW_COMPAT_WARP):
 *   BEVW_COMPAT_FILLPOLY    cv2.fillPoly (surroundBEV.py:159,234): 1 = OpenCV >= 4.5.2 edge collection (default),
 *                           0 = OpenCV 2.4 .. 4.5.1 (edges between the raw vertices, left span end rounded up)
 *   BEVW_COMPAT_ADDWEIGHTED cv2.addWeighted(ch, k, 0, 0, 0, ch) (surroundBEV.py:52-54): 1 = evaluated in CV_64F (default),
 *                           0 = in CV_32F
 *   BEVW_COMPAT_WARP        cv2.warpPerspective with INTER_LINEAR on 8UC3 (extrinsicCalib.py:166-169) and on the 16UC1 undistort map
 *                           (surroundBEV.py:105-108, the `bev_map2` half of the LUT quirk): 0 = the classic kernels of OpenCV 2.4 ... 4.10
 *                           (positions quantised to 1/32 pixel, fixed-point / tabulated weights; default), an ODD value < 64 = one member of
 *                           a family of float32 kernels in the style OpenCV 4.11 introduced (position kept in float32, cvRound of a float
 *                           lerp).  The bits: 1 float family, 2 fused multiply-adds in the coordinates, 4 fused multiply-adds in the
 *                           interpolation, 8 (1 - t) a + t b instead of a + t (b - a), 16 coordinates in double, 32 multiply by 1 / w.
 *                           Which member, if any, reproduces a given cv2 >= 4.11 is decided by the implementation probes of
 *                           tests/golden/ (tests/test_cv2_goldens.py tries all of them): candidates, NOT a parity claim.
 *   BEVW_COMPAT_REMAP       cv2.remap of 8UC3 through fixed-point maps (surroundBEV.py:110-117, intrinsicCalib.py:193-195): the weighted
 *                           sum is exact either way; 0 = rounded half up, (S + 512) >> 10, the classic kernels (default), 1 = rounded
 *                           half to even, what a float kernel ending in cvRound gives.  1 runs the per-pixel schedule (slower).
 * tests/golden/README.md: how a golden file from a real cv2 decides them. */
#define BEVW_COMPAT_FILLPOLY 0
#define BEVW_COMPAT_ADDWEIGHTED 1
#define BEVW_COMPAT_WARP 2
#define BEVW_COMPAT_REMAP 3
#define BEVW_COMPAT_KEYS 4
int bevw_set_compat(int key, int value);   /* 0 = OK */
int bevw_get_compat(int key);              /* current value, < 0 on an unknown key */
int bevw_device_count(void);               /* 0 when no GPU is visible (never negative) */
const char *bevw_last_error(void);
int bevw_device_name(int device, char *buf, size_t buflen);

/* Raw device memory for callers that keep batches resident in HBM without any GPU framework (bench.py, tests). */
int bevw_malloc(int device, size_t nbytes, void **dptr);
int bevw_free(int device, void *dptr);
int bevw_memcpy_h2d(int device, void *dst, const void *src, size_t nbytes);
int bevw_memcpy_d2h(int device, void *dst, const void *src, size_t nbytes);
int bevw_memset(int device, void *dst, int value, size_t nbytes);
/* Rate of a plain device-to-device copy kernel over two fresh buffers of nbytes (GB/s of bytes READ + WRITTEN; streaming != 0: non-temporal
 * loads and stores): the measured yardstick bench.py reports beside the 8 TB/s specification peak.  Additive: no reference
 * counterpart. */
int bevw_device_copy_rate(int device, size_t nbytes, int reps, int streaming, double *gb_per_s_moved);

/* ---- BevGenerator: construction = table build (surroundBEV.py:283-294) ----------------------------------- */
int bevw_create(const bevw_config *cfg, bevw_handle **out);
/* Camera.__init__ loads K (3x3), D (4), H (3x3) float64 (surroundBEV.py:83-85). */
int bevw_set_camera(bevw_handle *h, int cam, const double K[9], const double D[4], const double H[9]);
/* Camera.get_undistort_maps + get_bev_maps (surroundBEV.py:98-108), Mask.get_mask (:156-159) or
 * BlendMask.__init__ (:165-188): every table is built by HIP kernels and stays on the device. */
int bevw_build(bevw_handle *h);
void bevw_destroy(bevw_handle *h);

/* Table read-back (tests, inspection).  Shapes: undistort maps [int(FH*SS)][int(FW*SS)], BEV tables [BH][BW]. */
int bevw_get_undistort_map(bevw_handle *h, int cam, int16_t *map1, uint16_t *map2); /* Camera.undistort_maps */
int bevw_get_lut(bevw_handle *h, int cam, int16_t *map1, uint16_t *map2);           /* Camera.bev_maps       */
int bevw_get_mask(bevw_handle *h, int cam, uint8_t *mask);  /* Mask.mask / BlendMask.mask (u8, before /255.0)   */
/* Projection mode of bevw_run* (BASELINE.json north_star: "inverts H and applies the K/D fisheye model per output
 * pixel ... with LDS-staged input tiles").  BEVW_PROJ_LUT (default) is the reference's path: the tables Camera.__init__ builds
 * (surroundBEV.py:82-102) compiled into the contributor plan -- bit-exact against the oracle.  BEVW_PROJ_ANALYTIC samples every BEV pixel at
 * the position the camera model gives (inverse homography + fisheye model in fp64, fractions kept to 2^-21 pixel) with float-weight bilinear
 * interpolation: no fixed-point tables, not the reference's arithmetic; equal to its NumPy specification to <= 1 LSB and judged against the
 * table path by PSNR (tests/test_analytic.py).  The calibration of a handle is fixed, so the projection is evaluated once per handle and
 * mode (on the GPU) and compiled into a unit schedule with wide fractions (csrc/bevw_unit.h); the per-frame kernel stages source texels
 * through LDS exactly as the table mode does and interpolates in fp32.  Balance handles, BEV widths that are not a multiple of 4 and
 * unaligned buffers run the per-pixel kernel that evaluates the model itself (fp64 throughout).  Masks, blend weights, balance and the car
 * are unchanged.  Call before or after bevw_build; not available on camera-shard handles or with an output pitch. */
#define BEVW_PROJ_LUT 0
#define BEVW_PROJ_ANALYTIC 1       /* fp64 projection: equals its NumPy specification (oracle/np_analytic.py) to <= 1 LSB, >= 99.9 % of the bytes */
#define BEVW_PROJ_ANALYTIC_F32 2   /* the same formulas in fp32 (positions good to ~1e-4 pixel): held against the fp64 mode by PSNR */
int bevw_set_projection(bevw_handle *h, int mode);
int bevw_plan_info(bevw_handle *h, int32_t info[8]);        /* [0] max contributors/pixel, [1] plan usable, [2] schedule in use */

/* ---- BevGenerator.__call__ (surroundBEV.py:312-325) ------------------------------------------------------ */
/* frames: [batch][4][FH][FW][3]; car: [BH][BW][3] or NULL (already padded, surroundBEV.py:28-41); out: [batch][BH][BW][3] */
int bevw_run(bevw_handle *h, const uint8_t *frames, int batch, const uint8_t *car, uint8_t *out);
int bevw_run_device(bevw_handle *h, const void *d_frames, int batch, const void *d_car, void *d_out);
/* Row pitch of the DEVICE-side BEV images (cv::cuda::GpuMat's `step`; the reference's host arrays, surroundBEV.py:312-325, stay dense).
 * A 1080-pixel row is 3240 bytes: in a dense image only one row in eight starts on a 64-byte memory sector, every other row run a
 * kernel block writes starts and ends inside a sector it shares with its neighbour, and partially written sectors are what the
 * stitch pays most for.  bevw_set_output_pitch (before bevw_build) makes every device image of the handle
 * [BH][pitch][3] with `pitch_pixels` >= BEV_WIDTH pixels per row: BEVW_PITCH_DENSE (default) = BEV_WIDTH, BEVW_PITCH_ALIGNED = BEV_WIDTH
 * rounded up to 64 pixels (rows of whole sectors), or an explicit multiple of 4.  With a pitch other than BEV_WIDTH the host entry points
 * (bevw_run, bevw_run_cameras) still take and return dense arrays -- rows are compacted inside the device-to-host copy -- while
 * bevw_run_device writes d_out as [batch][BH][bevw_output_pitch()][3] (bytes of the padding columns are unspecified) and still reads a
 * dense d_car.  Needs the tile-plan schedule; not available on camera-shard handles. */
#define BEVW_PITCH_DENSE 0
#define BEVW_PITCH_ALIGNED (-1)
int bevw_set_output_pitch(bevw_handle *h, int pitch_pixels);
int bevw_output_pitch(bevw_handle *h);   /* pixels per row of the handle's device images */
/* The reference's own call shape, bev(front, back, left, right, car) (surroundBEV.py:312, main.py:84): four separate
 * [FH][FW][3] host arrays (no packing copy on the host), one frame set, out [BH][BW][3]. */
int bevw_run_cameras(bevw_handle *h, const uint8_t *front, const uint8_t *back, const uint8_t *left, const uint8_t *right,
                     const uint8_t *car, uint8_t *out);

/* Camera.undistort / Camera.warp_homography / Camera.raw2bev (surroundBEV.py:110-117) on host images.
 * undistort: src [batch][FH][FW][3] -> dst [batch][int(FH*SS)][int(FW*SS)][3]
 * warp_homography: src [batch][src_h][src_w][3] -> dst [batch][BH][BW][3]
 * raw2bev: src [batch][FH][FW][3] -> dst [batch][BH][BW][3] */
int bevw_camera_undistort(bevw_handle *h, int cam, const uint8_t *src, int batch, uint8_t *dst);
int bevw_camera_warp_homography(bevw_handle *h, int cam, const uint8_t *src, int src_w, int src_h, int batch,
                                uint8_t *dst);
int bevw_camera_raw2bev(bevw_handle *h, int cam, const uint8_t *src, int batch, uint8_t *dst);

/* Mask.__call__ (surroundBEV.py:161-162, blend = 0: cv2.bitwise_and(img, img, mask=mask)) and BlendMask.__call__
 * (surroundBEV.py:279-280, blend = 1: (img * float32(mask / 255.0)).astype(uint8)) with the handle's mask of `cam`.
 * img / out: [batch][BH][BW][3].  Inside bevw_run the same arithmetic is fused into the stitch kernels. */
int bevw_apply_mask(bevw_handle *h, int cam, const uint8_t *img, int batch, uint8_t *out);

/* ---- camera-per-GPU mode: every rank stitches the cameras it owns, one exchange, the stitch rank adds ---------- */
/* The reference adds the four masked BEV images with cv2.add (surroundBEV.py:318-320) and the car with another
 * (:323-324); cv2.add saturates, so the sum is min(255, total) however the terms are grouped -- a rank may add its own
 * cameras first and the stitch rank the parts.  Never reduce with a wrapping/plain sum collective.
 *
 * bevw_set_camera_shard (before bevw_build): the handle owns `cams` (ascending, 1..4 of front/back/left/right); only
 * those need bevw_set_camera, only their tables are built, and frame sets shrink to [batch][ncams][FH][FW][3]. */
int bevw_set_camera_shard(bevw_handle *h, const int32_t *cams, int ncams);
/* x0, y0, x1, y1 of the owned masks (x widened to multiples of 4 pixels): the part a rank sends. */
int bevw_shard_box(bevw_handle *h, int32_t box[4]);
/* luminance_balance statistics (surroundBEV.py:60-66): d_vsums[batch][ncams] (uint64) = sum of V over each owned
 * frame.  The host all-gathers them into [batch][4] (camera order) for bevw_shard_run_device. */
int bevw_shard_vsums_device(bevw_handle *h, const void *d_frames, int batch, void *d_vsums);
/* Partial BevGenerator.__call__ (surroundBEV.py:312-320): luminance shift with the global means (balance = 1,
 * d_all_vsums[batch][4] uint64, else NULL), raw2bev, mask / blend weight and cv2.add over the owned cameras.
 * d_out: full-size [batch][BH][BW][3], zero outside the owned masks.  No white balance, no car: bevw_combine_device. */
int bevw_shard_run_device(bevw_handle *h, const void *d_frames, int batch, const void *d_all_vsums, void *d_out);
/* d_full [batch][BH][BW][3] -> d_packed [batch][y1-y0][x1-x0][3] (the bevw_shard_box window). */
int bevw_shard_pack_device(bevw_handle *h, const void *d_full, int batch, void *d_packed);
/* Stitch rank: out = cv2.add over the packed parts (boxes[k] = x0,y0,x1,y1), then color_balance when the handle was
 * created with balance = 1 (surroundBEV.py:321-322), then cv2.add(surround, car) when d_car != NULL (:323-324).
 * Works on any built handle of the same BEV geometry (shard or not). */
int bevw_combine_device(bevw_handle *h, const void *const *d_parts, const int32_t *boxes, int nparts, int batch,
                        const void *d_car, void *d_out);

/* ---- the exchange step of the camera-per-GPU mode over RCCL / xGMI (no reference code: the reference is single
 * process; BASELINE.json north_star "sharded one-camera-per-GPU ... with an RCCL gather over xGMI for the final stitch").
 * librccl.so is dlopen'ed on first use, so the library loads and every single-GPU entry point works without it; the environment
 * variable BEVW_RCCL_LIB (read once per process) names the library to load instead -- another RCCL build, or the stand-in with which the
 * test-suite runs the multi-rank branches on a box with fewer GPUs than ranks (tests/native/rccl_standin.cpp).
 * A communicator spans ONE camera group (1, 2 or 4 ranks; rank order = ascending camera order).  Every call below is
 * enqueued on the handle's own HIP stream: rank-local stitch, exchange and combine need no host synchronisation.
 *   bevw_comm_unique_id  group rank 0 creates the 128-byte id; the caller carries it to the other ranks (any out-of-band
 *                        channel: cameraShard.RcclGroup broadcasts it over the SocketGroup control channel)
 *   bevw_comm_create     ncclCommInitRank on `device`
 *   bevw_shard_allgather_vsums   balance only: ncclAllGather of the per-frame V sums, laid out as [batch][4]
 *   bevw_shard_gather_parts      grouped ncclSend (non-root) / ncclRecv (root) of the packed mask boxes; d_recv / bytes are
 *                                indexed by group rank, the root's own entry is ignored.  NEVER a summing collective:
 *                                cv2.add saturates (surroundBEV.py:318-320), a wrapping u8 reduce is wrong on every seam pixel
 *   bevw_comm_selftest   1-rank communicator: all-gather + send/receive to self (what a 1-GPU box can exercise) */
int bevw_comm_available(void);                          /* 1 when librccl.so could be loaded */
int bevw_comm_unique_id(uint8_t id[128]);
int bevw_comm_create(int device, int rank, int world, const uint8_t id[128], bevw_comm **out);
void bevw_comm_destroy(bevw_comm *c);
int bevw_shard_allgather_vsums(bevw_handle *h, bevw_comm *c, const void *d_vsums, int batch, void *d_all_vsums);
int bevw_shard_gather_parts(bevw_handle *h, bevw_comm *c, const void *d_packed, size_t my_bytes, int root, void *const *d_recv,
                            const size_t *bytes);
int bevw_comm_selftest(bevw_handle *h, bevw_comm *c, const void *d_src, void *d_dst, size_t nbytes);

/* Module-level helpers of surroundBEV.py, exposed because the reference exports them:
 * luminance_balance(images) (:57-79): frames [batch][4][H][W][3] -> same shape;
 * color_balance(image) (:43-55): image [batch][H][W][3] -> same shape. */
int bevw_luminance_balance(int device, const uint8_t *frames, int batch, int width, int height, uint8_t *out);
int bevw_color_balance(int device, const uint8_t *images, int batch, int width, int height, uint8_t *out);

/* Stream timing for bench.py: HIP events recorded on the handle's own stream. */
int bevw_sync(bevw_handle *h);
int bevw_timer_start(bevw_handle *h);
int bevw_timer_stop(bevw_handle *h, float *elapsed_ms); /* records + synchronises the stop event */
/* Lap marks: an event recorded on the engine's stream WITHOUT synchronising (slot 0..65535); elapsed time between two marks
 * after the caller's sync.  bench.py marks every step: per-step durations and their median instead of one mean. */
int bevw_timer_mark(bevw_handle *h, int slot);
int bevw_timer_between(bevw_handle *h, int slot_a, int slot_b, float *elapsed_ms);

/* ---- cv2.remap with fixed-point maps: InCalibrator.undistort (intrinsicCalib.py:193-195), ----------------- */
/* ---- Tools/undistort.py:50-52,66, Camera.undistort (surroundBEV.py:110-111) ------------------------------- */
/* Builds cv2.fisheye.initUndistortRectifyMap(K, D, I, K', (int(fw*size_scale), int(fh*size_scale)), CV_16SC2) with
 * K' = K, f *= focal_scale, c = (fw/2*size_scale + offset_h, fh/2*size_scale + offset_v)
 * (intrinsicCalib.py:90-103; offsets: Tools/undistort.py:45-46). */
int bevw_fisheye_remapper_create(int device, int frame_width, int frame_height, const double K[9], const double D[4],
                                 double focal_scale, double size_scale, double offset_h, double offset_v,
                                 bevw_remapper **out);
/* Same for the pinhole ("normal") model: cv2.initUndistortRectifyMap(K, D, I, K', size, CV_16SC2) with
 * D = k1 k2 p1 p2 [k3 [k4 k5 k6]] (n_dist = 4, 5 or 8)  (Normal._get_undistort_maps, intrinsicCalib.py:150-163). */
int bevw_pinhole_remapper_create(int device, int frame_width, int frame_height, const double K[9], const double *D, int n_dist,
                                 double focal_scale, double size_scale, double offset_h, double offset_v,
                                 bevw_remapper **out);
/* Wraps caller-made maps (map1 CV_16SC2 [dh][dw][2], map2 CV_16UC1 [dh][dw]) for sources of size src_w x src_h. */
int bevw_remapper_from_maps(int device, int src_w, int src_h, const int16_t *map1, const uint16_t *map2, int dst_w,
                            int dst_h, bevw_remapper **out);
int bevw_remapper_dims(bevw_remapper *r, int32_t dims[4]); /* src_w, src_h, dst_w, dst_h */
int bevw_remapper_get_maps(bevw_remapper *r, int16_t *map1, uint16_t *map2);
/* src [batch][src_h][src_w][3] -> dst [batch][dst_h][dst_w][3]; INTER_LINEAR, BORDER_CONSTANT 0 */
int bevw_remap(bevw_remapper *r, const uint8_t *src, int batch, uint8_t *dst);
int bevw_remap_device(bevw_remapper *r, const void *d_src, int batch, void *d_dst);
int bevw_remapper_sync(bevw_remapper *r);
int bevw_remapper_timer_start(bevw_remapper *r);
int bevw_remapper_timer_stop(bevw_remapper *r, float *elapsed_ms);
int bevw_remapper_timer_mark(bevw_remapper *r, int slot);
int bevw_remapper_timer_between(bevw_remapper *r, int slot_a, int slot_b, float *elapsed_ms);
void bevw_remapper_destroy(bevw_remapper *r);

/* ---- cv2.warpPerspective(src_8UC3, H, (dst_w, dst_h)): ExCalibrator.warp (extrinsicCalib.py:166-169) ------ */
int bevw_warp_perspective_u8c3(int device, const uint8_t *src, int src_w, int src_h, const double H[9], int dst_w,
                               int dst_h, int batch, uint8_t *dst);

/* ---- ExCalibrator pre-processing warps --------------------------------------------------------------------- */
/* CenterImage.translate (extrinsicCalib.py:54-59): cv2.warpAffine(img, [[1,0,shift_x],[0,1,shift_y]], (w, h)) with the
 * integer shifts the reference builds (image centre minus the picked point): dst(x,y) = src(x-shift_x, y-shift_y) or 0. */
int bevw_translate_u8c3(int device, const uint8_t *src, int width, int height, int shift_x, int shift_y, int batch,
                        uint8_t *dst);
/* cv2.resize(img, (0,0), fx=fx, fy=fy), INTER_LINEAR, 8UC3 (ScaleImage.__call__, extrinsicCalib.py:125).
 * bevw_resize_dsize gives the output size (cvRound(w*fx), cvRound(h*fy)); dst: [batch][dsize[1]][dsize[0]][3].
 * The pad / centre-crop back to the input size (extrinsicCalib.py:101-120) is a host-side copy in the mirror. */
int bevw_resize_dsize(int src_w, int src_h, double fx, double fy, int32_t dsize[2]);
int bevw_resize_linear_u8c3(int device, const uint8_t *src, int src_w, int src_h, double fx, double fy, int batch,
                            uint8_t *dst);

/* ---- JPEG either side of the path: cv2.imread (main.py:74-77, Tools/undistort.py:65) and ------- */
/* ---- cv2.imwrite (surroundBEV.py:340, Tools/undistort.py:73, extrinsicCalib.py:211) -------------------------------------------------------- */
/* For ".jpg" both cv2 calls are libjpeg(-turbo) with its defaults; the kernels reproduce that library bit for bit (baseline Huffman,
 * jpeg_idct_islow, fancy upsampling, ycc_rgb_convert; encode: rgb_ycc_convert, h2v2_downsample, jpeg_fdct_islow, Annex-K tables at
 * cv2's quality 95, 4:2:0) -- oracle/jpegoracle.c, pinned against Pillow's libjpeg-turbo.  Un-stuffing and entropy decoding run ON THE GPU
 * (self-synchronising parallel Huffman decoding, csrc/bevw_jpeg.h); the host parses the markers in front of the scan and copies the
 * entropy-coded bytes as they are into pinned memory for the H2D transfer.  Supported: baseline / extended-sequential Huffman, 8 bit, one interleaved scan, grey (expanded to BGR,
 * as IMREAD_COLOR does) or YCbCr with luma 1x1 / 2x1 / 2x2, restart intervals.  Progressive, arithmetic, CMYK, 12-bit and multi-scan files
 * are REFUSED (BEVW_E_INVALID, reason in bevw_last_error()): there is no CPU decoder behind this.  An EXIF orientation tag is applied on the GPU as
 * cv2.imread applies it (one orientation per batch; orientations 5 .. 8 swap width and height of the decoded images: bevw_jpeg_decode_info reports
 * the size of the result, bevw_jpeg_probe the stored size and the tag).
 *
 *   bevw_jpeg_probe            header only: info = width, height, components, luma h, luma v, restart interval, EXIF orientation, 0
 *   bevw_jpeg_decode_stage     n files of ONE geometry: parse the headers, copy the entropy-coded bytes to pinned memory, enqueue the H2D copies
 *                              (what is resident afterwards is the files' entropy-coded data as it is in the files)
 *   bevw_jpeg_decode_run_device  enqueue the decode of the staged batch (un-stuffing, entropy decoding, inverse DCT, colour); image i is written as BGR rows of row_pitch_bytes at
 *                              d_out + i * image_stride_bytes -- with image_stride = FH*FW*3 the n = 4*batch images ARE the
 *                              [batch][4][FH][FW][3] frame sets bevw_run_device reads (files ordered front, back, left, right per set)
 *   bevw_jpeg_decode           both + copy to a dense host array [n][h][w][3]  (= [cv2.imread(f) for f in files])
 *   bevw_jpeg_decode_info      info = images, width, height, subsequences, fixed-point rounds (max over images), entropy-coded bytes,
 *                              images whose entropy-coded data ended before the image was complete (truncated / corrupt files: their
 *                              missing blocks are undefined; bevw_jpeg_decode turns a non-zero count into BEVW_E_INVALID), distinct
 *                              Huffman table sets
 *   bevw_jpeg_get_planes       test hook: the sample planes of image `index` after the inverse DCT (Y, Cb, Cr; whole MCUs)
 *   bevw_jpeg_encode_run_device  enqueue cv2.imwrite x n of device images (BGR rows of row_pitch_bytes, e.g. the padded rows of
 *                              BEVW_PITCH_ALIGNED); sampling 0x22 = 4:2:0 (cv2's default), 0x21 = 4:2:2, 0x11 = 4:4:4
 *   bevw_jpeg_encoded_sizes / _copy   synchronise; byte count of every file; one complete file (SOI ... EOI) to host memory
 *   bevw_jpeg_encoded_fetch    synchronise; ALL files of the last encode back to back in dst (file i at dst + offsets[i], offsets[n] = total):
 *                              a gather on the device and one device-to-host copy
 *   bevw_jpeg_wait_engine / bevw_wait_jpeg   order the codec's stream behind an engine's stream / the engine's behind the codec's without the
 *                              host: decode -> bevw_run_device -> encode as one asynchronous chain (BevGenerator.jpeg_stream)
 *   bevw_jpeg_encode           host images in, files out (out + i * cap_each, sizes[i]); cap_each >= bevw_jpeg_encode_bound
 * All "run" calls are asynchronous on the context's own stream; bevw_jpeg_sync waits; timer marks as for handles. */
#define BEVW_JPEG_420 0x22
#define BEVW_JPEG_422 0x21
#define BEVW_JPEG_444 0x11
int bevw_jpeg_probe(const uint8_t *data, size_t len, int32_t info[8]);
int bevw_jpeg_create(int device, bevw_jpeg **out);
void bevw_jpeg_destroy(bevw_jpeg *j);
int bevw_jpeg_decode_stage(bevw_jpeg *j, const uint8_t *const *data, const size_t *len, int n);
int bevw_jpeg_decode_run_device(bevw_jpeg *j, void *d_out, size_t image_stride_bytes, size_t row_pitch_bytes);
int bevw_jpeg_decode(bevw_jpeg *j, const uint8_t *const *data, const size_t *len, int n, uint8_t *out);
int bevw_jpeg_decode_info(bevw_jpeg *j, int64_t info[8]);
int bevw_jpeg_get_planes(bevw_jpeg *j, int index, uint8_t *planes);
int bevw_jpeg_encode_bound(int width, int height, int sampling, size_t *bound);
int bevw_jpeg_encode_run_device(bevw_jpeg *j, const void *d_bgr, int n, int width, int height, size_t image_stride_bytes,
                                size_t row_pitch_bytes, int quality, int sampling);
int bevw_jpeg_encoded_sizes(bevw_jpeg *j, size_t *sizes);
int bevw_jpeg_encoded_copy(bevw_jpeg *j, int index, uint8_t *dst, size_t cap);
int bevw_jpeg_encoded_fetch(bevw_jpeg *j, uint8_t *dst, size_t cap, size_t *offsets);
int bevw_jpeg_wait_engine(bevw_jpeg *j, bevw_handle *h);
int bevw_wait_jpeg(bevw_handle *h, bevw_jpeg *j);
int bevw_jpeg_encode(bevw_jpeg *j, const uint8_t *bgr, int n, int width, int height, int quality, int sampling, uint8_t *out,
                     size_t cap_each, size_t *sizes);
int bevw_jpeg_sync(bevw_jpeg *j);
int bevw_jpeg_timer_mark(bevw_jpeg *j, int slot);
int bevw_jpeg_timer_between(bevw_jpeg *j, int slot_a, int slot_b, float *elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* BEVWARP_H */
