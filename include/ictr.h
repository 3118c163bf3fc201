/*
 * ictr.h -- C-ABI of the MI355X-native Gauss-Newton photometric tracker.
 *
 * Drop-in boundary for the per-frame tracking hot path of catree/InvCompCamTrack. The reference
 * has NO C-ABI for this path: its boundary is the C++ class API in namespace CTR
 * (camera.h:19-31, pose.h:18-40, odometer.h:21-30, utilities.h:46-82) that the two CLI drivers
 * link statically (run_io_reprojection_test.cpp:189-223, run_track_nposes.cpp:185-259). This header
 * mirrors those classes one method per function, with opaque handles, plain pointers and sizes,
 * following the conventions of the reference's only real FFI (misc_src/triang.c + its ctypes
 * callers, func_util_geom.py:582-606): caller-owned C-contiguous float32/float64 buffers, results
 * written into caller-allocated outputs. Differences, all deliberate:
 *   - every function returns an int status (0 = ok) instead of void; nothing throws across the ABI;
 *     ictr_last_error() gives the message. (The reference ignores all errors.)
 *   - counts are int64_t (the reference's ctypes callers pass c_longlong against C int).
 *   - image pyramids are device-resident handles (ictr_pyramid) built by HIP kernels, replacing the
 *     cv::Mat-typed util_constructpyramide (utilities.h:63-64); ictr_odometer_setpose_host keeps the
 *     reference's literal "const float** level pointer" signature for callers that own host planes.
 *   - ictr_batch_* runs B independent tracking problems per launch (the run_track_nposes pose-sample
 *     axis, run_track_nposes.cpp:193); an ictr_odometer is a batch of one.
 *
 * include/ctr_shim.hpp re-creates namespace CTR {CamClass, PoseClass, OdometerClass} on top of
 * these functions; INTEGRATION.md shows the ctypes binding.
 *
 * All hot-path arithmetic runs in hand-written HIP kernels for gfx950; there is no CPU fallback:
 * every entry point that needs the GPU fails with ICTR_ERR_NO_DEVICE when none is usable.
 */
#ifndef ICTR_H
#define ICTR_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICTR_OK 0
#define ICTR_ERR_INVALID 1   /* bad argument */
#define ICTR_ERR_NO_DEVICE 2 /* no usable HIP device */
#define ICTR_ERR_HIP 3       /* a HIP runtime call failed */
#define ICTR_ERR_STATE 4     /* call order violated (e.g. TrackPose before SetPose) */

/* optparam, utilities.h:46-61 -- field order and types preserved verbatim */
typedef struct ictr_optparam {
  int maxpttrack;   /* SoA stride M of all point arrays; drivers round it up to a multiple of 4 */
  int psz;          /* patch size P */
  int pszd2;        /* P/2 */
  int pszd2m3;      /* P + P/2 - 1 */
  int novals;       /* P*P */
  int lv_f;         /* coarsest pyramid level (first processed) */
  int lv_l;         /* finest pyramid level (last processed) */
  bool donorm;      /* point cloud + pose normalisation */
  bool dopatchnorm; /* patch mean subtraction */
  int maxiter;
  float normdp_ratio;
  int verbosity;
} ictr_optparam;

/* fills the derived fields exactly as run_io_reprojection_test.cpp:112-126 does */
int ictr_optparam_init(ictr_optparam *op, int lv_f, int lv_l, int psz, int maxiter, float normdp_ratio,
                       int donorm, int dopatchnorm, int maxpttrack, int verbosity);

const char *ictr_last_error(void);
int ictr_version(void);
/* number of usable HIP devices (0 when none); never fails */
int ictr_device_count(void);
int ictr_set_device(int device);
/* measured streaming-read bandwidth of the current GPU in GB/s (bytes >= 1 MiB read reps times with wide loads):
 * the practical HBM ceiling to quote next to the vendor peak in roofline reports */
int ictr_stream_read_bandwidth(size_t bytes, int reps, double *gbps_out);
/* inspection: the transposing wave reduction of the resident-iteration kernel alone, on caller data.
 * vals[64 lanes][64 values], value index = 2 * patch + kind -> out[lane] = sum over the 64 lanes of value
 * 2 * patch_of_lane[lane] + kind_of_lane[lane] (all host arrays; 64 entries each) */
int ictr_debug_transpose_reduce(const float *vals, float *out, int *patch_of_lane, int *kind_of_lane,
                                int patches_per_wave /* 16 or 32: the kernel's two instantiations; 16 uses values 0..31 */);

/* ------------------------------------------------------------------ CamClass (camera.h:19-31, camera.cpp:14-45) */
typedef struct ictr_cam ictr_cam;
int ictr_cam_create(ictr_cam **out, int noscales, const float *fc, const float *cc, const int *wh, int padding);
void ictr_cam_destroy(ictr_cam *cam);
float ictr_cam_getfx(const ictr_cam *cam, int sc);
float ictr_cam_getfy(const ictr_cam *cam, int sc);
float ictr_cam_getcx(const ictr_cam *cam, int sc);
float ictr_cam_getcy(const ictr_cam *cam, int sc);
float ictr_cam_getswo(const ictr_cam *cam, int sc);
float ictr_cam_getsho(const ictr_cam *cam, int sc);
float ictr_cam_getsw(const ictr_cam *cam, int sc);
float ictr_cam_getsh(const ictr_cam *cam, int sc);

/* ------------------------------------------------------------------ utilities.h:84-241 (SE(3) exp / log, host) */
void ictr_se3_coeff_to_group_f(float *G12, const float *p6);
void ictr_se3_coeff_to_group_d(double *G12, const double *p6);
void ictr_se3_group_to_coeff_f(float *p6, const float *G12);
void ictr_se3_group_to_coeff_d(double *p6, const double *G12);
/* Hes.fullPivLu().solve(sumsd), odometer.cpp:509-515 (host copy of the device routine, for callers/tests) */
void ictr_solve6(const float *H36, const float *b6, float *x6);

/* ------------------------------------------------------------------ util_constructpyramide (utilities.cpp:14-52) */
typedef struct ictr_pyramid ictr_pyramid;
/* img: host, row-major f32, w x h. Builds lv_f+1 levels on the device: 2x2 box down-sampling, [-1 0 1]
 * gradients with reflect-101 border, replicate (image) / zero (gradient) padding by `pad` pixels.
 * getgrad: 0 image levels only (a "new" frame); 1 image + gradient planes (the reference's getgrad = true); 2 image
 * levels only, to be used as a REFERENCE frame: the tracker's 8x8 setup kernel forms the gradient patches on the fly
 * from the image plane (same subtraction, same blend, same bits as with the planes) -- a quarter of the memory and of
 * the build's bytes; accepted by trackings with psz 8 and no robustness option (they run the 8x8 per-iteration /
 * resident forms; small problems lose the one-launch tracker), refused with ICTR_ERR_STATE elsewhere. */
int ictr_pyramid_create(ictr_pyramid **out, const float *img, int w, int h, int lv_f, int getgrad, int pad);
/* same, img already in device memory (stays caller-owned; only read during the call) */
int ictr_pyramid_create_device(ictr_pyramid **out, const float *img_dev, int w, int h, int lv_f, int getgrad,
                               int pad, void *hip_stream);
/* Refill an existing pyramid from a new frame of the same size (the per-frame util_constructpyramide of a video loop,
 * run_track_nposes.cpp:180 once per image of a sequence): no allocation, the planes keep their addresses. img_dev: device memory, read by kernels
 * enqueued on hip_stream; img: host memory, copied to a device staging buffer first (has left `img` on return). */
int ictr_pyramid_rebuild_device(ictr_pyramid *pyr, const float *img_dev, void *hip_stream);
int ictr_pyramid_rebuild(ictr_pyramid *pyr, const float *img, void *hip_stream);
/* adopts caller-built host planes (the reference's img_pyr / dx_pyr / dy_pyr arrays); dx/dy may be NULL */
int ictr_pyramid_create_from_host_planes(ictr_pyramid **out, const float **img_pyr, const float **dx_pyr,
                                         const float **dy_pyr, int w, int h, int lv_f, int pad);
void ictr_pyramid_destroy(ictr_pyramid *pyr);
int ictr_pyramid_levels(const ictr_pyramid *pyr);
/* padded plane size of one level */
int ictr_pyramid_level_dims(const ictr_pyramid *pyr, int level, int *sw, int *sh);
/* which: 0 image, 1 dx, 2 dy */
int ictr_pyramid_download(const ictr_pyramid *pyr, int level, int which, float *host_out);
const float *ictr_pyramid_device_plane(const ictr_pyramid *pyr, int level, int which);

/* util_getPatch / util_getPatch_grad (utilities.cpp:55-113, 115-189), batched over K centres.
 * mids: host SoA x[K] then y[K]; outputs host, K*psz*psz each, patch-major. */
int ictr_get_patch(const ictr_pyramid *pyr, int level, const float *mids, int64_t K, int psz, int dopatchnorm,
                   float *out);
int ictr_get_patch_grad(const ictr_pyramid *pyr, int level, const float *mids, int64_t K, int psz, int dopatchnorm,
                        float *out, float *out_dx, float *out_dy);
/* run_track_nposes.cpp:271-355 -- the per-point patch correlation that scores a pose sample, on the device.
 * For each of K points: patches around its position in the backward-most, the reference and the forward-most frame
 * (util_getPatch with mean subtraction, :281), each divided by its norm; corr = max(0, (max(0, <b,r>) w_back +
 * max(0, <r,f>) w_fwd) / (w_back + w_fwd)) with the reference's strict validity tests (:290-305): -1 when the
 * reference position is outside, a term dropped (weight 0) when its frame's position is outside, 0 for NaN.
 * mids: host SoA x_back[K] y_back[K] x_ref[K] y_ref[K] x_fwd[K] y_fwd[K] at `level`; out_corr: K floats. */
int ictr_ncc_score(const ictr_pyramid *pyr_back, const ictr_pyramid *pyr_ref, const ictr_pyramid *pyr_fwd, int level,
                   const float *mids, int64_t K, int psz, float w_back, float w_fwd, float *out_corr);

/* ------------------------------------------------------------------ PoseClass (pose.h:18-40) */
typedef struct ictr_pose ictr_pose;
/* cam and op are held by pointer for the object's lifetime, like the reference (pose.cpp:14-18);
 * run_track_nposes.cpp:281 relies on that aliasing when it flips op.dopatchnorm. */
int ictr_pose_create(ictr_pose **out, const ictr_cam *cam, const ictr_optparam *op);
void ictr_pose_destroy(ictr_pose *pose);
int ictr_pose_setpose_se3(ictr_pose *pose, const double *p_in, const double *meanshift3, double varval);
int ictr_pose_addpose_se3(ictr_pose *pose, const float *dp6);
int ictr_pose_subpose_se3(ictr_pose *pose, const float *dp6);
int ictr_pose_getpose_se3(const ictr_pose *pose, double *p_out6);
/* host SoA buffers with stride op->maxpttrack: pt3d X[M] Y[M] Z[M] -> pt2d x[M] y[M] */
int ictr_pose_project_pt(const ictr_pose *pose, const float *pt3d, float *pt2d, int64_t nopoints, int sc);
int ictr_pose_project_pt_save_rotated(const ictr_pose *pose, const float *pt3d, float *pt3d_rot, float *pt2d,
                                      int64_t nopoints, int sc);
/* current cpos_p[6] / cpos_G[12] (host copies) */
int ictr_pose_get_state(const ictr_pose *pose, float *p6, float *G12);

/* ------------------------------------------------------------------ OdometerClass (odometer.h:21-30) */
typedef struct ictr_odometer ictr_odometer;
int ictr_odometer_create(ictr_odometer **out, ictr_pose *pose, const ictr_optparam *op);
void ictr_odometer_destroy(ictr_odometer *odo);
/* Set3Dpoints (odometer.cpp:171-239). pt_in: host f64 SoA X[n] Y[n] Z[n] (stride nopoints_in).
 * MUTATES pt_in when op->donorm, exactly like the reference (odometer.cpp:207-212). */
int ictr_odometer_set3dpoints(ictr_odometer *odo, double *pt_in, int64_t nopoints_in);
/* SetPose (odometer.cpp:241-255) with device pyramids; both are borrowed until the next SetPose */
int ictr_odometer_setpose(ictr_odometer *odo, const double *p_in, const ictr_pyramid *pyr_ref,
                          const ictr_pyramid *pyr_new);
/* the reference's literal signature: host level-pointer arrays (uploaded on every call: PCIe-bound) */
int ictr_odometer_setpose_host(ictr_odometer *odo, const double *p_in, const float **img_ref,
                               const float **img_ref_dx, const float **img_ref_dy, const float **img_new);
/* TrackPose (odometer.cpp:257-426): coarse-to-fine Gauss-Newton on the device, result in p_out[6] */
int ictr_odometer_trackpose(ictr_odometer *odo, double *p_out);
/* Get2DPoints (odometer.h:30): host SoA x[M] y[M] at level lv_l, valid after SetPose; owned by odo */
const float *ictr_odometer_get2dpoints(ictr_odometer *odo);
/* stream all of this odometer's kernels are enqueued on (hipStream_t); NULL = default stream */
int ictr_odometer_set_stream(ictr_odometer *odo, void *hip_stream);

/* ---- inspection (parity tests, profiling); not part of the reference surface ---- */
typedef struct ictr_trace_rec {
  int level;
  int iter;
  float H[36];
  float b[6];
  float dp[6];
  float p[6];
} ictr_trace_rec;
/* enable before trackpose; records every executed (level, iteration) of problem 0 */
int ictr_odometer_enable_trace(ictr_odometer *odo, int enable);
int ictr_odometer_trace(ictr_odometer *odo, ictr_trace_rec *out, int64_t capacity, int64_t *count);
/* which: 0 pat_ref 1 pat_ref_dx 2 pat_ref_dy (novals*M floats) 4 pt3d 5 pt3d_ref (3*M) 7 sd coefficients (16*M);
 * 100+l: pt2d of level l (2*M). Copies device state to host_out. */
int ictr_odometer_read_buffer(ictr_odometer *odo, int which, float *host_out, int64_t count);
/* normalisation parameters of the last Set3Dpoints */
int ictr_odometer_get_norm(const ictr_odometer *odo, double *meanshift3, double *varval);
/* Kernel-selection bits for A/B measurements and cross-checks (0 = the tuned default; results are the same up to
 * summation order unless noted):
 *   bit 1 (2)      any-size kernels for P = 8 instead of the wave64 = 8x8-patch fast path
 *   bits 4-5       patches per pipeline step of the iteration kernel: 1 -> 1, 2 -> 2, 3 -> 4 with temporal loads
 *   bits 6-7       patches per pipeline step of the setup kernel: 2 -> 1, 3 -> 4 (default 2)
 *   bit 8 (256)    H accumulated by the setup kernel instead of by the level's first iteration launch
 *   bits 9-11      ablation switches of the 8x8 setup kernel (512 no stores; with bit 12 also 1024 one plane's taps for
 *                  all three, 2048 no taps): WRONG RESULTS, timing only
 *   bit 12 (4096)  three separate reference planes instead of the packed {img,dx,dy,0} texels
 *   bit 13 (8192)  per-iteration launches whatever the problem size; bit 14 (16384) the one-launch tracker;
 *   bit 15 (32768) plain launches instead of the hipGraph replay; bit 18 (262144) begin phase as separate operations;
 *   bit 19 (524288) one workgroup per problem in the one-launch tracker (no teams, see ictr_batch_set_team)
 *   bit 21 (2097152) never the resident-iteration form (all iterations of a level in one launch, templates in
 *                  registers: the default for up to 8 dense frame pairs of >= 8193 8x8 patches); bit 23 (8388608) that
 *                  form whatever the batch size; bit 26 (67108864) that form with the level's setup inside the launch
 *                  (A/B: measured slower); bit 25 (33554432) debug: one workgroup of every problem skips its exchange
 *                  store (tests of the time-out path: the tracking FAILS)
 *   bit 27 (134217728) 8x8 setup kernel reads the reference pyramid's gradient planes instead of forming the gradient
 *                  patches on the fly from its image plane (the default for builder-made pyramids, the only way for
 *                  pyramids built with getgrad = 2; bit-identical patches either way) */
int ictr_odometer_set_variant(ictr_odometer *odo, int variant);
/* one-launch tracker, team form (see ictr_batch_set_team) */
int ictr_odometer_set_team(ictr_odometer *odo, int target_points, int min_points, int max_points);
int ictr_odometer_set_robust(ictr_odometer *odo, int flags, float huber_k); /* see ictr_batch_set_robust */

/* ------------------------------------------------------------------ batched engine (B independent problems) */
typedef struct ictr_batch ictr_batch;
/* All problems share cam and op; each has its own point set (<= op->maxpttrack), pose and frame pair. */
int ictr_batch_create(ictr_batch **out, const ictr_cam *cam, const ictr_optparam *op, int64_t nproblems);
void ictr_batch_destroy(ictr_batch *b);
int ictr_batch_set_stream(ictr_batch *b, void *hip_stream);
int ictr_batch_set3dpoints(ictr_batch *b, int64_t problem, double *pt_in, int64_t nopoints_in);
/* same, with the cloud normalisation supplied by the caller (sharded runs: every rank must use the mean /
 * mean-squared-radius of ALL points, not of its own slice). Ignored unless op->donorm. */
int ictr_batch_set3dpoints_norm(ictr_batch *b, int64_t problem, double *pt_in, int64_t nopoints_in,
                                const double *meanshift3, double varval);
int ictr_batch_get_norm(const ictr_batch *b, int64_t problem, double *meanshift3, double *varval);
int ictr_batch_setpose(ictr_batch *b, int64_t problem, const double *p_in, const ictr_pyramid *pyr_ref,
                       const ictr_pyramid *pyr_new);
/* SetPose of every problem in one call: p_all[6*nproblems], all problems on the same frame pair (pose samples) */
int ictr_batch_setpose_all(ictr_batch *b, const double *p_all, const ictr_pyramid *pyr_ref,
                           const ictr_pyramid *pyr_new);
/* enqueue SetPose's projection + the whole coarse-to-fine loop for every problem; asynchronous */
int ictr_batch_track_async(ictr_batch *b);
/* wait and fetch all poses: p_out[6*nproblems] */
int ictr_batch_get_poses(ictr_batch *b, double *p_out);
/* number of GN iterations each problem executed in the last track, per level summed: iters[nproblems] */
int ictr_batch_get_iterations(ictr_batch *b, int *iters);
int ictr_batch_get2dpoints(ictr_batch *b, int64_t problem, float *host_out /* 2*M */);
int ictr_batch_set_variant(ictr_batch *b, int variant); /* bits: see ictr_odometer_set_variant */
/* One-launch tracker, team form: a problem of min_points < nopoints <= max_points 8x8 patches is shared by several
 * workgroups (at most 64) that all-gather their partial sums inside the launch. target_points > 0: ceil(nopoints /
 * target_points) workgroups, a function of the problem's point count only; 0: automatic (shares of 40-128 points for a
 * lone problem, a batch sized so that its workgroups are resident together; depends on the batch size too); < 0: never.
 * Defaults 0 / 128 / 8192 (environment: ICTR_TEAM_TARGET / ICTR_TEAM_MINPTS / ICTR_TEAM_MAXPTS); tests use small
 * targets to exercise many, ragged and empty shares. Results equal the other launch forms up to summation order. */
int ictr_batch_set_team(ictr_batch *b, int target_points, int min_points, int max_points);
/* Behaviour-changing robustness options, all OFF by default (the default reproduces the reference, quirks included).
 * flags: ICTR_ROBUST_CLEAN  points outside the reference view at a level contribute nothing (the reference reuses
 *                           their stale patches and sd coefficients, odometer.cpp:304);
 *        ICTR_ROBUST_COMPOSE left-compositional update G <- exp(dp) G instead of p += dp (pose.cpp:118-123);
 *        ICTR_ROBUST_HUBER  residuals weighted min(1, huber_k / |r|) in J^T r (H stays the precomputed one).
 * With any flag set the 8x8 fast path is bypassed (any-size kernels). Oracle: oracle/np_oracle.py (same options). */
#define ICTR_ROBUST_CLEAN 1
#define ICTR_ROBUST_COMPOSE 2
#define ICTR_ROBUST_HUBER 4
int ictr_batch_set_robust(ictr_batch *b, int flags, float huber_k);
/* inspection, like ictr_odometer_read_buffer; additionally which = 8: the problem's device state as floats */
int ictr_batch_read_buffer(ictr_batch *b, int64_t problem, int which, float *host_out, int64_t count);
/* HIP-event timing on the batch's own stream: when enabled, ictr_batch_track_async brackets, per level, the
 * setup kernel (steps 4-6) and the block of maxiter iteration launches (steps 7-10) with events.
 * After the track has completed: ms_setup[l], ms_iters[l] for l in 0..lv_f (0 for levels not run). */
int ictr_batch_set_timing(ictr_batch *b, int enable);
int ictr_batch_get_level_times(ictr_batch *b, float *ms_setup, float *ms_iters);
/* ms_kernel[l]: summed duration of the level's maxiter accumulate-kernel launches ALONE (events around each launch,
 * excluding the tail kernels and the gaps) -- the figure comparable with rocprofv3 --kernel-trace --stats */
int ictr_batch_get_kernel_times(ictr_batch *b, float *ms_kernel);
/* ms per level of the level's FIRST accumulate launch alone (8x8 fast path: the instantiation that also sums H) */
int ictr_batch_get_first_iter_times(ictr_batch *b, float *ms_first);
/* For callers that run several engines concurrently on different streams: ictr_timebase_mark() sets a process-wide
 * time base (synchronises the device); ictr_batch_get_kernel_intervals returns, for the last completed tracking,
 * start and end of every accumulate launch in ms since that base, indexed [level * max(1, maxiter) + iteration]
 * (0, 0 for launches that did not run). Overlapping intervals of different engines = launches that shared the GPU. */
int ictr_timebase_mark(void);
int ictr_batch_get_kernel_intervals(ictr_batch *b, float *start_ms, float *end_ms);
int ictr_batch_get_setup_intervals(ictr_batch *b, float *start_ms, float *end_ms); /* [level]: the setup launches */
/* which launch form the last tracking used: 0 = per-iteration launches (plain), 1 = the one-launch tracker (whole
 * odometer.cpp:257-426 loop in one kernel; one workgroup or a team of workgroups per problem, see ictr_batch_last_team),
 * 2 = per-iteration launches replayed as one hipGraph, 3 = the one-launch tracker with the begin phase and the state
 * read-back inside the launch, 4 = the resident-iteration form (one setup launch + one k_level_resident launch per
 * level: dense problems of >= 8193 8x8 patches) */
int ictr_batch_last_path(const ictr_batch *b);
/* workgroups per problem of that launch when it was the one-launch tracker (1 otherwise): problems of a few hundred to a
 * few thousand 8x8 patches are shared by a team of workgroups that all-gather their partial sums inside the launch */
int ictr_batch_last_team(const ictr_batch *b);

/* ---- distributed (points sharded over ranks): split phases so the caller can all-reduce ----
 * The normal-equation block lives in a caller-visible device buffer: per problem 21 floats of H
 * (upper triangle, row-major) + 6 floats of b  => red[nproblems][27]. With sharding enabled the
 * accumulate kernels leave rank-local sums there and the *_finish kernels consume the reduced values. */
int ictr_batch_enable_sharding(ictr_batch *b, int enable);
float *ictr_batch_reduction_buffer(ictr_batch *b); /* device pointer, nproblems*27 floats */
/* use a caller-owned device buffer (e.g. a torch tensor handed to torch.distributed) instead; NULL = internal */
int ictr_batch_set_reduction_buffer(ictr_batch *b, float *dev_ptr);
int ictr_batch_begin(ictr_batch *b);                  /* SetPose projection, all problems */
/* 1 if the caller must all-reduce red[] between level_accumulate and level_finish; 0 on the 8x8 fast path, where H
 * is accumulated by the level's first iteration launch and travels with that iteration's b (27 floats, one message) */
int ictr_batch_level_allreduce_needed(ictr_batch *b);
int ictr_batch_level_accumulate(ictr_batch *b, int level);  /* steps 4-6 -> local H in red[] (unless deferred) */
int ictr_batch_level_finish(ictr_batch *b, int level);      /* adopt (reduced) H, reset iteration state */
int ictr_batch_iter_accumulate(ictr_batch *b, int level);   /* steps 7-9a -> local b in red[] */
int ictr_batch_iter_finish(ictr_batch *b, int level);       /* steps 9b-10 on the (reduced) b */

/* ------------------------------------------------------------------ Python flow-tracking surface on the device
 * misc_src/classoftrack.py:4-34 func_get_transf_position: K points (x, y) moved by a displacement field sampled
 * bilinearly at the points; float64 arithmetic in NumPy's operation order (bit-identical to the NumPy restatement and
 * to the goldens generated from the reference); NaN for points whose four taps are not all inside the field.
 * disp_u / disp_v: (H, W) planes, float32 (is_f64 = 0) or float64, host or device pointers (fields_on_device);
 * disp_v may be NULL (x only). xy and out: host (K, 2) float64. */
int ictr_flow_gather(const void *disp_u, const void *disp_v, int is_f64, int fields_on_device, int H, int W,
                     const double *xy, int64_t K, double *out);
/* misc_src/func_OF_util.py:87-129 func_extract_bil_patch, batched and without the post-processing options: K raw
 * bilinear patches (side x side x C, side = 2 (pz / 2): Python-2 integer division) of a host (H, W, C) float64 image
 * around host points (K, 2) float64; out: host (K, side, side, C) float64. Windows must lie inside the image. */
int ictr_extract_bil_patches(const double *img, int H, int W, int C, const double *pts, int64_t K, int pz, double *out);

/* ---- one-shot peer-to-peer all-reduce of the reduction buffer (latency-optimal on point-to-point xGMI) ----
 * Each rank stores its nproblems*27 floats straight into a mailbox slot in every peer's device memory (mapped with
 * hipIpc) and adds the world slots of its own mailbox in rank order: one hop instead of a ring's 2(N-1), one small
 * kernel on the compute stream, identical bits on every rank. Set-up: create -> exchange the local handles through any
 * host channel (torch.distributed all_gather) -> connect. See csrc/ictr_p2p.hip for the protocol. */
typedef struct ictr_p2p ictr_p2p;
int ictr_p2p_create(ictr_p2p **out, int rank, int world, int64_t count /* floats per exchange */);
int ictr_p2p_handle_bytes(void);                                  /* sizeof(hipIpcMemHandle_t) */
int ictr_p2p_local_handle(ictr_p2p *p, void *handle_out);          /* this rank's mailbox handle */
int ictr_p2p_connect(ictr_p2p *p, const void *all_handles);        /* world handles, rank order */
int ictr_p2p_allreduce(ictr_p2p *p, float *dev_buf, int64_t count, void *hip_stream); /* in place, asynchronous */
int ictr_p2p_error(ictr_p2p *p); /* 1 if an exchange timed out (a peer never arrived); synchronises the device */
void ictr_p2p_destroy(ictr_p2p *p);
/* Sharded RESIDENT form (r03): the batch holds this rank's shard of every problem's points, and its resident-iteration
 * launches (one per level, all iterations inside) add H -- once per level -- and b -- once per iteration -- over the ranks
 * THEMSELVES: a frame pair's solver workgroup writes its sums into every rank's mailbox and polls its own (the protocol
 * above, inside the launch: no kernel boundary, no host call, no communicator between two iterations). p: a connected
 * ictr_p2p with count >= 64 * nproblems, created alike on every rank; NULL switches the exchange off again. Every rank
 * must run the same trackings (same problems, levels, iteration limits); their loop decisions stay in lockstep because
 * every rank solves on identical sums. A tracking that cannot run in the resident form (see ictr_odometer_set_variant)
 * fails with ICTR_ERR_STATE instead of running unsynchronised; a peer that never arrives ends the launch after
 * ICTR_TEAM_TIMEOUT_S and the wait returns ICTR_ERR_HIP. */
int ictr_batch_set_peer_exchange(ictr_batch *b, ictr_p2p *p);

/* ------------------------------------------------------------------ flow producer for the misc_src/run_*OF* drivers
 * Those drivers shell out to an external optical-flow binary that is not in the reference repository
 * (misc_src/run_test_OF_track.py:90-108). ictr_patchflow is the in-tree replacement: K independent psz x psz patches
 * (psz <= 32), each with its own 2-parameter translation, pyramidal inverse-compositional Lucas-Kanade with
 * util_getPatch's sampling convention; one wave64 per patch runs every level and iteration inside one launch.
 * pts: host SoA x[K] y[K] at level 0 in frame A; out: host SoA positions in frame B (NaN when lost);
 * status (optional) 1/0; iters (optional) executed iterations. Build-defined algorithm: no reference pins it. */
int ictr_patchflow(const ictr_pyramid *pyr_a, const ictr_pyramid *pyr_b, const float *pts, int64_t K, int psz, int lv_f,
                   int lv_l, int maxiter, float eps, float *out, int *status, int *iters);
/* duration in ms of the k_patchflow launch of this thread's last ictr_patchflow call (HIP events), < 0 if unknown */
float ictr_patchflow_last_kernel_ms(void);

/* ------------------------------------------------------------------ full-frame parametric alignment (extension)
 * Inverse-compositional Gauss-Newton alignment of a whole template region under one parametric warp:
 * model 0 translation (2), 1 SE(2) (3), 2 affine (6), 3 homography (8). These are the warp models BASELINE.json's
 * configs 1, 2, 3 and 5 name; the reference itself has no such warp (its only warp is the SE(3) reprojection of
 * 3-D points, odometer.cpp:193-300), so this engine is build-defined: same Gauss-Newton skeleton (template
 * gradients + Hessian once per level, residual + J^T r per iteration, coarse-to-fine), Baker-Matthews update
 * M <- M * W(dp)^-1; oracle = oracle/np_icgn.py. nproblems independent frame pairs run in every launch.
 * Warps cross the boundary as row-major 3x3 matrices in level-0 pixel coordinates (template pixel -> current pixel). */
typedef struct ictr_icgn ictr_icgn;
#define ICTR_WARP_TRANSLATION 0
#define ICTR_WARP_SE2 1
#define ICTR_WARP_AFFINE 2
#define ICTR_WARP_HOMOGRAPHY 3
/* region_xywh: template rectangle at level 0, NULL = the frame minus a 2-pixel rim; eps: stop when |dp| <= eps */
int ictr_icgn_create(ictr_icgn **out, int model, int w, int h, int lv_f, int lv_l, int maxiter, float eps,
                     const int *region_xywh, int64_t nproblems);
void ictr_icgn_destroy(ictr_icgn *g);
int ictr_icgn_set_stream(ictr_icgn *g, void *hip_stream);
/* pyramids: lv_f+1 levels, gradients on the template, padding >= 2; borrowed, must outlive the run */
int ictr_icgn_set_frames(ictr_icgn *g, int64_t problem, const ictr_pyramid *tmpl, const ictr_pyramid *cur);
int ictr_icgn_set_warp(ictr_icgn *g, int64_t problem, const double *M9); /* initial warp, NULL = identity */
int ictr_icgn_set_timing(ictr_icgn *g, int enable);
int ictr_icgn_run_async(ictr_icgn *g); /* all levels, all iterations, no host synchronisation */
/* M9_out: nproblems*9; iters: nproblems; last_dp: nproblems*8 (any may be NULL). Synchronises the stream. */
int ictr_icgn_get_results(ictr_icgn *g, double *M9_out, int *iters, float *last_dp);
int ictr_icgn_get_kernel_times(ictr_icgn *g, float *ms_per_level); /* summed k_icgn_iter time per level */
/* row-band sharding (BASELINE config 5): each rank owns template rows [row_lo, row_hi) at level 0 and all-reduces
 * the 44-float record per problem (36 upper-triangle H + 8 b) between the accumulate and finish phases. */
int ictr_icgn_set_rows(ictr_icgn *g, int row_lo, int row_hi);
int ictr_icgn_enable_sharding(ictr_icgn *g, int enable, float *red_dev /* nproblems*44 floats or NULL */);
int ictr_icgn_begin(ictr_icgn *g);
int ictr_icgn_hess_accumulate(ictr_icgn *g, int level);
int ictr_icgn_hess_finish(ictr_icgn *g, int level);
int ictr_icgn_iter_accumulate(ictr_icgn *g, int level);
int ictr_icgn_iter_finish(ictr_icgn *g, int level);

#ifdef __cplusplus
}
#endif
#endif /* ICTR_H */
