/*
 * ictr_oracle.h -- CPU restatement of catree/InvCompCamTrack's Gauss-Newton photometric
 * tracker (CamClass / PoseClass / OdometerClass / utilities), in plain C.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT. Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library. The product path
 * (invcompcamtrack_amd/, include/ictr.h) never calls into it.
 *
 * Why a restatement and not the reference itself: the reference needs Eigen3 and OpenCV,
 * neither of which exists in the build image, so oracle/_ref cannot be built.
 * Every function below cites the reference file:line it follows. Pinning:
 *   - identity KAT  (run_io_reprojection_test.cpp:15)           tests/test_oracle_kats.py
 *   - Jacobian formula stated twice (odometer.cpp:313-326,
 *     run_odometer_test.m:151-152) vs finite differences of the exp map
 *   - solver input  (odometer.cpp:474-493) vs NumPy f64 solve
 *   - an independent NumPy restatement (oracle/np_oracle.py)
 * The reference's tests hold no numeric goldens for the alignment path, so beyond these
 * KATs the tracker oracle is "parity unpinned by reference outputs" (see DESIGN.md).
 *
 * Arithmetic: f32 everywhere the reference is f32, compiled with -ffp-contract=off so that,
 * like the reference's -msse4 -mavx build (no -mfma), no multiply-add is fused.
 */
#ifndef ICTR_ORACLE_H
#define ICTR_ORACLE_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* utilities.h:46-61 -- field order preserved */
typedef struct {
  int maxpttrack;
  int psz;
  int pszd2;
  int pszd2m3;
  int novals;
  int lv_f;
  int lv_l;
  bool donorm;
  bool dopatchnorm;
  int maxiter;
  float normdp_ratio;
  int verbosity;
} orc_optparam;

/* fills the derived fields the drivers compute (run_io_reprojection_test.cpp:112-126) */
void orc_optparam_init(orc_optparam *op, int lv_f, int lv_l, int psz, int maxiter, float normdp_ratio,
                       int donorm, int dopatchnorm, int maxpttrack, int verbosity);

/* ---- CamClass (camera.cpp:14-45) ---- */
typedef struct orc_cam orc_cam;
orc_cam *orc_cam_create(int noscales, const float *fc, const float *cc, const int *wh, int padding);
void orc_cam_destroy(orc_cam *c);
/* which: 0 fx 1 fy 2 cx 3 cy 4 swo 5 sho 6 sw 7 sh */
float orc_cam_get(const orc_cam *c, int which, int sc);

/* ---- utilities.h:84-241 ---- */
void orc_se3_exp_f(float *G, const float *p);
void orc_se3_exp_d(double *G, const double *p);
void orc_se3_log_f(float *p, const float *G);
void orc_se3_log_d(double *p, const double *G);

/* ---- utilities.cpp:14-52 (OpenCV semantics restated: 2x2 box, [-1 0 1], reflect-101, pad) ----
 * Level sizes: w_l = cvRound(w_{l-1}*0.5) as cv::resize(fx=.5) does. Output planes are
 * (w_l+2*pad) x (h_l+2*pad) row-major f32, caller-allocated. orc_pyramid_level_size gives dims. */
void orc_pyramid_level_size(int w, int h, int level, int *wl, int *hl);
void orc_pyramid_build(const float *img, int w, int h, int lv_f, int getgrad, int pad,
                       float **img_pyr, float **dx_pyr, float **dy_pyr);

/* ---- utilities.cpp:55-113, 115-189 ---- */
void orc_getpatch(const float *img, const float *mid, float *out, const orc_optparam *op, int width);
void orc_getpatch_grad(const float *img, const float *img_dx, const float *img_dy, const float *mid,
                       float *out, float *out_dx, float *out_dy, const orc_optparam *op, int width);

/* ---- PoseClass (pose.cpp) ---- */
typedef struct orc_pose orc_pose;
orc_pose *orc_pose_create(const orc_cam *cam, const orc_optparam *op);
void orc_pose_destroy(orc_pose *p);
void orc_pose_setpose_se3(orc_pose *p, const double *p_in, const double *meanshift, double varval);
void orc_pose_addpose_se3(orc_pose *p, const float *dp);
void orc_pose_subpose_se3(orc_pose *p, const float *dp);
void orc_pose_getpose_se3(const orc_pose *p, double *p_out);
void orc_pose_project_pt(const orc_pose *p, const float *pt3d, float *pt2d, int nopoints, int sc);
void orc_pose_project_pt_save_rotated(const orc_pose *p, const float *pt3d, float *pt3d_rot, float *pt2d,
                                      int nopoints, int sc);
const float *orc_pose_G(const orc_pose *p); /* cpos_G[12] */
const float *orc_pose_p(const orc_pose *p); /* cpos_p[6]  */

/* ---- Eigen fullPivLu().solve restated (odometer.cpp:509-515) ---- */
void orc_solve6_fullpivlu(const float *H /*36, symmetric*/, const float *b, float *x);

/* ---- OdometerClass (odometer.cpp) ---- */
typedef struct orc_odometer orc_odometer;
orc_odometer *orc_odometer_create(orc_pose *pose, const orc_optparam *op);
void orc_odometer_destroy(orc_odometer *o);
void orc_odometer_set3dpoints(orc_odometer *o, double *pt_in, int nopoints_in);
void orc_odometer_setpose(orc_odometer *o, const double *p_in, const float **img_ref, const float **img_ref_dx,
                          const float **img_ref_dy, const float **img_new);
void orc_odometer_trackpose(orc_odometer *o, double *p_out);
const float *orc_odometer_get2dpoints(const orc_odometer *o);

/* ---- inspection hooks for the parity tests (not in the reference) ----
 * The trace records, for every (level, iteration) executed by the last trackpose call:
 * Hes (36), sumsd (6), delta_p (6), cpos_p after the update (6). */
typedef struct {
  int level;
  int iter;
  float H[36];
  float b[6];
  float dp[6];
  float p[6];
} orc_trace_rec;
int orc_odometer_trace_count(const orc_odometer *o);
const orc_trace_rec *orc_odometer_trace(const orc_odometer *o);
/* which: 0 pat_ref 1 pat_ref_dx 2 pat_ref_dy 3 pat_new 4 pt3d 5 pt3d_ref 6 pt2d_new ; buffers after last call */
const float *orc_odometer_buffer(const orc_odometer *o, int which);
const float *orc_odometer_pt2d(const orc_odometer *o, int level);
const unsigned char *orc_odometer_ind(const orc_odometer *o, int which /*0 ref 1 new*/);
void orc_odometer_norm(const orc_odometer *o, double *meanshift3, double *varval);
/* accumulate the H / b sums in double instead of float (to size the f32 summation noise band) */
void orc_set_sum_mode(int use_double);

#ifdef __cplusplus
}
#endif
#endif
