// ctr_shim.hpp -- header-only C++ facade: namespace CTR { optparam, CamClass, PoseClass, OdometerClass, util_* } with the
// reference's method names and argument order (camera.h:19-31, pose.h:18-40, odometer.h:21-30), implemented on the
// C-ABI of include/ictr.h. A driver written against the reference's classes (run_io_reprojection_test.cpp:189-223,
// run_track_nposes.cpp:185-259) compiles against this header after two mechanical edits that the missing OpenCV
// types force:
//   * util_constructpyramide(img, w, h, lv_f, getgrad, pad) returns a CTR::Pyramid (device resident) instead of
//     filling cv::Mat arrays (utilities.h:63-64);
//   * OdometerClass::SetPose takes (p_in, const Pyramid& ref, const Pyramid& cur); the literal
//     (p_in, img_ref, img_ref_dx, img_ref_dy, img_new) overload with host level-pointer arrays is kept as well.
// Errors: the reference ignores all of them; here a failing call throws std::runtime_error with ictr_last_error().
#pragma once

#include <stdexcept>
#include <string>

#include "ictr.h"

namespace CTR {

typedef ictr_optparam optparam;  // utilities.h:46-61, same fields

inline void check(int rc, const char *what) {
  if (rc != ICTR_OK) throw std::runtime_error(std::string(what) + ": " + ictr_last_error());
}

class Pyramid {
 public:
  Pyramid(const float *img, int w, int h, int lv_f, bool getgrad, int imgpadding) : h_(nullptr) {
    check(ictr_pyramid_create(&h_, img, w, h, lv_f, getgrad ? 1 : 0, imgpadding), "util_constructpyramide");
  }
  ~Pyramid() { ictr_pyramid_destroy(h_); }
  // the next frame of a sequence into the same planes (no allocation)
  void rebuild(const float *img) { check(ictr_pyramid_rebuild(h_, img, nullptr), "util_constructpyramide (rebuild)"); }
  Pyramid(const Pyramid &) = delete;
  Pyramid &operator=(const Pyramid &) = delete;
  const ictr_pyramid *handle() const { return h_; }

 private:
  ictr_pyramid *h_;
};

// utilities.cpp:14-52
inline Pyramid *util_constructpyramide(const float *img, int w, int h, int lv_f, bool getgrad, int imgpadding) {
  return new Pyramid(img, w, h, lv_f, getgrad, imgpadding);
}
// utilities.cpp:55-113 / 115-189: one psz x psz patch around mid = (x, y) of a pyramid level (bilinear, patch-constant
// weights; mean subtracted when op->dopatchnorm). The reference takes the level's padded plane and its row stride
// (`img_ao_pyr[i][level]`, `camobj.getsw(level)`); the device-resident Pyramid carries both, so the call names the
// level instead. `out` points at psz * psz floats, row-major (what the reference's Eigen::Map refers to).
inline void util_getPatch(const Pyramid &pyr, int level, const float *mid, float *out, const optparam *op) {
  check(ictr_get_patch(pyr.handle(), level, mid, 1, op->psz, op->dopatchnorm ? 1 : 0, out), "util_getPatch");
}
inline void util_getPatch_grad(const Pyramid &pyr, int level, const float *mid, float *out, float *out_dx, float *out_dy,
                               const optparam *op) {
  check(ictr_get_patch_grad(pyr.handle(), level, mid, 1, op->psz, op->dopatchnorm ? 1 : 0, out, out_dx, out_dy),
        "util_getPatch_grad");
}
// run_track_nposes.cpp:271-355 for all points of a pose sample at once (fetch, zero-mean, unit norm, the two dot
// products, the weighting and the validity tests on the device): mids = x_back[K] y_back[K] x_ref[K] y_ref[K] x_fwd[K]
// y_fwd[K] at `level`; out_corr[K]. A caller may also restate those lines literally on util_getPatch (tests/cxx/
// nposes_driver.cpp does both and compares).
inline void util_patchNCC(const Pyramid &back, const Pyramid &ref, const Pyramid &fwd, int level, const float *mids,
                          int npoints, const optparam *op, float w_back, float w_fwd, float *out_corr) {
  check(ictr_ncc_score(back.handle(), ref.handle(), fwd.handle(), level, mids, npoints, op->psz, w_back, w_fwd, out_corr),
        "util_patchNCC");
}
template <typename T> inline void util_SE3_coeff_to_group(T *G, const T *p);
template <> inline void util_SE3_coeff_to_group<float>(float *G, const float *p) { ictr_se3_coeff_to_group_f(G, p); }
template <> inline void util_SE3_coeff_to_group<double>(double *G, const double *p) { ictr_se3_coeff_to_group_d(G, p); }
template <typename T> inline void util_SE3_group_to_coeff(T *p, const T *G);
template <> inline void util_SE3_group_to_coeff<float>(float *p, const float *G) { ictr_se3_group_to_coeff_f(p, G); }
template <> inline void util_SE3_group_to_coeff<double>(double *p, const double *G) { ictr_se3_group_to_coeff_d(p, G); }

class CamClass {
 public:
  CamClass(const int noscales_in, const float *fc_in, const float *cc_in, const int *wh_in, const int padding_in)
      : h_(nullptr) {
    check(ictr_cam_create(&h_, noscales_in, fc_in, cc_in, wh_in, padding_in), "CamClass");
  }
  ~CamClass() { ictr_cam_destroy(h_); }
  CamClass(const CamClass &) = delete;
  CamClass &operator=(const CamClass &) = delete;
  float getfx(int sc) const { return ictr_cam_getfx(h_, sc); }
  float getfy(int sc) const { return ictr_cam_getfy(h_, sc); }
  float getcx(int sc) const { return ictr_cam_getcx(h_, sc); }
  float getcy(int sc) const { return ictr_cam_getcy(h_, sc); }
  float getswo(int sc) const { return ictr_cam_getswo(h_, sc); }
  float getsho(int sc) const { return ictr_cam_getsho(h_, sc); }
  float getsw(int sc) const { return ictr_cam_getsw(h_, sc); }
  float getsh(int sc) const { return ictr_cam_getsh(h_, sc); }
  const ictr_cam *handle() const { return h_; }

 private:
  ictr_cam *h_;
};

class PoseClass {
 public:
  PoseClass(const CamClass *camobj_in, const optparam *op_in) : camobj(camobj_in), h_(nullptr) {
    check(ictr_pose_create(&h_, camobj_in->handle(), op_in), "PoseClass");
  }
  ~PoseClass() { ictr_pose_destroy(h_); }
  PoseClass(const PoseClass &) = delete;
  PoseClass &operator=(const PoseClass &) = delete;
  // the reference passes an Eigen::Vector3d by value; any 3 contiguous doubles do here
  void setpose_se3(const double *p_in, const double *meanshift_in, const double varval_in) {
    check(ictr_pose_setpose_se3(h_, p_in, meanshift_in, varval_in), "setpose_se3");
  }
  void addpose_se3(const float *p_in) { check(ictr_pose_addpose_se3(h_, p_in), "addpose_se3"); }
  void subpose_se3(const float *p_in) { check(ictr_pose_subpose_se3(h_, p_in), "subpose_se3"); }
  void getPose_se3(double *p_out) const { check(ictr_pose_getpose_se3(h_, p_out), "getPose_se3"); }
  void project_pt(const float *pt3d, float *pt2d, int nopoints, int sc) const {
    check(ictr_pose_project_pt(h_, pt3d, pt2d, nopoints, sc), "project_pt");
  }
  void project_pt_save_rotated(const float *pt3d, float *pt3d_rot, float *pt2d, int nopoints, int sc) const {
    check(ictr_pose_project_pt_save_rotated(h_, pt3d, pt3d_rot, pt2d, nopoints, sc), "project_pt_save_rotated");
  }
  const CamClass *camobj;  // public member in the reference too (pose.h:40)
  ictr_pose *handle() const { return h_; }

 private:
  ictr_pose *h_;
};

class OdometerClass {
 public:
  OdometerClass(PoseClass *pose_in, const optparam *op_in) : h_(nullptr) {
    check(ictr_odometer_create(&h_, pose_in->handle(), op_in), "OdometerClass");
  }
  ~OdometerClass() { ictr_odometer_destroy(h_); }
  OdometerClass(const OdometerClass &) = delete;
  OdometerClass &operator=(const OdometerClass &) = delete;
  void Set3Dpoints(double *pt_in, const int nopoints_in) {
    check(ictr_odometer_set3dpoints(h_, pt_in, nopoints_in), "Set3Dpoints");
  }
  void SetPose(const double *p_in, const Pyramid &img_ref, const Pyramid &img_new) {
    check(ictr_odometer_setpose(h_, p_in, img_ref.handle(), img_new.handle()), "SetPose");
  }
  void SetPose(const double *p_in, const float **img_ref_in, const float **img_ref_dx_in, const float **img_ref_dy_in,
               const float **img_new_in) {
    check(ictr_odometer_setpose_host(h_, p_in, img_ref_in, img_ref_dx_in, img_ref_dy_in, img_new_in), "SetPose");
  }
  void TrackPose(double *p_out) { check(ictr_odometer_trackpose(h_, p_out), "TrackPose"); }
  const float *Get2DPoints() const { return ictr_odometer_get2dpoints(h_); }

 private:
  ictr_odometer *h_;
};

}  // namespace CTR
