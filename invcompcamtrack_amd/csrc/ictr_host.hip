// ictr_host.hip -- implementation of the C-ABI in include/ictr.h: handle objects, device arenas and the
// launch sequences. Host-side maths here is only what the reference also does per call on a handful of
// scalars (pose normalisation pose.cpp:25-113, point-cloud normalisation odometer.cpp:171-239); everything
// that touches pixels or points runs in the kernels of ictr_kernels.hip. There is no CPU fallback.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "ictr_dev.h"
#include "se3_math.h"

namespace ictr {
void launch_pyr_copy(const float *, float *, int, int, int, int, hipStream_t);
void launch_pyr_down(const float *, int, int, int, float *, int, int, int, int, hipStream_t);
void launch_pyr_finish(float *, float *, float *, int, int, int, int, int, int, hipStream_t);
void launch_pyr_pack(const float *, const float *, const float *, float *, size_t, hipStream_t);
void launch_pyr_level(const float *, int, int, int, int, float *, float *, float *, float *, int, int, int, int, int, int,
                      hipStream_t);
void launch_stream_read(const float *, size_t, float *, hipStream_t);
void launch_getpatch(const float *, const float *, const float *, const float *, int, int, int, int, float *, float *,
                     float *, hipStream_t);
void launch_ncc(const float *, const float *, const float *, const float *, int, int, int, float, float, float, float,
                float *, hipStream_t);
void launch_flow_gather(const void *, const void *, int, int, int, const double *, int, double *, hipStream_t);
void launch_bil_patches(const double *, int, int, int, const double *, int, int, double *, hipStream_t);
void launch_project_generic(const float *, float *, float *, int, int, const float *, LevelCam, hipStream_t);
void launch_project_ref(const EngineDev &, const LevelCam *, int, hipStream_t);
void launch_ref_level(const EngineDev &, const LevelCam &, int, int, int, int, int, hipStream_t);
void launch_level_finish(const EngineDev &, int, hipStream_t);
void launch_iter(const EngineDev &, const LevelCam &, int, int, int, int, int, int, hipStream_t);
void launch_iter_main(const EngineDev &, const LevelCam &, int, int, int, int, int, int, hipStream_t, hipEvent_t,
                      hipEvent_t);
void launch_iter_tail(const EngineDev &, int, int, int, int, int, hipStream_t);
void launch_iter_finish(const EngineDev &, int, int, int, hipStream_t);
bool defer_h(const EngineDev &, int);
hipError_t launch_track1(const EngineDev &, const LevelCam *, int, int, const void *, ProbState *, hipStream_t,
                         const T1Team *, bool project_here = false);
hipError_t launch_level_resident(const EngineDev &, const LevelCam &, int, int, int, int, int, int, unsigned,
                                 unsigned long long, unsigned long long *, int *, int, int, const ResXchg *, hipStream_t);
hipError_t launch_debug_transpose_reduce(const float *, float *, int *, int *, int, hipStream_t);
size_t resident_mail_bytes(int, int);
int resident_points_per_workgroup(int);
int resident_blocks_per_cu(int, int);
int track1_team_q(int, int);
int track1_team_size(int, int);
size_t track1_team_mail_bytes(int, int);
size_t track1_blob_bytes(void);
size_t track1_plan(int, int, int, int, int *);
}  // namespace ictr

namespace ictr {
void launch_patchflow(const PFArgs &, hipStream_t);
}  // namespace ictr

using namespace ictr;

// ---------------------------------------------------------------- errors
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
int ictr_fail_(int code, const char *fmt, ...) {  // for the other translation units (ictr_icgn.hip)
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(expr)                                                                                  \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess) return fail(ICTR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));   \
  } while (0)

extern "C" const char *ictr_last_error(void) { return g_err.c_str(); }
extern "C" int ictr_version(void) { return 100; }
extern "C" int ictr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" int ictr_set_device(int device) {
  HIPCHK(hipSetDevice(device));
  return ICTR_OK;
}
static int need_device();
// Measured streaming-read bandwidth of this GPU in GB/s: `bytes` of freshly allocated memory (>> the 256 MB Infinity
// Cache) read `reps` times with plain wide loads. A yardstick for roofline reports next to the vendor peak.
extern "C" int ictr_stream_read_bandwidth(size_t bytes, int reps, double *gbps_out) {
  if (!gbps_out || bytes < (1u << 20) || reps < 1) return fail(ICTR_ERR_INVALID, "stream_read_bandwidth: bad arguments");
  if (int rc = need_device()) return rc;
  float *buf = nullptr, *sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc((void **)&buf, bytes);
  if (e == hipSuccess) e = hipMalloc((void **)&sink, sizeof(float) * 8192 * kBlock);
  if (e == hipSuccess) e = hipMemset(buf, 0, bytes);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  float ms = 0.0f;
  if (e == hipSuccess) {
    launch_stream_read(buf, bytes / 4, sink, nullptr);  // warm-up
    e = hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps && e == hipSuccess; ++r) launch_stream_read(buf, bytes / 4, sink, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (buf) (void)hipFree(buf);
  if (sink) (void)hipFree(sink);
  if (e != hipSuccess) return fail(ICTR_ERR_HIP, "stream_read_bandwidth: %s", hipGetErrorString(e));
  *gbps_out = (double)bytes * reps / (ms * 1e-3) / 1e9;
  return ICTR_OK;
}
// inspection: the resident-iteration kernel's transposing wave reduction alone (ictr_resident.hip, TrAcc) on caller data.
// vals[64 lanes][64 values = 2 patch + kind] -> out[lane] = the 64-lane sum of value 2 patch_of_lane + kind_of_lane
extern "C" int ictr_debug_transpose_reduce(const float *vals, float *out, int *patch_of_lane, int *kind_of_lane,
                                           int patches_per_wave) {
  if (!vals || !out || !patch_of_lane || !kind_of_lane || (patches_per_wave != 16 && patches_per_wave != 32))
    return fail(ICTR_ERR_INVALID, "debug_transpose_reduce: bad arguments (patches per wave 16 or 32)");
  if (int rc = need_device()) return rc;
  float *dv = nullptr, *dout = nullptr;
  int *dp = nullptr;
  hipError_t e = hipMalloc((void **)&dv, sizeof(float) * 64 * 64);
  if (e == hipSuccess) e = hipMalloc((void **)&dout, sizeof(float) * 64);
  if (e == hipSuccess) e = hipMalloc((void **)&dp, sizeof(int) * 128);
  if (e == hipSuccess) e = hipMemcpy(dv, vals, sizeof(float) * 64 * 64, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = launch_debug_transpose_reduce(dv, dout, dp, dp + 64, patches_per_wave, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out, dout, sizeof(float) * 64, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(patch_of_lane, dp, sizeof(int) * 64, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(kind_of_lane, dp + 64, sizeof(int) * 64, hipMemcpyDeviceToHost);
  for (void *p : {(void *)dv, (void *)dout, (void *)dp})
    if (p) (void)hipFree(p);
  if (e != hipSuccess) return fail(ICTR_ERR_HIP, "debug_transpose_reduce: %s", hipGetErrorString(e));
  return ICTR_OK;
}

static int need_device() {
  if (ictr_device_count() <= 0)
    return fail(ICTR_ERR_NO_DEVICE, "no usable HIP device: the tracker has no CPU fallback");
  return ICTR_OK;
}

extern "C" int ictr_optparam_init(ictr_optparam *op, int lv_f, int lv_l, int psz, int maxiter, float normdp_ratio,
                                  int donorm, int dopatchnorm, int maxpttrack, int verbosity) {
  if (!op) return fail(ICTR_ERR_INVALID, "op is NULL");
  memset(op, 0, sizeof(*op));
  op->lv_f = lv_f;
  op->lv_l = lv_l;
  op->psz = psz;
  op->pszd2 = psz / 2;
  op->pszd2m3 = psz + op->pszd2 - 1;
  op->novals = psz * psz;
  op->maxiter = maxiter;
  op->normdp_ratio = normdp_ratio;
  op->donorm = donorm != 0;
  op->dopatchnorm = dopatchnorm != 0;
  op->maxpttrack = maxpttrack;
  const int r = op->maxpttrack % 4;  // SSEMULTIPL padding of the drivers (run_io_reprojection_test.cpp:123-126)
  if (r > 0) op->maxpttrack += 4 - r;
  op->verbosity = verbosity;
  return ICTR_OK;
}

// ---------------------------------------------------------------- CamClass
struct ictr_cam {
  int noscales;
  int padding;
  int wh[2];
  std::vector<float> fx, fy, cx, cy, swo, sho, sw, sh;
};

extern "C" int ictr_cam_create(ictr_cam **out, int noscales, const float *fc, const float *cc, const int *wh,
                               int padding) {
  if (!out || !fc || !cc || !wh || noscales < 1 || noscales > 16 || padding < 0)
    return fail(ICTR_ERR_INVALID, "ictr_cam_create: bad arguments (noscales must be 1..16)");
  ictr_cam *c = new ictr_cam;
  c->noscales = noscales;
  c->padding = padding;
  c->wh[0] = wh[0];
  c->wh[1] = wh[1];
  for (auto *v : {&c->fx, &c->fy, &c->cx, &c->cy, &c->swo, &c->sho, &c->sw, &c->sh}) v->resize(noscales);
  for (int i = 0; i < noscales; ++i) {
    const float s = (float)(1 / pow(2, i));  // camera.cpp:33
    c->fx[i] = s * fc[0];
    c->fy[i] = s * fc[1];
    c->cx[i] = s * cc[0];
    c->cy[i] = s * cc[1];
    c->swo[i] = s * (float)wh[0];
    c->sho[i] = s * (float)wh[1];
    c->sw[i] = c->swo[i] + 2 * padding;
    c->sh[i] = c->sho[i] + 2 * padding;
  }
  *out = c;
  return ICTR_OK;
}
extern "C" void ictr_cam_destroy(ictr_cam *cam) { delete cam; }
#define CAMGET(name, field) \
  extern "C" float ictr_cam_##name(const ictr_cam *cam, int sc) { return cam->field[sc]; }
CAMGET(getfx, fx)
CAMGET(getfy, fy)
CAMGET(getcx, cx)
CAMGET(getcy, cy)
CAMGET(getswo, swo)
CAMGET(getsho, sho)
CAMGET(getsw, sw)
CAMGET(getsh, sh)

static LevelCam level_cam(const ictr_cam *c, int l) {
  LevelCam lc;
  lc.fx = c->fx[l];
  lc.fy = c->fy[l];
  lc.cx = c->cx[l];
  lc.cy = c->cy[l];
  lc.swo = c->swo[l];
  lc.sho = c->sho[l];
  lc.sw = (int)c->sw[l];
  return lc;
}

// ---------------------------------------------------------------- SE(3) helpers on the host
extern "C" void ictr_se3_coeff_to_group_f(float *G, const float *p) { se3_exp<float>(G, p); }
extern "C" void ictr_se3_coeff_to_group_d(double *G, const double *p) { se3_exp<double>(G, p); }
extern "C" void ictr_se3_group_to_coeff_f(float *p, const float *G) { se3_log<float>(p, G); }
extern "C" void ictr_se3_group_to_coeff_d(double *p, const double *G) { se3_log<double>(p, G); }
extern "C" void ictr_solve6(const float *H, const float *b, float *x) {
  // the device path (factor once per level + substitute per iteration), run back to back on the host
  float A[36], c[6];
  int piv[12], info[2];
  memcpy(A, H, sizeof(A));
  lu_factor_ws<6>(A, piv, info);
  lu_apply_ws<6>(A, piv, info, b, x, c);
}

// pose.cpp:25-76
static void host_setpose(bool donorm, const double *p_in, const double *ms, double varval, float *p_f, float *G_f) {
  double pn[6];
  memcpy(pn, p_in, sizeof(pn));
  if (donorm) {
    double G[12];
    se3_exp<double>(G, pn);
    double t[3];
    t[0] = -G[0] * G[3] - G[4] * G[7] - G[8] * G[11];
    t[1] = -G[1] * G[3] - G[5] * G[7] - G[9] * G[11];
    t[2] = -G[2] * G[3] - G[6] * G[7] - G[10] * G[11];
    t[0] = (t[0] - ms[0]) / varval;
    t[1] = (t[1] - ms[1]) / varval;
    t[2] = (t[2] - ms[2]) / varval;
    G[3] = -G[0] * t[0] - G[1] * t[1] - G[2] * t[2];
    G[7] = -G[4] * t[0] - G[5] * t[1] - G[6] * t[2];
    G[11] = -G[8] * t[0] - G[9] * t[1] - G[10] * t[2];
    se3_log<double>(pn, G);
  }
  for (int i = 0; i < 6; ++i) p_f[i] = (float)pn[i];
  se3_exp<float>(G_f, p_f);
}
// pose.cpp:79-113 (f32 G, f64 camera centre, f32 log: the reference's mixed precision is kept)
static void host_getpose(bool donorm, const float *p_f, const float *G_f, const double *ms, double varval,
                         double *p_out) {
  float pu[6];
  memcpy(pu, p_f, sizeof(pu));
  if (donorm) {
    float G[12];
    memcpy(G, G_f, sizeof(G));
    double t[3];
    t[0] = (double)(-G[0] * G[3] - G[4] * G[7] - G[8] * G[11]);
    t[1] = (double)(-G[1] * G[3] - G[5] * G[7] - G[9] * G[11]);
    t[2] = (double)(-G[2] * G[3] - G[6] * G[7] - G[10] * G[11]);
    t[0] = t[0] * varval + ms[0];
    t[1] = t[1] * varval + ms[1];
    t[2] = t[2] * varval + ms[2];
    G[3] = (float)(-G[0] * t[0] - G[1] * t[1] - G[2] * t[2]);
    G[7] = (float)(-G[4] * t[0] - G[5] * t[1] - G[6] * t[2]);
    G[11] = (float)(-G[8] * t[0] - G[9] * t[1] - G[10] * t[2]);
    se3_log<float>(pu, G);
  }
  for (int i = 0; i < 6; ++i) p_out[i] = (double)pu[i];
}

// ---------------------------------------------------------------- pyramid
struct ictr_pyramid {
  int nlev = 0, pad = 0, w0 = 0, h0 = 0, getgrad = 0;  // getgrad: 0 image only, 1 + dx / dy / packed planes,
                                                       // 2 image only, gradients formed on the fly by the consumers
  int builder_made = 0;  // the planes were computed by pyramid_build (dx / dy ARE the central differences of img)
  std::vector<int> w, h, sw, sh;
  std::vector<float *> img, dx, dy;  // device planes
  std::vector<float *> pack;         // with gradients: the level again as interleaved {img, dx, dy, 0} texels
  float *arena = nullptr;
  float *stage = nullptr;  // device copy of a host frame handed to ictr_pyramid_rebuild (allocated on first use)
};

struct ictr_pyramid_view {  // internal: what ictr_icgn.hip needs to know about a pyramid
  int nlev, pad;
  const int *w, *h, *sw;
  float *const *img, *const *dx, *const *dy;
  int getgrad;
};
extern "C" int ictr_pyramid_view_(const ictr_pyramid *p, ictr_pyramid_view *v) {
  v->nlev = p->nlev;
  v->pad = p->pad;
  v->w = p->w.data();
  v->h = p->h.data();
  v->sw = p->sw.data();
  v->img = p->img.data();
  v->dx = p->dx.data();
  v->dy = p->dy.data();
  v->getgrad = p->getgrad == 1 ? 1 : 0;  // (gradient PLANES; image-only pyramids of the tracker's OTF path have none)
  return 0;
}

static void level_size(int w, int h, int level, int *wl, int *hl) {
  auto half = [](int v) {  // cvRound(v*0.5), round-half-even, as cv::resize(dsize=Size(), fx=.5) sizes its output
    const int q = v / 2;
    if (v % 2 == 0) return q;
    return (q % 2 == 0) ? q : q + 1;
  };
  for (int i = 0; i < level; ++i) {
    w = half(w);
    h = half(h);
  }
  *wl = w;
  *hl = h;
}

static int pyramid_alloc(ictr_pyramid **out, int w, int h, int lv_f, int getgrad, int pad) {
  if (!out || w < 1 || h < 1 || lv_f < 0 || lv_f > 15 || pad < 0 || getgrad < 0 || getgrad > 2)
    return fail(ICTR_ERR_INVALID, "pyramid: bad arguments");
  if (int rc = need_device()) return rc;
  ictr_pyramid *p = new ictr_pyramid;
  p->nlev = lv_f + 1;
  p->pad = pad;
  p->w0 = w;
  p->h0 = h;
  p->getgrad = getgrad;
  size_t total = 0;
  for (int l = 0; l <= lv_f; ++l) {
    int wl, hl;
    level_size(w, h, l, &wl, &hl);
    if (wl < 1 || hl < 1) {
      delete p;
      return fail(ICTR_ERR_INVALID, "pyramid: level %d is empty", l);
    }
    p->w.push_back(wl);
    p->h.push_back(hl);
    p->sw.push_back(wl + 2 * pad);
    p->sh.push_back(hl + 2 * pad);
    size_t plane = (size_t)(wl + 2 * pad) * (hl + 2 * pad);
    plane = (plane + 63) / 64 * 64;  // 256-B aligned planes
    total += plane * (getgrad == 1 ? 7 : getgrad == 2 ? 1 : 3);
  }
  hipError_t e = hipMalloc((void **)&p->arena, total * sizeof(float));
  if (e != hipSuccess) {
    delete p;
    return fail(ICTR_ERR_HIP, "hipMalloc(%zu) failed: %s", total * sizeof(float), hipGetErrorString(e));
  }
  float *cur = p->arena;
  for (int l = 0; l <= lv_f; ++l) {
    size_t plane = (size_t)p->sw[l] * p->sh[l];
    plane = (plane + 63) / 64 * 64;
    p->img.push_back(cur);
    p->dx.push_back(getgrad == 2 ? nullptr : cur + plane);
    p->dy.push_back(getgrad == 2 ? nullptr : cur + 2 * plane);
    p->pack.push_back(getgrad == 1 ? cur + 3 * plane : nullptr);
    cur += (getgrad == 1 ? 7 : getgrad == 2 ? 1 : 3) * plane;
  }
  *out = p;
  return ICTR_OK;
}

static int pyramid_build(ictr_pyramid *p, const float *img_dev, hipStream_t s) {
  // one launch per level (k_pyr_level); ICTR_PYR_UNFUSED=1: the four-kernel form it replaced (bit-identical planes,
  // kept for the A/B and as a cross-check in the tests)
  static const bool unfused = [] {
    const char *v = getenv("ICTR_PYR_UNFUSED");
    return v && atoi(v) != 0;
  }();
  p->builder_made = 1;
  const int planes = p->getgrad == 1 ? 1 : 0;  // getgrad 2: the image levels only
  for (int l = 0; l < p->nlev; ++l) {
    if (!unfused || p->getgrad == 2) {
      if (l == 0)
        launch_pyr_level(img_dev, 1, p->w[0], p->h[0], p->w[0], p->img[0], p->dx[0], p->dy[0], p->pack[0], p->w[0],
                         p->h[0], p->pad, p->sw[0], p->sh[0], planes, s);
      else
        launch_pyr_level(p->img[l - 1], 0, p->w[l - 1], p->h[l - 1], p->sw[l - 1], p->img[l], p->dx[l], p->dy[l],
                         p->pack[l], p->w[l], p->h[l], p->pad, p->sw[l], p->sh[l], planes, s);
      continue;
    }
    if (l == 0)
      launch_pyr_copy(img_dev, p->img[0], p->w[0], p->h[0], p->pad, p->sw[0], s);
    else
      launch_pyr_down(p->img[l - 1], p->w[l - 1], p->h[l - 1], p->sw[l - 1], p->img[l], p->w[l], p->h[l], p->pad,
                      p->sw[l], s);
    launch_pyr_finish(p->img[l], p->dx[l], p->dy[l], p->w[l], p->h[l], p->pad, p->sw[l], p->sh[l], p->getgrad, s);
    if (p->getgrad) launch_pyr_pack(p->img[l], p->dx[l], p->dy[l], p->pack[l], (size_t)p->sw[l] * p->sh[l], s);
  }
  HIPCHK(hipGetLastError());
  return ICTR_OK;
}

// A video loop builds one pyramid per incoming frame (run_track_nposes.cpp:180: one per image of the sequence): refill an
// existing pyramid in place -- no allocation, every plane keeps its address (batches / graphs that hold them stay valid).
extern "C" int ictr_pyramid_rebuild_device(ictr_pyramid *p, const float *img_dev, void *hip_stream) {
  if (!p || !img_dev) return fail(ICTR_ERR_INVALID, "pyramid_rebuild: NULL argument");
  return pyramid_build(p, img_dev, (hipStream_t)hip_stream);
}
extern "C" int ictr_pyramid_rebuild(ictr_pyramid *p, const float *img, void *hip_stream) {
  if (!p || !img) return fail(ICTR_ERR_INVALID, "pyramid_rebuild: NULL argument");
  const size_t bytes = sizeof(float) * (size_t)p->w0 * p->h0;
  if (!p->stage) HIPCHK(hipMalloc((void **)&p->stage, bytes));
  // pageable host memory: the copy has left `img` when the call returns; the kernels are ordered behind it on the stream
  HIPCHK(hipMemcpyAsync(p->stage, img, bytes, hipMemcpyHostToDevice, (hipStream_t)hip_stream));
  return pyramid_build(p, p->stage, (hipStream_t)hip_stream);
}

extern "C" int ictr_pyramid_create_device(ictr_pyramid **out, const float *img_dev, int w, int h, int lv_f, int getgrad,
                                          int pad, void *hip_stream) {
  if (!img_dev) return fail(ICTR_ERR_INVALID, "pyramid: img is NULL");
  ictr_pyramid *p = nullptr;
  if (int rc = pyramid_alloc(&p, w, h, lv_f, getgrad, pad)) return rc;
  int rc = pyramid_build(p, img_dev, (hipStream_t)hip_stream);
  if (rc) {
    ictr_pyramid_destroy(p);
    return rc;
  }
  *out = p;
  return ICTR_OK;
}

extern "C" int ictr_pyramid_create(ictr_pyramid **out, const float *img, int w, int h, int lv_f, int getgrad,
                                   int pad) {
  if (!img) return fail(ICTR_ERR_INVALID, "pyramid: img is NULL");
  if (int rc = need_device()) return rc;
  float *d = nullptr;
  HIPCHK(hipMalloc((void **)&d, sizeof(float) * (size_t)w * h));
  hipError_t e = hipMemcpy(d, img, sizeof(float) * (size_t)w * h, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    hipFree(d);
    return fail(ICTR_ERR_HIP, "hipMemcpy H2D failed: %s", hipGetErrorString(e));
  }
  int rc = ictr_pyramid_create_device(out, d, w, h, lv_f, getgrad, pad, nullptr);
  hipError_t e2 = hipDeviceSynchronize();
  hipFree(d);
  if (rc) return rc;
  if (e2 != hipSuccess) return fail(ICTR_ERR_HIP, "pyramid kernels failed: %s", hipGetErrorString(e2));
  return ICTR_OK;
}

extern "C" int ictr_pyramid_create_from_host_planes(ictr_pyramid **out, const float **img_pyr, const float **dx_pyr,
                                                    const float **dy_pyr, int w, int h, int lv_f, int pad) {
  if (!img_pyr) return fail(ICTR_ERR_INVALID, "pyramid: img_pyr is NULL");
  ictr_pyramid *p = nullptr;
  if (int rc = pyramid_alloc(&p, w, h, lv_f, dx_pyr && dy_pyr, pad)) return rc;
  for (int l = 0; l <= lv_f; ++l) {
    const size_t bytes = sizeof(float) * (size_t)p->sw[l] * p->sh[l];
    hipError_t e = hipMemcpy(p->img[l], img_pyr[l], bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && dx_pyr && dy_pyr) {
      e = hipMemcpy(p->dx[l], dx_pyr[l], bytes, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(p->dy[l], dy_pyr[l], bytes, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
      ictr_pyramid_destroy(p);
      return fail(ICTR_ERR_HIP, "hipMemcpy H2D failed: %s", hipGetErrorString(e));
    }
    if (p->getgrad) launch_pyr_pack(p->img[l], p->dx[l], p->dy[l], p->pack[l], (size_t)p->sw[l] * p->sh[l], nullptr);
  }
  if (p->getgrad && hipStreamSynchronize(nullptr) != hipSuccess) {
    ictr_pyramid_destroy(p);
    return fail(ICTR_ERR_HIP, "pyramid: packing kernels failed");
  }
  *out = p;
  return ICTR_OK;
}

extern "C" void ictr_pyramid_destroy(ictr_pyramid *p) {
  if (!p) return;
  if (p->arena) hipFree(p->arena);
  if (p->stage) hipFree(p->stage);
  delete p;
}
extern "C" int ictr_pyramid_levels(const ictr_pyramid *p) { return p ? p->nlev : 0; }
extern "C" int ictr_pyramid_level_dims(const ictr_pyramid *p, int level, int *sw, int *sh) {
  if (!p || level < 0 || level >= p->nlev) return fail(ICTR_ERR_INVALID, "pyramid: bad level");
  if (sw) *sw = p->sw[level];
  if (sh) *sh = p->sh[level];
  return ICTR_OK;
}
static float *pyr_plane(const ictr_pyramid *p, int level, int which) {
  if (!p || level < 0 || level >= p->nlev) return nullptr;
  return which == 0 ? p->img[level] : which == 1 ? p->dx[level] : which == 2 ? p->dy[level] : nullptr;
}
extern "C" const float *ictr_pyramid_device_plane(const ictr_pyramid *p, int level, int which) {
  return pyr_plane(p, level, which);
}
extern "C" int ictr_pyramid_download(const ictr_pyramid *p, int level, int which, float *host_out) {
  const float *d = pyr_plane(p, level, which);
  if (!d || !host_out) return fail(ICTR_ERR_INVALID, "pyramid_download: bad arguments");
  HIPCHK(hipMemcpy(host_out, d, sizeof(float) * (size_t)p->sw[level] * p->sh[level], hipMemcpyDeviceToHost));
  return ICTR_OK;
}

static int get_patch_impl(const ictr_pyramid *pyr, int level, const float *mids, int64_t K, int psz, int dopatchnorm,
                          float *out, float *out_dx, float *out_dy, bool grad) {
  if (!pyr || level < 0 || level >= pyr->nlev || !mids || !out || K < 0 || psz < 1 || psz > pyr->pad)
    return fail(ICTR_ERR_INVALID, "get_patch: bad arguments (psz must be <= pyramid padding)");
  if (grad && (!out_dx || !out_dy || pyr->getgrad != 1)) return fail(ICTR_ERR_INVALID, "get_patch_grad: no gradient planes");
  if (K == 0) return ICTR_OK;
  // centres must lie inside [0,swo] x [0,sho] like the callers guarantee (odometer.cpp:273-276)
  for (int64_t i = 0; i < K; ++i)
    if (!(mids[i] >= 0 && mids[i] <= (float)pyr->w[level] && mids[i + K] >= 0 && mids[i + K] <= (float)pyr->h[level]))
      return fail(ICTR_ERR_INVALID, "get_patch: centre %lld outside the image", (long long)i);
  const size_t nf = (size_t)K * psz * psz;
  float *d_m = nullptr, *d_o = nullptr;
  HIPCHK(hipMalloc((void **)&d_m, sizeof(float) * 2 * K));
  hipError_t e = hipMalloc((void **)&d_o, sizeof(float) * nf * (grad ? 3 : 1));
  if (e != hipSuccess) {
    hipFree(d_m);
    return fail(ICTR_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
  }
  e = hipMemcpy(d_m, mids, sizeof(float) * 2 * K, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    launch_getpatch(pyr->img[level], grad ? pyr->dx[level] : nullptr, grad ? pyr->dy[level] : nullptr, d_m, (int)K, psz,
                    pyr->sw[level], dopatchnorm, d_o, grad ? d_o + nf : nullptr, grad ? d_o + 2 * nf : nullptr, nullptr);
    e = hipMemcpy(out, d_o, sizeof(float) * nf, hipMemcpyDeviceToHost);
    if (e == hipSuccess && grad) {
      e = hipMemcpy(out_dx, d_o + nf, sizeof(float) * nf, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(out_dy, d_o + 2 * nf, sizeof(float) * nf, hipMemcpyDeviceToHost);
    }
  }
  hipFree(d_m);
  hipFree(d_o);
  if (e != hipSuccess) return fail(ICTR_ERR_HIP, "get_patch failed: %s", hipGetErrorString(e));
  return ICTR_OK;
}
extern "C" int ictr_get_patch(const ictr_pyramid *pyr, int level, const float *mids, int64_t K, int psz,
                              int dopatchnorm, float *out) {
  return get_patch_impl(pyr, level, mids, K, psz, dopatchnorm, out, nullptr, nullptr, false);
}
extern "C" int ictr_get_patch_grad(const ictr_pyramid *pyr, int level, const float *mids, int64_t K, int psz,
                                   int dopatchnorm, float *out, float *out_dx, float *out_dy) {
  return get_patch_impl(pyr, level, mids, K, psz, dopatchnorm, out, out_dx, out_dy, true);
}

// run_track_nposes.cpp:271-355: per-point patch correlation, computed on the device (k_ncc)
extern "C" int ictr_ncc_score(const ictr_pyramid *pyr_back, const ictr_pyramid *pyr_ref, const ictr_pyramid *pyr_fwd,
                              int level, const float *mids, int64_t K, int psz, float w_back, float w_fwd,
                              float *out_corr) {
  if (!pyr_back || !pyr_ref || !pyr_fwd || K < 0 || (K > 0 && (!mids || !out_corr)) || psz < 1 || psz > 64)
    return fail(ICTR_ERR_INVALID, "ncc_score: bad arguments");
  for (const ictr_pyramid *py : {pyr_back, pyr_ref, pyr_fwd})
    if (level < 0 || level >= py->nlev || psz > py->pad || py->sw[level] != pyr_ref->sw[level] ||
        py->w[level] != pyr_ref->w[level] || py->h[level] != pyr_ref->h[level])
      return fail(ICTR_ERR_INVALID, "ncc_score: pyramids differ at level %d or padding < psz", level);
  if (K == 0) return ICTR_OK;
  if (int rc = need_device()) return rc;
  float *d = nullptr;
  HIPCHK(hipMalloc((void **)&d, sizeof(float) * 7 * K));
  hipError_t e = hipMemcpy(d, mids, sizeof(float) * 6 * K, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    launch_ncc(pyr_back->img[level], pyr_ref->img[level], pyr_fwd->img[level], d, (int)K, psz, pyr_ref->sw[level],
               (float)pyr_ref->w[level], (float)pyr_ref->h[level], w_back, w_fwd, d + 6 * K, nullptr);
    e = hipMemcpy(out_corr, d + 6 * K, sizeof(float) * K, hipMemcpyDeviceToHost);
  }
  hipFree(d);
  if (e != hipSuccess) return fail(ICTR_ERR_HIP, "ncc_score failed: %s", hipGetErrorString(e));
  return ICTR_OK;
}

// ---------------------------------------------------------------- Python flow-tracking surface (misc_src) on the device
// classoftrack.func_get_transf_position (classoftrack.py:4-34). disp_u / disp_v: (H, W) planes, float32 or float64,
// on the host or (fields_on_device) already on the device; disp_v may be NULL. xy, out: host (K, 2) float64.
extern "C" int ictr_flow_gather(const void *disp_u, const void *disp_v, int is_f64, int fields_on_device, int H, int W,
                                const double *xy, int64_t K, double *out) {
  if (!disp_u || H < 1 || W < 1 || K < 0 || (K > 0 && (!xy || !out)))
    return fail(ICTR_ERR_INVALID, "flow_gather: bad arguments");
  if (K == 0) return ICTR_OK;
  if (int rc = need_device()) return rc;
  const size_t fb = (size_t)H * W * (is_f64 ? 8 : 4);
  char *d_f = nullptr;
  double *d_xy = nullptr;
  hipError_t e = hipMalloc((void **)&d_xy, sizeof(double) * 4 * K);
  const void *du = disp_u, *dv = disp_v;
  if (e == hipSuccess && !fields_on_device) {
    e = hipMalloc((void **)&d_f, fb * 2);
    if (e == hipSuccess) e = hipMemcpy(d_f, disp_u, fb, hipMemcpyHostToDevice);
    if (e == hipSuccess && disp_v) e = hipMemcpy(d_f + fb, disp_v, fb, hipMemcpyHostToDevice);
    du = d_f;
    dv = disp_v ? d_f + fb : nullptr;
  }
  if (e == hipSuccess) e = hipMemcpy(d_xy, xy, sizeof(double) * 2 * K, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    launch_flow_gather(du, dv, is_f64, H, W, d_xy, (int)K, d_xy + 2 * K, nullptr);
    e = hipMemcpy(out, d_xy + 2 * K, sizeof(double) * 2 * K, hipMemcpyDeviceToHost);
  }
  if (d_f) hipFree(d_f);
  if (d_xy) hipFree(d_xy);
  if (e != hipSuccess) return fail(ICTR_ERR_HIP, "flow_gather failed: %s", hipGetErrorString(e));
  return ICTR_OK;
}

// func_OF_util.func_extract_bil_patch (func_OF_util.py:87-129), batched: img host (H, W, C) float64, pts host (K, 2)
// float64 (x, y), out host (K, side, side, C) float64 with side = 2 (pz / 2). Every window must lie inside the image.
extern "C" int ictr_extract_bil_patches(const double *img, int H, int W, int C, const double *pts, int64_t K, int pz,
                                        double *out) {
  const int half = pz / 2;
  if (!img || H < 2 || W < 2 || C < 1 || K < 0 || pz < 2 || (K > 0 && (!pts || !out)))
    return fail(ICTR_ERR_INVALID, "extract_bil_patches: bad arguments");
  for (int64_t k = 0; k < K; ++k) {
    const double fx = floor(pts[2 * k]), fy = floor(pts[2 * k + 1]);
    if (!(fx - half >= 0 && fy - half >= 0 && fx + half < W && fy + half < H))  // taps x0 .. x0 + side (ceil window)
      return fail(ICTR_ERR_INVALID, "extract_bil_patches: the window of point %lld leaves the image", (long long)k);
  }
  if (K == 0) return ICTR_OK;
  if (int rc = need_device()) return rc;
  const size_t ib = sizeof(double) * (size_t)H * W * C, ob = sizeof(double) * (size_t)K * 4 * half * half * C;
  double *d_img = nullptr, *d_pts = nullptr, *d_out = nullptr;
  hipError_t e = hipMalloc((void **)&d_img, ib);
  if (e == hipSuccess) e = hipMalloc((void **)&d_pts, sizeof(double) * 2 * K);
  if (e == hipSuccess) e = hipMalloc((void **)&d_out, ob);
  if (e == hipSuccess) e = hipMemcpy(d_img, img, ib, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_pts, pts, sizeof(double) * 2 * K, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    launch_bil_patches(d_img, H, W, C, d_pts, (int)K, half, d_out, nullptr);
    e = hipMemcpy(out, d_out, ob, hipMemcpyDeviceToHost);
  }
  for (void *p_ : {(void *)d_img, (void *)d_pts, (void *)d_out})
    if (p_) hipFree(p_);
  if (e != hipSuccess) return fail(ICTR_ERR_HIP, "extract_bil_patches failed: %s", hipGetErrorString(e));
  return ICTR_OK;
}

// ---------------------------------------------------------------- PoseClass
struct ictr_pose {
  const ictr_cam *cam;
  const ictr_optparam *op;
  double meanshift[3] = {0, 0, 0};
  double varval = 0;
  float p[6] = {0, 0, 0, 0, 0, 0};
  float G[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
};

extern "C" int ictr_pose_create(ictr_pose **out, const ictr_cam *cam, const ictr_optparam *op) {
  if (!out || !cam || !op) return fail(ICTR_ERR_INVALID, "ictr_pose_create: NULL argument");
  ictr_pose *p = new ictr_pose;
  p->cam = cam;
  p->op = op;
  *out = p;
  return ICTR_OK;
}
extern "C" void ictr_pose_destroy(ictr_pose *pose) { delete pose; }
extern "C" int ictr_pose_setpose_se3(ictr_pose *pose, const double *p_in, const double *meanshift3, double varval) {
  if (!pose || !p_in) return fail(ICTR_ERR_INVALID, "setpose_se3: NULL argument");
  if (pose->op->donorm) {
    if (!meanshift3) return fail(ICTR_ERR_INVALID, "setpose_se3: donorm needs meanshift");
    pose->varval = varval;
    memcpy(pose->meanshift, meanshift3, sizeof(double) * 3);
  }
  host_setpose(pose->op->donorm, p_in, pose->meanshift, pose->varval, pose->p, pose->G);
  return ICTR_OK;
}
extern "C" int ictr_pose_addpose_se3(ictr_pose *pose, const float *dp) {
  if (!pose || !dp) return fail(ICTR_ERR_INVALID, "addpose_se3: NULL argument");
  for (int i = 0; i < 6; ++i) pose->p[i] += dp[i];
  se3_exp<float>(pose->G, pose->p);
  return ICTR_OK;
}
extern "C" int ictr_pose_subpose_se3(ictr_pose *pose, const float *dp) {
  if (!pose || !dp) return fail(ICTR_ERR_INVALID, "subpose_se3: NULL argument");
  for (int i = 0; i < 6; ++i) pose->p[i] -= dp[i];
  se3_exp<float>(pose->G, pose->p);
  return ICTR_OK;
}
extern "C" int ictr_pose_getpose_se3(const ictr_pose *pose, double *p_out) {
  if (!pose || !p_out) return fail(ICTR_ERR_INVALID, "getpose_se3: NULL argument");
  host_getpose(pose->op->donorm, pose->p, pose->G, pose->meanshift, pose->varval, p_out);
  return ICTR_OK;
}
extern "C" int ictr_pose_get_state(const ictr_pose *pose, float *p6, float *G12) {
  if (!pose) return fail(ICTR_ERR_INVALID, "pose is NULL");
  if (p6) memcpy(p6, pose->p, sizeof(float) * 6);
  if (G12) memcpy(G12, pose->G, sizeof(float) * 12);
  return ICTR_OK;
}

static int project_impl(const ictr_pose *pose, const float *pt3d, float *pt3d_rot, float *pt2d, int64_t nopoints,
                        int sc) {
  if (!pose || !pt3d || !pt2d || nopoints < 0 || sc < 0 || sc >= pose->cam->noscales)
    return fail(ICTR_ERR_INVALID, "project_pt: bad arguments");
  if (int rc = need_device()) return rc;
  const int M = pose->op->maxpttrack;
  if (nopoints > M) return fail(ICTR_ERR_INVALID, "project_pt: nopoints > maxpttrack");
  if (nopoints == 0) return ICTR_OK;
  float *d = nullptr;
  HIPCHK(hipMalloc((void **)&d, sizeof(float) * (size_t)(8 * M + 12)));
  float *d3 = d, *dr = d + 3 * M, *d2 = d + 6 * M, *dG = d + 8 * M;
  hipError_t e = hipMemcpy(d3, pt3d, sizeof(float) * 3 * M, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dG, pose->G, sizeof(float) * 12, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d2, pt2d, sizeof(float) * 2 * M, hipMemcpyHostToDevice);
  if (e == hipSuccess && pt3d_rot) e = hipMemcpy(dr, pt3d_rot, sizeof(float) * 3 * M, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    launch_project_generic(d3, pt3d_rot ? dr : nullptr, d2, (int)nopoints, M, dG, level_cam(pose->cam, sc), nullptr);
    e = hipMemcpy(pt2d, d2, sizeof(float) * 2 * M, hipMemcpyDeviceToHost);
    if (e == hipSuccess && pt3d_rot) e = hipMemcpy(pt3d_rot, dr, sizeof(float) * 3 * M, hipMemcpyDeviceToHost);
  }
  hipFree(d);
  if (e != hipSuccess) return fail(ICTR_ERR_HIP, "project_pt failed: %s", hipGetErrorString(e));
  return ICTR_OK;
}
extern "C" int ictr_pose_project_pt(const ictr_pose *pose, const float *pt3d, float *pt2d, int64_t nopoints, int sc) {
  return project_impl(pose, pt3d, nullptr, pt2d, nopoints, sc);
}
extern "C" int ictr_pose_project_pt_save_rotated(const ictr_pose *pose, const float *pt3d, float *pt3d_rot,
                                                 float *pt2d, int64_t nopoints, int sc) {
  if (!pt3d_rot) return fail(ICTR_ERR_INVALID, "project_pt_save_rotated: pt3d_rot is NULL");
  return project_impl(pose, pt3d, pt3d_rot, pt2d, nopoints, sc);
}

// ---------------------------------------------------------------- batched engine
struct ProbHost {
  int npts = 0;
  double meanshift[3] = {0, 0, 0};
  double varval = 0;
  float p[6] = {0, 0, 0, 0, 0, 0};
  float G[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  const ictr_pyramid *ref = nullptr, *cur = nullptr;
  bool pose_set = false;
  int iters = 0;
};

struct ictr_batch {
  const ictr_cam *cam = nullptr;
  const ictr_optparam *op = nullptr;
  int B = 0, M = 0, n = 0, nlev = 0, P = 0;
  hipStream_t stream = nullptr;
  int variant = 0;
  int robust = 0;        // ICTR_ROBUST_* (off by default)
  float huber_k = 0.0f;
  int sharded = 0;
  int gridx = 1;
  int cpw = 64, gridx8 = 1;  // P=8 fast path: points per wave chunk, workgroups per problem
  bool t1_project_here = false;  // this tracking's one-launch tracker projects itself and mirrors the final records
  bool trace_on = false;
  bool projected = false;
  // device
  float *d_pt3d = nullptr, *d_pt3d_ref = nullptr, *d_pt2d = nullptr, *d_T = nullptr, *d_Gx = nullptr,
        *d_Gy = nullptr, *d_coef = nullptr, *d_partH = nullptr, *d_partb = nullptr, *d_red = nullptr;
  ProbState *d_st = nullptr;
  PlaneSet *d_planes = nullptr;
  ictr_trace_rec *d_trace = nullptr;
  int *d_trace_count = nullptr;
  int trace_cap = 0;
  // host mirrors
  std::vector<ProbHost> probs;
  std::vector<ProbState> h_st;
  std::vector<PlaneSet> h_planes;
  std::vector<float> h_pt2d;  // Get2DPoints mirror of problem 0 (or scratch)
  std::vector<float> h_stage;
  // optional HIP-event timing (bench.py): per level e0 -> setup kernel -> e1 -> iteration launches -> e2
  bool timing = false;
  std::vector<hipEvent_t> ev;  // 3 per level
  std::vector<char> ev_used;
  std::vector<hipEvent_t> evk;  // 2 per (level, iteration): around the accumulate kernel alone
  int evk_iters = 0;
  bool evk_valid = true;  // the per-iteration kernel events of the last tracking were recorded (not in the resident form)
  ResXchg xchg = {};  // sharded resident form (ictr_batch_set_peer_exchange): xchg.world > 1 = the resident launches sum
                     // H and b over the ranks themselves
  unsigned *d_xseq = nullptr;  // [B] exchange counters of that form
  int packed = 0;    // every reference pyramid of the current tracking has the interleaved planes
  int otf = 0;       // ... is builder-made (1), and some of them image-only (2): see EngineDev.otf
  int maxpts = 0;    // largest nopoints over the problems of the current tracking (set by ictr_batch_begin)
  int last_path = 0; // 0: per-iteration launches, 1: one-launch tracker (ictr_track1.hip), 2: launches replayed as a graph,
                     // 3: one-launch tracker that also carried the begin phase and wrote the host mirror
  // the per-iteration launch sequence of one tracking as an instantiated hipGraph (launch-bound sizes, enqueue_levels)
  hipGraphExec_t gexec = nullptr;
  std::string gkey;               // everything the captured launches depend on; a change rebuilds the graph
  hipStream_t cap_stream = nullptr;  // capture needs a non-null stream; nothing ever executes on it
  bool graph_broken = false;      // capture / instantiate failed once: plain launches from then on
  int phase_it = 0;  // iteration counter of the phase API (event slot of the next iter_accumulate)
  float *d_red_own = nullptr;
  // results of the last track_async: the final states are copied to pinned host memory in-stream and an event marks
  // the end, so that get_poses / the timing getters wait for THIS tracking only and the caller may already have
  // enqueued the next one on the same stream (another engine): the host runs one step ahead of the GPU
  ProbState *h_st_pin = nullptr;
  ProbState *d_st_mirror = nullptr;  // h_st_pin as the device sees it (the one-launch tracker stores final states there)
  char *h_up_pin = nullptr;  // pinned staging of the per-tracking uploads (states + plane table): truly asynchronous
  // team form of the one-launch tracker (several workgroups per problem, ictr_track1.hip "Teams")
  unsigned long long *d_team_mail = nullptr;  // granule mailboxes [B][2][team][32]; allocated on first use
  size_t team_mail_bytes = 0;
  unsigned team_epoch = 0;    // tags of a launch: epoch << 12 | exchange number
  // resident-iteration form (ictr_resident.hip): all iterations of a level in one launch, templates in registers
  unsigned long long *d_res_mail = nullptr;  // per slot: gather box + broadcast box
  size_t res_mail_bytes = 0;
  unsigned res_epoch = 0;
  int last_team = 1;          // workgroups per problem of the last one-launch tracking
  int team_target = 0;        // points per workgroup aimed at (0: automatic, < 0: no teams); ictr_batch_set_team
  int team_lo = 128, team_hi = 8192;  // problem sizes (points) served by teams: lo < maxpts <= hi
  int *h_team_err = nullptr;  // pinned: an exchange of some launch timed out (sticky)
  int *d_team_err = nullptr;  // ... as the device sees it
  hipEvent_t done_ev = nullptr, up_ev = nullptr;
  bool done_valid = false, up_pending = false;
};

// the plane table's place behind the B records in their common device block / staging buffer (16-byte aligned)
static size_t up_planes_offset(int B) { return (sizeof(ProbState) * (size_t)B + 15) / 16 * 16; }
static int env_int(const char *name, int dflt) {
  const char *s = getenv(name);
  return s ? atoi(s) : dflt;
}

static void batch_free(ictr_batch *b) {
  if (!b) return;
  for (hipEvent_t e : b->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : b->evk) (void)hipEventDestroy(e);
  if (b->gexec) (void)hipGraphExecDestroy(b->gexec);
  if (b->cap_stream) (void)hipStreamDestroy(b->cap_stream);
  if (b->done_ev) (void)hipEventDestroy(b->done_ev);
  if (b->up_ev) (void)hipEventDestroy(b->up_ev);
  if (b->h_st_pin) (void)hipHostFree(b->h_st_pin);
  if (b->h_up_pin) (void)hipHostFree(b->h_up_pin);
  if (b->h_team_err) (void)hipHostFree(b->h_team_err);
  if (b->d_team_mail) (void)hipFree(b->d_team_mail);
  if (b->d_res_mail) (void)hipFree(b->d_res_mail);
  if (b->d_xseq) (void)hipFree(b->d_xseq);
  b->d_red = b->d_red_own;
  for (void *p : {(void *)b->d_pt3d, (void *)b->d_pt3d_ref, (void *)b->d_pt2d, (void *)b->d_T, (void *)b->d_Gx,
                  (void *)b->d_Gy, (void *)b->d_coef, (void *)b->d_partH, (void *)b->d_partb, (void *)b->d_red,
                  (void *)b->d_st, (void *)b->d_trace, (void *)b->d_trace_count})  // (d_planes lives in d_st's block)
    if (p) hipFree(p);
  delete b;
}

// kernel-selection bits as the launchers see them: any robustness option routes P = 8 through the any-size kernels
static int engine_variant(const ictr_batch *b) {
  static const int env_or = [] {  // ICTR_VARIANT_OR: OR extra selection bits into every engine (whole-suite A/B runs)
    const char *v = getenv("ICTR_VARIANT_OR");
    return v ? atoi(v) : 0;
  }();
  return b->variant | env_or | (b->robust ? 2 : 0);
}

static EngineDev engine_dev(const ictr_batch *b) {
  EngineDev e;
  e.B = b->B;
  e.M = b->M;
  e.P = b->P;
  e.n = b->n;
  e.nlev = b->nlev;
  e.lv_f = b->op->lv_f;
  e.lv_l = b->op->lv_l;
  e.maxiter = b->op->maxiter;
  e.ratio = b->op->normdp_ratio;
  e.dopatchnorm = b->op->dopatchnorm ? 1 : 0;
  e.sharded = b->sharded;
  e.packed = b->packed;
  e.otf = b->otf;
  e.robust = b->robust;
  e.huber_k = b->huber_k;
  e.pt3d = b->d_pt3d;
  e.pt3d_ref = b->d_pt3d_ref;
  e.pt2d = b->d_pt2d;
  e.T = b->d_T;
  e.Gx = b->d_Gx;
  e.Gy = b->d_Gy;
  e.coef = b->d_coef;
  e.st = b->d_st;
  e.planes = b->d_planes;
  e.partH = b->d_partH;
  e.partb = b->d_partb;
  e.red = b->d_red;
  e.trace.rec = b->trace_on ? b->d_trace : nullptr;
  e.trace.count = b->d_trace_count;
  e.trace.capacity = b->trace_cap;
  return e;
}

static int check_op(const ictr_optparam *op, const ictr_cam *cam) {
  if (op->psz < 1 || op->psz > 64) return fail(ICTR_ERR_INVALID, "psz must be 1..64");
  if (op->novals != op->psz * op->psz || op->pszd2 != op->psz / 2)
    return fail(ICTR_ERR_INVALID, "optparam derived fields inconsistent (use ictr_optparam_init)");
  if (op->lv_l < 0 || op->lv_f < op->lv_l || op->lv_f >= cam->noscales)
    return fail(ICTR_ERR_INVALID, "need 0 <= lv_l <= lv_f < cam.noscales");
  if (op->maxpttrack < 1) return fail(ICTR_ERR_INVALID, "maxpttrack must be >= 1");
  if (cam->padding < op->psz) return fail(ICTR_ERR_INVALID, "camera padding must be >= psz");
  return ICTR_OK;
}

extern "C" int ictr_batch_create(ictr_batch **out, const ictr_cam *cam, const ictr_optparam *op, int64_t nproblems) {
  if (!out || !cam || !op || nproblems < 1 || nproblems > 65535)
    return fail(ICTR_ERR_INVALID, "batch_create: bad arguments (1..65535 problems)");
  if (int rc = check_op(op, cam)) return rc;
  if (int rc = need_device()) return rc;
  ictr_batch *b = new ictr_batch;
  b->cam = cam;
  b->op = op;
  b->B = (int)nproblems;
  b->M = op->maxpttrack;
  b->P = op->psz;
  b->n = op->novals;
  b->nlev = op->lv_f + 1;
  b->team_target = env_int("ICTR_TEAM_TARGET", 0);  // 0: automatic (team_points), < 0: never
  b->team_lo = env_int("ICTR_TEAM_MINPTS", 128);    // up to here ONE workgroup per problem is as fast (tools/team_sweep.py)
  b->team_hi = env_int("ICTR_TEAM_MAXPTS", 8192);   // beyond: the per-iteration kernels
  const size_t B = b->B, M = b->M, n = b->n, L = b->nlev;
  const int ppw = (b->n <= 64 && 64 % b->n == 0) ? 64 / b->n : 1;
  const int64_t groups = (b->M + ppw - 1) / ppw;
  // workgroups per problem: enough to cover the points once, capped so that B problems together stay near
  // 256 CUs x 8 resident workgroups x 2 (the rest is grid-strided)
  const int64_t cap = std::min<int64_t>(kMaxGridX, std::max<int64_t>(64, 2 * kMaxGridX / (int64_t)B));
  b->gridx = (int)std::min<int64_t>(std::max<int64_t>((groups + kWaves - 1) / kWaves, 1), cap);
  b->trace_cap = std::max(1, (int)L * std::max(1, op->maxiter));
  hipError_t e = hipSuccess;
  auto alloc = [&](void **p, size_t bytes) {
    if (e == hipSuccess) e = hipMalloc(p, bytes);
    if (e == hipSuccess) e = hipMemset(*p, 0, bytes);
  };
  alloc((void **)&b->d_pt3d, sizeof(float) * B * 3 * M);
  alloc((void **)&b->d_pt3d_ref, sizeof(float) * B * 3 * M);
  alloc((void **)&b->d_pt2d, sizeof(float) * B * L * 2 * M);
  alloc((void **)&b->d_T, sizeof(float) * B * M * n);
  alloc((void **)&b->d_Gx, sizeof(float) * B * M * n);
  alloc((void **)&b->d_Gy, sizeof(float) * B * M * n);
  alloc((void **)&b->d_coef, sizeof(float) * B * M * kCoefStride);
  alloc((void **)&b->d_partH, sizeof(float) * B * b->gridx * kPartHStride);
  alloc((void **)&b->d_partb, sizeof(float) * B * b->gridx * kPartBStride);
  alloc((void **)&b->d_red, sizeof(float) * B * kRedStride);
  // the records and the plane table in ONE block: one upload per SetPose round instead of two (each small transfer is an
  // engine switch of 5-8 us in front of the tracking's first kernel)
  alloc((void **)&b->d_st, up_planes_offset(B) + sizeof(PlaneSet) * B * L);
  if (e == hipSuccess) b->d_planes = reinterpret_cast<PlaneSet *>(reinterpret_cast<char *>(b->d_st) + up_planes_offset(B));
  alloc((void **)&b->d_trace, sizeof(ictr_trace_rec) * b->trace_cap);
  alloc((void **)&b->d_trace_count, sizeof(int));
  if (e == hipSuccess) e = hipHostMalloc((void **)&b->h_st_pin, sizeof(ProbState) * B, hipHostMallocDefault);
  if (e == hipSuccess)
    e = hipHostMalloc((void **)&b->h_up_pin, up_planes_offset(B) + sizeof(PlaneSet) * B * L, hipHostMallocDefault);
  if (e == hipSuccess && hipHostGetDevicePointer((void **)&b->d_st_mirror, b->h_st_pin, 0) != hipSuccess) {
    (void)hipGetLastError();
    b->d_st_mirror = nullptr;  // no mapped view: final states come back by copy
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&b->done_ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&b->up_ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    batch_free(b);
    return fail(ICTR_ERR_HIP, "batch_create: device allocation failed: %s", hipGetErrorString(e));
  }
  b->d_red_own = b->d_red;
  b->probs.resize(B);
  b->h_st.resize(B);
  b->h_planes.resize(B * L);
  b->h_pt2d.assign(2 * M, 0.0f);
  b->h_stage.assign(3 * M, 0.0f);
  *out = b;
  return ICTR_OK;
}
extern "C" void ictr_batch_destroy(ictr_batch *b) { batch_free(b); }
extern "C" int ictr_batch_set_stream(ictr_batch *b, void *hip_stream) {
  if (!b) return fail(ICTR_ERR_INVALID, "batch is NULL");
  b->stream = (hipStream_t)hip_stream;
  return ICTR_OK;
}
extern "C" int ictr_batch_set_robust(ictr_batch *b, int flags, float huber_k) {
  if (!b) return fail(ICTR_ERR_INVALID, "batch is NULL");
  if (flags & ~(ICTR_ROBUST_CLEAN | ICTR_ROBUST_COMPOSE | ICTR_ROBUST_HUBER))
    return fail(ICTR_ERR_INVALID, "set_robust: unknown flag bits 0x%x", flags);
  if ((flags & ICTR_ROBUST_HUBER) && !(huber_k > 0.0f))
    return fail(ICTR_ERR_INVALID, "set_robust: the Huber threshold must be positive");
  b->robust = flags;
  b->huber_k = huber_k;
  return ICTR_OK;
}
extern "C" int ictr_batch_set_team(ictr_batch *b, int target_points, int min_points, int max_points) {
  if (!b) return fail(ICTR_ERR_INVALID, "batch is NULL");
  if (min_points < 0 || max_points < min_points)
    return fail(ICTR_ERR_INVALID, "set_team: need 0 <= min_points <= max_points");
  b->team_target = target_points;
  b->team_lo = min_points;
  b->team_hi = max_points;
  return ICTR_OK;
}
extern "C" int ictr_p2p_fill_xchg_(const ictr_p2p *p, ictr::ResXchg *x);
// Sharded resident form: the batch holds this rank's SHARD of every problem's points; its resident-iteration launches
// then add H (once per level) and b (once per iteration) over the ranks themselves -- the solver workgroup of a frame
// pair writes its sums into every rank's mailbox and polls its own (ictr_p2p.hip's one-hop protocol, inside the launch).
// p: a connected ictr_p2p of at least 64 granules per problem, the same on every rank; NULL: back to a plain batch.
// Every rank must track the same sequence of (problems, levels, iteration limits); the ranks' loop decisions stay in
// lockstep because every rank solves on identical sums. Needs the resident form (8x8 patches, no robustness option, no
// patch normalisation); a tracking that cannot take it fails with ICTR_ERR_STATE rather than run unsynchronised.
extern "C" int ictr_batch_set_peer_exchange(ictr_batch *b, ictr_p2p *p) {
  if (!b) return fail(ICTR_ERR_INVALID, "batch is NULL");
  if (!p) {
    memset(&b->xchg, 0, sizeof(b->xchg));
    return ICTR_OK;
  }
  ResXchg x;
  if (ictr_p2p_fill_xchg_(p, &x)) return fail(ICTR_ERR_STATE, "set_peer_exchange: the p2p object is not connected");
  if (x.cap < (long long)kXchgPerPair * b->B)
    return fail(ICTR_ERR_INVALID, "set_peer_exchange: the mailboxes hold %lld granules per rank, %d x %d needed", x.cap,
                kXchgPerPair, b->B);
  if (!b->d_xseq) HIPCHK(hipMalloc((void **)&b->d_xseq, sizeof(unsigned) * b->B));
  HIPCHK(hipMemsetAsync(b->d_xseq, 0, sizeof(unsigned) * b->B, b->stream));
  x.xseq = b->d_xseq;
  b->xchg = x;
  return ICTR_OK;
}

extern "C" int ictr_batch_set_variant(ictr_batch *b, int variant) {
  if (!b) return fail(ICTR_ERR_INVALID, "batch is NULL");
  b->variant = variant;
  return ICTR_OK;
}

// odometer.cpp:171-239 (ResetOdometer + normalisation + f64->f32 SoA).
// given_ms/given_var: normalisation computed elsewhere (sharded runs need the GLOBAL mean / variance).
static int set3dpoints_impl(ictr_batch *b, int64_t problem, double *pt_in, int64_t nopoints_in, const double *given_ms,
                            double given_var) {
  if (!b || problem < 0 || problem >= b->B || nopoints_in < 0 || (nopoints_in > 0 && !pt_in))
    return fail(ICTR_ERR_INVALID, "set3dpoints: bad arguments");
  ProbHost &ph = b->probs[problem];
  const size_t M = b->M, n = b->n;
  // ResetOdometer (odometer.cpp:580-609): patch / sd state of this problem back to zero
  HIPCHK(hipMemsetAsync(b->d_T + problem * M * n, 0, sizeof(float) * M * n, b->stream));
  HIPCHK(hipMemsetAsync(b->d_Gx + problem * M * n, 0, sizeof(float) * M * n, b->stream));
  HIPCHK(hipMemsetAsync(b->d_Gy + problem * M * n, 0, sizeof(float) * M * n, b->stream));
  HIPCHK(hipMemsetAsync(b->d_coef + problem * M * kCoefStride, 0, sizeof(float) * M * kCoefStride, b->stream));
  ph.meanshift[0] = ph.meanshift[1] = ph.meanshift[2] = 0;
  ph.varval = 0;
  ph.npts = (int)std::min<int64_t>(nopoints_in, b->M);
  ph.pose_set = false;
  const int np = ph.npts;
  double *p1 = pt_in, *p2 = pt_in + nopoints_in, *p3 = pt_in + 2 * nopoints_in;
  std::fill(b->h_stage.begin(), b->h_stage.end(), 0.0f);
  float *s = b->h_stage.data();
  if (b->op->donorm) {
    const double nd = (double)np;
    if (given_ms) {
      memcpy(ph.meanshift, given_ms, sizeof(double) * 3);
    } else {
      for (int i = 0; i < np; ++i) ph.meanshift[0] += p1[i];
      for (int i = 0; i < np; ++i) ph.meanshift[1] += p2[i];
      for (int i = 0; i < np; ++i) ph.meanshift[2] += p3[i];
      ph.meanshift[0] /= nd;
      ph.meanshift[1] /= nd;
      ph.meanshift[2] /= nd;
    }
    for (int i = 0; i < np; ++i) {  // writes back into the caller's array, like odometer.cpp:207-212
      p1[i] -= ph.meanshift[0];
      p2[i] -= ph.meanshift[1];
      p3[i] -= ph.meanshift[2];
      ph.varval += p1[i] * p1[i] + p2[i] * p2[i] + p3[i] * p3[i];
    }
    ph.varval /= nd;  // mean squared radius (no sqrt), odometer.cpp:214
    if (given_ms) ph.varval = given_var;
    for (int i = 0; i < np; ++i) {
      s[i] = (float)(p1[i] / ph.varval);
      s[i + M] = (float)(p2[i] / ph.varval);
      s[i + 2 * M] = (float)(p3[i] / ph.varval);
    }
  } else {
    for (int i = 0; i < np; ++i) {
      s[i] = (float)p1[i];
      s[i + M] = (float)p2[i];
      s[i + 2 * M] = (float)p3[i];
    }
  }
  HIPCHK(hipMemcpyAsync(b->d_pt3d + problem * 3 * M, s, sizeof(float) * 3 * M, hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));  // h_stage is reused by the next call
  return ICTR_OK;
}
extern "C" int ictr_batch_set3dpoints(ictr_batch *b, int64_t problem, double *pt_in, int64_t nopoints_in) {
  return set3dpoints_impl(b, problem, pt_in, nopoints_in, nullptr, 0.0);
}
extern "C" int ictr_batch_set3dpoints_norm(ictr_batch *b, int64_t problem, double *pt_in, int64_t nopoints_in,
                                           const double *meanshift3, double varval) {
  if (!meanshift3) return fail(ICTR_ERR_INVALID, "set3dpoints_norm: meanshift is NULL");
  return set3dpoints_impl(b, problem, pt_in, nopoints_in, meanshift3, varval);
}
extern "C" int ictr_batch_get_norm(const ictr_batch *b, int64_t problem, double *meanshift3, double *varval) {
  if (!b || problem < 0 || problem >= b->B) return fail(ICTR_ERR_INVALID, "get_norm: bad arguments");
  if (meanshift3) memcpy(meanshift3, b->probs[problem].meanshift, sizeof(double) * 3);
  if (varval) *varval = b->probs[problem].varval;
  return ICTR_OK;
}

extern "C" int ictr_batch_setpose(ictr_batch *b, int64_t problem, const double *p_in, const ictr_pyramid *pyr_ref,
                                  const ictr_pyramid *pyr_new) {
  if (!b || problem < 0 || problem >= b->B || !p_in || !pyr_ref || !pyr_new)
    return fail(ICTR_ERR_INVALID, "setpose: bad arguments");
  for (const ictr_pyramid *py : {pyr_ref, pyr_new}) {
    if (py->nlev < b->nlev || py->pad != b->cam->padding)
      return fail(ICTR_ERR_INVALID, "setpose: pyramid has %d levels / pad %d, engine needs %d / %d", py->nlev, py->pad,
                  b->nlev, b->cam->padding);
    for (int l = 0; l < b->nlev; ++l)
      if (py->sw[l] != (int)b->cam->sw[l] || py->sh[l] < (int)b->cam->sh[l])
        return fail(ICTR_ERR_INVALID, "setpose: pyramid level %d is %dx%d, camera expects %dx%d", l, py->sw[l],
                    py->sh[l], (int)b->cam->sw[l], (int)b->cam->sh[l]);
  }
  if (!pyr_ref->getgrad) return fail(ICTR_ERR_INVALID, "setpose: reference pyramid has no gradients");
  ProbHost &ph = b->probs[problem];
  host_setpose(b->op->donorm, p_in, ph.meanshift, ph.varval, ph.p, ph.G);
  ph.ref = pyr_ref;
  ph.cur = pyr_new;
  ph.pose_set = true;
  b->projected = false;
  return ICTR_OK;
}

// SetPose for every problem of the batch in one call: p_all[6 * nproblems], one frame pair shared by all (the
// run_track_nposes shape: every pose sample tracks the same pair, run_track_nposes.cpp:232-258)
extern "C" int ictr_batch_setpose_all(ictr_batch *b, const double *p_all, const ictr_pyramid *pyr_ref,
                                      const ictr_pyramid *pyr_new) {
  if (!b || !p_all) return fail(ICTR_ERR_INVALID, "setpose_all: bad arguments");
  for (int64_t k = 0; k < b->B; ++k)
    if (int rc = ictr_batch_setpose(b, k, p_all + 6 * k, pyr_ref, pyr_new)) return rc;
  return ICTR_OK;
}

static bool use_track1(const ictr_batch *b);
// ictr_batch_begin, host part: initial states, plane table and launch geometry of the coming tracking
static int begin_prepare(ictr_batch *b) {
  if (int rc = check_op(b->op, b->cam)) return rc;
  if (b->op->maxpttrack != b->M || b->op->psz != b->P || b->op->lv_f + 1 != b->nlev)
    return fail(ICTR_ERR_STATE, "optparam maxpttrack/psz/lv_f changed after creation");
  b->done_valid = false;
  b->phase_it = 0;
  bool all_packed = true, all_builder = true, any_image_only = false;
  if (b->timing) std::fill(b->ev_used.begin(), b->ev_used.end(), 0);
  int maxpts = 0;
  for (int i = 0; i < b->B; ++i) {
    const ProbHost &ph = b->probs[i];
    if (!ph.pose_set) return fail(ICTR_ERR_STATE, "problem %d: SetPose has not been called", i);
    ProbState &st = b->h_st[i];
    memset(&st, 0, sizeof(st));
    memcpy(st.p, ph.p, sizeof(st.p));
    memcpy(st.G, ph.G, sizeof(st.G));
    st.npts = ph.npts;
    st.normdp = st.normdp_init = 1e-10f;
    maxpts = std::max(maxpts, ph.npts);
    for (int l = 0; l < b->nlev; ++l) {
      PlaneSet &ps = b->h_planes[(size_t)i * b->nlev + l];
      ps.ref = ph.ref->img[l];
      ps.dx = ph.ref->dx[l];
      ps.dy = ph.ref->dy[l];
      ps.cur = ph.cur->img[l];
      ps.pack = ph.ref->pack[l];
      if (!ps.pack) all_packed = false;
    }
    if (!ph.ref->builder_made) all_builder = false;
    if (ph.ref->getgrad == 2) any_image_only = true;
  }
  b->packed = all_packed ? 1 : 0;
  if (any_image_only && !all_builder)
    return fail(ICTR_ERR_STATE, "a batch cannot mix image-only reference pyramids (getgrad = 2) with pyramids made from "
                                "caller-supplied planes: the setup kernel reads either the planes or the image, for all "
                                "problems of a launch");
  b->otf = !all_builder ? 0 : (any_image_only ? 2 : 1);

  b->maxpts = maxpts;
  {
    // P=8 fast path geometry: a wave owns `cpw` consecutive points. Small problems get small chunks (more
    // waves, latency hidden by occupancy); large batches get 64-point chunks (coalesced stage 1, deep ILP).
    const int64_t total = (int64_t)std::max(maxpts, 1) * b->B;
    int cpw = 4;
    while (cpw < 64 && total / cpw > 32768) cpw *= 2;
    if (const char *env = getenv("ICTR_CPW")) {  // experiments only
      const int v = atoi(env);
      if (v >= 1 && v <= 64) cpw = v;
    }
    b->cpw = cpw;
    const int64_t chunks = ((int64_t)std::max(maxpts, 1) + cpw - 1) / cpw;
    const int64_t want = (chunks + kWaves - 1) / kWaves;
    const int64_t capx = std::max<int64_t>(1, (int64_t)b->gridx);  // partial buffers are sized for gridx blocks
    b->gridx8 = (int)std::min<int64_t>(std::max<int64_t>(want, 1), capx);
    if (b->gridx8 >= 64 && !getenv("ICTR_NO_XCD_BANDS"))  // multiple of 8: XCD-aware order (xcd_band_block)
      b->gridx8 = (int)std::min<int64_t>((b->gridx8 + 7) / 8 * 8, capx / 8 * 8);
  }
  if (b->otf == 2 && (b->P != 8 || b->robust || (engine_variant(b) & 2)))
    return fail(ICTR_ERR_STATE, "a reference pyramid without gradient planes (getgrad = 2: gradients formed on the fly) is "
                                "served by the 8x8 setup kernel k_ref8 only: psz 8, no robustness option, variant bit 1 "
                                "clear (small problems then run the per-iteration launches instead of the one-launch "
                                "tracker)");
  return ICTR_OK;
}
// ... device part: upload states + plane table, clear the trace counter, run step 3 for every problem
static int begin_device(ictr_batch *b, bool project = true) {
  const int maxpts = project ? b->maxpts : 0;  // (!project: the tracking's own launch projects, see track_enqueue)
  {
    const size_t nst = sizeof(ProbState) * b->B, npl = sizeof(PlaneSet) * b->h_planes.size();
    if (b->up_pending) HIPCHK(hipEventSynchronize(b->up_ev));  // the previous upload has left the staging buffer
    const size_t off = up_planes_offset(b->B);
    memcpy(b->h_up_pin, b->h_st.data(), nst);
    memcpy(b->h_up_pin + off, b->h_planes.data(), npl);
    HIPCHK(hipMemcpyAsync(b->d_st, b->h_up_pin, off + npl, hipMemcpyHostToDevice, b->stream));  // (d_planes follows d_st)
    HIPCHK(hipEventRecord(b->up_ev, b->stream));
    b->up_pending = true;
  }
  if (maxpts > 0) {  // (the launch also clears the trace counter)
    LevelCam cams[16];
    for (int l = 0; l < b->nlev; ++l) cams[l] = level_cam(b->cam, l);
    launch_project_ref(engine_dev(b), cams, maxpts, b->stream);
  } else {
    HIPCHK(hipMemsetAsync(b->d_trace_count, 0, sizeof(int), b->stream));
  }
  HIPCHK(hipGetLastError());
  return ICTR_OK;
}
extern "C" int ictr_batch_begin(ictr_batch *b) {
  if (!b) return fail(ICTR_ERR_INVALID, "batch is NULL");
  if (int rc = begin_prepare(b)) return rc;
  if (int rc = begin_device(b)) return rc;
  b->projected = true;
  return ICTR_OK;
}

extern "C" int ictr_batch_enable_sharding(ictr_batch *b, int enable) {
  if (!b) return fail(ICTR_ERR_INVALID, "batch is NULL");
  b->sharded = enable ? 1 : 0;
  return ICTR_OK;
}
extern "C" float *ictr_batch_reduction_buffer(ictr_batch *b) { return b ? b->d_red : nullptr; }

static int level_ok(ictr_batch *b, int level) {
  if (!b) return fail(ICTR_ERR_INVALID, "batch is NULL");
  if (!b->projected) return fail(ICTR_ERR_STATE, "ictr_batch_begin has not run since the last SetPose");
  if (level < b->op->lv_l || level > b->op->lv_f) return fail(ICTR_ERR_INVALID, "level out of range");
  return ICTR_OK;
}
// 1: the level phase leaves a partial H in the reduction buffer that must be summed over ranks before level_finish;
// 0 (P = 8 fast path, deferred H): H travels with the first iteration's b, the level phase needs no collective
extern "C" int ictr_batch_level_allreduce_needed(ictr_batch *b) {
  if (!b) return 1;
  return defer_h(engine_dev(b), engine_variant(b)) ? 0 : 1;
}
extern "C" int ictr_batch_level_accumulate(ictr_batch *b, int level) {
  if (int rc = level_ok(b, level)) return rc;
  b->phase_it = 0;
  launch_ref_level(engine_dev(b), level_cam(b->cam, level), level, b->gridx, engine_variant(b), b->cpw, b->gridx8, b->stream);
  HIPCHK(hipGetLastError());
  return ICTR_OK;
}
extern "C" int ictr_batch_level_finish(ictr_batch *b, int level) {
  if (int rc = level_ok(b, level)) return rc;
  b->phase_it = 0;
  if (b->sharded) launch_level_finish(engine_dev(b), engine_variant(b), b->stream);
  HIPCHK(hipGetLastError());
  return ICTR_OK;
}
extern "C" int ictr_batch_iter_accumulate(ictr_batch *b, int level) {
  if (int rc = level_ok(b, level)) return rc;
  // with timing on, HIP events bracket the accumulate kernel alone, as in the fused run (get_kernel_times)
  const bool tk = b->timing && b->phase_it < b->evk_iters && (int)b->evk.size() >= 2 * b->nlev * b->evk_iters;
  const EngineDev e = engine_dev(b);
  const LevelCam lc = level_cam(b->cam, level);
  const int first = b->phase_it == 0;
  const int ke = 2 * (level * b->evk_iters + b->phase_it);
  launch_iter_main(e, lc, level, b->gridx, engine_variant(b), b->cpw, b->gridx8, first, b->stream,
                   tk ? b->evk[ke] : nullptr, tk ? b->evk[ke + 1] : nullptr);
  if (tk && b->phase_it + 1 == std::min(b->op->maxiter, b->evk_iters)) b->ev_used[level] = 2;  // kernel events complete
  b->phase_it++;
  launch_iter_tail(e, level, b->gridx, engine_variant(b), b->gridx8, first, b->stream);
  HIPCHK(hipGetLastError());
  return ICTR_OK;
}
extern "C" int ictr_batch_iter_finish(ictr_batch *b, int level) {
  if (int rc = level_ok(b, level)) return rc;
  if (b->sharded) launch_iter_finish(engine_dev(b), level, engine_variant(b), b->phase_it == 1, b->stream);
  HIPCHK(hipGetLastError());
  return ICTR_OK;
}

// Small problems (the reference's own sizes: 50-1000 points per pair) run the whole coarse-to-fine loop in ONE launch,
// one workgroup per problem (ictr_track1.hip): below ~1000 points the per-iteration launch pairs are pure
// dependent-launch latency. Not for sharded batches (they need collectives between phases) and not when per-launch
// event timing is on. Variant bit 13 forces the per-iteration launches, bit 14 the one-launch tracker.
// Team form ("Teams", ictr_track1.hip): 8x8 problems above the one-workgroup range are shared by several workgroups
// that all-gather their partial sums through a device mailbox -- still one launch per tracking. Variant bit 19
// (524288) switches the form off (A/B).
// Points per workgroup aimed at for the current tracking; 0 = no teams. Explicit (ictr_batch_set_team, ICTR_TEAM_TARGET):
// a function of the problem size alone. Automatic (measured, tools/team_sweep.py, profiles/r02_notes.md): a lone problem
// is fastest in shares of 40 points (five patches per wave; the all-gather of up to 64 shares is one round trip);
// a batch wants all its workgroups resident at once, one per CU, down to the largest share whose patches fit the LDS
// (160 points); when even that does not fit the chip, problems of up to 384 points go back to one workgroup each
// (two per CU in the 128-register build), larger ones take the smallest team.
static int team_points(const ictr_batch *b) {
  if (b->team_target != 0) return b->team_target >= 8 ? b->team_target : 0;
  const int n = b->maxpts;
  const int upper = std::min(64, (n + 39) / 40), lower = (n + 159) / 160;
  static const int n_cu = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1)
      v = 256;
    return v;
  }();
  const int fit = std::max(1, n_cu / std::max(1, b->B));
  if (fit < lower && n <= 384) return 0;
  const int team = std::min(upper, std::max(lower, fit));
  return team < 2 ? 0 : (n + team - 1) / team;
}
// workgroups per problem of the current tracking, 1 = not the team form
static int track1_team(const ictr_batch *b) {
  static const int maxwg = env_int("ICTR_TEAM_MAXWG", 4096);   // workgroups of one launch
  const int v = engine_variant(b);
  if ((v & (1 << 19)) || b->P != 8 || b->robust) return 1;
  if (b->maxpts <= b->team_lo || b->maxpts > b->team_hi) return 1;
  if ((int64_t)b->nlev * (1 + std::max(0, b->op->maxiter)) >= 4000) return 1;  // exchange number: 12 bits of the tag
  const int target = team_points(b);
  if (target < 1) return 1;
  const int team = track1_team_size(b->maxpts, target);
  if (team < 2 || (int64_t)team * b->B > maxwg) return 1;
  return team;
}
static bool resident_takes(const ictr_batch *b);  // (resident_plan(b).parts > 0, defined with the plan below)
static bool use_track1(const ictr_batch *b) {
  const int v = engine_variant(b);
  if (b->sharded || b->timing || (v & 8192)) return false;
  if (b->xchg.world > 1) return false;  // sharded resident form: only k_level_resident exchanges with the peer ranks
  if (b->otf == 2) return false;  // image-only reference pyramids: only k_ref8 forms the gradient patches on the fly
  if (b->maxpts < 8193 && resident_takes(b)) return false;  // a large batch of mid-size problems: the resident form
  if (b->maxpts < 1) return false;
  if (track1_team(b) > 1) return true;
  if ((size_t)b->maxpts * 64 > 128 * 1024) return false;  // point records must fit in LDS
  if (v & 16384) return true;
  // Measured (tools/latency.py, r02): one problem costs 0.17 ms + 1.9 us per 8x8 patch in one launch against a flat
  // 0.52 ms of dependent launches -> cross-over near 190 points; a batch of independent problems (run_track_nposes:
  // one workgroup per pose sample, all CUs busy) still wins at 300 points each (64 x 300: 0.87 vs 0.98 ms).
  static const int forced = [] {
    const char *s = getenv("ICTR_TRACK1_MAXPTS");
    return s ? atoi(s) : 0;
  }();
  const int limit = forced > 0 ? forced : (b->B >= 16 ? 384 : 192);
  return (int64_t)b->maxpts * b->n <= (int64_t)limit * 64;
}
static int track1_waves(const ictr_batch *b) {
  // Always the same workgroup shape: which wave owns which patch -- and with it the order of every sum -- then depends
  // on the problem's own point count only, so a problem gives the same bits whatever else shares its launch
  // (run_track_nposes: any split of the pose samples over batches or ranks writes the same file).
  static const int forced = [] {
    const char *s = getenv("ICTR_TRACK1_WAVES");
    return s ? atoi(s) : 0;
  }();
  (void)b;
  return forced > 0 ? forced : 8;
}

// Admission of team launches. A team's workgroups wait for each other inside the kernel, so they must all become
// resident. One launch alone is safe whatever its size (in-order dispatch: the lowest unfinished team always gets its
// CUs). Several team launches on different streams are dispatched interleaved, each with at most ONE partly resident
// team at its dispatch front; if those fronts could fill every CU, nobody would ever be complete. So the launches in
// flight (process-wide, any batch, any stream) are kept to sum(team - 1) < CUs: a launch that would exceed it first
// makes its stream wait (hipStreamWaitEvent, the host does not block) for the oldest team launches still in flight.
// With teams of at most 64 workgroups and 256 CUs that is four concurrent launches of the largest team, more of smaller.
struct TeamFlight {
  hipEvent_t ev;
  int weight;  // quarter-CU slots, see team_launch
};
struct TeamDevice {  // per device: launches in flight (oldest first), recycled events, CU count
  std::deque<TeamFlight> flights;
  std::vector<hipEvent_t> events;
  int n_cu = 0;
};
static std::mutex g_team_mu;
static std::map<int, TeamDevice> g_team_dev;
static int cu_count_of(int dev) {
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) {
    (void)hipGetLastError();
    v = 256;
  }
  return v;
}
static int team_cu_count() {  // of the calling thread's current device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    dev = 0;
  }
  std::lock_guard<std::mutex> lk(g_team_mu);
  TeamDevice &d = g_team_dev[dev];
  if (d.n_cu == 0) d.n_cu = cu_count_of(dev);
  return d.n_cu;
}
// Admit + launch + record as ONE critical section (two host threads driving two engines must not both pass the budget
// test before either launch is visible): make `s` wait until the launch fits beside the launches in flight on this
// device, run `launch` (which enqueues the kernel on `s`), record an event behind it and enter it in the flight list.
// weight, in quarter-CU slots (four workgroups of the resident-iteration kernel share a CU): a team launch
// 4 (team - 1) -- its partly resident dispatch front, a whole CU per workgroup --, a resident-iteration launch one per
// workgroup x (4 / workgroups per CU): ALL of them must be resident. Budget 4 CUs - 4: sum(team - 1) < CUs as before;
// a resident launch that fills every slot but four fits alone.
template <class F>
static int team_launch(int weight, hipStream_t s, F &&launch) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    dev = 0;
  }
  std::lock_guard<std::mutex> lk(g_team_mu);
  TeamDevice &d = g_team_dev[dev];
  if (d.n_cu == 0) d.n_cu = cu_count_of(dev);
  while (!d.flights.empty() && hipEventQuery(d.flights.front().ev) == hipSuccess) {  // retire finished ones
    d.events.push_back(d.flights.front().ev);
    d.flights.pop_front();
  }
  (void)hipGetLastError();  // hipEventQuery's "not ready" is not an error
  int load = 0;
  for (const TeamFlight &f : d.flights) load += f.weight;
  static const int off = env_int("ICTR_TEAM_NO_ADMISSION", 0);  // A/B only: shows what the admission is for
  const int budget = off ? (1 << 30) : 4 * d.n_cu - 4;
  for (size_t i = 0; i < d.flights.size() && load + weight > budget; ++i) {
    HIPCHK(hipStreamWaitEvent(s, d.flights[i].ev, 0));  // this launch starts behind flight i
    load -= d.flights[i].weight;
  }
  hipEvent_t ev = nullptr;
  if (!d.events.empty()) {
    ev = d.events.back();
    d.events.pop_back();
  } else {
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  }
  const int rc = launch();
  hipError_t er = rc == ICTR_OK ? hipEventRecord(ev, s) : hipSuccess;
  if (rc != ICTR_OK || er != hipSuccess) {
    d.events.push_back(ev);  // nothing is in flight behind this event: keep it for the next launch
    if (rc != ICTR_OK) return rc;
    return fail(ICTR_ERR_HIP, "team_launch: hipEventRecord failed: %s", hipGetErrorString(er));
  }
  d.flights.push_back(TeamFlight{ev, weight});
  return ICTR_OK;
}

// bound of an in-launch poll (team form, resident-iteration form), seconds. Read at every launch: tests shorten it
static double team_timeout_s() {
  const char *s = getenv("ICTR_TEAM_TIMEOUT_S");
  return s ? std::max(0.001, atof(s)) : 5.0;
}
// mailbox, tag epoch and error flag of the next team launch (tm->team == 1: not a team launch, nothing allocated)
static int team_prepare(ictr_batch *b, T1Team *tm) {
  memset(tm, 0, sizeof(*tm));
  tm->team = track1_team(b);
  if (tm->team < 2) {
    tm->team = 1;
    return ICTR_OK;
  }
  tm->q = track1_team_q(b->maxpts, team_points(b));
  const size_t need = track1_team_mail_bytes(b->B, tm->team);
  if (need > b->team_mail_bytes) {
    if (b->d_team_mail) {
      HIPCHK(hipStreamSynchronize(b->stream));  // an earlier launch may still be polling the old mailbox
      HIPCHK(hipFree(b->d_team_mail));
      b->d_team_mail = nullptr;
      b->team_mail_bytes = 0;
    }
    // granules are written and polled with agent-scope accesses; uncached device memory keeps them out of the L2s
    hipError_t e = hipExtMallocWithFlags((void **)&b->d_team_mail, need, hipDeviceMallocUncached);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      e = hipMalloc((void **)&b->d_team_mail, need);
    }
    if (e == hipSuccess) e = hipMemsetAsync(b->d_team_mail, 0, need, b->stream);  // tag 0: "nothing yet"
    if (e != hipSuccess) return fail(ICTR_ERR_HIP, "team mailbox allocation failed: %s", hipGetErrorString(e));
    b->team_mail_bytes = need;
    b->team_epoch = 0;
  }
  if (!b->h_team_err) {
    HIPCHK(hipHostMalloc((void **)&b->h_team_err, sizeof(int), hipHostMallocDefault));
    *b->h_team_err = 0;
    HIPCHK(hipHostGetDevicePointer((void **)&b->d_team_err, b->h_team_err, 0));
  }
  b->team_epoch += 1;
  if (b->team_epoch >= (1u << 20)) {  // the epoch field wrapped: forget every old tag
    HIPCHK(hipMemsetAsync(b->d_team_mail, 0, b->team_mail_bytes, b->stream));
    b->team_epoch = 1;
  }
  const double limit_s = team_timeout_s();
  tm->tag0 = b->team_epoch << 12;
  tm->mute = (engine_variant(b) & (1 << 25)) ? 1 : 0;  // debug: part 0 never posts (the time-out path's test)
  tm->limit = (unsigned long long)(limit_s * 1e8);
  tm->mail = b->d_team_mail;
  tm->err = b->d_team_err;
  return ICTR_OK;
}

// Launch-bound sizes (a few hundred to a few thousand points: every kernel of the per-iteration form runs 2-5 us)
// replay the whole launch sequence of a tracking -- (setup + tail) per level, (accumulate + tail) per iteration, 111
// kernels for 5 levels x 10 iterations -- as ONE instantiated hipGraph: the host pays one graph launch instead of 111
// kernel launches and the GPU finds the next packet already queued. The graph is captured once per batch and reused for
// as long as nothing the launches depend on changes (kernel arguments are passed by value: pointers, sizes, options,
// camera, grid shapes -- all of it goes into the key). Variant bit 15 (32768) keeps the plain launches (A/B).
struct ResPlan {  // resident-iteration form (below): worker workgroups per frame pair, pairs in flight; 0 = not this form
  int parts = 0, slots = 0;
  int np = 32;    // patches per wave of the kernel instantiation (32: four 1080p pairs in flight; 16: one or two pairs)
  int fused = 0;  // variant bit 26 (67108864): the level's setup inside the launch (no k_ref8 launch, no template round trip
                  // through HBM). Measured slower -- 4.39 against 4.04 ms per 32 pairs: the setup is ~3500 instructions per
                  // wave and runs on ONE wave per SIMD there (80 us per pair and level) instead of on every wave slot of the
                  // chip in k_ref8 (48 us per pair equivalent) -- so it stays an A/B form (profiles/r03_notes.md)
};
static ResPlan resident_plan(const ictr_batch *b);
static bool use_graph(const ictr_batch *b) {
  if (b->sharded || b->timing || b->graph_broken || (engine_variant(b) & 32768)) return false;
  if (resident_plan(b).parts > 0) return false;  // three launches per level, admission events: nothing to replay
  static const int64_t limit = [] {
    const char *s = getenv("ICTR_GRAPH_MAXPTS");  // total points of a batch up to which the graph is used; 0 = never
    return s ? (int64_t)atoll(s) : (int64_t)65536;
  }();
  return (int64_t)b->maxpts * b->B <= limit;
}
template <class T>
static void key_put(std::string &k, const T &v) { k.append(reinterpret_cast<const char *>(&v), sizeof(T)); }
static std::string graph_key(const ictr_batch *b, const EngineDev &e) {
  std::string k;
  for (int v : {e.B, e.M, e.P, e.n, e.nlev, e.lv_f, e.lv_l, e.maxiter, e.dopatchnorm, e.sharded, e.packed, e.robust,
                e.trace.capacity, engine_variant(b), b->cpw, b->gridx, b->gridx8})
    key_put(k, v);
  key_put(k, e.ratio);
  key_put(k, e.huber_k);
  for (const void *q : {(const void *)e.pt3d, (const void *)e.pt3d_ref, (const void *)e.pt2d, (const void *)e.T,
                        (const void *)e.Gx, (const void *)e.Gy, (const void *)e.coef, (const void *)e.st,
                        (const void *)e.planes, (const void *)e.partH, (const void *)e.partb, (const void *)e.red,
                        (const void *)e.trace.rec, (const void *)e.trace.count})
    key_put(k, q);
  for (int l = 0; l < b->nlev; ++l) {
    const LevelCam lc = level_cam(b->cam, l);
    for (float v : {lc.fx, lc.fy, lc.cx, lc.cy, lc.swo, lc.sho}) key_put(k, v);
    key_put(k, lc.sw);
  }
  return k;
}

// Resident-iteration form (ictr_resident.hip): problems of thousands of 8x8 patches run all iterations of a level in
// ONE launch with their templates resident in registers -- `parts` worker workgroups of 128 points + one solver
// workgroup per frame pair, `slots` pairs in flight (two workgroups per CU) -- instead of streaming T/Gx/Gy from HBM in
// every iteration. Needs every workgroup of the launch resident at once: slots * (parts + 1) <= CUs * occupancy.
// An iteration is then a latency chain of ~10 us per pair with two pairs in flight: the form of choice for ONE or a few
// dense frame pairs (one 1080p pair: 0.42 against 0.72 ms; 8 pairs 1.66 against 1.82), while a large batch on two
// streams is served as well by the streaming kernels at the HBM roofline (32 pairs: 6.5 against 6.2-6.4 ms), so the
// default is this form up to ICTR_RESIDENT_MAXB = 8 pairs per engine. Variant bit 21 (2097152) or ICTR_RESIDENT=0:
// never; variant bit 23 (8388608): whatever the batch size (A/B).
static ResPlan resident_plan(const ictr_batch *b) {
  ResPlan p;
  static const int on = env_int("ICTR_RESIDENT", 1);
  static const int min_pts = env_int("ICTR_RESIDENT_MINPTS", 8193);  // below: the one-launch tracker's team form
  const int v = engine_variant(b);
  if (!on || (v & ((1 << 21) | 8192 | 4096)) || (v & 2) || b->P != 8 || b->robust || b->sharded || b->op->dopatchnorm || !(b->packed || b->otf == 2))
    return p;
  const bool xchg = b->xchg.world > 1;  // sharded resident form: any shard size (an empty shard still runs its solvers)
  // batches of mid-size problems with >= 48 000 points together also run faster here than as teams of the one-launch
  // tracker (r03, tools/mid_ab.py: 64 x 1000 points 0.88 -> 0.76 ms, 16 x 3000 0.91 -> 0.56, 12 x 4000 0.91 -> 0.56,
  // 8 x 6000 0.97 -> 0.58, 128 x 500 0.85 -> 0.75, 256 x 1000 3.39 -> 2.06; below that total the teams win: 8 x 5000
  // 0.49 against 0.53, 16 x 2500 0.49 / 0.54, 32 x 800 0.43 / 0.48; so do problems of 300 points at any batch size)
  static const int64_t batch_total = env_int("ICTR_RESIDENT_BATCH_MINTOTAL", 48000);
  const bool big_batch = batch_total > 0 && b->maxpts >= 500 && (int64_t)b->B * b->maxpts >= batch_total;
  if ((b->maxpts < min_pts && !xchg && !big_batch) || b->op->maxiter < 1) return p;
  static const int max_b = env_int("ICTR_RESIDENT_MAXB", 1 << 20);
  if (b->B > max_b && !(v & (1 << 23))) return p;
  static const int max_slots = env_int("ICTR_RESIDENT_SLOTS", 1 << 20);  // experiments: pairs in flight per launch
  static const int force_np = env_int("ICTR_RESIDENT_NP", 0);             // experiments: 16 or 32
  // sixteen patches per wave (twice the workgroups, half the patch loop) when ALL pairs of the batch are then in flight
  // at once; thirty-two (the most templates a CU can hold: four 1080p pairs in flight) otherwise
  // (the fused setup gathers from the three reference planes: not with image-only pyramids, not with a peer exchange)
  p.fused = ((v & (1 << 26)) && !xchg && b->otf != 2) ? 1 : 0;
  for (int np : {16, 32}) {
    if (force_np && np != force_np) continue;
    const int bpc = resident_blocks_per_cu(np, p.fused);
    if (bpc < 1) continue;
    const int q = resident_points_per_workgroup(np);
    const int parts = std::max(1, (b->maxpts + q - 1) / q);
    const int64_t capacity = (int64_t)bpc * team_cu_count();
    const int slots = (int)std::min<int64_t>(std::min<int64_t>(b->B, max_slots), capacity / (parts + 1));
    if (slots < 1) continue;
    if (np == 16 && slots < b->B && !force_np && !xchg) continue;
    if ((int64_t)((b->B + slots - 1) / slots) * b->op->maxiter >= 4000) return p;  // exchange number: 12 bits of the tag
    p.parts = parts;
    p.slots = slots;
    p.np = np;
    return p;
  }
  return p;
}
static bool resident_takes(const ictr_batch *b) { return resident_plan(b).parts > 0; }
// one level's iterations as ONE resident launch (behind the level's setup launches on the same stream)
static int launch_resident(ictr_batch *b, const EngineDev &e, const LevelCam &lc, int level, const ResPlan &p, int nblk,
                           hipStream_t s) {
  const size_t need = resident_mail_bytes(p.parts, p.slots);
  if (need > b->res_mail_bytes) {
    if (b->d_res_mail) {
      HIPCHK(hipStreamSynchronize(s));
      HIPCHK(hipFree(b->d_res_mail));
      b->d_res_mail = nullptr;
      b->res_mail_bytes = 0;
    }
    hipError_t er = hipExtMallocWithFlags((void **)&b->d_res_mail, need, hipDeviceMallocUncached);
    if (er != hipSuccess) {
      (void)hipGetLastError();
      er = hipMalloc((void **)&b->d_res_mail, need);
    }
    if (er == hipSuccess) er = hipMemsetAsync(b->d_res_mail, 0, need, s);  // tag 0: "nothing yet"
    if (er != hipSuccess) return fail(ICTR_ERR_HIP, "resident mailbox allocation failed: %s", hipGetErrorString(er));
    b->res_mail_bytes = need;
    b->res_epoch = 0;
  }
  if (!b->h_team_err) {
    HIPCHK(hipHostMalloc((void **)&b->h_team_err, sizeof(int), hipHostMallocDefault));
    *b->h_team_err = 0;
    HIPCHK(hipHostGetDevicePointer((void **)&b->d_team_err, b->h_team_err, 0));
  }
  b->res_epoch += 1;
  if (b->res_epoch >= (1u << 20)) {
    HIPCHK(hipMemsetAsync(b->d_res_mail, 0, b->res_mail_bytes, s));
    b->res_epoch = 1;
  }
  const double limit_s = team_timeout_s();
  // every workgroup of the launch must be resident: it starts when its slots are free of team / resident launches
  const int bpc = std::max(1, std::min(4, resident_blocks_per_cu(p.np, p.fused)));
  const int weight = p.slots * (p.parts + 1) * (4 / bpc);
  const int mute = (engine_variant(b) & (1 << 25)) ? 1 : 0;  // debug: worker 0 never posts its sums (time-out test)
  static const int prio_mode = env_int("ICTR_RESIDENT_PRIO", 2);  // rotating wave priorities: 4.29 -> 4.03 ms per 32 pairs (r03 notes)
  return team_launch(weight, s, [&]() -> int {
    HIPCHK(launch_level_resident(e, lc, level, p.np, p.fused, p.parts, p.slots, nblk, b->res_epoch << 12,
                                 (unsigned long long)(limit_s * 1e8), b->d_res_mail, b->d_team_err, mute, prio_mode,
                                 b->xchg.world > 1 ? &b->xchg : nullptr, s));
    return ICTR_OK;
  });
}

// split launchers (ictr_kernels.hip): accumulate kernel and tail kernel separately, so that events can bracket
// the accumulate kernel alone
static int enqueue_level_kernels(ictr_batch *b, const EngineDev &e, hipStream_t s, bool events) {
  const int mi = b->op->maxiter;
  const ResPlan rp = resident_plan(b);
  if (rp.parts > 0) {  // the setup launches (H included), then ONE launch for all iterations of the level
    for (int sl = b->op->lv_f; sl >= b->op->lv_l; --sl) {
      const LevelCam lc = level_cam(b->cam, sl);
      if (events) HIPCHK(hipEventRecord(b->ev[3 * sl + 0], s));
      // The setup launch's chunk size in the resident path (the resident launch itself has its own geometry; the
      // batch's chunk size serves the per-iteration kernels). (1) At least 16 points per wave chunk (32 from three
      // problems on): the setup leaves one H partial per workgroup, and the pair's solver workgroup sums them before the
      // first iteration -- 2025 of them with the 4-point chunks a single dense pair gets. One / two / four dense 1080p
      // pairs: 0.42 / 0.52 / 0.73 -> 0.38 / 0.45 / 0.59 ms per tracking. (2) 64 at the coarser levels of batches of
      // eight or more -- frames that fit the caches: 32 pairs x 32 400 points 339-345 / 265-269 / 240-250 us at levels
      // 0 / 1 / 2 with 32, 364-375 / 253-260 / 214-225 with 64 (profiles/r03_notes.md 10, 12).
      int cpw_l = b->cpw, g8_l = b->gridx8;
      static const int split = env_int("ICTR_REF8_CPW_BY_LEVEL", 1);
      auto blocks_for = [&](int c) {  // workgroups per problem with c points per wave chunk
        const int64_t want = (((int64_t)std::max(b->maxpts, 1) + c - 1) / c + kWaves - 1) / kWaves;
        int g = (int)std::min<int64_t>(std::max<int64_t>(want, 1), std::max(b->gridx8, 1));
        if (g >= 64) g = std::min((g + 7) / 8 * 8, b->gridx8);
        return g;
      };
      if (split && !getenv("ICTR_CPW")) {
        // ... as long as the launch keeps about two workgroups per CU (16 x 3000 points: 32 would leave 384)
        const int64_t enough = 2 * (int64_t)team_cu_count() - 16;
        const int floor_cpw = b->B <= 2 ? 16 : 32;
        for (int c = floor_cpw; c > cpw_l; c /= 2)
          if ((int64_t)b->B * blocks_for(c) >= enough) {
            cpw_l = c;
            break;
          }
        if (sl > 0 && b->B >= 8 && cpw_l >= 16 && (int64_t)b->B * blocks_for(64) >= enough) cpw_l = 64;
      }
      if (cpw_l != b->cpw) g8_l = blocks_for(cpw_l);
      if (!rp.fused) launch_ref_level(e, lc, sl, b->gridx, engine_variant(b) | 256 | (1 << 24), cpw_l, g8_l, s);
      if (events) HIPCHK(hipEventRecord(b->ev[3 * sl + 1], s));
      if (int rc = launch_resident(b, e, lc, sl, rp, g8_l, s)) return rc;
      if (events) {
        HIPCHK(hipEventRecord(b->ev[3 * sl + 2], s));
        b->ev_used[sl] = 1;
      }
    }
    b->evk_valid = false;  // no per-iteration launches: the kernel-time getters report zeros
    b->last_path = 4;
    return ICTR_OK;
  }
  const bool tk = events && (int)b->evk.size() >= 2 * b->nlev * mi && mi <= b->evk_iters;
  b->evk_valid = tk;
  for (int sl = b->op->lv_f; sl >= b->op->lv_l; --sl) {
    const LevelCam lc = level_cam(b->cam, sl);
    if (events) HIPCHK(hipEventRecord(b->ev[3 * sl + 0], s));
    launch_ref_level(e, lc, sl, b->gridx, engine_variant(b), b->cpw, b->gridx8, s);
    if (events) HIPCHK(hipEventRecord(b->ev[3 * sl + 1], s));
    for (int it = 0; it < mi; ++it) {
      const int ke = 2 * (sl * b->evk_iters + it);
      launch_iter_main(e, lc, sl, b->gridx, engine_variant(b), b->cpw, b->gridx8, it == 0, s,
                       tk ? b->evk[ke] : nullptr, tk ? b->evk[ke + 1] : nullptr);
      launch_iter_tail(e, sl, b->gridx, engine_variant(b), b->gridx8, it == 0, s);
    }
    if (events) {
      HIPCHK(hipEventRecord(b->ev[3 * sl + 2], s));
      b->ev_used[sl] = 1;
    }
  }
  return ICTR_OK;
}

// (re)build the graph of the current tracking's launches; false: fall back to plain launches for good
static bool build_graph(ictr_batch *b, const EngineDev &e, const std::string &key) {
  if (b->gexec) {
    (void)hipGraphExecDestroy(b->gexec);
    b->gexec = nullptr;
  }
  b->gkey.clear();
  if (!b->cap_stream && hipStreamCreateWithFlags(&b->cap_stream, hipStreamNonBlocking) != hipSuccess) return false;
  if (hipStreamBeginCapture(b->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return false;
  const int rc = enqueue_level_kernels(b, e, b->cap_stream, false);
  hipGraph_t g = nullptr;
  const hipError_t ec = hipStreamEndCapture(b->cap_stream, &g);
  bool ok = rc == ICTR_OK && ec == hipSuccess && g != nullptr;
  if (ok) ok = hipGraphInstantiate(&b->gexec, g, nullptr, nullptr, 0) == hipSuccess;
  if (g) (void)hipGraphDestroy(g);
  if (!ok) {
    (void)hipGetLastError();  // clear the sticky error of the failed capture
    b->gexec = nullptr;
    return false;
  }
  b->gkey = key;
  return true;
}

static int enqueue_levels(ictr_batch *b) {
  const EngineDev e = engine_dev(b);
  b->last_path = 0;
  if (b->xchg.world > 1 && resident_plan(b).parts < 1)
    return fail(ICTR_ERR_STATE, "a peer exchange is set (sharded resident form) but this tracking cannot run in the "
                                "resident-iteration form (8x8 patches, no robustness option, no patch normalisation, builder-"
                                "made pyramids, at most 4000 pair-rounds x iterations per level)");
  if (use_track1(b)) {
    LevelCam cams[16];
    for (int l = 0; l < b->nlev; ++l) cams[l] = level_cam(b->cam, l);
    T1Team tm;
    if (int rc = team_prepare(b, &tm)) return rc;
    auto launch = [&]() -> int {
      HIPCHK(launch_track1(e, cams, b->maxpts, track1_waves(b), nullptr, b->t1_project_here ? b->d_st_mirror : nullptr,
                           b->stream, tm.team > 1 ? &tm : nullptr, b->t1_project_here));
      return ICTR_OK;
    };
    if (tm.team > 1) {
      if (int rc = team_launch(4 * (tm.team - 1), b->stream, launch)) return rc;
    } else if (int rc = launch()) {
      return rc;
    }
    b->last_team = tm.team;
    b->last_path = 1;
    return ICTR_OK;
  }
  if (use_graph(b)) {
    const std::string key = graph_key(b, e);
    if ((b->gexec && key == b->gkey) || build_graph(b, e, key)) {
      HIPCHK(hipGraphLaunch(b->gexec, b->stream));
      b->last_path = 2;
      return ICTR_OK;
    }
    b->graph_broken = true;
  }
  if (b->timing) std::fill(b->ev_used.begin(), b->ev_used.end(), 0);
  if (int rc = enqueue_level_kernels(b, e, b->stream, b->timing)) return rc;
  HIPCHK(hipGetLastError());
  return ICTR_OK;
}

// One tracking of every problem, enqueued on the stream up to and including the final states' way into the pinned host
// mirror and the event that marks them. When SetPose has not been followed by an explicit begin and the one-launch
// tracker is the form to use, a small batch goes out as ONE launch that also carries ictr_batch_begin's device part in
// its arguments (T1Args in ictr_track1.hip) and writes the final states to the mirror itself: no upload copies, no fill,
// no projection launch, no read-back copy. Variant bit 18 (262144) keeps the separate operations (A/B).
static int track_enqueue(ictr_batch *b) {
  if (b->h_team_err && *(volatile int *)b->h_team_err) {
    // the previous tracking of this batch ran into an exchange time-out (reported by its wait): let whatever it left on
    // the stream finish, then start clean -- mailbox tags carry the launch epoch, nothing of the failed launch survives
    HIPCHK(hipStreamSynchronize(b->stream));
    *(volatile int *)b->h_team_err = 0;
  }
  bool fused = false;
  if (!b->projected) {
    if (int rc = begin_prepare(b)) return rc;
    const size_t nst = sizeof(ProbState) * b->B, npl = sizeof(PlaneSet) * b->h_planes.size();
    fused = use_track1(b) && !b->trace_on && b->d_st_mirror && nst + npl <= track1_blob_bytes() &&
            !(engine_variant(b) & (1 << 18));
    // the one-launch tracker with a batch too large for the kernel arguments: the records are uploaded, the launch
    // itself projects (step 3) and writes the final records into the host's pinned mirror -- no k_project_ref launch in
    // front, no read-back copy behind (500 pose samples x 60 points: 0.42 -> 0.39 ms per frame pair)
    b->t1_project_here = !fused && use_track1(b) && !b->trace_on && b->d_st_mirror && !(engine_variant(b) & (1 << 18));
    if (!fused)
      if (int rc = begin_device(b, !b->t1_project_here)) return rc;
    b->projected = true;
  } else {
    b->t1_project_here = false;
  }
  bool mirrored = false;
  if (fused) {
    const size_t nst = sizeof(ProbState) * b->B, npl = sizeof(PlaneSet) * b->h_planes.size();
    unsigned char blob[4096];
    memcpy(blob, b->h_st.data(), nst);
    memcpy(blob + nst, b->h_planes.data(), npl);
    LevelCam cams[16];
    for (int l = 0; l < b->nlev; ++l) cams[l] = level_cam(b->cam, l);
    T1Team tm;
    if (int rc = team_prepare(b, &tm)) return rc;
    auto launch = [&]() -> int {
      HIPCHK(launch_track1(engine_dev(b), cams, b->maxpts, track1_waves(b), blob, b->d_st_mirror, b->stream,
                           tm.team > 1 ? &tm : nullptr));
      return ICTR_OK;
    };
    if (tm.team > 1) {
      if (int rc = team_launch(4 * (tm.team - 1), b->stream, launch)) return rc;
    } else if (int rc = launch()) {
      return rc;
    }
    b->last_team = tm.team;
    b->last_path = 3;
    mirrored = true;
  } else {
    if (int rc = enqueue_levels(b)) return rc;
    mirrored = b->t1_project_here && b->last_path == 1;
  }
  if (!mirrored)
    HIPCHK(hipMemcpyAsync(b->h_st_pin, b->d_st, sizeof(ProbState) * b->B, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipEventRecord(b->done_ev, b->stream));
  b->done_valid = true;
  return ICTR_OK;
}

extern "C" int ictr_batch_track_async(ictr_batch *b) {
  if (!b) return fail(ICTR_ERR_INVALID, "batch is NULL");
  if (b->sharded) return fail(ICTR_ERR_STATE, "sharded batches are driven phase by phase (see ictr.h)");
  b->projected = false;  // a batch tracking always starts from the poses of the last SetPose calls
  return track_enqueue(b);
}
// an in-launch exchange of the last tracking timed out (team form of the one-launch tracker, resident-iteration form):
// its results are invalid. The flag stays set until the next tracking is enqueued (track_enqueue).
static int team_error_check(const ictr_batch *b) {
  if (b->h_team_err && *(volatile int *)b->h_team_err)
    return fail(ICTR_ERR_HIP, "%s: a workgroup waited in vain for its peers' partial sums (in-launch exchange timed out "
                              "after %.3f s; are all workgroups of the launch resident?); the results of this tracking "
                              "are invalid",
                b->last_path == 4 ? "resident-iteration form (k_level_resident)" : "one-launch tracker, team form (k_track1_p8)",
                team_timeout_s());
  return ICTR_OK;
}
// wait for the engine's last tracking (not for whatever else was enqueued on the stream after it)
static int batch_wait(ictr_batch *b) {
  if (b->done_valid)
    HIPCHK(hipEventSynchronize(b->done_ev));
  else
    HIPCHK(hipStreamSynchronize(b->stream));
  return team_error_check(b);
}

extern "C" int ictr_batch_set_timing(ictr_batch *b, int enable) {
  if (!b) return fail(ICTR_ERR_INVALID, "batch is NULL");
  if (enable && b->ev.empty()) {
    b->ev.resize(3 * b->nlev);
    b->ev_used.assign(b->nlev, 0);
    for (auto &e : b->ev) HIPCHK(hipEventCreate(&e));
    b->evk_iters = std::max(1, b->op->maxiter);
    b->evk.resize((size_t)2 * b->nlev * b->evk_iters);
    for (auto &e : b->evk) HIPCHK(hipEventCreate(&e));
  }
  b->timing = enable != 0;
  return ICTR_OK;
}
extern "C" int ictr_batch_get_level_times(ictr_batch *b, float *ms_setup, float *ms_iters) {
  if (!b || !ms_setup || !ms_iters) return fail(ICTR_ERR_INVALID, "get_level_times: NULL argument");
  if (b->ev.empty()) return fail(ICTR_ERR_STATE, "timing was never enabled");
  if (int rc = batch_wait(b)) return rc;
  for (int l = 0; l < b->nlev; ++l) {
    ms_setup[l] = ms_iters[l] = 0.0f;
    if (b->ev_used[l] != 1) continue;
    HIPCHK(hipEventElapsedTime(&ms_setup[l], b->ev[3 * l + 0], b->ev[3 * l + 1]));
    HIPCHK(hipEventElapsedTime(&ms_iters[l], b->ev[3 * l + 1], b->ev[3 * l + 2]));
  }
  return ICTR_OK;
}
extern "C" int ictr_batch_get_kernel_times(ictr_batch *b, float *ms_kernel) {
  if (!b || !ms_kernel) return fail(ICTR_ERR_INVALID, "get_kernel_times: NULL argument");
  if (b->evk.empty()) return fail(ICTR_ERR_STATE, "timing was never enabled");
  if (int rc = batch_wait(b)) return rc;
  const int mi = std::min(b->op->maxiter, b->evk_iters);
  for (int l = 0; l < b->nlev; ++l) {
    ms_kernel[l] = 0.0f;
    if (!b->ev_used[l] || !b->evk_valid) continue;
    for (int it = 0; it < mi; ++it) {
      float ms = 0.0f;
      HIPCHK(hipEventElapsedTime(&ms, b->evk[2 * (l * b->evk_iters + it)], b->evk[2 * (l * b->evk_iters + it) + 1]));
      ms_kernel[l] += ms;
    }
  }
  return ICTR_OK;
}
// the first accumulate launch of each level alone (on the 8x8 fast path it is a different kernel instantiation: it
// also accumulates the 21 H sums), so that callers can report the regular iteration kernel separately
extern "C" int ictr_batch_get_first_iter_times(ictr_batch *b, float *ms_first) {
  if (!b || !ms_first) return fail(ICTR_ERR_INVALID, "get_first_iter_times: NULL argument");
  if (b->evk.empty()) return fail(ICTR_ERR_STATE, "timing was never enabled");
  if (int rc = batch_wait(b)) return rc;
  for (int l = 0; l < b->nlev; ++l) {
    ms_first[l] = 0.0f;
    if (!b->ev_used[l] || !b->evk_valid || b->op->maxiter < 1) continue;
    HIPCHK(hipEventElapsedTime(&ms_first[l], b->evk[2 * (l * b->evk_iters)], b->evk[2 * (l * b->evk_iters) + 1]));
  }
  return ICTR_OK;
}
// Absolute launch intervals (for callers that run several engines concurrently on different streams and need to know
// which launches overlapped): ms since the process-wide time base set by ictr_timebase_mark().
static hipEvent_t g_timebase = nullptr;
extern "C" int ictr_timebase_mark(void) {
  if (int rc = need_device()) return rc;
  if (!g_timebase) HIPCHK(hipEventCreate(&g_timebase));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipEventRecord(g_timebase, nullptr));
  HIPCHK(hipEventSynchronize(g_timebase));
  return ICTR_OK;
}
extern "C" int ictr_batch_get_kernel_intervals(ictr_batch *b, float *start_ms, float *end_ms) {
  if (!b || !start_ms || !end_ms) return fail(ICTR_ERR_INVALID, "get_kernel_intervals: NULL argument");
  if (b->evk.empty()) return fail(ICTR_ERR_STATE, "timing was never enabled");
  if (!g_timebase) return fail(ICTR_ERR_STATE, "ictr_timebase_mark has not been called");
  if (int rc = batch_wait(b)) return rc;
  const int mi = std::min(b->op->maxiter, b->evk_iters);
  for (int l = 0; l < b->nlev; ++l)
    for (int it = 0; it < b->evk_iters; ++it) {
      const int k = l * b->evk_iters + it;
      start_ms[k] = end_ms[k] = 0.0f;
      if (!b->ev_used[l] || !b->evk_valid || it >= mi) continue;
      HIPCHK(hipEventElapsedTime(&start_ms[k], g_timebase, b->evk[2 * k]));
      HIPCHK(hipEventElapsedTime(&end_ms[k], g_timebase, b->evk[2 * k + 1]));
    }
  return ICTR_OK;
}
// the same for the per-level setup launches (k_ref* + k_level_tail): [level]
extern "C" int ictr_batch_get_setup_intervals(ictr_batch *b, float *start_ms, float *end_ms) {
  if (!b || !start_ms || !end_ms) return fail(ICTR_ERR_INVALID, "get_setup_intervals: NULL argument");
  if (b->ev.empty()) return fail(ICTR_ERR_STATE, "timing was never enabled");
  if (!g_timebase) return fail(ICTR_ERR_STATE, "ictr_timebase_mark has not been called");
  if (int rc = batch_wait(b)) return rc;
  for (int l = 0; l < b->nlev; ++l) {
    start_ms[l] = end_ms[l] = 0.0f;
    if (b->ev_used[l] != 1) continue;
    HIPCHK(hipEventElapsedTime(&start_ms[l], g_timebase, b->ev[3 * l + 0]));
    HIPCHK(hipEventElapsedTime(&end_ms[l], g_timebase, b->ev[3 * l + 1]));
  }
  return ICTR_OK;
}
extern "C" int ictr_batch_last_path(const ictr_batch *b) { return b ? b->last_path : -1; }
extern "C" int ictr_batch_last_team(const ictr_batch *b) {
  return b ? ((b->last_path == 1 || b->last_path == 3) ? b->last_team : 1) : -1;
}
extern "C" int ictr_batch_set_reduction_buffer(ictr_batch *b, float *dev_ptr) {
  if (!b) return fail(ICTR_ERR_INVALID, "batch is NULL");
  b->d_red = dev_ptr ? dev_ptr : b->d_red_own;
  return ICTR_OK;
}

static int batch_fetch_state(ictr_batch *b) {
  if (b->done_valid) {
    HIPCHK(hipEventSynchronize(b->done_ev));
    memcpy(b->h_st.data(), b->h_st_pin, sizeof(ProbState) * b->B);
  } else {
    HIPCHK(hipMemcpyAsync(b->h_st.data(), b->d_st, sizeof(ProbState) * b->B, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
  }
  if (int rc = team_error_check(b)) return rc;  // never hand out the poses of a tracking whose exchanges timed out
  for (int i = 0; i < b->B; ++i) {
    memcpy(b->probs[i].p, b->h_st[i].p, sizeof(float) * 6);
    memcpy(b->probs[i].G, b->h_st[i].G, sizeof(float) * 12);
    b->probs[i].iters = b->h_st[i].total_iters;
  }
  return ICTR_OK;
}
extern "C" int ictr_batch_get_poses(ictr_batch *b, double *p_out) {
  if (!b || !p_out) return fail(ICTR_ERR_INVALID, "get_poses: NULL argument");
  if (int rc = batch_fetch_state(b)) return rc;
  for (int i = 0; i < b->B; ++i) {
    const ProbHost &ph = b->probs[i];
    host_getpose(b->op->donorm, ph.p, ph.G, ph.meanshift, ph.varval, p_out + 6 * i);
  }
  return ICTR_OK;
}
extern "C" int ictr_batch_get_iterations(ictr_batch *b, int *iters) {
  if (!b || !iters) return fail(ICTR_ERR_INVALID, "get_iterations: NULL argument");
  for (int i = 0; i < b->B; ++i) iters[i] = b->probs[i].iters;
  return ICTR_OK;
}
extern "C" int ictr_batch_get2dpoints(ictr_batch *b, int64_t problem, float *host_out) {
  if (!b || problem < 0 || problem >= b->B || !host_out) return fail(ICTR_ERR_INVALID, "get2dpoints: bad arguments");
  if (!b->projected)
    if (int rc = ictr_batch_begin(b)) return rc;
  const size_t M = b->M;
  HIPCHK(hipMemcpyAsync(host_out, b->d_pt2d + ((size_t)problem * b->nlev + b->op->lv_l) * 2 * M, sizeof(float) * 2 * M,
                        hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return ICTR_OK;
}

// ---------------------------------------------------------------- OdometerClass = batch of one + a PoseClass
struct ictr_odometer {
  ictr_batch *b = nullptr;
  ictr_pose *pose = nullptr;
  ictr_pyramid *own_ref = nullptr, *own_new = nullptr;  // uploads made by setpose_host
};

extern "C" int ictr_odometer_create(ictr_odometer **out, ictr_pose *pose, const ictr_optparam *op) {
  if (!out || !pose || !op) return fail(ICTR_ERR_INVALID, "odometer_create: NULL argument");
  ictr_batch *b = nullptr;
  if (int rc = ictr_batch_create(&b, pose->cam, op, 1)) return rc;
  ictr_odometer *o = new ictr_odometer;
  o->b = b;
  o->pose = pose;
  *out = o;
  return ICTR_OK;
}
extern "C" void ictr_odometer_destroy(ictr_odometer *o) {
  if (!o) return;
  ictr_pyramid_destroy(o->own_ref);
  ictr_pyramid_destroy(o->own_new);
  batch_free(o->b);
  delete o;
}
extern "C" int ictr_odometer_set_stream(ictr_odometer *o, void *s) {
  if (!o) return fail(ICTR_ERR_INVALID, "odometer is NULL");
  return ictr_batch_set_stream(o->b, s);
}
extern "C" int ictr_odometer_set_robust(ictr_odometer *o, int flags, float huber_k) {
  if (!o) return fail(ICTR_ERR_INVALID, "odometer is NULL");
  return ictr_batch_set_robust(o->b, flags, huber_k);
}
extern "C" int ictr_odometer_set_team(ictr_odometer *o, int target_points, int min_points, int max_points) {
  if (!o) return fail(ICTR_ERR_INVALID, "odometer is NULL");
  return ictr_batch_set_team(o->b, target_points, min_points, max_points);
}
extern "C" int ictr_odometer_set_variant(ictr_odometer *o, int v) {
  if (!o) return fail(ICTR_ERR_INVALID, "odometer is NULL");
  return ictr_batch_set_variant(o->b, v);
}
extern "C" int ictr_odometer_set3dpoints(ictr_odometer *o, double *pt_in, int64_t nopoints_in) {
  if (!o) return fail(ICTR_ERR_INVALID, "odometer is NULL");
  return ictr_batch_set3dpoints(o->b, 0, pt_in, nopoints_in);
}
extern "C" int ictr_odometer_setpose(ictr_odometer *o, const double *p_in, const ictr_pyramid *pyr_ref,
                                     const ictr_pyramid *pyr_new) {
  if (!o) return fail(ICTR_ERR_INVALID, "odometer is NULL");
  if (int rc = ictr_batch_setpose(o->b, 0, p_in, pyr_ref, pyr_new)) return rc;
  // PoseClass state follows (pose.cpp:25-76 is invoked through the odometer in the reference)
  ProbHost &ph = o->b->probs[0];
  memcpy(o->pose->meanshift, ph.meanshift, sizeof(ph.meanshift));
  o->pose->varval = ph.varval;
  memcpy(o->pose->p, ph.p, sizeof(ph.p));
  memcpy(o->pose->G, ph.G, sizeof(ph.G));
  // Step 3 (the projections) is deferred to whoever needs it first: Get2DPoints right after SetPose runs it on its own
  // (ictr_batch_get2dpoints), TrackPose folds it into its launch (track_enqueue). Argument errors surface here.
  return begin_prepare(o->b);
}
extern "C" int ictr_odometer_setpose_host(ictr_odometer *o, const double *p_in, const float **img_ref,
                                          const float **img_ref_dx, const float **img_ref_dy, const float **img_new) {
  if (!o || !img_ref || !img_ref_dx || !img_ref_dy || !img_new)
    return fail(ICTR_ERR_INVALID, "setpose_host: NULL argument");
  const ictr_cam *c = o->b->cam;
  ictr_pyramid_destroy(o->own_ref);
  ictr_pyramid_destroy(o->own_new);
  o->own_ref = o->own_new = nullptr;
  if (int rc = ictr_pyramid_create_from_host_planes(&o->own_ref, img_ref, img_ref_dx, img_ref_dy, c->wh[0], c->wh[1],
                                                    o->b->op->lv_f, c->padding))
    return rc;
  if (int rc = ictr_pyramid_create_from_host_planes(&o->own_new, img_new, nullptr, nullptr, c->wh[0], c->wh[1],
                                                    o->b->op->lv_f, c->padding))
    return rc;
  return ictr_odometer_setpose(o, p_in, o->own_ref, o->own_new);
}
extern "C" int ictr_odometer_trackpose(ictr_odometer *o, double *p_out) {
  if (!o || !p_out) return fail(ICTR_ERR_INVALID, "trackpose: NULL argument");
  ictr_batch *b = o->b;
  if (!b->probs[0].pose_set) return fail(ICTR_ERR_STATE, "TrackPose before SetPose");
  // verbosity == 2: the reference prints |delta_p|_1 after every iteration (odometer.cpp:416-417); the device
  // records every iteration (trace), printed below in the reference's format
  const bool verbose = b->op->verbosity == 2, trace_was_on = b->trace_on;
  if (verbose && !trace_was_on) {
    b->trace_on = true;
    if (b->projected) HIPCHK(hipMemsetAsync(b->d_trace_count, 0, sizeof(int), b->stream));  // else: the begin phase
  }
  int rc_enq = track_enqueue(b);
  b->trace_on = trace_was_on;
  if (rc_enq) return rc_enq;
  // like the reference, a second TrackPose without SetPose continues from the current pose with the old
  // reference projections (device state persists)
  if (int rc = batch_fetch_state(b)) return rc;
  if (verbose) {
    int c = 0;
    HIPCHK(hipMemcpy(&c, b->d_trace_count, sizeof(int), hipMemcpyDeviceToHost));
    c = std::min(c, b->trace_cap);
    std::vector<ictr_trace_rec> recs((size_t)std::max(c, 0));
    if (c > 0) HIPCHK(hipMemcpy(recs.data(), b->d_trace, sizeof(ictr_trace_rec) * c, hipMemcpyDeviceToHost));
    for (const ictr_trace_rec &r : recs) {
      const float *d = r.dp;  // delta_p.lpNorm<1>() in Eigen's redux order, as on the device
      const float nd = (fabsf(d[0]) + (fabsf(d[1]) + fabsf(d[2]))) + (fabsf(d[3]) + (fabsf(d[4]) + fabsf(d[5])));
      printf("Sc%02i,It%02i: %g\n", r.level, r.iter, nd);
    }
    fflush(stdout);
  }
  memcpy(o->pose->p, b->probs[0].p, sizeof(float) * 6);
  memcpy(o->pose->G, b->probs[0].G, sizeof(float) * 12);
  return ictr_pose_getpose_se3(o->pose, p_out);
}
extern "C" const float *ictr_odometer_get2dpoints(ictr_odometer *o) {
  if (!o) return nullptr;
  if (ictr_batch_get2dpoints(o->b, 0, o->b->h_pt2d.data()) != ICTR_OK) return nullptr;
  return o->b->h_pt2d.data();
}
extern "C" int ictr_odometer_enable_trace(ictr_odometer *o, int enable) {
  if (!o) return fail(ICTR_ERR_INVALID, "odometer is NULL");
  o->b->trace_on = enable != 0;
  return ICTR_OK;
}
extern "C" int ictr_odometer_trace(ictr_odometer *o, ictr_trace_rec *out, int64_t capacity, int64_t *count) {
  if (!o || !count) return fail(ICTR_ERR_INVALID, "trace: NULL argument");
  int c = 0;
  HIPCHK(hipMemcpy(&c, o->b->d_trace_count, sizeof(int), hipMemcpyDeviceToHost));
  c = std::min(c, o->b->trace_cap);
  *count = c;
  const int64_t ncopy = std::min<int64_t>(c, capacity);
  if (out && ncopy > 0) HIPCHK(hipMemcpy(out, o->b->d_trace, sizeof(ictr_trace_rec) * ncopy, hipMemcpyDeviceToHost));
  return ICTR_OK;
}
// which: 0 T, 1 Gx, 2 Gy (novals*M), 4 pt3d, 5 pt3d_ref (3*M), 7 sd coefficients (16*M), 8 ProbState as floats,
// 100+l: pt2d of level l (2*M)
extern "C" int ictr_batch_read_buffer(ictr_batch *b, int64_t problem, int which, float *host_out, int64_t count) {
  if (!b || !host_out || count < 0 || problem < 0 || problem >= b->B)
    return fail(ICTR_ERR_INVALID, "read_buffer: bad arguments");
  const size_t M = b->M, n = b->n, pr = (size_t)problem;
  const float *src = nullptr;
  size_t avail = 0;
  switch (which) {
    case 0: src = b->d_T + pr * M * n; avail = M * n; break;
    case 1: src = b->d_Gx + pr * M * n; avail = M * n; break;
    case 2: src = b->d_Gy + pr * M * n; avail = M * n; break;
    case 4: src = b->d_pt3d + pr * 3 * M; avail = 3 * M; break;
    case 5: src = b->d_pt3d_ref + pr * 3 * M; avail = 3 * M; break;
    case 7: src = b->d_coef + pr * M * kCoefStride; avail = M * kCoefStride; break;
    case 8: src = reinterpret_cast<const float *>(b->d_st + pr); avail = sizeof(ProbState) / sizeof(float); break;
    case 9: src = b->d_partH + pr * 8; avail = pr == 0 ? 16 : 8; break;  // k_track1 phase cycle counters (ICTR_T1_PROF builds only)
    case 10: src = b->d_partH + (size_t)b->B * 8 + pr * 4; avail = 4; break;  // ... and the solver's
    default:
      if (which >= 100 && which < 100 + b->nlev) {
        src = b->d_pt2d + (pr * b->nlev + (size_t)(which - 100)) * 2 * M;
        avail = 2 * M;
      }
  }
  if (!src || (size_t)count > avail) return fail(ICTR_ERR_INVALID, "read_buffer: unknown buffer or count too large");
  if (!b->projected && (which == 5 || which == 8 || which >= 100)) {  // what SetPose's deferred step 3 produces
    bool all_set = true;
    for (const ProbHost &ph : b->probs) all_set = all_set && ph.pose_set;
    if (all_set)
      if (int rc = ictr_batch_begin(b)) return rc;
  }
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(host_out, src, sizeof(float) * count, hipMemcpyDeviceToHost));
  return ICTR_OK;
}
extern "C" int ictr_odometer_read_buffer(ictr_odometer *o, int which, float *host_out, int64_t count) {
  if (!o) return fail(ICTR_ERR_INVALID, "odometer is NULL");
  return ictr_batch_read_buffer(o->b, 0, which, host_out, count);
}
extern "C" int ictr_odometer_get_norm(const ictr_odometer *o, double *meanshift3, double *varval) {
  if (!o) return fail(ICTR_ERR_INVALID, "odometer is NULL");
  if (meanshift3) memcpy(meanshift3, o->b->probs[0].meanshift, sizeof(double) * 3);
  if (varval) *varval = o->b->probs[0].varval;
  return ICTR_OK;
}

// ---------------------------------------------------------------- per-patch translation IC-LK (flow producer)
static thread_local float g_pf_ms = -1.0f;
extern "C" float ictr_patchflow_last_kernel_ms(void) { return g_pf_ms; }
extern "C" int ictr_patchflow(const ictr_pyramid *pa, const ictr_pyramid *pb, const float *pts, int64_t K, int psz,
                              int lv_f, int lv_l, int maxiter, float eps, float *out, int *status, int *iters) {
  if (!pa || !pb || K < 0 || (K > 0 && (!pts || !out)) || psz < 1 || psz > 32 || lv_l < 0 || lv_f < lv_l || maxiter < 0)
    return fail(ICTR_ERR_INVALID, "patchflow: bad arguments (psz must be 1..32)");
  if (lv_f >= pa->nlev || lv_f >= pb->nlev || lv_f > 15)
    return fail(ICTR_ERR_INVALID, "patchflow: pyramids have fewer than lv_f+1 levels");
  if (pa->getgrad != 1) return fail(ICTR_ERR_INVALID, "patchflow: the first pyramid needs gradient planes (getgrad = 1)");
  if (pa->pad < psz || pb->pad < psz) return fail(ICTR_ERR_INVALID, "patchflow: pyramid padding must be >= psz");
  for (int l = lv_l; l <= lv_f; ++l)
    if (pa->w[l] != pb->w[l] || pa->h[l] != pb->h[l] || pa->sw[l] != pb->sw[l])
      return fail(ICTR_ERR_INVALID, "patchflow: the two pyramids differ in size at level %d", l);
  if (K == 0) return ICTR_OK;
  if (int rc = need_device()) return rc;
  PFArgs a;
  memset(&a, 0, sizeof(a));
  for (int l = lv_l; l <= lv_f; ++l) {
    a.lv[l].a = pa->img[l];
    a.lv[l].ax = pa->dx[l];
    a.lv[l].ay = pa->dy[l];
    a.lv[l].b = pb->img[l];
    a.lv[l].sw = pa->sw[l];
    a.lv[l].swo = (float)pa->w[l];
    a.lv[l].sho = (float)pa->h[l];
    a.lv[l].scale = (float)(1 / pow(2, l));
  }
  a.lv_f = lv_f;
  a.lv_l = lv_l;
  a.P = psz;
  a.maxiter = maxiter;
  a.K = (int)K;
  a.eps2 = eps * eps;
  a.min_det = 1e-4f;
  float *d = nullptr;
  HIPCHK(hipMalloc((void **)&d, sizeof(float) * 6 * K));
  float *d_pts = d, *d_out = d + 2 * K;
  int *d_status = reinterpret_cast<int *>(d + 4 * K), *d_iters = reinterpret_cast<int *>(d + 5 * K);
  a.pts = d_pts;
  a.out = d_out;
  a.status = d_status;
  a.iters = d_iters;
  hipError_t e = hipMemcpy(d_pts, pts, sizeof(float) * 2 * K, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    static thread_local hipEvent_t ev0 = nullptr, ev1 = nullptr;  // duration of the one kernel, for callers that report it
    if (!ev0 && (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess)) ev0 = ev1 = nullptr;
    if (ev0) (void)hipEventRecord(ev0, nullptr);
    launch_patchflow(a, nullptr);
    if (ev0) (void)hipEventRecord(ev1, nullptr);
    e = hipMemcpy(out, d_out, sizeof(float) * 2 * K, hipMemcpyDeviceToHost);
    g_pf_ms = -1.0f;
    if (ev0 && e == hipSuccess && hipEventElapsedTime(&g_pf_ms, ev0, ev1) != hipSuccess) g_pf_ms = -1.0f;
    if (e == hipSuccess && status) e = hipMemcpy(status, d_status, sizeof(int) * K, hipMemcpyDeviceToHost);
    if (e == hipSuccess && iters) e = hipMemcpy(iters, d_iters, sizeof(int) * K, hipMemcpyDeviceToHost);
  }
  hipFree(d);
  if (e != hipSuccess) return fail(ICTR_ERR_HIP, "patchflow failed: %s", hipGetErrorString(e));
  return ICTR_OK;
}
