/*
 * ictr_oracle.c -- CPU restatement of the reference tracker. TEST INFRASTRUCTURE (see header).
 * All file:line citations are relative to /root/reference.
 *
 * Faithful to the reference's implementation choices on purpose (it doubles as the
 * "port" CPU baseline in bench.py): patch-major buffers, six materialised steepest-descent
 * planes, six sd_proj planes zeroed every iteration, whole-buffer sums, full-pivot LU.
 * Quirks reproduced: ceil(x+1e-5f) tap selection, stale patches / sd images for points that
 * leave the reference view, additive se(3) update, varval = mean squared radius, the f64 "1.0 +"
 * Jacobian terms narrowed to f32, f32 round trip in getPose_se3.
 */
#define _POSIX_C_SOURCE 200112L
#include "ictr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ optparam */
void orc_optparam_init(orc_optparam *op, int lv_f, int lv_l, int psz, int maxiter, float normdp_ratio,
                       int donorm, int dopatchnorm, int maxpttrack, int verbosity) {
  /* run_io_reprojection_test.cpp:112-126 */
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
  if (op->maxpttrack % 4 > 0) op->maxpttrack += 4 - op->maxpttrack % 4; /* SSEMULTIPL==4 padding */
  op->verbosity = verbosity;
}

/* ------------------------------------------------------------------ CamClass */
struct orc_cam {
  int noscales;
  float *v[8]; /* fx fy cx cy swo sho sw sh */
};

orc_cam *orc_cam_create(int noscales, const float *fc, const float *cc, const int *wh, int padding) {
  /* camera.cpp:14-45 */
  orc_cam *c = (orc_cam *)calloc(1, sizeof(orc_cam));
  c->noscales = noscales;
  for (int k = 0; k < 8; ++k) c->v[k] = (float *)calloc((size_t)noscales, sizeof(float));
  for (int i = 0; i < noscales; ++i) {
    float sc_fct = (float)(1 / pow(2, i)); /* camera.cpp:33 : double pow, narrowed */
    c->v[0][i] = sc_fct * fc[0];
    c->v[1][i] = sc_fct * fc[1];
    c->v[2][i] = sc_fct * cc[0];
    c->v[3][i] = sc_fct * cc[1];
    c->v[4][i] = sc_fct * (float)wh[0];
    c->v[5][i] = sc_fct * (float)wh[1];
    c->v[6][i] = c->v[4][i] + 2 * padding;
    c->v[7][i] = c->v[5][i] + 2 * padding;
  }
  return c;
}
void orc_cam_destroy(orc_cam *c) {
  if (!c) return;
  for (int k = 0; k < 8; ++k) free(c->v[k]);
  free(c);
}
float orc_cam_get(const orc_cam *c, int which, int sc) { return c->v[which][sc]; }

/* ------------------------------------------------------------------ SE(3) exp/log */
#define T float
#define FN(n) n##_f
#define SQRT sqrtf
#define SIN sinf
#define COS cosf
#define ACOS acosf
#define TAN tanf
#include "se3_tmpl.inc"
#undef T
#undef FN
#undef SQRT
#undef SIN
#undef COS
#undef ACOS
#undef TAN
#define T double
#define FN(n) n##_d
#define SQRT sqrt
#define SIN sin
#define COS cos
#define ACOS acos
#define TAN tan
#include "se3_tmpl.inc"
#undef T
#undef FN
#undef SQRT
#undef SIN
#undef COS
#undef ACOS
#undef TAN

/* ------------------------------------------------------------------ pyramid */
static int cv_round_half(int v) {
  /* cvRound(v*0.5): round-half-to-even, as cv::resize(dsize=Size(), fx=.5) sizes its output */
  int q = v / 2;
  if (v % 2 == 0) return q;
  return (q % 2 == 0) ? q : q + 1;
}
void orc_pyramid_level_size(int w, int h, int level, int *wl, int *hl) {
  for (int i = 0; i < level; ++i) {
    w = cv_round_half(w);
    h = cv_round_half(h);
  }
  *wl = w;
  *hl = h;
}

static void downsample_half(const float *src, int sw, int sh, float *dst, int dw, int dh) {
  /* utilities.cpp:24 cv::resize(.., .5, .5, INTER_LINEAR). For even sizes OpenCV maps this to the
   * 2x2 area-fast path: mean of the 2x2 block. For an odd size it is true bilinear at
   * src = 2*d + 0.5 with the last tap clamped. Exact for 8-bit sourced images either way. */
  if (sw == 2 * dw && sh == 2 * dh) {
    for (int y = 0; y < dh; ++y) {
      const float *r0 = src + (size_t)(2 * y) * sw;
      const float *r1 = r0 + sw;
      for (int x = 0; x < dw; ++x)
        dst[(size_t)y * dw + x] = ((r0[2 * x] + r1[2 * x]) + (r0[2 * x + 1] + r1[2 * x + 1])) * 0.25f;
    }
    return;
  }
  for (int y = 0; y < dh; ++y) {
    int y0 = 2 * y, y1 = y0 + 1;
    if (y0 > sh - 1) y0 = sh - 1;
    if (y1 > sh - 1) y1 = sh - 1;
    for (int x = 0; x < dw; ++x) {
      int x0 = 2 * x, x1 = x0 + 1;
      if (x0 > sw - 1) x0 = sw - 1;
      if (x1 > sw - 1) x1 = sw - 1;
      float top = src[(size_t)y0 * sw + x0] * 0.5f + src[(size_t)y0 * sw + x1] * 0.5f;
      float bot = src[(size_t)y1 * sw + x0] * 0.5f + src[(size_t)y1 * sw + x1] * 0.5f;
      dst[(size_t)y * dw + x] = top * 0.5f + bot * 0.5f;
    }
  }
}

static inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  if (i < 0) return -i;
  if (i >= n) return 2 * n - 2 - i;
  return i;
}

void orc_pyramid_build(const float *img, int w, int h, int lv_f, int getgrad, int pad, float **img_pyr,
                       float **dx_pyr, float **dy_pyr) {
  /* utilities.cpp:14-52 */
  float *prev = NULL;
  int pw = 0, ph = 0;
  for (int l = 0; l <= lv_f; ++l) {
    int lw, lh;
    orc_pyramid_level_size(w, h, l, &lw, &lh);
    float *cur = (float *)malloc(sizeof(float) * (size_t)lw * lh);
    if (l == 0)
      memcpy(cur, img, sizeof(float) * (size_t)lw * lh);
    else
      downsample_half(prev, pw, ph, cur, lw, lh);

    int W = lw + 2 * pad;
    /* image: BORDER_REPLICATE (utilities.cpp:39) */
    for (int y = 0; y < lh + 2 * pad; ++y) {
      int sy = y - pad;
      sy = sy < 0 ? 0 : (sy > lh - 1 ? lh - 1 : sy);
      for (int x = 0; x < W; ++x) {
        int sx = x - pad;
        sx = sx < 0 ? 0 : (sx > lw - 1 ? lw - 1 : sx);
        img_pyr[l][(size_t)y * W + x] = cur[(size_t)sy * lw + sx];
      }
    }
    if (getgrad) {
      /* cv::Sobel ksize=1: [-1 0 1], BORDER_DEFAULT = reflect-101 (utilities.cpp:30-31);
       * then zero padding (utilities.cpp:44-45) */
      memset(dx_pyr[l], 0, sizeof(float) * (size_t)W * (lh + 2 * pad));
      memset(dy_pyr[l], 0, sizeof(float) * (size_t)W * (lh + 2 * pad));
      for (int y = 0; y < lh; ++y)
        for (int x = 0; x < lw; ++x) {
          float gx = cur[(size_t)y * lw + reflect101(x + 1, lw)] - cur[(size_t)y * lw + reflect101(x - 1, lw)];
          float gy = cur[(size_t)reflect101(y + 1, lh) * lw + x] - cur[(size_t)reflect101(y - 1, lh) * lw + x];
          dx_pyr[l][(size_t)(y + pad) * W + x + pad] = gx;
          dy_pyr[l][(size_t)(y + pad) * W + x + pad] = gy;
        }
    }
    free(prev);
    prev = cur;
    pw = lw;
    ph = lh;
  }
  free(prev);
}

/* ------------------------------------------------------------------ patch fetch */
static int g_sum_double = 0;
void orc_set_sum_mode(int use_double) { g_sum_double = use_double; }

/* Whole-buffer sum. Eigen's .sum() order is unspecified; this mimics its AVX linear-vectorised
 * redux (8-lane packets, two packet accumulators, horizontal add, scalar tail). */
static float sum_f32(const float *a, size_t n) {
  if (g_sum_double) {
    double s = 0;
    for (size_t i = 0; i < n; ++i) s += a[i];
    return (float)s;
  }
  float acc0[8] = {0}, acc1[8] = {0};
  size_t n16 = n / 16 * 16, i = 0;
  for (; i < n16; i += 16)
    for (int k = 0; k < 8; ++k) {
      acc0[k] += a[i + k];
      acc1[k] += a[i + 8 + k];
    }
  size_t n8 = n / 8 * 8;
  if (i < n8) {
    for (int k = 0; k < 8; ++k) acc0[k] += a[i + k];
    i += 8;
  }
  for (int k = 0; k < 8; ++k) acc0[k] += acc1[k];
  float s = ((acc0[0] + acc0[4]) + (acc0[1] + acc0[5])) + ((acc0[2] + acc0[6]) + (acc0[3] + acc0[7]));
  for (; i < n; ++i) s += a[i];
  return s;
}
static float dot_f32(const float *a, const float *b, size_t n) {
  if (g_sum_double) {
    double s = 0;
    for (size_t i = 0; i < n; ++i) s += (double)(a[i] * b[i]);
    return (float)s;
  }
  float acc0[8] = {0}, acc1[8] = {0};
  size_t n16 = n / 16 * 16, i = 0;
  for (; i < n16; i += 16)
    for (int k = 0; k < 8; ++k) {
      acc0[k] += a[i + k] * b[i + k];
      acc1[k] += a[i + 8 + k] * b[i + 8 + k];
    }
  size_t n8 = n / 8 * 8;
  if (i < n8) {
    for (int k = 0; k < 8; ++k) acc0[k] += a[i + k] * b[i + k];
    i += 8;
  }
  for (int k = 0; k < 8; ++k) acc0[k] += acc1[k];
  float s = ((acc0[0] + acc0[4]) + (acc0[1] + acc0[5])) + ((acc0[2] + acc0[6]) + (acc0[3] + acc0[7]));
  for (; i < n; ++i) s += a[i] * b[i];
  return s;
}

typedef struct {
  float we[4];
  int col0, row0; /* first tap 'a' column / row in the padded plane */
} tapinfo;

static inline tapinfo taps(const float *mid, const orc_optparam *op) {
  /* utilities.cpp:66-77,91-95 */
  tapinfo t;
  int pos0 = (int)ceilf(mid[0] + .00001f);
  int pos1 = (int)ceilf(mid[1] + .00001f);
  int pos2 = (int)floorf(mid[0]);
  int pos3 = (int)floorf(mid[1]);
  float r0 = mid[0] - (float)pos2;
  float r1 = mid[1] - (float)pos3;
  t.we[0] = r0 * r1;
  t.we[1] = (1 - r0) * r1;
  t.we[2] = r0 * (1 - r1);
  t.we[3] = (1 - r0) * (1 - r1);
  t.col0 = pos0 + op->pszd2;
  t.row0 = pos1 + op->pszd2;
  return t;
}

static inline void fetch_plane(const float *img, const tapinfo *t, float *out, int psz, int width) {
  /* utilities.cpp:97-109 : a=(col,row) b=(col-1,row) c=(col,row-1) d=(col-1,row-1) */
  for (int j = 0; j < psz; ++j) {
    const float *a = img + (size_t)(t->row0 + j) * width + t->col0;
    const float *c = a - width;
    for (int i = 0; i < psz; ++i)
      out[j * psz + i] = t->we[0] * a[i] + t->we[1] * a[i - 1] + t->we[2] * c[i] + t->we[3] * c[i - 1];
  }
}

void orc_getpatch(const float *img, const float *mid, float *out, const orc_optparam *op, int width) {
  tapinfo t = taps(mid, op);
  fetch_plane(img, &t, out, op->psz, width);
  if (op->dopatchnorm) { /* utilities.cpp:111-112 */
    float m = sum_f32(out, (size_t)op->novals) / op->novals;
    for (int i = 0; i < op->novals; ++i) out[i] -= m;
  }
}

void orc_getpatch_grad(const float *img, const float *img_dx, const float *img_dy, const float *mid, float *out,
                       float *out_dx, float *out_dy, const orc_optparam *op, int width) {
  tapinfo t = taps(mid, op);
  fetch_plane(img, &t, out, op->psz, width);
  fetch_plane(img_dx, &t, out_dx, op->psz, width);
  fetch_plane(img_dy, &t, out_dy, op->psz, width);
  if (op->dopatchnorm) { /* utilities.cpp:187-188 : intensity patch only */
    float m = sum_f32(out, (size_t)op->novals) / op->novals;
    for (int i = 0; i < op->novals; ++i) out[i] -= m;
  }
}

/* ------------------------------------------------------------------ PoseClass */
struct orc_pose {
  const orc_cam *cam;
  const orc_optparam *op;
  double meanshift[3];
  double varval;
  float G[12];
  float p[6];
};

orc_pose *orc_pose_create(const orc_cam *cam, const orc_optparam *op) {
  orc_pose *p = (orc_pose *)calloc(1, sizeof(orc_pose));
  p->cam = cam;
  p->op = op;
  return p;
}
void orc_pose_destroy(orc_pose *p) { free(p); }
const float *orc_pose_G(const orc_pose *p) { return p->G; }
const float *orc_pose_p(const orc_pose *p) { return p->p; }

void orc_pose_setpose_se3(orc_pose *P, const double *p_in, const double *meanshift, double varval) {
  /* pose.cpp:25-76 */
  double pn[6];
  memcpy(pn, p_in, sizeof(double) * 6);
  if (P->op->donorm) {
    P->varval = varval;
    memcpy(P->meanshift, meanshift, sizeof(double) * 3);
    double G[12];
    orc_se3_exp_d(G, pn);
    double t[3];
    t[0] = -G[0] * G[3] - G[4] * G[7] - G[8] * G[11];
    t[1] = -G[1] * G[3] - G[5] * G[7] - G[9] * G[11];
    t[2] = -G[2] * G[3] - G[6] * G[7] - G[10] * G[11];
    t[0] = (t[0] - meanshift[0]) / varval;
    t[1] = (t[1] - meanshift[1]) / varval;
    t[2] = (t[2] - meanshift[2]) / varval;
    G[3] = -G[0] * t[0] - G[1] * t[1] - G[2] * t[2];
    G[7] = -G[4] * t[0] - G[5] * t[1] - G[6] * t[2];
    G[11] = -G[8] * t[0] - G[9] * t[1] - G[10] * t[2];
    orc_se3_log_d(pn, G);
  }
  for (int i = 0; i < 6; ++i) P->p[i] = (float)pn[i];
  orc_se3_exp_f(P->G, P->p);
}

void orc_pose_getpose_se3(const orc_pose *P, double *p_out) {
  /* pose.cpp:79-113 : mixed precision kept (f32 G, f64 camera centre, f32 log) */
  float pu[6];
  memcpy(pu, P->p, sizeof(float) * 6);
  if (P->op->donorm) {
    float G[12];
    memcpy(G, P->G, sizeof(float) * 12);
    double t[3];
    t[0] = (double)(-G[0] * G[3] - G[4] * G[7] - G[8] * G[11]); /* f32 expression, widened */
    t[1] = (double)(-G[1] * G[3] - G[5] * G[7] - G[9] * G[11]);
    t[2] = (double)(-G[2] * G[3] - G[6] * G[7] - G[10] * G[11]);
    t[0] = t[0] * P->varval + P->meanshift[0];
    t[1] = t[1] * P->varval + P->meanshift[1];
    t[2] = t[2] * P->varval + P->meanshift[2];
    G[3] = (float)(-G[0] * t[0] - G[1] * t[1] - G[2] * t[2]); /* f64 expression, narrowed */
    G[7] = (float)(-G[4] * t[0] - G[5] * t[1] - G[6] * t[2]);
    G[11] = (float)(-G[8] * t[0] - G[9] * t[1] - G[10] * t[2]);
    orc_se3_log_f(pu, G);
  }
  for (int i = 0; i < 6; ++i) p_out[i] = (double)pu[i];
}

void orc_pose_addpose_se3(orc_pose *P, const float *dp) {
  /* pose.cpp:116-129 : additive in se(3) coordinates, then re-exp */
  for (int i = 0; i < 6; ++i) P->p[i] += dp[i];
  orc_se3_exp_f(P->G, P->p);
}
void orc_pose_subpose_se3(orc_pose *P, const float *dp) {
  for (int i = 0; i < 6; ++i) P->p[i] -= dp[i];
  orc_se3_exp_f(P->G, P->p);
}

static void project_impl(const orc_pose *P, const float *pt3d, float *pt3d_rot, float *pt2d, int nopoints, int sc) {
  /* pose.cpp:307-397 / 400-488. The 4-wide SSE form rounds nopoints up to a multiple of 4 and
   * also processes the pad lanes; element-wise results are identical to this scalar loop. */
  const int M = P->op->maxpttrack;
  float fx = P->cam->v[0][sc], fy = P->cam->v[1][sc], cx = P->cam->v[2][sc], cy = P->cam->v[3][sc];
  const float *G = P->G;
  int n = nopoints;
  if (n % 4 > 0) n += 4 - n % 4;
  for (int i = 0; i < n; ++i) {
    float X = pt3d[i], Y = pt3d[i + M], Z = pt3d[i + 2 * M];
    float tx = G[0] * X + G[1] * Y + G[2] * Z + G[3];
    float ty = G[4] * X + G[5] * Y + G[6] * Z + G[7];
    float tz = G[8] * X + G[9] * Y + G[10] * Z + G[11];
    if (pt3d_rot) {
      pt3d_rot[i] = tx;
      pt3d_rot[i + M] = ty;
      pt3d_rot[i + 2 * M] = tz;
    }
    pt2d[i] = (tx / tz) * fx + cx;
    pt2d[i + M] = (ty / tz) * fy + cy;
  }
}
void orc_pose_project_pt(const orc_pose *P, const float *pt3d, float *pt2d, int nopoints, int sc) {
  project_impl(P, pt3d, NULL, pt2d, nopoints, sc);
}
void orc_pose_project_pt_save_rotated(const orc_pose *P, const float *pt3d, float *pt3d_rot, float *pt2d,
                                      int nopoints, int sc) {
  project_impl(P, pt3d, pt3d_rot, pt2d, nopoints, sc);
}

/* ------------------------------------------------------------------ 6x6 full-pivot LU */
void orc_solve6_fullpivlu(const float *Hin, const float *b, float *x) {
  /* Eigen::FullPivLU<Matrix<float,6,6>>::compute + solve restated (odometer.cpp:514).
   * Column-major scan for the pivot with strict '>', rank threshold eps*6*maxpivot,
   * free variables set to zero for a rank-deficient system. */
  enum { N = 6 };
  float A[N][N]; /* A[r][c] */
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) A[r][c] = Hin[r * N + c];
  int rt[N], ct[N];
  int nonzero = N;
  float maxpivot = 0.0f;
  for (int k = 0; k < N; ++k) {
    int br = k, bc = k;
    float best = fabsf(A[k][k]);
    for (int c = k; c < N; ++c)
      for (int r = k; r < N; ++r) {
        float v = fabsf(A[r][c]);
        if (v > best) {
          best = v;
          br = r;
          bc = c;
        }
      }
    if (best == 0.0f) {
      nonzero = k;
      for (int i = k; i < N; ++i) {
        rt[i] = i;
        ct[i] = i;
      }
      break;
    }
    if (best > maxpivot) maxpivot = best;
    rt[k] = br;
    ct[k] = bc;
    if (br != k)
      for (int c = 0; c < N; ++c) {
        float t = A[k][c];
        A[k][c] = A[br][c];
        A[br][c] = t;
      }
    if (bc != k)
      for (int r = 0; r < N; ++r) {
        float t = A[r][k];
        A[r][k] = A[r][bc];
        A[r][bc] = t;
      }
    if (k < N - 1) {
      for (int r = k + 1; r < N; ++r) A[r][k] /= A[k][k];
      for (int c = k + 1; c < N; ++c)
        for (int r = k + 1; r < N; ++r) A[r][c] -= A[r][k] * A[k][c];
    }
  }
  for (int i = 0; i < N; ++i) x[i] = 0.0f;
  if (nonzero == 0) return;
  float thr = maxpivot * (1.1920929e-07f * N);
  int rank = 0;
  for (int i = 0; i < nonzero; ++i) rank += (fabsf(A[i][i]) > thr);
  float c[N];
  for (int i = 0; i < N; ++i) c[i] = b[i];
  for (int k = 0; k < N; ++k)
    if (rt[k] != k) {
      float t = c[k];
      c[k] = c[rt[k]];
      c[rt[k]] = t;
    }
  for (int i = 0; i < N; ++i) /* unit lower, column oriented */
    for (int r = i + 1; r < N; ++r) c[r] -= c[i] * A[r][i];
  for (int i = rank - 1; i >= 0; --i) { /* upper, top-left rank x rank */
    c[i] /= A[i][i];
    for (int r = 0; r < i; ++r) c[r] -= c[i] * A[r][i];
  }
  for (int i = 0; i < rank; ++i) x[i] = c[i];
  for (int k = N - 1; k >= 0; --k)
    if (ct[k] != k) {
      float t = x[k];
      x[k] = x[ct[k]];
      x[ct[k]] = t;
    }
}

/* ------------------------------------------------------------------ OdometerClass */
struct orc_odometer {
  orc_pose *pose;
  const orc_optparam *op;
  double meanshift[3];
  double varval;
  const float **img_ref, **img_ref_dx, **img_ref_dy, **img_new;
  int nopoints;
  float Hes[36], sumsd[6], delta_p[6];
  unsigned char *ind_ref, *ind_new;
  float *pt3d, *pt3d_ref, **pt2d, *pt2d_new;
  float *pat_ref, *pat_ref_dx, *pat_ref_dy, *pat_new;
  float *sd[6], *sdp[6];
  float *pdiff;
  orc_trace_rec *trace;
  int ntrace, captrace;
};

static float *alloc32(size_t n) {
  void *p = NULL;
  if (posix_memalign(&p, 32, sizeof(float) * (n ? n : 1)) != 0) return NULL;
  return (float *)p;
}

static void reset_odometer(orc_odometer *o) {
  /* odometer.cpp:580-609 */
  const orc_optparam *op = o->op;
  size_t nm = (size_t)op->novals * op->maxpttrack;
  memset(o->Hes, 0, sizeof(o->Hes));
  memset(o->ind_ref, 1, (size_t)op->maxpttrack);
  memset(o->ind_new, 1, (size_t)op->maxpttrack);
  memset(o->pat_ref, 0, sizeof(float) * nm);
  memset(o->pat_ref_dx, 0, sizeof(float) * nm);
  memset(o->pat_ref_dy, 0, sizeof(float) * nm);
  memset(o->pat_new, 0, sizeof(float) * nm);
  for (int k = 0; k < 6; ++k) {
    memset(o->sd[k], 0, sizeof(float) * nm);
    memset(o->sdp[k], 0, sizeof(float) * nm);
  }
}

orc_odometer *orc_odometer_create(orc_pose *pose, const orc_optparam *op) {
  /* odometer.cpp:19-154 */
  orc_odometer *o = (orc_odometer *)calloc(1, sizeof(orc_odometer));
  o->pose = pose;
  o->op = op;
  size_t M = (size_t)op->maxpttrack, nm = (size_t)op->novals * M;
  o->pt2d = (float **)calloc((size_t)op->lv_f + 1, sizeof(float *));
  o->pt3d = alloc32(3 * M);
  o->pt3d_ref = alloc32(3 * M);
  o->pt2d_new = alloc32(2 * M);
  memset(o->pt3d, 0, sizeof(float) * 3 * M); /* reference leaves these uninitialised; zero is benign */
  memset(o->pt3d_ref, 0, sizeof(float) * 3 * M);
  memset(o->pt2d_new, 0, sizeof(float) * 2 * M);
  for (int i = 0; i <= op->lv_f; ++i) {
    o->pt2d[i] = alloc32(2 * M);
    memset(o->pt2d[i], 0, sizeof(float) * 2 * M);
  }
  o->pat_ref = alloc32(nm);
  o->pat_ref_dx = alloc32(nm);
  o->pat_ref_dy = alloc32(nm);
  o->pat_new = alloc32(nm);
  for (int k = 0; k < 6; ++k) {
    o->sd[k] = alloc32(nm);
    o->sdp[k] = alloc32(nm);
  }
  o->pdiff = alloc32((size_t)op->novals);
  o->ind_ref = (unsigned char *)malloc(M);
  o->ind_new = (unsigned char *)malloc(M);
  reset_odometer(o);
  return o;
}

void orc_odometer_destroy(orc_odometer *o) {
  if (!o) return;
  free(o->pt3d);
  free(o->pt3d_ref);
  free(o->pt2d_new);
  for (int i = 0; i <= o->op->lv_f; ++i) free(o->pt2d[i]);
  free(o->pt2d);
  free(o->pat_ref);
  free(o->pat_ref_dx);
  free(o->pat_ref_dy);
  free(o->pat_new);
  for (int k = 0; k < 6; ++k) {
    free(o->sd[k]);
    free(o->sdp[k]);
  }
  free(o->pdiff);
  free(o->ind_ref);
  free(o->ind_new);
  free(o->trace);
  free(o);
}

void orc_odometer_set3dpoints(orc_odometer *o, double *pt_in, int nopoints_in) {
  /* odometer.cpp:171-239 */
  const orc_optparam *op = o->op;
  reset_odometer(o);
  o->meanshift[0] = o->meanshift[1] = o->meanshift[2] = 0;
  o->varval = 0;
  const int M = op->maxpttrack;
  o->nopoints = nopoints_in < M ? nopoints_in : M;
  const int n = o->nopoints;
  double *p1 = pt_in, *p2 = pt_in + nopoints_in, *p3 = pt_in + 2 * (size_t)nopoints_in;
  if (op->donorm) {
    double nd = (double)n;
    for (int i = 0; i < n; ++i) o->meanshift[0] += p1[i];
    for (int i = 0; i < n; ++i) o->meanshift[1] += p2[i];
    for (int i = 0; i < n; ++i) o->meanshift[2] += p3[i];
    o->meanshift[0] /= nd;
    o->meanshift[1] /= nd;
    o->meanshift[2] /= nd;
    for (int i = 0; i < n; ++i) { /* mutates the caller's array, as the reference does (:207-212) */
      p1[i] -= o->meanshift[0];
      p2[i] -= o->meanshift[1];
      p3[i] -= o->meanshift[2];
      o->varval += p1[i] * p1[i] + p2[i] * p2[i] + p3[i] * p3[i];
    }
    o->varval /= nd; /* mean SQUARED radius, no sqrt (:214) */
    for (int i = 0; i < n; ++i) {
      o->pt3d[i] = (float)(p1[i] / o->varval);
      o->pt3d[i + M] = (float)(p2[i] / o->varval);
      o->pt3d[i + 2 * M] = (float)(p3[i] / o->varval);
    }
  } else {
    for (int i = 0; i < n; ++i) {
      o->pt3d[i] = (float)p1[i];
      o->pt3d[i + M] = (float)p2[i];
      o->pt3d[i + 2 * M] = (float)p3[i];
    }
  }
}

void orc_odometer_setpose(orc_odometer *o, const double *p_in, const float **img_ref, const float **img_ref_dx,
                          const float **img_ref_dy, const float **img_new) {
  /* odometer.cpp:241-255 */
  o->img_ref = img_ref;
  o->img_ref_dx = img_ref_dx;
  o->img_ref_dy = img_ref_dy;
  o->img_new = img_new;
  orc_pose_setpose_se3(o->pose, p_in, o->meanshift, o->varval);
  orc_pose_project_pt_save_rotated(o->pose, o->pt3d, o->pt3d_ref, o->pt2d[o->op->lv_f], o->nopoints, o->op->lv_f);
  for (int sl = o->op->lv_f - 1; sl >= o->op->lv_l; --sl)
    orc_pose_project_pt(o->pose, o->pt3d, o->pt2d[sl], o->nopoints, sl);
}

static void compute_hessian(orc_odometer *o) {
  /* odometer.cpp:428-472 : 21 whole-buffer dot products, mirrored */
  size_t nm = (size_t)o->op->novals * o->op->maxpttrack;
  for (int j = 0; j < 6; ++j)
    for (int k = j; k < 6; ++k) {
      float v = dot_f32(o->sd[j], o->sd[k], nm);
      o->Hes[j * 6 + k] = v;
      o->Hes[k * 6 + j] = v;
    }
}

static void trace_push(orc_odometer *o, int level, int iter) {
  if (o->ntrace == o->captrace) {
    o->captrace = o->captrace ? 2 * o->captrace : 64;
    o->trace = (orc_trace_rec *)realloc(o->trace, sizeof(orc_trace_rec) * (size_t)o->captrace);
  }
  orc_trace_rec *r = &o->trace[o->ntrace++];
  r->level = level;
  r->iter = iter;
  memcpy(r->H, o->Hes, sizeof(r->H));
  memcpy(r->b, o->sumsd, sizeof(r->b));
  memcpy(r->dp, o->delta_p, sizeof(r->dp));
  memcpy(r->p, o->pose->p, sizeof(r->p));
}

void orc_odometer_trackpose(orc_odometer *o, double *p_out) {
  /* odometer.cpp:257-426 */
  const orc_optparam *op = o->op;
  const orc_cam *cam = o->pose->cam;
  const int M = op->maxpttrack, nv = op->novals, n = o->nopoints;
  const size_t nm = (size_t)nv * M;
  o->ntrace = 0;

  for (int sl = op->lv_f; sl >= op->lv_l; --sl) {
    const float swo = cam->v[4][sl], sho = cam->v[5][sl];
    const int width = (int)cam->v[6][sl]; /* getsw() float -> const int width (:286) */

    /* step 4 (:268-298) */
    for (int i = 0; i < n; ++i) {
      float mid[2] = {o->pt2d[sl][i], o->pt2d[sl][i + M]};
      if ((mid[0] < 0) | (mid[1] < 0) | (mid[0] > swo) | (mid[1] > sho)) {
        o->ind_ref[i] = 0;
      } else {
        o->ind_ref[i] = 1;
        orc_getpatch_grad(o->img_ref[sl], o->img_ref_dx[sl], o->img_ref_dy[sl], mid, o->pat_ref + (size_t)i * nv,
                          o->pat_ref_dx + (size_t)i * nv, o->pat_ref_dy + (size_t)i * nv, op, width);
      }
    }

    /* step 5 (:302-328). The "1.0 +" terms are evaluated in double and narrowed to float when
     * Eigen multiplies a float expression by the scalar. */
    for (int i = 0; i < n; ++i) {
      if (!o->ind_ref[i]) continue; /* stale sd images stay (reference quirk) */
      float pt_x = o->pt3d_ref[i], pt_y = o->pt3d_ref[i + M], pt_z = o->pt3d_ref[i + 2 * M];
      float pt_zsq = pt_z * pt_z;
      float fx = cam->v[0][sl], fy = cam->v[1][sl];
      float cxk[6], cyk[6];
      cxk[0] = fx / pt_z;
      cyk[0] = 0.0f;
      cxk[1] = 0.0f;
      cyk[1] = fy / pt_z;
      cxk[2] = -pt_x / pt_zsq * fx;
      cyk[2] = -pt_y / pt_zsq * fy;
      cxk[3] = -pt_x * pt_y / pt_zsq * fx;
      cyk[3] = (float)((-(1.0 + pt_y * pt_y / pt_zsq)) * fy);
      cxk[4] = (float)((1.0 + pt_x * pt_x / pt_zsq) * fx);
      cyk[4] = pt_x * pt_y / pt_zsq * fy;
      cxk[5] = -pt_y / pt_z * fx;
      cyk[5] = pt_x / pt_z * fy;
      const float *gx = o->pat_ref_dx + (size_t)i * nv, *gy = o->pat_ref_dy + (size_t)i * nv;
      float *s0 = o->sd[0] + (size_t)i * nv, *s1 = o->sd[1] + (size_t)i * nv;
      for (int q = 0; q < nv; ++q) {
        s0[q] = gx[q] * cxk[0];
        s1[q] = gy[q] * cyk[1];
      }
      for (int k = 2; k < 6; ++k) {
        float *s = o->sd[k] + (size_t)i * nv;
        for (int q = 0; q < nv; ++q) s[q] = gx[q] * cxk[k] + gy[q] * cyk[k];
      }
    }

    /* step 6 */
    compute_hessian(o);

    float normdp_init = 1e-10f;
    float normdp = normdp_init;
    for (int it = 0; (it < op->maxiter) & ((normdp / normdp_init) > op->normdp_ratio); ++it) {
      for (int k = 0; k < 6; ++k) memset(o->sdp[k], 0, sizeof(float) * nm); /* :352-357 */
      orc_pose_project_pt(o->pose, o->pt3d, o->pt2d_new, n, sl);         /* step 7 */
      for (int i = 0; i < n; ++i) {                                       /* step 8 */
        float mid[2] = {o->pt2d_new[i], o->pt2d_new[i + M]};
        if ((mid[0] < 0) | (mid[1] < 0) | (mid[0] > swo) | (mid[1] > sho)) {
          o->ind_new[i] = 0;
        } else {
          o->ind_new[i] = 1;
          float *pn = o->pat_new + (size_t)i * nv;
          orc_getpatch(o->img_new[sl], mid, pn, op, width);
          const float *pr = o->pat_ref + (size_t)i * nv;
          for (int q = 0; q < nv; ++q) o->pdiff[q] = pr[q] - pn[q];
          for (int k = 0; k < 6; ++k) {
            const float *s = o->sd[k] + (size_t)i * nv;
            float *d = o->sdp[k] + (size_t)i * nv;
            for (int q = 0; q < nv; ++q) d[q] = s[q] * o->pdiff[q];
          }
        }
      }
      for (int k = 0; k < 6; ++k) o->sumsd[k] = sum_f32(o->sdp[k], nm);   /* step 9a */
      orc_solve6_fullpivlu(o->Hes, o->sumsd, o->delta_p);                   /* step 9b */
      orc_pose_addpose_se3(o->pose, o->delta_p);                            /* step 10 */
      normdp = 0.0f;
      for (int k = 0; k < 6; ++k) normdp += fabsf(o->delta_p[k]);
      if (it == 0) normdp_init = normdp;
      if (op->verbosity == 2) printf("Sc%02i,It%02i: %g\n", sl, it, normdp);
      trace_push(o, sl, it);
    }
  }
  orc_pose_getpose_se3(o->pose, p_out);
}

const float *orc_odometer_get2dpoints(const orc_odometer *o) { return o->pt2d[o->op->lv_l]; }
int orc_odometer_trace_count(const orc_odometer *o) { return o->ntrace; }
const orc_trace_rec *orc_odometer_trace(const orc_odometer *o) { return o->trace; }
const float *orc_odometer_buffer(const orc_odometer *o, int which) {
  switch (which) {
    case 0: return o->pat_ref;
    case 1: return o->pat_ref_dx;
    case 2: return o->pat_ref_dy;
    case 3: return o->pat_new;
    case 4: return o->pt3d;
    case 5: return o->pt3d_ref;
    case 6: return o->pt2d_new;
    default: return NULL;
  }
}
const float *orc_odometer_pt2d(const orc_odometer *o, int level) { return o->pt2d[level]; }
const unsigned char *orc_odometer_ind(const orc_odometer *o, int which) { return which ? o->ind_new : o->ind_ref; }
void orc_odometer_norm(const orc_odometer *o, double *meanshift3, double *varval) {
  memcpy(meanshift3, o->meanshift, sizeof(double) * 3);
  *varval = o->varval;
}
