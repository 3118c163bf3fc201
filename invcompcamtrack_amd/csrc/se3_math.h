// se3_math.h -- SE(3) exp/log and the 6x6 (NxN) full-pivot LU solve, usable from host and gfx950 device code.
//
// Product code (part of libictr_hip.so). Mirrors, for parity with the reference:
//   se3_exp  : util_SE3_coeff_to_group   utilities.h:84-145  (Eade closed form, Taylor branch for sigma<=1e-4)
//   se3_log  : util_SE3_group_to_coeff   utilities.h:149-241
//   lu_solve : Hes.fullPivLu().solve(b)  odometer.cpp:509-515 (Eigen FullPivLU: column-major pivot scan,
//              rank threshold eps*N*|maxpivot|, free variables of a rank-deficient system set to 0)
// The translation unit is compiled with -ffp-contract=off: the reference's build (-msse4 -mavx, no FMA)
// rounds every product, and so do we.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ICTR_HD __host__ __device__ inline
#else
#define ICTR_HD inline
#endif

namespace ictr {

template <typename T> struct mathfn;
template <> struct mathfn<float> {
  static ICTR_HD float sqrt_(float x) { return sqrtf(x); }
  static ICTR_HD float sin_(float x) { return sinf(x); }
  static ICTR_HD float cos_(float x) { return cosf(x); }
  static ICTR_HD void sincos_(float x, float *sn, float *cs) {
#if defined(__HIP_DEVICE_COMPILE__)
    sincosf(x, sn, cs);  // one argument reduction for both (same results as sinf / cosf)
#else
    *sn = sinf(x);
    *cs = cosf(x);
#endif
  }
  static ICTR_HD float acos_(float x) { return acosf(x); }
  static ICTR_HD float tan_(float x) { return tanf(x); }
};
template <> struct mathfn<double> {
  static ICTR_HD double sqrt_(double x) { return sqrt(x); }
  static ICTR_HD double sin_(double x) { return sin(x); }
  static ICTR_HD double cos_(double x) { return cos(x); }
  static ICTR_HD void sincos_(double x, double *sn, double *cs) {
    *sn = sin(x);
    *cs = cos(x);
  }
  static ICTR_HD double acos_(double x) { return acos(x); }
  static ICTR_HD double tan_(double x) { return tan(x); }
};

// p = (t0,t1,t2, w0,w1,w2)  ->  G = [R | V t], 3x4 row-major
template <typename T> ICTR_HD void se3_exp(T *G, const T *p) {
  using M = mathfn<T>;
  const T w0 = p[3], w1 = p[4], w2 = p[5];
  const T q0 = w0 * w0, q1 = w1 * w1, q2 = w2 * w2;
  const T sig = M::sqrt_(q0 + q1 + q2);
  const T s2 = (sig * sig);
  const T s3 = (sig * sig * sig);
  T sa, sb, sc;  // sin(s)/s, (1-cos s)/s^2, (s-sin s)/s^3
  if (sig > 1e-4) {
    T sn, cs;
    M::sincos_(sig, &sn, &cs);
    sa = sn / sig;
    sb = (1 - cs) / s2;
    sc = (sig - sn) / s3;
  } else {
    sa = 1 - s2 / 6 * (1 - s2 / 20 * (1 - s2 / 42));
    sb = (T)(.5 * (1 - s2 / 12 * (1 - s2 / 30 * (1 - s2 / 56))));
    sc = (1 - s2 / 20 * (1 - s2 / 42 * (1 - s2 / 72))) / 6;
  }
  // R = I + sa [w]x + sb [w]x^2
  {
    const T a = q1 * sb, b = q2 * sb, c = q0 * sb;
    const T w01 = w0 * w1 * sb, w2a = w2 * sa, w02 = w0 * w2 * sb, w1a = w1 * sa, w0a = w0 * sa,
            w12 = w1 * w2 * sb;
    G[0] = 1 - a - b;
    G[1] = w01 - w2a;
    G[2] = w1a + w02;
    G[4] = w2a + w01;
    G[5] = 1 - c - b;
    G[6] = w12 - w0a;
    G[8] = w02 - w1a;
    G[9] = w0a + w12;
    G[10] = 1 - c - a;
  }
  // V = I + sb [w]x + sc [w]x^2 applied to t
  {
    const T a = w2 * sb, b = w0 * w1 * sc, c = w1 * sb, d = w0 * w2 * sc, e = w0 * sb, f = w1 * w2 * sc;
    G[3] = (1 - (q1 + q2) * sc) * p[0] + (b - a) * p[1] + (c + d) * p[2];
    G[7] = (a + b) * p[0] + (1 - (q0 + q2) * sc) * p[1] + (f - e) * p[2];
    G[11] = (d - c) * p[0] + (e + f) * p[1] + (1 - (q0 + q1) * sc) * p[2];
  }
}

template <typename T> ICTR_HD void se3_log(T *p, const T *G) {
  using M = mathfn<T>;
  const T tr = G[0] + G[5] + G[10];
  const T theta = M::acos_((T)(0.5f * (tr - 1)));
  // W = theta/(2 sin theta) (R - R^T), stored as its three independent entries
  T o1 = 0, o2 = 0, o5 = 0;     // W(0,1), W(0,2), W(1,2)
  T S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // W^2
  if (theta < 1e-10) {
    p[3] = 0.0f;
    p[4] = 0.0f;
    p[5] = 0.0f;
  } else {
    const T coef = theta / (2.0f * M::sin_(theta));
    o1 = coef * (G[1] - G[4]);
    o2 = coef * (G[2] - G[8]);
    o5 = coef * (G[6] - G[9]);
    p[3] = -o5;
    p[4] = o2;
    p[5] = -o1;
    const T a = o1 * o1, b = o2 * o2, c = o5 * o5;
    S[0] = -a - b;
    S[1] = -o2 * o5;
    S[3] = S[1];
    S[2] = o1 * o5;
    S[6] = S[2];
    S[4] = -a - c;
    S[5] = -o1 * o2;
    S[7] = S[5];
    S[8] = -b - c;
  }
  T h;
  if (theta < 1e-4)
    h = (T)(1.0f / 12.0f);
  else
    h = (1.0f - theta / (2.0f * M::tan_(theta / 2.0f))) / (theta * theta);
  const T W[9] = {0, o1, o2, -o1, 0, o5, -o2, -o5, 0};
  T Vi[9];
  for (int i = 0; i < 9; ++i) {
    const bool diag = (i == 0) | (i == 4) | (i == 8);
    Vi[i] = diag ? (1.0f + h * S[i]) : (-0.5f * W[i] + h * S[i]);
  }
  p[0] = Vi[0] * G[3] + Vi[1] * G[7] + Vi[2] * G[11];
  p[1] = Vi[3] * G[3] + Vi[4] * G[7] + Vi[5] * G[11];
  p[2] = Vi[6] * G[3] + Vi[7] * G[7] + Vi[8] * G[11];
}

// Full-pivot LU solve of a symmetric NxN float system, Eigen FullPivLU semantics.
template <int N> ICTR_HD void lu_solve(const float *Hin, const float *bin, float *x) {
  float A[N * N];
  for (int i = 0; i < N * N; ++i) A[i] = Hin[i];
  int rowsw[N], colsw[N];
  int nonzero = N;
  float maxpiv = 0.0f;
  for (int k = 0; k < N; ++k) {
    int br = k, bc = k;
    float best = fabsf(A[k * N + k]);
    for (int c = k; c < N; ++c)       // column-major scan, strict '>' keeps the first maximum
      for (int r = k; r < N; ++r) {
        const float v = fabsf(A[r * N + c]);
        if (v > best) {
          best = v;
          br = r;
          bc = c;
        }
      }
    if (best == 0.0f) {
      nonzero = k;
      for (int i = k; i < N; ++i) rowsw[i] = colsw[i] = i;
      break;
    }
    if (best > maxpiv) maxpiv = best;
    rowsw[k] = br;
    colsw[k] = bc;
    if (br != k)
      for (int c = 0; c < N; ++c) {
        const float t = A[k * N + c];
        A[k * N + c] = A[br * N + c];
        A[br * N + c] = t;
      }
    if (bc != k)
      for (int r = 0; r < N; ++r) {
        const float t = A[r * N + k];
        A[r * N + k] = A[r * N + bc];
        A[r * N + bc] = t;
      }
    if (k < N - 1) {
      const float piv = A[k * N + k];
      for (int r = k + 1; r < N; ++r) A[r * N + k] /= piv;
      for (int c = k + 1; c < N; ++c)
        for (int r = k + 1; r < N; ++r) A[r * N + c] -= A[r * N + k] * A[k * N + c];
    }
  }
  for (int i = 0; i < N; ++i) x[i] = 0.0f;
  if (nonzero == 0) return;
  const float thr = maxpiv * (1.1920929e-07f * N);
  int rank = 0;
  for (int i = 0; i < nonzero; ++i) rank += (fabsf(A[i * N + i]) > thr) ? 1 : 0;
  float c[N];
  for (int i = 0; i < N; ++i) c[i] = bin[i];
  for (int k = 0; k < N; ++k) {
    const int r = rowsw[k];
    if (r != k) {
      const float t = c[k];
      c[k] = c[r];
      c[r] = t;
    }
  }
  for (int i = 0; i < N; ++i)
    for (int r = i + 1; r < N; ++r) c[r] -= c[i] * A[r * N + i];
  for (int i = N - 1; i >= 0; --i) {
    if (i < rank) {
      c[i] /= A[i * N + i];
      for (int r = 0; r < i; ++r) c[r] -= c[i] * A[r * N + i];
    }
  }
  for (int i = 0; i < N; ++i) x[i] = (i < rank) ? c[i] : 0.0f;
  for (int k = N - 1; k >= 0; --k) {
    const int q = colsw[k];
    if (q != k) {
      const float t = x[k];
      x[k] = x[q];
      x[q] = t;
    }
  }
}

// The same algorithm in two halves, every runtime-indexed array in caller-provided storage (LDS on the device).
// H is constant during the Gauss-Newton iterations of one pyramid level, so the device factors it once per level
// and only substitutes per iteration; the arithmetic (and therefore every bit of the result) is that of lu_solve.
//   lu_factor_ws: A (N*N) is overwritten with the LU factors; piv gets N row + N column transpositions;
//                 info[0] = number of non-zero pivots, info[1] = rank (Eigen threshold eps*N*|maxpivot|).
//   lu_apply_ws : x = solve with the stored factors; c is N floats of workspace.
template <int N> ICTR_HD void lu_factor_ws(float *A, int *piv, int *info) {
  int *rowsw = piv, *colsw = piv + N;
  int nonzero = N;
  float maxpiv = 0.0f;
  for (int k = 0; k < N; ++k) {
    int br = k, bc = k;
    float best = fabsf(A[k * N + k]);
    for (int cc = k; cc < N; ++cc)
      for (int r = k; r < N; ++r) {
        const float v = fabsf(A[r * N + cc]);
        if (v > best) {
          best = v;
          br = r;
          bc = cc;
        }
      }
    if (best == 0.0f) {
      nonzero = k;
      for (int i = k; i < N; ++i) rowsw[i] = colsw[i] = i;
      break;
    }
    if (best > maxpiv) maxpiv = best;
    rowsw[k] = br;
    colsw[k] = bc;
    if (br != k)
      for (int cc = 0; cc < N; ++cc) {
        const float t = A[k * N + cc];
        A[k * N + cc] = A[br * N + cc];
        A[br * N + cc] = t;
      }
    if (bc != k)
      for (int r = 0; r < N; ++r) {
        const float t = A[r * N + k];
        A[r * N + k] = A[r * N + bc];
        A[r * N + bc] = t;
      }
    if (k < N - 1) {
      const float pv = A[k * N + k];
      for (int r = k + 1; r < N; ++r) A[r * N + k] /= pv;
      for (int cc = k + 1; cc < N; ++cc)
        for (int r = k + 1; r < N; ++r) A[r * N + cc] -= A[r * N + k] * A[k * N + cc];
    }
  }
  int rank = 0;
  if (nonzero > 0) {
    const float thr = maxpiv * (1.1920929e-07f * N);
    for (int i = 0; i < nonzero; ++i) rank += (fabsf(A[i * N + i]) > thr) ? 1 : 0;
  }
  info[0] = nonzero;
  info[1] = rank;
}

template <int N>
ICTR_HD void lu_apply_ws(const float *A, const int *piv, const int *info, const float *bin, float *x, float *c) {
  const int *rowsw = piv, *colsw = piv + N;
  const int nonzero = info[0], rank = info[1];
  for (int i = 0; i < N; ++i) x[i] = 0.0f;
  if (nonzero == 0) return;
  for (int i = 0; i < N; ++i) c[i] = bin[i];
  for (int k = 0; k < N; ++k) {
    const int r = rowsw[k];
    if (r != k) {
      const float t = c[k];
      c[k] = c[r];
      c[r] = t;
    }
  }
  for (int i = 0; i < N; ++i)
    for (int r = i + 1; r < N; ++r) c[r] -= c[i] * A[r * N + i];
  for (int i = N - 1; i >= 0; --i) {
    if (i < rank) {
      c[i] /= A[i * N + i];
      for (int r = 0; r < i; ++r) c[r] -= c[i] * A[r * N + i];
    }
  }
  for (int i = 0; i < N; ++i) c[i] = (i < rank) ? c[i] : 0.0f;
  for (int k = N - 1; k >= 0; --k) {
    const int q = colsw[k];
    if (q != k) {
      const float t = c[k];
      c[k] = c[q];
      c[q] = t;
    }
  }
  for (int i = 0; i < N; ++i) x[i] = c[i];
}

}  // namespace ictr
