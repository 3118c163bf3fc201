// ictr_icgn.hip -- full-frame inverse-compositional Gauss-Newton alignment with a parametric warp
// (translation 2, SE(2) 3, affine 6, homography 8 parameters), batched, pyramidal.
//
// Status: EXTENSION. The reference contains none of these warp models (its only warp is the pinhole
// reprojection of 3-D points under SE(3)); BASELINE.json's configs 1, 2, 3 and 5 name them, so they are built on the
// same Gauss-Newton skeleton as the reference-faithful tracker (precomputed template gradients and Hessian per
// level, per-iteration residual and J^T r, tiny solve, coarse-to-fine) -- with the classic Baker-Matthews
// inverse-compositional update  M <- M * W(dp)^-1. Oracle: oracle/np_icgn.py (NumPy f64). Parity is therefore
// "unpinned by the reference".
//
// Conventions: the warp is a 3x3 matrix M acting on NORMALISED template coordinates n = ((x - w/2)/f, (y - h/2)/f),
// f = max(w,h)/2 at level 0; a level-l pixel is the mean of a 2^l x 2^l block, x_0 = 2^l x_l + (2^l - 1)/2, so with
// c_l = (c_0 + 1/2)/2^l - 1/2 and f_l = f_0/2^l the normalised coordinates -- and therefore M -- are the same at
// every pyramid level. Template T = frame A, gradients = the pyramid's
// central differences (I(x+1) - I(x-1), the reference's convention), current image I = frame B sampled bilinearly
// at K_l * M * n. With G = 2 dI/dx the true steepest-descent image is (f_l/2) * sd, so dp = (2/f_l) H^-1 b.
//
// MI355X mapping: memory-bound streaming, 16 B per template pixel per iteration (T, Gx, Gy + one current texel).
// One thread per pixel, x fastest (coalesced T/Gx/Gy rows; the four bilinear taps of a near-identity warp are
// near-coalesced and L1/L2 resident), N per-lane accumulators, one shuffle reduction per wave per launch, one
// partial per workgroup; a second launch with one workgroup per problem reduces in fixed order (f64), solves and
// composes the warp on the device (same accumulate/tail split as the tracker, ictr_kernels.hip).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "ictr_dev.h"
#include "se3_math.h"

// Nothing in this file has to round like the reference's CPU build (there is no reference for it): let the compiler
// contract a*b+c into FMAs here although the library is built with -ffp-contract=off. The kernels are within 2x of
// being VALU-bound (about 80 lane-instructions per pixel for the homography without FMA), so this matters.
#pragma clang fp contract(fast)

extern "C" const char *ictr_last_error(void);
int ictr_fail_(int code, const char *fmt, ...);  // ictr_host.hip

namespace ictr {

enum { kTrans = 0, kSE2 = 1, kAffine = 2, kHomog = 3 };
__host__ __device__ constexpr int model_np(int m) { return m == kTrans ? 2 : m == kSE2 ? 3 : m == kAffine ? 6 : 8; }
constexpr int kIcMaxN = 8;
constexpr int kIcNH = 36;      // 8*9/2
constexpr int kIcPartH = 40;   // floats per workgroup partial of H
constexpr int kIcPartB = 8;
constexpr int kIcRed = 44;     // sharded reduction record: 36 + 8

struct IcState {
  float M[9];     // current warp, normalised coordinates
  float H[64];
  float LU[64];
  int piv[16];
  int luinfo[2];
  float b[8];
  float dp[8];
  int it, active, total_iters, pad_[2];
  float Hinv[64];  // H^-1 (row-major), computed once per level when H has full rank (hinv_ok); see k_icgn_hess_tail
  int hinv_ok, pad2_[3];
};

struct IcLevel {
  int w, h, sw, pad;
  float cx, cy, f;  // pixel = n * f + c
};

struct IcDev {
  int B, model, n, nh, maxiter, sharded;
  int dbg;  // experiments only (ICTR_ICGN_DBG): bit 0 skip the bilinear taps, bit 1 non-temporal tap loads
  int x0, y0, x1, y1;  // template region at level 0 (inclusive-exclusive), scaled per level
  int row_lo, row_hi;  // this rank's rows of the region at level 0 (sharding)
  float eps;
  const PlaneSet *planes;  // [B][nlev]
  IcState *st;
  float *partH, *partb, *red;
  int nlev;
};

// steepest-descent row of one pixel: sd_k = gx * Jx_k + gy * Jy_k, J = dW/dp at the identity
template <int MODEL> __device__ __forceinline__ void ic_sd(float gx, float gy, float x, float y, float *sd) {
  if constexpr (MODEL == kTrans) {
    sd[0] = gx;
    sd[1] = gy;
  } else if constexpr (MODEL == kSE2) {
    sd[0] = gy * x - gx * y;
    sd[1] = gx;
    sd[2] = gy;
  } else if constexpr (MODEL == kAffine) {
    sd[0] = gx * x;
    sd[1] = gy * x;
    sd[2] = gx * y;
    sd[3] = gy * y;
    sd[4] = gx;
    sd[5] = gy;
  } else {
    const float q = -(gx * x + gy * y);
    sd[0] = gx * x;
    sd[1] = gy * x;
    sd[2] = q * x;
    sd[3] = gx * y;
    sd[4] = gy * y;
    sd[5] = q * y;
    sd[6] = gx;
    sd[7] = gy;
  }
}

// plane pointers come out of a table in memory, so the compiler cannot prove their address space and would emit
// flat_load; say "global" explicitly
typedef const float __attribute__((address_space(1))) *ic_gf32;
typedef float ic_f4 __attribute__((ext_vector_type(4)));
typedef const ic_f4 __attribute__((address_space(1))) *ic_gf32x4;

__device__ __forceinline__ float ic_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

struct IcRegion {
  int x0, y0, w, h;  // region at this level: x in [x0, x0+w), y in [y0, y0+h)
};
__device__ __forceinline__ IcRegion ic_region(const IcDev &e, int level) {
  IcRegion r;
  const int s = 1 << level;
  r.x0 = (e.x0 + s - 1) / s;
  const int xe = e.x1 / s;
  const int ylo = max(e.y0, e.row_lo), yhi = min(e.y1, e.row_hi);
  r.y0 = (ylo + s - 1) / s;
  const int ye = (yhi + s - 1) / s;  // rows are split at level 0; a level-l row belongs to the rank owning s*row
  r.w = max(xe - r.x0, 0);
  r.h = max(ye - r.y0, 0);
  return r;
}

// H = sum sd^T sd over the template region (once per level)
template <int MODEL>
__global__ __launch_bounds__(kBlock) void k_icgn_hess(IcDev e, IcLevel L, int level) {
  constexpr int N = model_np(MODEL), NH = N * (N + 1) / 2;
  __shared__ float sW[kWaves][kIcPartH];
  const int b = blockIdx.y;
  const PlaneSet pl = e.planes[b * e.nlev + level];
  const IcRegion R = ic_region(e, level);
  const long npx = (long)R.w * R.h;
  float acc[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) acc[j] = 0.0f;
  const float inv_f = 1.0f / L.f;
  for (long t = (long)blockIdx.x * kBlock + threadIdx.x; t < npx; t += (long)gridDim.x * kBlock) {
    const int y = R.y0 + (int)(t / R.w), x = R.x0 + (int)(t % R.w);
    const size_t o = (size_t)(y + L.pad) * L.sw + x + L.pad;
    const float gx = ((ic_gf32)pl.dx)[o], gy = ((ic_gf32)pl.dy)[o];
    float sd[N];
    ic_sd<MODEL>(gx, gy, ((float)x - L.cx) * inv_f, ((float)y - L.cy) * inv_f, sd);
    int jk = 0;
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
      for (int c = a; c < N; ++c) acc[jk++] += sd[a] * sd[c];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    const float v = ic_wave_sum(acc[j]);
    if (lane == 0) sW[wave][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < NH)
    e.partH[((size_t)b * gridDim.x + blockIdx.x) * kIcPartH + threadIdx.x] =
        (sW[0][threadIdx.x] + sW[1][threadIdx.x]) + (sW[2][threadIdx.x] + sW[3][threadIdx.x]);
}

// b = sum sd^T (I(W(x)) - T(x)) for the current warp.
// V = 4: a lane owns four consecutive pixels of a row, read as one aligned 16-byte load per plane (a wave streams
// 1 KB per plane per step); rows are cut into quads aligned to the padded plane, pixels of a quad outside the region
// are masked. V = 1: scalar form for planes whose stride is not a multiple of four floats.
template <int MODEL>
__device__ __forceinline__ void ic_pixel(ic_gf32 cur, const IcLevel &L, const float *M, float inv_f,
                                         float xmax, float ymax, int x, float ny, float tv, float gx, float gy,
                                         float *acc, int dbg) {
  constexpr int N = model_np(MODEL);
  const float nx = ((float)x - L.cx) * inv_f;
  const float u = M[0] * nx + M[1] * ny + M[2];
  const float v = M[3] * nx + M[4] * ny + M[5];
  float iw = 1.0f;
  if constexpr (MODEL == kHomog) iw = 1.0f / (M[6] * nx + M[7] * ny + M[8]);
  const float px = u * iw * L.f + L.cx, py = v * iw * L.f + L.cy;
  if ((px >= 0.0f) & (py >= 0.0f) & (px <= xmax) & (py <= ymax)) {  // NaN-safe; outside pixels contribute nothing
    const float fxf = floorf(px), fyf = floorf(py);
    const float ax = px - fxf, ay = py - fyf;
    ic_gf32 q = cur + ((size_t)((int)fyf + L.pad) * L.sw + (int)fxf + L.pad);
    float i00, i01, i10, i11;
    if (dbg & 1) {
      i00 = i01 = i10 = i11 = tv + ax;
    } else if (dbg & 2) {
      i00 = __builtin_nontemporal_load(q), i01 = __builtin_nontemporal_load(q + 1);
      i10 = __builtin_nontemporal_load(q + L.sw), i11 = __builtin_nontemporal_load(q + L.sw + 1);
    } else {
      i00 = q[0], i01 = q[1], i10 = q[L.sw], i11 = q[L.sw + 1];
    }
    const float iv = (i00 * (1.0f - ax) + i01 * ax) * (1.0f - ay) + (i10 * (1.0f - ax) + i11 * ax) * ay;
    const float r = iv - tv;
    float sd[N];
    ic_sd<MODEL>(gx, gy, nx, ny, sd);
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] += sd[k] * r;
  }
}

template <int MODEL, int V>
__global__ __launch_bounds__(kBlock) void k_icgn_iter(IcDev e, IcLevel L, int level) {
  constexpr int N = model_np(MODEL);
  __shared__ float sW[kWaves][kIcPartB];
  const int b = blockIdx.y;
  const IcState &st = e.st[b];
  if (!st.active) return;
  const PlaneSet pl = e.planes[b * e.nlev + level];
  const ic_gf32 T = (ic_gf32)pl.ref, Gx = (ic_gf32)pl.dx, Gy = (ic_gf32)pl.dy, cur = (ic_gf32)pl.cur;
  const IcRegion R = ic_region(e, level);
  float M[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) M[k] = st.M[k];  // affine models keep the last row at (0,0,1): M is normalised by M[8]
  float acc[N];
#pragma unroll
  for (int k = 0; k < N; ++k) acc[k] = 0.0f;
  const float inv_f = 1.0f / L.f;
  const float xmax = (float)(L.w - 1), ymax = (float)(L.h - 1);
  if constexpr (V == 4) {
    // a workgroup owns a tile of 256 pixels x 4 rows (wave w = row w, lane = quad): the four waves' bilinear taps
    // overlap in 3 of their 5 rows, which the CU's L1 serves
    // first quad on a 16-byte boundary of the padded row; when the row pitch is a multiple of 128 bytes (e.g. width
    // % 32 == 0 and padding 16) on a cache-line boundary, so that no line is shared between two tiles (measured with
    // padding 4: 19 % HBM over-fetch, every 1-KB tile row touching 9 lines instead of 8)
    const int xs = R.x0 - ((R.x0 + L.pad) & ((L.sw & 31) == 0 ? 31 : 3));
    const int nq = (R.x0 + R.w - xs + 3) >> 2;
    const int ntx = (nq + 63) >> 6, nty = (R.h + kWaves - 1) / kWaves;
    const int xe = R.x0 + R.w;
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
    typedef ic_f4 f4;
    for (int t = blockIdx.x; t < ntx * nty; t += gridDim.x) {
      const int ty = t / ntx, tx = t - ty * ntx;
      const int row = ty * kWaves + wave_, q = tx * 64 + lane_;
      if ((row >= R.h) | (q >= nq)) continue;
      const int y = R.y0 + row, x = xs + 4 * q;
      const size_t o = (size_t)(y + L.pad) * L.sw + (x + L.pad);
      const f4 tv = __builtin_nontemporal_load((ic_gf32x4)(T + o));
      const f4 gx = __builtin_nontemporal_load((ic_gf32x4)(Gx + o));
      const f4 gy = __builtin_nontemporal_load((ic_gf32x4)(Gy + o));
      const float ny = ((float)y - L.cy) * inv_f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((x + j >= R.x0) & (x + j < xe))
          ic_pixel<MODEL>(cur, L, M, inv_f, xmax, ymax, x + j, ny, tv[j], gx[j], gy[j], acc, e.dbg);
    }
  } else {
    const long npx = (long)R.w * R.h;
    for (long t = (long)blockIdx.x * kBlock + threadIdx.x; t < npx; t += (long)gridDim.x * kBlock) {
      const int y = R.y0 + (int)(t / R.w), x = R.x0 + (int)(t % R.w);
      const size_t o = (size_t)(y + L.pad) * L.sw + x + L.pad;
      const float tv = __builtin_nontemporal_load(T + o), gx = __builtin_nontemporal_load(Gx + o),
                  gy = __builtin_nontemporal_load(Gy + o);
      ic_pixel<MODEL>(cur, L, M, inv_f, xmax, ymax, x, ((float)y - L.cy) * inv_f, tv, gx, gy, acc, e.dbg);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float v = ic_wave_sum(acc[k]);
    if (lane == 0) sW[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < N)
    e.partb[((size_t)b * gridDim.x + blockIdx.x) * kIcPartB + threadIdx.x] =
        (sW[0][threadIdx.x] + sW[1][threadIdx.x]) + (sW[2][threadIdx.x] + sW[3][threadIdx.x]);
}

// LDS-staged form of k_icgn_iter: the gathered bilinear taps were the inefficient stream of the direct form (25 % of
// the bytes, 33 % of the time; profiles/r01_icgn.md), so the workgroup first copies the bounding box of its tile's
// warped footprint into LDS with the same aligned 16-byte row loads the template planes get, then every pixel takes
// its four taps from LDS. Tile = 256 pixels x 4 rows (wave = row, lane + 64 j = pixel: adjacent lanes read adjacent
// LDS words, no bank conflicts). A projective warp maps the convex tile into the convex hull of its warped corners, so
// the corners bound the footprint; one pixel of slack absorbs rounding. Tiles whose footprint does not fit the LDS
// window (strong rotation / scale) fall back to direct gathers, so the result never depends on the window size.
constexpr int kIcLW = 288, kIcLH = 12;  // LDS window: 288 x 12 floats = 13.5 KB per workgroup

template <int MODEL>
__global__ __launch_bounds__(kBlock) void k_icgn_iter_lds(IcDev e, IcLevel L, int level) {
  constexpr int N = model_np(MODEL);
  __shared__ __attribute__((aligned(16))) float sTile[kIcLH * kIcLW];
  __shared__ float sW[kWaves][kIcPartB];
  const int b = blockIdx.y;
  const IcState &st = e.st[b];
  if (!st.active) return;
  const PlaneSet pl = e.planes[b * e.nlev + level];
  const ic_gf32 T = (ic_gf32)pl.ref, Gx = (ic_gf32)pl.dx, Gy = (ic_gf32)pl.dy, cur = (ic_gf32)pl.cur;
  const IcRegion R = ic_region(e, level);
  float M[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) M[k] = st.M[k];
  float acc[N];
#pragma unroll
  for (int k = 0; k < N; ++k) acc[k] = 0.0f;
  const float inv_f = 1.0f / L.f;
  const float xmax = (float)(L.w - 1), ymax = (float)(L.h - 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntx = (R.w + 255) >> 8, nty = (R.h + kWaves - 1) / kWaves;
  const int xe = R.x0 + R.w;

  auto warp_pt = [&](float x, float y, float &px, float &py) -> bool {
    const float nx = (x - L.cx) * inv_f, ny = (y - L.cy) * inv_f;
    const float u = M[0] * nx + M[1] * ny + M[2], v = M[3] * nx + M[4] * ny + M[5];
    float wq = 1.0f;
    if constexpr (MODEL == kHomog) wq = M[6] * nx + M[7] * ny + M[8];
    const float iw = 1.0f / wq;
    px = u * iw * L.f + L.cx;
    py = v * iw * L.f + L.cy;
    return wq > 1e-6f;
  };

  for (int t = blockIdx.x; t < ntx * nty; t += gridDim.x) {
    const int ty = t / ntx, tx = t - ty * ntx;
    const int tx0 = R.x0 + (tx << 8), ty0 = R.y0 + ty * kWaves;
    const int txl = min(tx0 + 255, xe - 1), tyl = min(ty0 + kWaves - 1, R.y0 + R.h - 1);
    // ---- footprint of the tile in the current frame (wave-uniform arithmetic)
    float cxs[4], cys[4];
    bool okc = warp_pt((float)tx0, (float)ty0, cxs[0], cys[0]);
    okc &= warp_pt((float)txl, (float)ty0, cxs[1], cys[1]);
    okc &= warp_pt((float)tx0, (float)tyl, cxs[2], cys[2]);
    okc &= warp_pt((float)txl, (float)tyl, cxs[3], cys[3]);
    const float fxmin = fminf(fminf(cxs[0], cxs[1]), fminf(cxs[2], cxs[3]));
    const float fxmax = fmaxf(fmaxf(cxs[0], cxs[1]), fmaxf(cxs[2], cxs[3]));
    const float fymin = fminf(fminf(cys[0], cys[1]), fminf(cys[2], cys[3]));
    const float fymax = fmaxf(fmaxf(cys[0], cys[1]), fmaxf(cys[2], cys[3]));
    okc &= (fxmax - fxmin < 4096.0f) & (fymax - fymin < 4096.0f) & (fabsf(fxmin) < 1e6f) & (fabsf(fymin) < 1e6f);
    int bx0 = 0, by0 = 0, ncols = 0, nrows = 0;
    if (okc) {
      const int cx0 = max((int)floorf(fxmin) - 1, -L.pad), cx1 = min((int)floorf(fxmax) + 2, L.w - 1 + L.pad);
      const int cy0 = max((int)floorf(fymin) - 1, -L.pad), cy1 = min((int)floorf(fymax) + 2, L.h - 1 + L.pad);
      bx0 = cx0 - ((cx0 + L.pad) & 3);  // 16-byte aligned start inside the padded row
      by0 = cy0;
      ncols = cx1 - bx0 + 1;
      nrows = cy1 - cy0 + 1;
    }
    const bool staged = okc & (ncols > 0) & (nrows > 0) & (ncols <= kIcLW) & (nrows <= kIcLH);
    // ---- this lane's template pixels: row ty0 + wave, columns tx0 + lane + 64 j (issued before the LDS fill)
    const int y = ty0 + wave;
    const bool rowok = y <= tyl;
    float tv[4], gxv[4], gyv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = tx0 + lane + 64 * j;
      tv[j] = gxv[j] = gyv[j] = 0.0f;
      if (rowok & (x < xe)) {
        const size_t o = (size_t)(y + L.pad) * L.sw + (x + L.pad);
        tv[j] = __builtin_nontemporal_load(T + o);
        gxv[j] = __builtin_nontemporal_load(Gx + o);
        gyv[j] = __builtin_nontemporal_load(Gy + o);
      }
    }
    if (staged) {
      const int n4 = (ncols + 3) >> 2;
      for (int r = wave; r < nrows; r += kWaves) {
        const ic_gf32 src = cur + ((size_t)(by0 + r + L.pad) * L.sw + (bx0 + L.pad));
        for (int c4 = lane; c4 < n4; c4 += 64)
          *reinterpret_cast<ic_f4 *>(&sTile[r * kIcLW + 4 * c4]) = *(ic_gf32x4)(src + 4 * c4);
      }
    }
    __syncthreads();
    const float ny = ((float)y - L.cy) * inv_f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = tx0 + lane + 64 * j;
      if (!(rowok & (x < xe))) continue;
      if (!staged) {
        ic_pixel<MODEL>(cur, L, M, inv_f, xmax, ymax, x, ny, tv[j], gxv[j], gyv[j], acc, 0);
        continue;
      }
      const float nx = ((float)x - L.cx) * inv_f;
      const float u = M[0] * nx + M[1] * ny + M[2];
      const float v = M[3] * nx + M[4] * ny + M[5];
      float iw = 1.0f;
      if constexpr (MODEL == kHomog) iw = 1.0f / (M[6] * nx + M[7] * ny + M[8]);
      const float px = u * iw * L.f + L.cx, py = v * iw * L.f + L.cy;
      if ((px >= 0.0f) & (py >= 0.0f) & (px <= xmax) & (py <= ymax)) {
        const float fxf = floorf(px), fyf = floorf(py);
        const float ax = px - fxf, ay = py - fyf;
        int q = ((int)fyf - by0) * kIcLW + ((int)fxf - bx0);
        q = min(max(q, 0), (kIcLH - 1) * kIcLW - 2);  // never outside the window, whatever the rounding did
        const float i00 = sTile[q], i01 = sTile[q + 1], i10 = sTile[q + kIcLW], i11 = sTile[q + kIcLW + 1];
        const float iv = (i00 * (1.0f - ax) + i01 * ax) * (1.0f - ay) + (i10 * (1.0f - ax) + i11 * ax) * ay;
        const float r = iv - tv[j];
        float sd[N];
        ic_sd<MODEL>(gxv[j], gyv[j], nx, ny, sd);
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] += sd[k] * r;
      }
    }
    __syncthreads();  // the window is refilled by the next tile
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float v = ic_wave_sum(acc[k]);
    if (lane == 0) sW[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < N)
    e.partb[((size_t)b * gridDim.x + blockIdx.x) * kIcPartB + threadIdx.x] =
        (sW[0][threadIdx.x] + sW[1][threadIdx.x]) + (sW[2][threadIdx.x] + sW[3][threadIdx.x]);
}

// ---- 3x3 helpers (double) for the compositional update
__device__ __host__ inline void m3_mul(const double *A, const double *B, double *C) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
__device__ __host__ inline void m3_inv(const double *A, double *I) {
  const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  const double id = 1.0 / det;
  I[0] = c00 * id;
  I[1] = (A[2] * A[7] - A[1] * A[8]) * id;
  I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  I[3] = c01 * id;
  I[4] = (A[0] * A[8] - A[2] * A[6]) * id;
  I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  I[6] = c02 * id;
  I[7] = (A[1] * A[6] - A[0] * A[7]) * id;
  I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}
__device__ __host__ inline void ic_param_matrix(int model, const double *p, double *W) {
  for (int i = 0; i < 9; ++i) W[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (model == kTrans) {
    W[2] = p[0];
    W[5] = p[1];
  } else if (model == kSE2) {
    const double c = cos(p[0]), s = sin(p[0]);
    W[0] = c; W[1] = -s; W[2] = p[1];
    W[3] = s; W[4] = c;  W[5] = p[2];
  } else if (model == kAffine) {
    W[0] = 1 + p[0]; W[3] = p[1]; W[1] = p[2]; W[4] = 1 + p[3]; W[2] = p[4]; W[5] = p[5];
  } else {
    W[0] = 1 + p[0]; W[3] = p[1]; W[6] = p[2]; W[1] = p[3]; W[4] = 1 + p[4]; W[7] = p[5]; W[2] = p[6]; W[5] = p[7];
  }
}

__device__ void ic_level_reset(IcState &st, const IcDev &e) {
  st.it = 0;
  st.active = e.maxiter > 0 ? 1 : 0;
}

// one workgroup per problem: fixed-order f64 reduction of the H partials, LU factorisation once per level
__global__ __launch_bounds__(kBlock) void k_icgn_hess_tail(IcDev e, int nblk, int finish_only) {
  __shared__ double sRed[kBlock / 64][64];
  __shared__ float sA[64];
  __shared__ int sI[20];
  const int b = blockIdx.x;
  const int N = e.n, NH = e.nh;
  IcState &st = e.st[b];
  if (!finish_only) {
    const int j = threadIdx.x & 63, sl = threadIdx.x >> 6;
    double s = 0.0;
    if (j < NH) {  // (eight loads in flight, additions in the same order: the loop is pure load latency otherwise)
#pragma unroll 8
      for (int k = sl; k < nblk; k += kBlock / 64) s += (double)e.partH[((size_t)b * nblk + k) * kIcPartH + j];
    }
    sRed[sl][j] = s;
    __syncthreads();
    if (threadIdx.x < NH) {
      double t = 0.0;
      for (int q = 0; q < kBlock / 64; ++q) t += sRed[q][threadIdx.x];
      sA[threadIdx.x] = (float)t;  // upper triangle, row-major
    }
    __syncthreads();
    if (e.sharded) {
      if (threadIdx.x < NH) e.red[(size_t)b * kIcRed + threadIdx.x] = sA[threadIdx.x];
      return;
    }
  } else {
    if (threadIdx.x < NH) sA[threadIdx.x] = e.red[(size_t)b * kIcRed + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < kIcRed) e.red[(size_t)b * kIcRed + threadIdx.x] = 0.0f;
  }
  __shared__ float sFull[64];
  if (threadIdx.x < N * N) {
    const int r = threadIdx.x / N, c = threadIdx.x % N;
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    const float v = sA[lo * N - lo * (lo - 1) / 2 + (hi - lo)];
    st.H[threadIdx.x] = v;
    sFull[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    switch (N) {
      case 2: lu_factor_ws<2>(sFull, sI, sI + 16); break;
      case 3: lu_factor_ws<3>(sFull, sI, sI + 16); break;
      case 6: lu_factor_ws<6>(sFull, sI, sI + 16); break;
      default: lu_factor_ws<8>(sFull, sI, sI + 16); break;
    }
    for (int k = 0; k < N * N; ++k) st.LU[k] = sFull[k];
    for (int k = 0; k < 2 * N; ++k) st.piv[k] = sI[k];
    st.luinfo[0] = sI[16];
    st.luinfo[1] = sI[17];
    st.hinv_ok = (sI[17] == N) ? 1 : 0;
    ic_level_reset(st, e);
  }
  // Full rank (the normal case): the inverse once per level -- thread j solves for the unit vector e_j with the factors
  // just stored -- so that an iteration's solve is N dot products on N threads instead of one thread's substitution
  // through LDS arrays (the classic inverse-compositional form keeps H^-1 too). Rank-deficient: the iterations keep
  // the full-pivot substitution and its particular solution.
  __syncthreads();
  __shared__ float sUnit[8][24];  // per solving thread: right-hand side [0..7], solution [8..15], workspace [16..23]
  if (threadIdx.x < N && sI[17] == N) {
    float *u = sUnit[threadIdx.x];
    for (int k = 0; k < N; ++k) u[k] = (k == (int)threadIdx.x) ? 1.0f : 0.0f;
    switch (N) {
      case 2: lu_apply_ws<2>(sFull, sI, sI + 16, u, u + 8, u + 16); break;
      case 3: lu_apply_ws<3>(sFull, sI, sI + 16, u, u + 8, u + 16); break;
      case 6: lu_apply_ws<6>(sFull, sI, sI + 16, u, u + 8, u + 16); break;
      default: lu_apply_ws<8>(sFull, sI, sI + 16, u, u + 8, u + 16); break;
    }
    for (int r = 0; r < N; ++r) st.Hinv[r * N + threadIdx.x] = u[8 + r];  // column j of the inverse
  }
}

// one workgroup per problem: reduce b, dp = (2/f) H^-1 b, M <- M * W(dp)^-1, loop condition
__global__ __launch_bounds__(kBlock) void k_icgn_iter_tail(IcDev e, float f_level, int nblk, int finish_only) {
  __shared__ double sRed[kBlock / 8][8];
  __shared__ float sLU[96];
  const int b = blockIdx.x;
  const int N = e.n;
  IcState &st = e.st[b];
  if (!st.active) return;
  if (!finish_only) {
    const int j = threadIdx.x & 7, sl = threadIdx.x >> 3;
    double s = 0.0;
    if (j < N) {
#pragma unroll 8
      for (int k = sl; k < nblk; k += kBlock / 8) s += (double)e.partb[((size_t)b * nblk + k) * kIcPartB + j];
    }
    sRed[sl][j] = s;
    __syncthreads();
    if (threadIdx.x < N) {
      double t = 0.0;
      for (int q = 0; q < kBlock / 8; ++q) t += sRed[q][threadIdx.x];
      if (e.sharded)
        e.red[(size_t)b * kIcRed + kIcNH + threadIdx.x] = (float)t;
      else
        sLU[64 + threadIdx.x] = (float)t;
    }
    if (e.sharded) return;
  } else if (threadIdx.x < N) {
    sLU[64 + threadIdx.x] = e.red[(size_t)b * kIcRed + kIcNH + threadIdx.x];
    e.red[(size_t)b * kIcRed + kIcNH + threadIdx.x] = 0.0f;
  }
  const int use_inv = st.hinv_ok;  // workgroup-uniform
  __shared__ int sI[20];
  float *dpf = sLU + 80;  // keep every runtime-indexed array in LDS (no scratch)
  if (use_inv) {
    __syncthreads();  // b in sLU[64..]
    if (threadIdx.x < N) {  // dp_j = sum_k Hinv[j][k] b[k], one row per thread
      float d = 0.0f;
      for (int k = 0; k < N; ++k) d += st.Hinv[threadIdx.x * N + k] * sLU[64 + k];
      dpf[threadIdx.x] = d;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
  } else {
    if (threadIdx.x < N * N) sLU[threadIdx.x] = st.LU[threadIdx.x];
    if (threadIdx.x < 2 * N) sI[threadIdx.x] = st.piv[threadIdx.x];
    if (threadIdx.x < 2) sI[16 + threadIdx.x] = st.luinfo[threadIdx.x];
    __syncthreads();
    if (threadIdx.x != 0) return;
    switch (N) {
      case 2: lu_apply_ws<2>(sLU, sI, sI + 16, sLU + 64, dpf, sLU + 72); break;
      case 3: lu_apply_ws<3>(sLU, sI, sI + 16, sLU + 64, dpf, sLU + 72); break;
      case 6: lu_apply_ws<6>(sLU, sI, sI + 16, sLU + 64, dpf, sLU + 72); break;
      default: lu_apply_ws<8>(sLU, sI, sI + 16, sLU + 64, dpf, sLU + 72); break;
    }
  }
  const double sc = 2.0 / (double)f_level;
  double nrm = 0.0;
  for (int k = 0; k < N; ++k) {
    const double d = (double)dpf[k] * sc;
    st.b[k] = sLU[64 + k];
    st.dp[k] = (float)d;
    nrm += d * d;
  }
  // the eight scalars of the parametrisation (unused ones are zero), then W(dp) with compile-time indices only
  const double p0 = dpf[0] * sc, p1 = dpf[1] * sc, p2 = N > 2 ? dpf[2] * sc : 0.0, p3 = N > 3 ? dpf[3] * sc : 0.0,
               p4 = N > 4 ? dpf[4] * sc : 0.0, p5 = N > 5 ? dpf[5] * sc : 0.0, p6 = N > 6 ? dpf[6] * sc : 0.0,
               p7 = N > 7 ? dpf[7] * sc : 0.0;
  double W[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (e.model == kTrans) {
    W[2] = p0;
    W[5] = p1;
  } else if (e.model == kSE2) {
    const double c = cos(p0), s = sin(p0);
    W[0] = c;
    W[1] = -s;
    W[3] = s;
    W[4] = c;
    W[2] = p1;
    W[5] = p2;
  } else if (e.model == kAffine) {
    W[0] = 1 + p0;
    W[3] = p1;
    W[1] = p2;
    W[4] = 1 + p3;
    W[2] = p4;
    W[5] = p5;
  } else {
    W[0] = 1 + p0;
    W[3] = p1;
    W[6] = p2;
    W[1] = p3;
    W[4] = 1 + p4;
    W[7] = p5;
    W[2] = p6;
    W[5] = p7;
  }
  double Wi[9], Mo[9], Mn[9];
  m3_inv(W, Wi);
#pragma unroll
  for (int k = 0; k < 9; ++k) Mo[k] = (double)st.M[k];
  m3_mul(Mo, Wi, Mn);
  const double inv8 = 1.0 / Mn[8];
#pragma unroll
  for (int k = 0; k < 9; ++k) st.M[k] = (float)(Mn[k] * inv8);
  st.it += 1;
  st.total_iters += 1;
  st.active = ((st.it < e.maxiter) & (nrm > (double)e.eps * (double)e.eps)) ? 1 : 0;
}

}  // namespace ictr

using namespace ictr;

// ---------------------------------------------------------------- host side
#define HIPCHK_IC(expr)                                                                                     \
  do {                                                                                                      \
    hipError_t _e = (expr);                                                                                 \
    if (_e != hipSuccess) return ictr_fail_(ICTR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));   \
  } while (0)

struct ictr_pyramid_view {  // what ictr_host.hip exposes about a pyramid
  int nlev, pad;
  const int *w, *h, *sw;
  float *const *img, *const *dx, *const *dy;
  int getgrad;
};
extern "C" int ictr_pyramid_view_(const ictr_pyramid *p, ictr_pyramid_view *v);  // ictr_host.hip

struct ictr_icgn {
  int model = 0, w = 0, h = 0, lv_f = 0, lv_l = 0, maxiter = 0, B = 0, nlev = 0, pad = -1;
  float eps = 0;
  int region[4] = {0, 0, 0, 0};
  int rows[2] = {0, 0};
  int sharded = 0;
  int gridx = 1;
  int dbg = 0;
  bool lds = false;          // ICTR_ICGN_LDS=1: LDS-staged footprint instead of direct tap gathers (measured: no gain)
  bool scalar_only = false;  // ICTR_ICGN_SCALAR=1: force the one-pixel-per-lane kernels (A/B measurements)
  hipStream_t stream = nullptr;
  IcState *d_st = nullptr;
  PlaneSet *d_planes = nullptr;
  float *d_partH = nullptr, *d_partb = nullptr, *d_red = nullptr, *d_red_own = nullptr;
  std::vector<IcState> h_st;
  std::vector<PlaneSet> h_planes;
  std::vector<char> frames_set;
  std::vector<int> lw, lh, lsw;
  bool timing = false;
  std::vector<hipEvent_t> ev;  // 2 per (level, iteration)
  std::vector<float> h_M0;     // initial warps (normalised), 9 per problem
};

static void icgn_K(const ictr_icgn *g, double *K, double *Ki) {
  const double f = std::max(g->w, g->h) / 2.0, cx = g->w / 2.0, cy = g->h / 2.0;
  const double k[9] = {f, 0, cx, 0, f, cy, 0, 0, 1};
  memcpy(K, k, sizeof(k));
  m3_inv(K, Ki);
}

static IcDev icgn_dev(const ictr_icgn *g) {
  IcDev e;
  e.B = g->B;
  e.model = g->model;
  e.n = model_np(g->model);
  e.nh = e.n * (e.n + 1) / 2;
  e.maxiter = g->maxiter;
  e.sharded = g->sharded;
  e.dbg = g->dbg;
  e.x0 = g->region[0];
  e.y0 = g->region[1];
  e.x1 = g->region[0] + g->region[2];
  e.y1 = g->region[1] + g->region[3];
  e.row_lo = g->rows[0];
  e.row_hi = g->rows[1];
  e.eps = g->eps;
  e.planes = g->d_planes;
  e.st = g->d_st;
  e.partH = g->d_partH;
  e.partb = g->d_partb;
  e.red = g->d_red;
  e.nlev = g->nlev;
  return e;
}
static IcLevel icgn_level(const ictr_icgn *g, int l) {
  IcLevel L;
  L.w = g->lw[l];
  L.h = g->lh[l];
  L.sw = g->lsw[l];
  L.pad = g->pad;
  // a level-l pixel is the mean of a 2^l x 2^l block: x_0 = 2^l x_l + (2^l - 1)/2, so the centre moves accordingly
  const float s = (float)(1 / pow(2, l));
  L.cx = s * (g->w / 2.0f + 0.5f) - 0.5f;
  L.cy = s * (g->h / 2.0f + 0.5f) - 0.5f;
  L.f = s * (std::max(g->w, g->h) / 2.0f);
  return L;
}

extern "C" int ictr_icgn_create(ictr_icgn **out, int model, int w, int h, int lv_f, int lv_l, int maxiter, float eps,
                                const int *region_xywh, int64_t nproblems) {
  if (!out || model < 0 || model > 3 || w < 8 || h < 8 || lv_l < 0 || lv_f < lv_l || lv_f > 15 || maxiter < 0 ||
      nproblems < 1 || nproblems > 65535)
    return ictr_fail_(ICTR_ERR_INVALID, "icgn_create: bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return ictr_fail_(ICTR_ERR_NO_DEVICE, "no usable HIP device: the alignment engine has no CPU fallback");
  ictr_icgn *g = new ictr_icgn;
  g->model = model;
  g->w = w;
  g->h = h;
  g->lv_f = lv_f;
  g->lv_l = lv_l;
  g->maxiter = maxiter;
  g->eps = eps;
  g->B = (int)nproblems;
  g->nlev = lv_f + 1;
  if (region_xywh) {
    memcpy(g->region, region_xywh, sizeof(int) * 4);
  } else {  // whole frame minus a 2-pixel rim (the gradient is zero on the border)
    g->region[0] = g->region[1] = 2;
    g->region[2] = w - 4;
    g->region[3] = h - 4;
  }
  if (g->region[0] < 0 || g->region[1] < 0 || g->region[2] < 1 || g->region[3] < 1 || g->region[0] + g->region[2] > w ||
      g->region[1] + g->region[3] > h) {
    delete g;
    return ictr_fail_(ICTR_ERR_INVALID, "icgn_create: template region outside the frame");
  }
  g->rows[0] = 0;
  g->rows[1] = h;
  if (const char *sv = getenv("ICTR_ICGN_SCALAR")) g->scalar_only = atoi(sv) != 0;
  if (const char *sv = getenv("ICTR_ICGN_DBG")) g->dbg = atoi(sv);
  if (const char *sv = getenv("ICTR_ICGN_LDS")) g->lds = atoi(sv) != 0;
  const long npx = (long)g->region[2] * g->region[3];
  // workgroups per problem: enough to fill the chip a few times over, few enough that the tails' reductions and the
  // idle-block prologues stay cheap on the small levels (measured: 8 K total is the best level-0 choice, 32 K costs
  // 5 % there and 4x on the coarse levels); smaller levels launch fewer (icgn_grid)
  long want = std::max<long>(64, 8192 / g->B);
  if (const char *sv = getenv("ICTR_ICGN_GRIDX")) want = atol(sv);
  g->gridx = (int)std::max<long>(want, 1);
  hipError_t e = hipSuccess;
  auto alloc = [&](void **p, size_t bytes) {
    if (e == hipSuccess) e = hipMalloc(p, bytes);
    if (e == hipSuccess) e = hipMemset(*p, 0, bytes);
  };
  alloc((void **)&g->d_st, sizeof(IcState) * g->B);
  alloc((void **)&g->d_planes, sizeof(PlaneSet) * g->B * g->nlev);
  alloc((void **)&g->d_partH, sizeof(float) * (size_t)g->B * g->gridx * kIcPartH);
  alloc((void **)&g->d_partb, sizeof(float) * (size_t)g->B * g->gridx * kIcPartB);
  alloc((void **)&g->d_red, sizeof(float) * (size_t)g->B * kIcRed);
  if (e != hipSuccess) {
    delete g;
    return ictr_fail_(ICTR_ERR_HIP, "icgn_create: device allocation failed: %s", hipGetErrorString(e));
  }
  g->d_red_own = g->d_red;
  g->h_st.resize(g->B);
  g->h_planes.resize((size_t)g->B * g->nlev);
  g->frames_set.assign(g->B, 0);
  g->h_M0.assign((size_t)9 * g->B, 0.0f);
  for (int b = 0; b < g->B; ++b) g->h_M0[9 * b] = g->h_M0[9 * b + 4] = g->h_M0[9 * b + 8] = 1.0f;
  *out = g;
  return ICTR_OK;
}
extern "C" void ictr_icgn_destroy(ictr_icgn *g) {
  if (!g) return;
  for (hipEvent_t e : g->ev) (void)hipEventDestroy(e);
  for (void *p : {(void *)g->d_st, (void *)g->d_planes, (void *)g->d_partH, (void *)g->d_partb, (void *)g->d_red_own})
    if (p) hipFree(p);
  delete g;
}
extern "C" int ictr_icgn_set_stream(ictr_icgn *g, void *s) {
  if (!g) return ictr_fail_(ICTR_ERR_INVALID, "icgn is NULL");
  g->stream = (hipStream_t)s;
  return ICTR_OK;
}
extern "C" int ictr_icgn_set_frames(ictr_icgn *g, int64_t problem, const ictr_pyramid *tmpl, const ictr_pyramid *cur) {
  if (!g || problem < 0 || problem >= g->B || !tmpl || !cur) return ictr_fail_(ICTR_ERR_INVALID, "icgn_set_frames: bad arguments");
  ictr_pyramid_view a, c;
  ictr_pyramid_view_(tmpl, &a);
  ictr_pyramid_view_(cur, &c);
  if (a.nlev < g->nlev || c.nlev < g->nlev || !a.getgrad || a.pad < 2 || a.pad != c.pad || a.w[0] != g->w || a.h[0] != g->h ||
      c.w[0] != g->w || c.h[0] != g->h)
    return ictr_fail_(ICTR_ERR_INVALID, "icgn_set_frames: pyramids do not match the engine (size, levels, pad >= 2, gradients)");
  if (g->pad >= 0 && g->pad != a.pad) return ictr_fail_(ICTR_ERR_INVALID, "icgn_set_frames: all pyramids must share one padding");
  g->pad = a.pad;
  if (g->lw.empty()) {
    g->lw.assign(a.w, a.w + g->nlev);
    g->lh.assign(a.h, a.h + g->nlev);
    g->lsw.assign(a.sw, a.sw + g->nlev);
  }
  for (int l = 0; l < g->nlev; ++l) {
    if (a.w[l] != g->lw[l] || c.w[l] != g->lw[l] || a.h[l] != g->lh[l] || c.h[l] != g->lh[l])
      return ictr_fail_(ICTR_ERR_INVALID, "icgn_set_frames: level %d size mismatch", l);
    PlaneSet &ps = g->h_planes[(size_t)problem * g->nlev + l];
    ps.ref = a.img[l];
    ps.dx = a.dx[l];
    ps.dy = a.dy[l];
    ps.cur = c.img[l];
  }
  g->frames_set[problem] = 1;
  return ICTR_OK;
}
// M9: row-major 3x3 in level-0 PIXEL coordinates (template pixel -> current-frame pixel); NULL = identity
extern "C" int ictr_icgn_set_warp(ictr_icgn *g, int64_t problem, const double *M9) {
  if (!g || problem < 0 || problem >= g->B) return ictr_fail_(ICTR_ERR_INVALID, "icgn_set_warp: bad arguments");
  double K[9], Ki[9], t[9], Mn[9];
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (M9 && g->model != kHomog && (M9[6] != 0.0 || M9[7] != 0.0 || M9[8] == 0.0))
    return ictr_fail_(ICTR_ERR_INVALID, "icgn_set_warp: a projective initial warp needs the homography model");
  icgn_K(g, K, Ki);
  m3_mul(Ki, M9 ? M9 : I, t);
  m3_mul(t, K, Mn);
  for (int k = 0; k < 9; ++k) g->h_M0[9 * problem + k] = (float)(Mn[k] / Mn[8]);
  return ICTR_OK;
}
extern "C" int ictr_icgn_set_rows(ictr_icgn *g, int row_lo, int row_hi) {
  if (!g || row_lo < 0 || row_hi < row_lo) return ictr_fail_(ICTR_ERR_INVALID, "icgn_set_rows: bad arguments");
  g->rows[0] = row_lo;
  g->rows[1] = row_hi;
  return ICTR_OK;
}
extern "C" int ictr_icgn_enable_sharding(ictr_icgn *g, int enable, float *red_dev) {
  if (!g) return ictr_fail_(ICTR_ERR_INVALID, "icgn is NULL");
  g->sharded = enable ? 1 : 0;
  g->d_red = red_dev ? red_dev : g->d_red_own;
  return ICTR_OK;
}
extern "C" int ictr_icgn_set_timing(ictr_icgn *g, int enable) {
  if (!g) return ictr_fail_(ICTR_ERR_INVALID, "icgn is NULL");
  if (enable && g->ev.empty()) {
    g->ev.resize((size_t)2 * g->nlev * std::max(1, g->maxiter));
    for (auto &e : g->ev) HIPCHK_IC(hipEventCreate(&e));
  }
  g->timing = enable != 0;
  return ICTR_OK;
}

extern "C" int ictr_icgn_begin(ictr_icgn *g) {
  if (!g) return ictr_fail_(ICTR_ERR_INVALID, "icgn is NULL");
  for (int b = 0; b < g->B; ++b) {
    if (!g->frames_set[b]) return ictr_fail_(ICTR_ERR_STATE, "icgn: frames of problem %d not set", b);
    IcState &st = g->h_st[b];
    memset(&st, 0, sizeof(st));
    memcpy(st.M, &g->h_M0[9 * b], sizeof(float) * 9);
  }
  HIPCHK_IC(hipMemcpyAsync(g->d_st, g->h_st.data(), sizeof(IcState) * g->B, hipMemcpyHostToDevice, g->stream));
  HIPCHK_IC(hipMemcpyAsync(g->d_planes, g->h_planes.data(), sizeof(PlaneSet) * g->h_planes.size(), hipMemcpyHostToDevice,
                           g->stream));
  return ICTR_OK;
}

// workgroups per problem at one level: at most gridx, at most one per work unit, and balanced (every workgroup gets
// the same number of units, +-1)
static int icgn_grid(const ictr_icgn *g, int level, bool vec) {
  const int s = 1 << level;
  const long w = std::max(g->region[2] / s, 1), h = std::max(g->region[3] / s + 1, 1);
  const long units = vec ? ((w + 255) / 256 + 1) * ((h + 3) / 4) : (w * h + kBlock - 1) / kBlock;
  const long per = (units + g->gridx - 1) / g->gridx;
  return (int)std::max<long>(1, std::min<long>(g->gridx, (units + per - 1) / per));
}

template <typename F> static void icgn_dispatch(int model, F &&f) {
  switch (model) {
    case kTrans: f(std::integral_constant<int, kTrans>()); break;
    case kSE2: f(std::integral_constant<int, kSE2>()); break;
    case kAffine: f(std::integral_constant<int, kAffine>()); break;
    default: f(std::integral_constant<int, kHomog>()); break;
  }
}

extern "C" int ictr_icgn_hess_accumulate(ictr_icgn *g, int level) {
  if (!g || level < g->lv_l || level > g->lv_f) return ictr_fail_(ICTR_ERR_INVALID, "icgn: bad level");
  const IcDev e = icgn_dev(g);
  const IcLevel L = icgn_level(g, level);
  const int nblk = icgn_grid(g, level, false);
  const dim3 grid(nblk, g->B), blk(kBlock);
  icgn_dispatch(g->model, [&](auto m) { hipLaunchKernelGGL((k_icgn_hess<decltype(m)::value>), grid, blk, 0, g->stream, e, L, level); });
  hipLaunchKernelGGL(k_icgn_hess_tail, dim3(g->B), blk, 0, g->stream, e, nblk, 0);
  HIPCHK_IC(hipGetLastError());
  return ICTR_OK;
}
extern "C" int ictr_icgn_hess_finish(ictr_icgn *g, int level) {
  if (!g) return ictr_fail_(ICTR_ERR_INVALID, "icgn is NULL");
  if (g->sharded) hipLaunchKernelGGL(k_icgn_hess_tail, dim3(g->B), dim3(kBlock), 0, g->stream, icgn_dev(g), 0, 1);
  HIPCHK_IC(hipGetLastError());
  return ICTR_OK;
}
static int icgn_iter_main(ictr_icgn *g, const IcDev &e, const IcLevel &L, int level) {  // returns workgroups per problem
  const bool vec = g->pad >= 4 && (L.sw & 3) == 0 && !g->scalar_only;
  const int nblk = icgn_grid(g, level, vec);
  const dim3 grid(nblk, g->B), blk(kBlock);
  icgn_dispatch(g->model, [&](auto m) {
    if (vec && g->lds)
      hipLaunchKernelGGL((k_icgn_iter_lds<decltype(m)::value>), grid, blk, 0, g->stream, e, L, level);
    else if (vec)
      hipLaunchKernelGGL((k_icgn_iter<decltype(m)::value, 4>), grid, blk, 0, g->stream, e, L, level);
    else
      hipLaunchKernelGGL((k_icgn_iter<decltype(m)::value, 1>), grid, blk, 0, g->stream, e, L, level);
  });
  return nblk;
}
extern "C" int ictr_icgn_iter_accumulate(ictr_icgn *g, int level) {
  if (!g || level < g->lv_l || level > g->lv_f) return ictr_fail_(ICTR_ERR_INVALID, "icgn: bad level");
  const IcDev e = icgn_dev(g);
  const IcLevel L = icgn_level(g, level);
  const int nblk = icgn_iter_main(g, e, L, level);
  hipLaunchKernelGGL(k_icgn_iter_tail, dim3(g->B), dim3(kBlock), 0, g->stream, e, L.f, nblk, 0);
  HIPCHK_IC(hipGetLastError());
  return ICTR_OK;
}
extern "C" int ictr_icgn_iter_finish(ictr_icgn *g, int level) {
  if (!g) return ictr_fail_(ICTR_ERR_INVALID, "icgn is NULL");
  if (g->sharded) {
    const IcLevel L = icgn_level(g, level);
    hipLaunchKernelGGL(k_icgn_iter_tail, dim3(g->B), dim3(kBlock), 0, g->stream, icgn_dev(g), L.f, 0, 1);
  }
  HIPCHK_IC(hipGetLastError());
  return ICTR_OK;
}

extern "C" int ictr_icgn_run_async(ictr_icgn *g) {
  if (!g) return ictr_fail_(ICTR_ERR_INVALID, "icgn is NULL");
  if (g->sharded) return ictr_fail_(ICTR_ERR_STATE, "sharded engines are driven phase by phase");
  if (int rc = ictr_icgn_begin(g)) return rc;
  const IcDev e = icgn_dev(g);
  for (int l = g->lv_f; l >= g->lv_l; --l) {
    const IcLevel L = icgn_level(g, l);
    if (int rc = ictr_icgn_hess_accumulate(g, l)) return rc;
    for (int it = 0; it < g->maxiter; ++it) {
      const bool tk = g->timing && !g->ev.empty();
      if (tk) HIPCHK_IC(hipEventRecord(g->ev[2 * ((size_t)l * g->maxiter + it)], g->stream));
      const int nblk = icgn_iter_main(g, e, L, l);
      if (tk) HIPCHK_IC(hipEventRecord(g->ev[2 * ((size_t)l * g->maxiter + it) + 1], g->stream));
      hipLaunchKernelGGL(k_icgn_iter_tail, dim3(g->B), dim3(kBlock), 0, g->stream, e, L.f, nblk, 0);
    }
  }
  HIPCHK_IC(hipGetLastError());
  return ICTR_OK;
}

// results: warps (row-major 3x3, level-0 pixel coordinates), iterations, last dp
extern "C" int ictr_icgn_get_results(ictr_icgn *g, double *M9_out, int *iters, float *last_dp) {
  if (!g) return ictr_fail_(ICTR_ERR_INVALID, "icgn is NULL");
  HIPCHK_IC(hipMemcpyAsync(g->h_st.data(), g->d_st, sizeof(IcState) * g->B, hipMemcpyDeviceToHost, g->stream));
  HIPCHK_IC(hipStreamSynchronize(g->stream));
  double K[9], Ki[9], t[9], Mp[9], Mn[9];
  icgn_K(g, K, Ki);
  for (int b = 0; b < g->B; ++b) {
    const IcState &st = g->h_st[b];
    if (M9_out) {
      for (int k = 0; k < 9; ++k) Mn[k] = (double)st.M[k];
      m3_mul(K, Mn, t);
      m3_mul(t, Ki, Mp);
      for (int k = 0; k < 9; ++k) M9_out[9 * b + k] = Mp[k] / Mp[8];
    }
    if (iters) iters[b] = st.total_iters;
    if (last_dp) memcpy(last_dp + 8 * b, st.dp, sizeof(float) * 8);
  }
  return ICTR_OK;
}
extern "C" int ictr_icgn_get_kernel_times(ictr_icgn *g, float *ms_per_level) {
  if (!g || !ms_per_level) return ictr_fail_(ICTR_ERR_INVALID, "icgn_get_kernel_times: NULL argument");
  if (g->ev.empty()) return ictr_fail_(ICTR_ERR_STATE, "timing was never enabled");
  HIPCHK_IC(hipStreamSynchronize(g->stream));
  for (int l = 0; l < g->nlev; ++l) {
    ms_per_level[l] = 0.0f;
    if (l < g->lv_l) continue;
    for (int it = 0; it < g->maxiter; ++it) {
      float ms = 0.0f;
      HIPCHK_IC(hipEventElapsedTime(&ms, g->ev[2 * ((size_t)l * g->maxiter + it)], g->ev[2 * ((size_t)l * g->maxiter + it) + 1]));
      ms_per_level[l] += ms;
    }
  }
  return ICTR_OK;
}
