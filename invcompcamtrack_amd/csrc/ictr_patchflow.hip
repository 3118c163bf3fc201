// ictr_patchflow.hip -- per-patch translation inverse-compositional Lucas-Kanade ("point tracker").
//
// Role (BASELINE config 4): the reference's misc_src/run_*OF* drivers obtain their
// displacement fields from an EXTERNAL optical-flow binary (run_OF_RGB / run_DE_RGB of the author's OF_DIS repo,
// misc_src/run_test_OF_track.py:90-108) that is not part of the repository. This kernel is the in-tree
// producer that replaces it: K independent P x P patches, each with its own 2-parameter translation, pyramidal,
// same Gauss-Newton skeleton as the camera tracker with J = identity:
//     H = sum [Gx Gy]^T [Gx Gy] over the template patch (once per level),
//     b = sum [Gx Gy]^T (T - I(x + p)),   p += H^-1 b.
// Sampling convention = util_getPatch's (utilities.cpp:55-113: patch-constant bilinear weights, ceil(x+1e-5f) taps,
// offsets -(P - P/2)...), so the patches it sees are the tracker's patches. The algorithm itself is build-defined:
// there is no reference implementation to pin it ("parity unpinned by the reference"); the oracle is
// oracle/np_patchflow.py.
//
// MI355X mapping: one wave64 per patch does ALL levels and ALL iterations of its patch inside one launch (the
// problems are independent: no global reduction, no launch per iteration). The template T/Gx/Gy of the patch lives
// in registers (<= 16 pixels per lane => P <= 32), H and b are wave shuffle reductions, every lane solves the 2x2.
#include "ictr_dev.h"

namespace ictr {

__device__ __forceinline__ float pf_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

struct PFTaps {
  float w0, w1, w2, w3;
  int base;
};
__device__ __forceinline__ PFTaps pf_taps(float mx, float my, int P, int sw) {
  PFTaps t;
  const int p0 = (int)ceilf(mx + .00001f), p1 = (int)ceilf(my + .00001f);
  const float r0 = mx - floorf(mx), r1 = my - floorf(my);
  t.w0 = r0 * r1;
  t.w1 = (1 - r0) * r1;
  t.w2 = r0 * (1 - r1);
  t.w3 = (1 - r0) * (1 - r1);
  t.base = (p1 + P / 2) * sw + p0 + P / 2;
  return t;
}
__device__ __forceinline__ float pf_fetch(const float *__restrict__ img, int idx, int sw, const PFTaps &t) {
  return t.w0 * img[idx] + t.w1 * img[idx - 1] + t.w2 * img[idx - sw] + t.w3 * img[idx - sw - 1];
}
__device__ __forceinline__ bool pf_in_view(float x, float y, float swo, float sho) {
  return (x >= 0.0f) & (y >= 0.0f) & (x <= swo) & (y <= sho);
}

template <int NPL>  // pixels per lane = ceil(P*P / 64)
__global__ __launch_bounds__(kBlock) void k_patchflow(PFArgs a) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (k >= a.K) return;
  const int P = a.P, n = P * P;
  const float x0 = a.pts[k], y0 = a.pts[k + a.K];
  float px = 0.0f, py = 0.0f;
  bool ok = (x0 == x0) & (y0 == y0);
  int nit = 0;
  int off[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) off[i] = 0;

  for (int l = a.lv_f; l >= a.lv_l && ok; --l) {
    const PFLevel L = a.lv[l];
    if (l != a.lv_f) {
      px *= 2.0f;
      py *= 2.0f;
    }
    const float xl = x0 * L.scale, yl = y0 * L.scale;
    if (!pf_in_view(xl, yl, L.swo, L.sho)) {
      ok = false;
      break;
    }
    const PFTaps ta = pf_taps(xl, yl, P, L.sw);
    float T[NPL], Gx[NPL], Gy[NPL];
    float hxx = 0.0f, hxy = 0.0f, hyy = 0.0f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int q = lane + 64 * i;
      T[i] = Gx[i] = Gy[i] = 0.0f;
      off[i] = (q / P) * L.sw + (q % P);
      if (q < n) {
        const int idx = ta.base + off[i];
        T[i] = pf_fetch(L.a, idx, L.sw, ta);
        Gx[i] = pf_fetch(L.ax, idx, L.sw, ta);
        Gy[i] = pf_fetch(L.ay, idx, L.sw, ta);
      }
      hxx += Gx[i] * Gx[i];
      hxy += Gx[i] * Gy[i];
      hyy += Gy[i] * Gy[i];
    }
    hxx = pf_wave_sum(hxx);
    hxy = pf_wave_sum(hxy);
    hyy = pf_wave_sum(hyy);
    const float det = hxx * hyy - hxy * hxy;
    const float tr = hxx + hyy;
    if (!(det > a.min_det * tr * tr) || !(tr > 0.0f)) {  // textureless or 1-D structure: the 2x2 is not solvable
      ok = false;
      break;
    }
    const float idet = 1.0f / det;
    for (int it = 0; it < a.maxiter; ++it) {
      const float cx = xl + px, cy = yl + py;
      if (!pf_in_view(cx, cy, L.swo, L.sho)) {
        ok = false;
        break;
      }
      const PFTaps tb = pf_taps(cx, cy, P, L.sw);
      float bx = 0.0f, by = 0.0f;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int q = lane + 64 * i;
        if (q < n) {
          const float r = T[i] - pf_fetch(L.b, tb.base + off[i], L.sw, tb);
          bx += Gx[i] * r;
          by += Gy[i] * r;
        }
      }
      bx = pf_wave_sum(bx);
      by = pf_wave_sum(by);
      const float dx = (hyy * bx - hxy * by) * idet;
      const float dy = (hxx * by - hxy * bx) * idet;
      px += dx;
      py += dy;
      ++nit;
      if (dx * dx + dy * dy < a.eps2) break;
    }
  }
  if (lane == 0) {
    const float s = 1.0f / a.lv[a.lv_l].scale;  // back to level-0 pixels
    const float nanv = __int_as_float(0x7fc00000);
    a.out[k] = ok ? x0 + px * s : nanv;
    a.out[k + a.K] = ok ? y0 + py * s : nanv;
    a.status[k] = ok ? 1 : 0;
    a.iters[k] = nit;
  }
}

void launch_patchflow(const PFArgs &a, hipStream_t s) {
  const int n = a.P * a.P;
  const int npl = (n + 63) / 64;
  const dim3 g((a.K + kWaves - 1) / kWaves), blk(kBlock);
  if (npl <= 1)
    hipLaunchKernelGGL(k_patchflow<1>, g, blk, 0, s, a);
  else if (npl <= 4)
    hipLaunchKernelGGL(k_patchflow<4>, g, blk, 0, s, a);
  else
    hipLaunchKernelGGL(k_patchflow<16>, g, blk, 0, s, a);
}

}  // namespace ictr
