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
//
// r02: an iteration is a latency chain per wave (4096 patches = 4096 waves = one round of four per SIMD), so what
// counts is how few dependent steps it has: the four taps of a pixel are TWO 8-byte loads ((x-1,x) at y and at y-1),
// every lane's loads of an iteration are issued before the first is consumed (pixels beyond the patch read the patch's
// first pixel and carry zero templates: no branch in the loop), and the wave sums use DPP (11 instructions) instead of
// six dependent ds_bpermute round trips each. Same expressions in the same order per pixel: same values.
#include <stdlib.h>

#include "ictr_dev.h"
#include "ictr_devfn.h"

namespace ictr {

__device__ __forceinline__ float pf_wave_sum(float v) { return wave_sum_dpp(v); }

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
struct PFWin {
  f32x2_a4 ab, cd;  // (x-1,y),(x,y) and (x-1,y-1),(x,y-1)
};
__device__ __forceinline__ PFWin pf_load(gconst_f32 img, int idx, int sw) {
  PFWin w;
  w.ab = *reinterpret_cast<gconst_f32x2>(img + (idx - 1));
  w.cd = *reinterpret_cast<gconst_f32x2>(img + (idx - sw - 1));
  return w;
}
__device__ __forceinline__ float pf_blend(const PFWin &w, const PFTaps &t) {
  return t.w0 * w.ab.y + t.w1 * w.ab.x + t.w2 * w.cd.y + t.w3 * w.cd.x;
}
__device__ __forceinline__ float pf_fetch(gconst_f32 img, int idx, int sw, const PFTaps &t) {
  return pf_blend(pf_load(img, idx, sw), t);
}
__device__ __forceinline__ bool pf_in_view(float x, float y, float swo, float sho) {
  return (x >= 0.0f) & (y >= 0.0f) & (x <= swo) & (y <= sho);
}

// WPP waves share a patch (pixels per lane NPL = ceil(P*P / (64 WPP))). One wave per patch needs no synchronisation at
// all; for large patches (more than 4 pixels per lane) TWO waves per patch halve each wave's chain of pixels and double
// the waves that hide each other's loads (+2 % at 4096 patches of 31x31, +9 % at 65536): the wave sums then meet in LDS (partials of the two waves added in wave order by
// both: the same bits in both), one workgroup barrier per reduction. Barriers need a uniform loop structure: every
// patch of a workgroup runs every level and iteration slot; a patch that is lost or converged only stops updating.
template <int NPL, int WPP>
__global__ __launch_bounds__(kBlock) void k_patchflow(PFArgs a) {
  constexpr int PPB = kWaves / WPP;  // patches per workgroup
  __shared__ float sSum[WPP > 1 ? kWaves : 1][4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wsub = wave % WPP;
  const int k = blockIdx.x * PPB + wave / WPP;
  const bool valid = k < a.K;
  if (WPP == 1 && !valid) return;
  const int P = a.P, n = P * P;
  const int kk = valid ? k : 0;
  const float x0 = a.pts[kk], y0 = a.pts[kk + a.K];
  float px = 0.0f, py = 0.0f;
  bool ok = valid & (x0 == x0) & (y0 == y0);
  int nit = 0;
  int off[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) off[i] = 0;
  // sum over the patch's WPP waves of three / two per-wave totals (all waves of the patch get the same bits)
  auto patch_sum3 = [&](float &u, float &v, float &w) {
    u = pf_wave_sum(u);
    v = pf_wave_sum(v);
    w = pf_wave_sum(w);
    if constexpr (WPP > 1) {
      __syncthreads();  // (the previous reduction's partials have been read)
      if (lane == 0) {
        sSum[wave][0] = u;
        sSum[wave][1] = v;
        sSum[wave][2] = w;
      }
      __syncthreads();
      const int w0 = wave - wsub;
      u = v = w = 0.0f;
#pragma unroll
      for (int q = 0; q < WPP; ++q) {
        u += sSum[w0 + q][0];
        v += sSum[w0 + q][1];
        w += sSum[w0 + q][2];
      }
    }
  };

  for (int l = a.lv_f; l >= a.lv_l && (WPP > 1 || ok); --l) {
    const PFLevel L = a.lv[l];
    if (l != a.lv_f) {
      px *= 2.0f;
      py *= 2.0f;
    }
    const float xl = x0 * L.scale, yl = y0 * L.scale;
    if (ok && !pf_in_view(xl, yl, L.swo, L.sho)) ok = false;
    if (WPP == 1 && !ok) break;
    const PFTaps ta = pf_taps(ok ? xl : 1.0f, ok ? yl : 1.0f, P, L.sw);  // (1,1): a harmless in-plane window
    gconst_f32 pa = (gconst_f32)L.a, pax = (gconst_f32)L.ax, pay = (gconst_f32)L.ay, pb = (gconst_f32)L.b;
    float T[NPL], Gx[NPL], Gy[NPL];
    float hxx = 0.0f, hxy = 0.0f, hyy = 0.0f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int q = wsub * 64 + lane + 64 * WPP * i;
      const bool in = q < n;
      off[i] = in ? (q / P) * L.sw + (q % P) : 0;  // beyond the patch: its first pixel, with zero templates
      const int idx = ta.base + off[i];
      const float t = pf_fetch(pa, idx, L.sw, ta), gx = pf_fetch(pax, idx, L.sw, ta), gy = pf_fetch(pay, idx, L.sw, ta);
      T[i] = in ? t : 0.0f;
      Gx[i] = in ? gx : 0.0f;
      Gy[i] = in ? gy : 0.0f;
      hxx += Gx[i] * Gx[i];
      hxy += Gx[i] * Gy[i];
      hyy += Gy[i] * Gy[i];
    }
    patch_sum3(hxx, hxy, hyy);
    const float det = hxx * hyy - hxy * hxy;
    const float tr = hxx + hyy;
    if (ok && (!(det > a.min_det * tr * tr) || !(tr > 0.0f))) ok = false;  // textureless or 1-D structure
    if (WPP == 1 && !ok) break;
    const float idet = 1.0f / det;
    bool run = ok;  // this patch still iterates at this level
    for (int it = 0; it < a.maxiter; ++it) {
      if (WPP == 1 && !run) break;
      const float cx = xl + px, cy = yl + py;
      if (run && !pf_in_view(cx, cy, L.swo, L.sho)) {
        ok = false;
        run = false;
        if (WPP == 1) break;
      }
      const PFTaps tb = pf_taps(run ? cx : 1.0f, run ? cy : 1.0f, P, L.sw);
      float bx = 0.0f, by = 0.0f, dummy = 0.0f;
      PFWin w[NPL];
#pragma unroll
      for (int i = 0; i < NPL; ++i) w[i] = pf_load(pb, tb.base + off[i], L.sw);  // all in flight before the first use
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        float r = T[i] - pf_blend(w[i], tb);
        r = (wsub * 64 + lane + 64 * WPP * i < n) ? r : 0.0f;  // (T = 0 there, but the frame value is not)
        bx += Gx[i] * r;
        by += Gy[i] * r;
      }
      patch_sum3(bx, by, dummy);
      if (run) {
        const float dx = (hyy * bx - hxy * by) * idet;
        const float dy = (hxx * by - hxy * bx) * idet;
        px += dx;
        py += dy;
        ++nit;
        if (dx * dx + dy * dy < a.eps2) run = false;
      }
    }
  }
  if (lane == 0 && wsub == 0 && valid) {
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
  const dim3 blk(kBlock);
  // few large patches: two waves per patch (see k_patchflow); ICTR_PF_WPP=1 keeps one wave per patch (A/B)
  static const int wpp_env = [] {
    const char *e = getenv("ICTR_PF_WPP");
    return e ? atoi(e) : 0;
  }();
  // measured (31x31 patches, 3 levels x 10 iterations): 4096 patches 168 against 173 us, 65536 patches 2.74 against 2.99 ms
  const bool two = wpp_env ? wpp_env == 2 : npl > 4;
  if (two) {
    const dim3 g((a.K + kWaves / 2 - 1) / (kWaves / 2));
    hipLaunchKernelGGL((k_patchflow<8, 2>), g, blk, 0, s, a);
    return;
  }
  const dim3 g((a.K + kWaves - 1) / kWaves);
  if (npl <= 1)
    hipLaunchKernelGGL((k_patchflow<1, 1>), g, blk, 0, s, a);
  else if (npl <= 4)
    hipLaunchKernelGGL((k_patchflow<4, 1>), g, blk, 0, s, a);
  else
    hipLaunchKernelGGL((k_patchflow<16, 1>), g, blk, 0, s, a);
}

}  // namespace ictr
