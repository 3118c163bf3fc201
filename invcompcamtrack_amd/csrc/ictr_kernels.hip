// ictr_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the Gauss-Newton photometric tracker.
//
// Reference steps (odometer.cpp:257-426, /root/reference) -> kernels:
//   step 3   project at the reference pose, all levels      k_project_ref      (pose.cpp:307-488)
//   step 4-6 reference patches + gradients, sd coefficients,
//            Hessian                                        k_ref_level        (utilities.cpp:115-189, odometer.cpp:268-334,428-472)
//   step 7-10 project, fetch current patch, residual,
//            J^T r, solve, pose update, loop condition       k_iter             (utilities.cpp:55-113, odometer.cpp:344-418,509-515)
//   pyramid  2x2 box + [-1 0 1] gradients + padding          k_pyr_*            (utilities.cpp:14-52)
//
// Design: memory-bound gather + rank-1 accumulate, no MFMA. One wave64 owns one 8x8 patch
// (four 4x4 patches): the 64 lanes ARE the 64 patch pixels, so T/Gx/Gy are read as 256-B coalesced rows, the
// bilinear taps are 8 x 32-B row segments, and the six J^T r sums live in per-lane registers across all the
// patches a wave visits -- one shuffle reduction per wave per launch, not per patch. The steepest-descent
// images of the reference (6 planes + 6 projected planes, re-zeroed every iteration) are never materialised:
// a patch carries 12 scalar coefficients. Every workgroup leaves one partial sum per component; a second, tiny
// launch (one workgroup per problem: k_level_tail / k_iter_tail) adds them in a fixed order in f64, solves the
// 6x6 system and updates the pose on the device, so the host never reads anything back inside the loop and the
// sums are reproducible bit for bit. Doing that reduction inside the big kernel ("last workgroup to arrive")
// was measured and rejected: it needs one agent-scope release (an L2 write-back scan, buffer_wbl2 sc1) per
// workgroup, ~4000 per launch, which throttled the L2 for every wave: 310 us per iteration against 141 us for
// the two-launch form (profiles/r01_notes.md).
//
// Arithmetic parity: compiled with -ffp-contract=off; every expression below keeps the reference's
// operand order, so patches, projections and coefficients are bit-identical to the CPU path; only the
// order of the big sums differs.
#include <algorithm>

#include <hip/hip_ext.h>

#include "ictr_dev.h"
#include "ictr_devfn.h"
#include "se3_math.h"

namespace ictr {

// ---------------------------------------------------------------- step 3: projection at the reference pose
// pose.cpp:400-488 (save rotated, level lv_f) + pose.cpp:307-397 for the other levels (odometer.cpp:251-254).
// The camera-frame point is level independent, so one pass writes every level.
struct AllCams {
  LevelCam lc[16];
};

__global__ __launch_bounds__(kBlock) void k_project_ref(EngineDev e, AllCams cams) {
  const int b = blockIdx.y;
  const ProbState &st = e.st[b];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i == 0 && b == 0 && e.trace.count != nullptr) *e.trace.count = 0;  // the tracking's trace starts here (begin phase)
  if (i >= st.npts) return;
  const float *p3 = e.pt3d + (size_t)b * 3 * e.M;
  float *p3r = e.pt3d_ref + (size_t)b * 3 * e.M;
  const float X = p3[i], Y = p3[i + e.M], Z = p3[i + 2 * e.M];
  const float tx = st.G[0] * X + st.G[1] * Y + st.G[2] * Z + st.G[3];
  const float ty = st.G[4] * X + st.G[5] * Y + st.G[6] * Z + st.G[7];
  const float tz = st.G[8] * X + st.G[9] * Y + st.G[10] * Z + st.G[11];
  p3r[i] = tx;
  p3r[i + e.M] = ty;
  p3r[i + 2 * e.M] = tz;
  for (int l = e.lv_l; l <= e.lv_f; ++l) {
    float *p2 = e.pt2d + ((size_t)b * e.nlev + l) * 2 * e.M;
    p2[i] = (tx / tz) * cams.lc[l].fx + cams.lc[l].cx;
    p2[i + e.M] = (ty / tz) * cams.lc[l].fy + cams.lc[l].cy;
  }
}

// PoseClass::project_pt / project_pt_save_rotated on caller buffers (device copies), SoA stride M
__global__ __launch_bounds__(kBlock) void k_project_generic(const float *__restrict__ pt3d, float *pt3d_rot,
                                                            float *pt2d, int n, int M, const float *__restrict__ G,
                                                            LevelCam lc) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float X = pt3d[i], Y = pt3d[i + M], Z = pt3d[i + 2 * M];
  const float tx = G[0] * X + G[1] * Y + G[2] * Z + G[3];
  const float ty = G[4] * X + G[5] * Y + G[6] * Z + G[7];
  const float tz = G[8] * X + G[9] * Y + G[10] * Z + G[11];
  if (pt3d_rot) {
    pt3d_rot[i] = tx;
    pt3d_rot[i + M] = ty;
    pt3d_rot[i + 2 * M] = tz;
  }
  pt2d[i] = (tx / tz) * lc.fx + lc.cx;
  pt2d[i + M] = (ty / tz) * lc.fy + lc.cy;
}

// ---------------------------------------------------------------- steps 4-6: per level setup (any patch size)
// PT = compile-time patch size (4), or 0 = run-time e.P (any size; a wave loops over the pixels). Writes one
// partial of the 21 unique H entries per workgroup; k_level_tail finishes.
template <int PT>
__global__ __launch_bounds__(kBlock) void k_ref_level(EngineDev e, LevelCam lc, int level) {
  __shared__ float sW[kWaves][kPartHStride];
  const int b = blockIdx.y;
  const ProbState &st = e.st[b];
  const int npts = st.npts;
  const int P = PT ? PT : e.P;
  const int n = P * P;
  const int pszd2 = P / 2;
  const int M = e.M;
  const PlaneSet pl = e.planes[b * e.nlev + level];
  const float *pt2d = e.pt2d + ((size_t)b * e.nlev + level) * 2 * M;
  const float *p3r = e.pt3d_ref + (size_t)b * 3 * M;
  float *T = e.T + (size_t)b * M * n;
  float *Gx = e.Gx + (size_t)b * M * n;
  float *Gy = e.Gy + (size_t)b * M * n;
  float *coefb = e.coef + (size_t)b * M * kCoefStride;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int ppw = (n <= 64 && (64 % n) == 0) ? 64 / n : 1;  // patches per wave
  const int sub = ppw > 1 ? lane / n : 0;
  const int q0 = ppw > 1 ? lane % n : lane;
  const int qstride = ppw > 1 ? n : 64;
  const int gwidth = ppw > 1 ? n : 64;

  float acc[kHUnique];
#pragma unroll
  for (int j = 0; j < kHUnique; ++j) acc[j] = 0.0f;

  const int nw = gridDim.x * kWaves;
  for (int g = blockIdx.x * kWaves + wave; g * ppw < npts; g += nw) {
    const int i = g * ppw + sub;
    const bool valid = i < npts;
    const float mx = valid ? pt2d[i] : -1.0f;
    const float my = valid ? pt2d[i + M] : -1.0f;
    const bool vis = valid && in_view(mx, my, lc.swo, lc.sho);
    float cx[6], cy[6];
    if (vis) {
      sd_coefs(p3r[i], p3r[i + M], p3r[i + 2 * M], lc.fx, lc.fy, cx, cy);
      if (q0 == 0) {
        float *c = coefb + (size_t)i * kCoefStride;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          c[k] = cx[k];
          c[6 + k] = cy[k];
        }
      }
    } else if (valid) {  // out of the reference view at this level: keep the stale coefficients (quirk, odometer.cpp:304)
      const float *c = coefb + (size_t)i * kCoefStride;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        cx[k] = c[k];
        cy[k] = c[6 + k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) cx[k] = cy[k] = 0.0f;
    }
    const Taps tp = make_taps(vis ? mx : 0.0f, vis ? my : 0.0f, pszd2);
    const int base = (tp.row0)*lc.sw + tp.col0;

    float mean = 0.0f;
    if (e.dopatchnorm) {  // utilities.cpp:187-188 : intensity patch only
      float s = 0.0f;
      for (int q = q0; q < n; q += qstride)
        if (vis) s += tap4(pl.ref, base + (q / P) * lc.sw + (q % P), lc.sw, tp);
      s = group_sum(s, gwidth);
      mean = s / (float)n;
    }
    for (int q = q0; q < n; q += qstride) {
      float gx = 0.0f, gy = 0.0f;
      const size_t o = (size_t)i * n + q;
      if (vis) {
        const int idx = base + (q / P) * lc.sw + (q % P);
        float t = tap4(pl.ref, idx, lc.sw, tp);
        if (e.dopatchnorm) t -= mean;
        gx = tap4(pl.dx, idx, lc.sw, tp);
        gy = tap4(pl.dy, idx, lc.sw, tp);
        T[o] = t;
        Gx[o] = gx;
        Gy[o] = gy;
      } else if (valid) {
        if (e.robust & ICTR_ROBUST_CLEAN) {  // option: no stale contributions, neither to H nor (through sd) to b
          Gx[o] = 0.0f;
          Gy[o] = 0.0f;
        } else {
          gx = Gx[o];
          gy = Gy[o];
        }
      }
      float sd[6];
      sd_values(gx, gy, cx, cy, sd);
      int jk = 0;
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int k = j; k < 6; ++k) acc[jk++] += sd[j] * sd[k];
    }
  }

#pragma unroll
  for (int j = 0; j < kHUnique; ++j) {
    const float v = wave_sum(acc[j]);
    if (lane == 0) sW[wave][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < kHUnique) {
    const float v = (sW[0][threadIdx.x] + sW[1][threadIdx.x]) + (sW[2][threadIdx.x] + sW[3][threadIdx.x]);
    e.partH[((size_t)b * gridDim.x + blockIdx.x) * kPartHStride + threadIdx.x] = v;
  }
}

// One workgroup per problem: fixed-order f64 reduction of the H partials of `nblk` workgroups (8 slices x 32
// components, then the slices in order), publish H (or the rank-local sum when sharded), reset the loop state.
// fixed-order f64 reduction of the H partials of `nblk` workgroups into sH[0..20] (all threads of the workgroup)
__device__ __forceinline__ void reduce_partH(const EngineDev &e, int b, int nblk, double (*sRed)[32], float *sH) {
  {
    const int j = threadIdx.x & 31, sl = threadIdx.x >> 5;
    double s = 0.0;
    const float *ph = e.partH + (size_t)b * nblk * kPartHStride + j;
    // (unrolled: eight loads in flight, the additions stay in the same order -- with thousands of partials, e.g. one
    // dense 1080p pair in 4-point chunks, the serial load latency of this loop was 60 us per level)
    if (j < kHUnique) {
#pragma unroll 8
      for (int k = sl; k < nblk; k += kBlock / 32) s += (double)ph[(size_t)k * kPartHStride];
    }
    sRed[sl][j] = s;
  }
  __syncthreads();
  if (threadIdx.x < kHUnique) {
    double s = 0.0;
#pragma unroll
    for (int sl = 0; sl < kBlock / 32; ++sl) s += sRed[sl][threadIdx.x];
    sH[threadIdx.x] = (float)s;
  }
  __syncthreads();
}
// defer_h: the P = 8 fast path accumulates H inside the level's FIRST iteration launch (k_iter8<.., WH = true>: it
// streams Gx, Gy and the coefficients anyway), so the setup kernel is a pure gather/store kernel and this tail only
// resets the loop state; the first k_iter_tail / k_iter_finish of the level reduces and factors H.
// The factorisation and, per iteration, the solve / pose update / loop condition run on ONE wave in registers
// (WaveSolver, ictr_devfn.h) instead of one thread on LDS arrays: 3-4 k cycles instead of ~14 k on the critical path
// between two accumulate launches.
__global__ __launch_bounds__(kBlock) void k_level_tail(EngineDev e, int nblk, int defer_h) {
  __shared__ double sRed[kBlock / 32][32];
  __shared__ float sH[32];
  const int b = blockIdx.x;
  ProbState &st = e.st[b];
  if (defer_h) {
    if (threadIdx.x == 0) level_reset(st, e);
    return;
  }
  reduce_partH(e, b, nblk, sRed, sH);
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  if (e.sharded) {
    if (lane < kHUnique) e.red[(size_t)b * kRedStride + lane] = sH[lane];
  } else {
    WaveSolver S;
    ws_factor(S, sH[h_unique_index(lane)], lane);
    ws_store_factor(S, st, lane);
    if (lane == 0) level_reset(st, e);
  }
}

// sharded mode: adopt the all-reduced H (red[b][0..20]) and reset the iteration state (one wave per problem)
__global__ __launch_bounds__(64) void k_level_finish(EngineDev e, int defer_h) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  ProbState &st = e.st[b];
  if (defer_h) {
    if (lane == 0) level_reset(st, e);
    return;
  }
  const float a = e.red[(size_t)b * kRedStride + h_unique_index(lane)];
  __builtin_amdgcn_wave_barrier();
  // leave the slot zeroed: the caller may all-reduce the whole buffer again before it is rewritten
  if (lane < kRedStride) e.red[(size_t)b * kRedStride + lane] = 0.0f;
  WaveSolver S;
  ws_factor(S, a, lane);
  ws_store_factor(S, st, lane);
  if (lane == 0) level_reset(st, e);
}

// ---------------------------------------------------------------- steps 7-9a: one Gauss-Newton iteration (any patch size)
template <int PT>
__global__ __launch_bounds__(kBlock) void k_iter(EngineDev e, LevelCam lc, int level) {
  __shared__ float sW[kWaves][kPartBStride];
  const int b = blockIdx.y;
  const ProbState &st = e.st[b];
  if (!st.active) return;  // loop condition of odometer.cpp:344-346, decided on the device by the previous tail
  const int npts = st.npts;
  const int P = PT ? PT : e.P;
  const int n = P * P;
  const int pszd2 = P / 2;
  const int M = e.M;
  const PlaneSet pl = e.planes[b * e.nlev + level];
  const float *__restrict__ p3 = e.pt3d + (size_t)b * 3 * M;
  const float *__restrict__ T = e.T + (size_t)b * M * n;
  const float *__restrict__ Gx = e.Gx + (size_t)b * M * n;
  const float *__restrict__ Gy = e.Gy + (size_t)b * M * n;
  const float *__restrict__ coefb = e.coef + (size_t)b * M * kCoefStride;
  const float *__restrict__ cur = pl.cur;

  float G[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) G[k] = st.G[k];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int ppw = (n <= 64 && (64 % n) == 0) ? 64 / n : 1;
  const int sub = ppw > 1 ? lane / n : 0;
  const int q0 = ppw > 1 ? lane % n : lane;
  const int qstride = ppw > 1 ? n : 64;
  const int gwidth = ppw > 1 ? n : 64;

  float acc[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) acc[k] = 0.0f;

  const int nw = gridDim.x * kWaves;
  for (int g = blockIdx.x * kWaves + wave; g * ppw < npts; g += nw) {
    const int i = g * ppw + sub;
    const bool valid = i < npts;
    // step 7 (pose.cpp:384-391)
    const float X = valid ? p3[i] : 0.0f, Y = valid ? p3[i + M] : 0.0f, Z = valid ? p3[i + 2 * M] : 1.0f;
    const float tx = G[0] * X + G[1] * Y + G[2] * Z + G[3];
    const float ty = G[4] * X + G[5] * Y + G[6] * Z + G[7];
    const float tz = G[8] * X + G[9] * Y + G[10] * Z + G[11];
    const float mx = (tx / tz) * lc.fx + lc.cx;
    const float my = (ty / tz) * lc.fy + lc.cy;
    const bool vis = valid && in_view(mx, my, lc.swo, lc.sho);  // ind_new (odometer.cpp:369-377)
    float cx[6], cy[6];
    {
      const float4 *c4 = reinterpret_cast<const float4 *>(coefb + (size_t)(valid ? i : 0) * kCoefStride);
      const float4 c0 = c4[0], c1 = c4[1], c2 = c4[2];
      cx[0] = c0.x; cx[1] = c0.y; cx[2] = c0.z; cx[3] = c0.w; cx[4] = c1.x; cx[5] = c1.y;
      cy[0] = c1.z; cy[1] = c1.w; cy[2] = c2.x; cy[3] = c2.y; cy[4] = c2.z; cy[5] = c2.w;
    }
    const Taps tp = make_taps(vis ? mx : 0.0f, vis ? my : 0.0f, pszd2);
    const int base = tp.row0 * lc.sw + tp.col0;

    float mean = 0.0f;
    if (e.dopatchnorm) {  // utilities.cpp:111-112
      float s = 0.0f;
      for (int q = q0; q < n; q += qstride)
        if (vis) s += tap4(cur, base + (q / P) * lc.sw + (q % P), lc.sw, tp);
      s = group_sum(s, gwidth);
      mean = s / (float)n;
    }
    for (int q = q0; q < n; q += qstride) {
      if (vis) {
        const size_t o = (size_t)i * n + q;
        float inew = tap4(cur, base + (q / P) * lc.sw + (q % P), lc.sw, tp);
        if (e.dopatchnorm) inew -= mean;
        float r = T[o] - inew;  // pdiff (odometer.cpp:381)
        if (e.robust & ICTR_ROBUST_HUBER) {
          const float ar = fabsf(r);
          if (ar > e.huber_k) r *= e.huber_k / ar;
        }
        float sd[6];
        sd_values(Gx[o], Gy[o], cx, cy, sd);
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] += sd[k] * r;  // sd*_proj summed (odometer.cpp:386-404)
      }
    }
  }

#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float v = wave_sum(acc[k]);
    if (lane == 0) sW[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const float v = (sW[0][threadIdx.x] + sW[1][threadIdx.x]) + (sW[2][threadIdx.x] + sW[3][threadIdx.x]);
    e.partb[((size_t)b * gridDim.x + blockIdx.x) * kPartBStride + threadIdx.x] = v;
  }
}

// One workgroup per problem: fixed-order f64 reduction of the b partials (32 slices x 8 components, then the
// slices in order) and steps 9b-10 (solve, pose update, loop condition) on wave 0 -- or, when the points are sharded
// over ranks, just the rank-local sums into red[] for the all-reduce.
__global__ __launch_bounds__(kBlock) void k_iter_tail(EngineDev e, int level, int nblk, int first_h) {
  __shared__ double sRed[kBlock / 8][8];
  __shared__ double sRedH[kBlock / 32][32];
  __shared__ float sH[32];
  const int b = blockIdx.x;
  ProbState &st = e.st[b];
  if (!st.active) return;
  // deferred H: the launch before this one also wrote the H partials (workgroup-uniform branch)
  if (first_h) reduce_partH(e, b, nblk, sRedH, sH);
  {
    const int j = threadIdx.x & 7, sl = threadIdx.x >> 3;
    double s = 0.0;
    const float *pb = e.partb + (size_t)b * nblk * kPartBStride + j;
    if (j < 6) {
#pragma unroll 8
      for (int k = sl; k < nblk; k += kBlock / 8) s += (double)pb[(size_t)k * kPartBStride];
    }
    sRed[sl][j] = s;
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  double bs = 0.0;
  if (lane < 6) {
#pragma unroll
    for (int sl = 0; sl < kBlock / 8; ++sl) bs += sRed[sl][lane];
  }
  if (e.sharded) {
    if (first_h && lane < kHUnique) e.red[(size_t)b * kRedStride + lane] = sH[lane];
    if (lane < 6) e.red[(size_t)b * kRedStride + kHUnique + lane] = (float)bs;
    return;
  }
  WaveSolver S;
  float G[12];
  ws_load_state(S, st, lane, G);
  if (first_h) {
    ws_factor(S, sH[h_unique_index(lane)], lane);
    ws_store_factor(S, st, lane);
  } else {
    ws_load_factor(S, st, lane);
  }
  ws_iterate(S, (float)bs, solve_opts(e), level, b, lane, G);
  ws_store_state(S, st, lane, G);
}

// sharded mode: steps 9b-10 on the all-reduced b (red[b][21..26]); every rank does the same arithmetic
// (one wave per problem)
__global__ __launch_bounds__(64) void k_iter_finish(EngineDev e, int level, int first_h) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  ProbState &st = e.st[b];
  if (!st.active) return;
  WaveSolver S;
  float G[12];
  ws_load_state(S, st, lane, G);
  float a = 0.0f, bi = 0.0f;
  if (first_h) a = e.red[(size_t)b * kRedStride + h_unique_index(lane)];  // deferred H: adopt the all-reduced sums
  if (lane < 6) bi = e.red[(size_t)b * kRedStride + kHUnique + lane];
  __builtin_amdgcn_wave_barrier();
  if (lane < kRedStride && (first_h || lane >= kHUnique)) e.red[(size_t)b * kRedStride + lane] = 0.0f;
  if (first_h) {
    ws_factor(S, a, lane);
    ws_store_factor(S, st, lane);
  } else {
    ws_load_factor(S, st, lane);
  }
  ws_iterate(S, bi, solve_opts(e), level, b, lane, G);
  ws_store_state(S, st, lane, G);
}

// ================================================================ P = 8 fast path (wave64 == one 8x8 patch)
// Two stages per chunk of up to 64 consecutive points owned by one wave:
//   stage 1  one POINT per lane: coalesced reads of the point arrays; projection / visibility / bilinear weights /
//            tap base index computed once per point (not 64x redundantly per patch), in the reference's exact
//            arithmetic; the per-point constants go into a wave-private LDS record;
//   stage 2  one PATCH per step, the lanes are its 64 pixels: T/Gx/Gy are 256-B coalesced rows with a scalar
//            base, the frame window is read with every cache line requested once, the per-patch constants come
//            back from LDS as broadcast ds_read_b128. The steps are software-pipelined (double-buffered
//            registers): the loads of step s+1 are in flight while step s is reduced.
// XCD-aware workgroup order. The hardware deals consecutive workgroup ids round-robin to the 8 XCDs, each with its
// own L2. Re-labelling id -> (id % 8) * (n / 8) + id / 8 gives every XCD a contiguous band of chunks, so neighbouring
// chunks (which share frame cache lines at their borders) meet in one L2 instead of being fetched twice from the
// fabric. Needs a grid that is a multiple of 8 (the launchers round up; surplus workgroups find no chunk).
__device__ __forceinline__ int xcd_band_block(int bx, int gx) { return (gx & 7) ? bx : (bx & 7) * (gx >> 3) + (bx >> 3); }

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef const f32x4_t __attribute__((address_space(1))) *gconst_f32x4;

// Packed taps: the setup kernel gathers the SAME 9x9 window from three planes. With the planes interleaved as one
// {img, dx, dy, 0} texel per pixel a window row is 144 contiguous bytes (2 cache lines) instead of 3 x 36 bytes in
// 3-6 lines, and two 16-byte loads per lane replace three 8-byte ones. Measured reason: the three separate gathers
// pulled ~7 KB of lines per patch through an L1 that cannot hold the 20 waves' windows, and the kernel ran at the
// L2's pace, not HBM's (ablations in profiles/r01_notes.md).
struct TapLoads4 {
  f32x4_t l, r, tl, tr;  // texels at (x-1, y), (x, y) and, for lanes 0-7, the row above
};
__device__ __forceinline__ TapLoads4 taps_issue4(gconst_f32x4 pack_at_base, int loff, int sw, int lane) {
  TapLoads4 t;
  t.l = pack_at_base[loff - 1];
  t.r = pack_at_base[loff];
  f32x4_t z = {0.0f, 0.0f, 0.0f, 0.0f};
  t.tl = z;
  t.tr = z;
  if (lane < 8) {
    t.tl = pack_at_base[loff - sw - 1];
    t.tr = pack_at_base[loff - sw];
  }
  return t;
}
// same operand order as taps_blend (utilities.cpp:107), plane k of the texels
__device__ __forceinline__ float taps_blend4(const TapLoads4 &t, int k, float w0, float w1, float w2, float w3, int lane) {
  const float a = t.r[k], b = t.l[k];
  const float cu = __shfl_up(a, 8, 64), du = __shfl_up(b, 8, 64);
  const float c = lane < 8 ? t.tr[k] : cu, d = lane < 8 ? t.tl[k] : du;
  return w0 * a + w1 * b + w2 * c + w3 * d;
}

template <int kU>
struct PatchLoads {  // raw load results of one stage-2 step of kU patches (consumers belong to the reduce phase)
  float t[kU], gx[kU], gy[kU];
  TapLoads cur[kU];  // .ab = (x-1,y),(x,y) and .top = (x-1,y-1),(x,y-1) of the lane's own pixel
  int rec[kU];       // LDS record index of the patch, or 64 (the zero record) for the padding of a partial step
};

// Stage-2 addressing (what survived round 2's A/B runs, profiles/r02_notes.md): buffer loads -- plane / patch-buffer
// descriptor in SGPRs, wave-uniform offset in an SGPR, per-lane constant byte offset in a VGPR, no vector address
// arithmetic per load; every lane loads its own two row pairs (x-1,x at y and y-1): no cross-lane traffic, no selects,
// neighbouring lanes' requests hit the same cache lines. The LDS record of a point is
// [w1 w0 w3 w2][cx2..5][cy2..5][cx0 cy1 vis -]: four ds_read_b128 at one address, operands already paired for the packed
// multiply-adds; the window's byte offset is computed per POINT in stage 1 (one v_readlane per patch); the padding of a
// partial pipeline step points at a zero record (an exact zero contribution to b) instead of being branched around.
// A wave issues at most one instruction of any kind per ~4 cycles, so at the coarse levels (frames cached) the
// kernel's pace is its instruction count per patch, not bytes.
// (H is no longer summed here: the setup kernel k_ref8 leaves the level's H partials, from three sums per patch.)
template <bool PN, int kU>
__global__ __launch_bounds__(kBlock) void k_iter8(EngineDev e, LevelCam lc, int level, int cpw) {
  __shared__ __attribute__((aligned(16))) float sRec[kWaves][65 * kRec];  // record 64: zeros
  __shared__ float sW[kWaves][kPartBStride];
  const int b = blockIdx.y;
  const ProbState &st = e.st[b];
  if (!st.active) return;
  const int npts = st.npts;
  const int M = e.M;
  const PlaneSet pl = e.planes[b * e.nlev + level];
  const float *__restrict__ p3 = e.pt3d + (size_t)b * 3 * M;
  const float *__restrict__ T = e.T + (size_t)b * M * 64;
  const float *__restrict__ Gx = e.Gx + (size_t)b * M * 64;
  const float *__restrict__ Gy = e.Gy + (size_t)b * M * 64;
  const float *__restrict__ coefb = e.coef + (size_t)b * M * kCoefStride;
  const int sw = lc.sw;
  // buffer descriptors (base, no stride, max range, raw dword format) of the current frame and the patches
  const __amdgpu_buffer_rsrc_t rcur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pl.cur), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(T), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rGx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Gx), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rGy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Gy), 0, 0x7fffffff, 0x00020000);

  float G[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) G[k] = st.G[k];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *rec = sRec[wave];
  const float4 *rec4 = reinterpret_cast<const float4 *>(rec);
  if (lane < kRec) rec[64 * kRec + lane] = 0.0f;  // (first read behind stage 1's wave barrier)

  f32x2_t acc01 = {0.0f, 0.0f}, acc23 = {0.0f, 0.0f}, acc45 = {0.0f, 0.0f};  // the six J^T r sums as register pairs

  const int nchunks = (npts + cpw - 1) / cpw;
  for (int ch = xcd_band_block(blockIdx.x, gridDim.x) * kWaves + wave; ch < nchunks; ch += gridDim.x * kWaves) {
    const int i0 = ch * cpw;
    const int cnt = min(cpw, npts - i0);
    // ---- stage 1: lane j <-> point i0 + j  (step 7, pose.cpp:384-391; ind_new, odometer.cpp:369-377)
    const bool pv = lane < cnt;
    const int ip = i0 + (pv ? lane : 0);
    const float X = p3[ip], Y = p3[ip + M], Z = p3[ip + 2 * M];
    const float4 *c4 = reinterpret_cast<const float4 *>(coefb + (size_t)ip * kCoefStride);
    const float4 q0 = c4[0], q1 = c4[1], q2 = c4[2];  // cx0..3 | cx4 cx5 cy0 cy1 | cy2..5
    const float tx = G[0] * X + G[1] * Y + G[2] * Z + G[3];
    const float ty = G[4] * X + G[5] * Y + G[6] * Z + G[7];
    const float tz = G[8] * X + G[9] * Y + G[10] * Z + G[11];
    const float mx = (tx / tz) * lc.fx + lc.cx;
    const float my = (ty / tz) * lc.fy + lc.cy;
    const bool vis = pv && in_view(mx, my, lc.swo, lc.sho);
    const Taps tp = make_taps(vis ? mx : 1.0f, vis ? my : 1.0f, 4);  // (1,1): a harmless in-plane window
    const int base_v = tp.row0 * sw + tp.col0;
    const int so_v = (base_v - sw - 1) * 4;  // bytes to the window's top-left texel (tap d of pixel 0)
    {
      float4 *r4 = reinterpret_cast<float4 *>(rec + lane * kRec);
      r4[0] = make_float4(tp.w1, tp.w0, tp.w3, tp.w2);
      r4[1] = make_float4(q0.z, q0.w, q1.x, q1.y);  // cx2 cx3 cx4 cx5
      r4[2] = make_float4(q2.x, q2.y, q2.z, q2.w);  // cy2 cy3 cy4 cy5
      r4[3] = make_float4(q0.x, q1.w, vis ? 1.0f : 0.0f, 0.0f);  // cx0 cy1 vis -
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- stage 2. The kU patches of a step are nsteps apart (neighbouring points share frame cache lines).
    const int nsteps = (cnt + kU - 1) / kU;
    auto issue = [&](PatchLoads<kU> &L, int sidx) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int jraw = sidx + u * nsteps;
        const int jj = min(jraw, cnt - 1);
        constexpr int aux = 2;  // gfx950 cache policy bits of a buffer load: bit 1 = nt (streamed once per launch:
                                // keep them from evicting the re-used frame lines)
        L.rec[u] = (jraw < cnt) ? jj : 64;  // padding: the zero record; its loads repeat the chunk's last patch
        const int so = rlane(so_v, jj);
        const int po4 = (i0 + jj) * 256;  // bytes: one 8x8 float patch = 256 B
        const int off_cd = ((lane >> 3) * sw + (lane & 7)) * 4, off_ab = off_cd + sw * 4;
        L.t[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rT, lane * 4, po4, aux));
        L.gx[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rGx, lane * 4, po4, aux));
        L.gy[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rGy, lane * 4, po4, aux));
        L.cur[u].ab = __builtin_bit_cast(f32x2_a4, __builtin_amdgcn_raw_buffer_load_b64(rcur, off_ab, so, 0));
        L.cur[u].top = __builtin_bit_cast(f32x2_a4, __builtin_amdgcn_raw_buffer_load_b64(rcur, off_cd, so, 0));
      }
    };
    auto reduce = [&](const PatchLoads<kU> &L) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        // (compiler barrier tied to the accumulators: keeps the four record reads of patch u behind the sums of patch
        // u - 1 -- hoisted together they cost 64 registers and two waves of occupancy)
        asm volatile("" : "+v"(acc01), "+v"(acc23), "+v"(acc45) : : "memory");
        const float4 *r4 = rec4 + L.rec[u] * 4;
        const float4 wv = r4[0], qx = r4[1], qy = r4[2], qz = r4[3];
        // utilities.cpp:107 in the reference's operand order, never contracted: ((w0 a + w1 b) + w2 c) + w3 d -- template
        // and current patch must round identically so that identical frames give a residual of exactly zero
        float inew = wv.y * L.cur[u].ab.y + wv.x * L.cur[u].ab.x + wv.w * L.cur[u].top.y + wv.z * L.cur[u].top.x;
        if constexpr (PN) inew -= wave_sum(inew) / 64.0f;  // utilities.cpp:111-112
        const float gx = L.gx[u], gy = L.gy[u];
        const float r = (L.t[u] - inew) * qz.z;  // pdiff (odometer.cpp:381); 0 out of view and for padding
        // the J^T r sums are compared to tolerance only: explicit multiply-adds
        // (as register PAIRS: v_pk_fma_f32 does two of the six sums per instruction)
        const f32x2_t g2 = {gx * r, gy * r}, gr2 = {g2.x, g2.x}, hr2 = {g2.y, g2.y};
        acc01 = __builtin_elementwise_fma(g2, (f32x2_t){qz.x, qz.y}, acc01);  // sd1 = Gx cx0, sd2 = Gy cy1
        acc23 = __builtin_elementwise_fma(gr2, (f32x2_t){qx.x, qx.y},         // sd3..sd6 = Gx cxk + Gy cyk
                                          __builtin_elementwise_fma(hr2, (f32x2_t){qy.x, qy.y}, acc23));
        acc45 = __builtin_elementwise_fma(gr2, (f32x2_t){qx.z, qx.w},         // (odometer.cpp:319-326)
                                          __builtin_elementwise_fma(hr2, (f32x2_t){qy.z, qy.w}, acc45));
      }
    };
    PatchLoads<kU> A, B;
    issue(A, 0);
    for (int sidx = 0; sidx < nsteps; sidx += 2) {
      if (sidx + 1 < nsteps) issue(B, sidx + 1);
      reduce(A);
      if (sidx + 2 < nsteps) issue(A, sidx + 2);
      if (sidx + 1 < nsteps) reduce(B);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the records are rewritten by the next chunk
  }

  const float acc[6] = {acc01.x, acc01.y, acc23.x, acc23.y, acc45.x, acc45.y};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float v = wave_sum(acc[k]);
    if (lane == 0) sW[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const float v = (sW[0][threadIdx.x] + sW[1][threadIdx.x]) + (sW[2][threadIdx.x] + sW[3][threadIdx.x]);
    e.partb[((size_t)b * gridDim.x + blockIdx.x) * kPartBStride + threadIdx.x] = v;
  }
}

// steps 4-6 for 8x8 patches, same two-stage, software-pipelined form: reference patches + gradients (bit-exact:
// un-contracted blends), sd coefficients, and the level's H partials in the form SURVEY.md §8 a18 prescribes. The
// steepest-descent images are sd_k = Gx cx_k + Gy cy_k with per-POINT constants cx, cy (odometer.cpp:313-326), so
//   H_jk = sum_pixels sd_j sd_k (odometer.cpp:428-455) = sum_points [cx_j cx_k Sxx + (cx_j cy_k + cy_j cx_k) Sxy + cy_j cy_k Syy]
// with three sums per patch, Sxx = sum Gx^2, Sxy = sum Gx Gy, Syy = sum Gy^2 (wave reductions while the kernel waits on
// memory anyway), instead of 21 multiply-adds per pixel in a separate instantiation of the iteration kernel. A point out
// of the reference view keeps its stale patch and coefficients (odometer.cpp:304): its S is summed from the stored
// gradients, so nothing but the patch buffers and the coefficient line carries state from level to level.
// H is compared to tolerance only (summation order differs from the CPU path's whole-buffer sums anyway).
// Gradients on the fly (OTF): a pyramid the builder made has dx = I(x+1) - I(x-1), dy = I(y+1) - I(y-1) inside the image,
// 0 on its border columns / rows (reflect-101) and in the padding (utilities.cpp:30-45), so the blended gradient patches
// can be formed from the IMAGE plane alone: the same f32 subtraction the builder did, the same blend -- the same bits --
// from a 4 x 4 neighbourhood per lane (12 texels: two 16-byte and two 8-byte loads) instead of the three planes' taps
// (or the packed 16-byte texels, a third of which is padding). The reference frame is then read at 4 B per pixel, and a
// pyramid built for this path (getgrad = 2) holds nothing but the image levels.
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef const f32x4_a4 __attribute__((address_space(1))) *gconst_f32x4a;
struct TapLoadsOTF {
  f32x4_a4 r0, r1;   // rows y and y-1: columns x-2 .. x+1   (b = r0[1], a = r0[2], d = r1[1], c = r1[2])
  f32x2_a4 up, dn;   // rows y+1 and y-2: columns x-1, x
};
__device__ __forceinline__ TapLoadsOTF taps_issue_otf(gconst_f32 img_at_base, int loff, int sw) {
  TapLoadsOTF t;
  gconst_f32 q = img_at_base + loff;
  t.r0 = *reinterpret_cast<gconst_f32x4a>(q - 2);
  t.r1 = *reinterpret_cast<gconst_f32x4a>(q - sw - 2);
  t.up = *reinterpret_cast<gconst_f32x2>(q + sw - 1);
  t.dn = *reinterpret_cast<gconst_f32x2>(q - 2 * sw - 1);
  return t;
}
// ix, iy: unpadded image coordinates of tap a of this lane's pixel; w, h: the level's size. interior: every tap of
// every lane of the patch is at least one pixel inside (wave-uniform fast path: no predicates)
__device__ __forceinline__ void taps_blend_otf(const TapLoadsOTF &t, float w0, float w1, float w2, float w3, bool interior,
                                               int ix, int iy, int w, int h, float &tv, float &gx, float &gy) {
  const float a = t.r0[2], b = t.r0[1], c = t.r1[2], d = t.r1[1];
  tv = w0 * a + w1 * b + w2 * c + w3 * d;
  float xa = t.r0[3] - t.r0[1], xb = t.r0[2] - t.r0[0], xc = t.r1[3] - t.r1[1], xd = t.r1[2] - t.r1[0];
  float ya = t.up.y - t.r1[2], yb = t.up.x - t.r1[1], yc = t.r0[2] - t.dn.y, yd = t.r0[1] - t.dn.x;
  if (!interior) {  // the builder's border rules: dx is 0 outside 1 <= X <= w-2 (and outside the image rows), dy alike
    const bool colA = ix >= 0 && ix < w, colB = ix - 1 >= 0 && ix - 1 < w, rowA = iy >= 0 && iy < h, rowC = iy - 1 >= 0 && iy - 1 < h;
    const bool dxA = ix >= 1 && ix <= w - 2, dxB = ix - 1 >= 1 && ix - 1 <= w - 2;
    const bool dyA = iy >= 1 && iy <= h - 2, dyC = iy - 1 >= 1 && iy - 1 <= h - 2;
    xa = (dxA && rowA) ? xa : 0.0f;
    xb = (dxB && rowA) ? xb : 0.0f;
    xc = (dxA && rowC) ? xc : 0.0f;
    xd = (dxB && rowC) ? xd : 0.0f;
    ya = (dyA && colA) ? ya : 0.0f;
    yb = (dyA && colB) ? yb : 0.0f;
    yc = (dyC && colA) ? yc : 0.0f;
    yd = (dyC && colB) ? yd : 0.0f;
  }
  gx = w0 * xa + w1 * xb + w2 * xc + w3 * xd;
  gy = w0 * ya + w1 * yb + w2 * yc + w3 * yd;
}

template <int I, int N, class Fn>
__device__ __forceinline__ void res_static_for(Fn &&fn) {  // fn(integral_constant<I>) ... fn(integral_constant<N - 1>)
  if constexpr (I < N) {
    fn(std::integral_constant<int, I>{});
    res_static_for<I + 1, N>(fn);
  }
}

template <int kU>
struct RefLoads {
  TapLoads r[kU], x[kU], y[kU];
  TapLoads4 p4[kU];
  TapLoadsOTF o[kU];
  int rec[kU];
  int vis[kU];
  int inner[kU];  // OTF: the patch's taps are all at least one pixel inside the image (wave-uniform)
  int tx[kU], ty[kU];  // OTF: unpadded coordinates of tap a of pixel (0, 0)
};

// ST (with OTF, without PN): the chunk's patches in statically unrolled groups of SIXTEEN -- three windows in flight,
// and the three sums of S per patch by the transposing wave reduction of the resident kernel (TrAcc, ictr_devfn.h: ~10
// instructions per patch instead of the 42 of three wave sums + selects); everything else as in the dynamic form.
template <bool PN, int kU, bool PK, bool OTF = false, bool ST = false>  // PK: packed {img, dx, dy, 0} planes; OTF: image plane only
__global__ __launch_bounds__(kBlock) void k_ref8(EngineDev e, LevelCam lc, int level, int cpw) {
  static_assert(!ST || (OTF && !PN), "the static form exists for on-the-fly gradients without patch normalisation");
  __shared__ __attribute__((aligned(16))) float sRec[kWaves][64 * kRec];
  __shared__ float sW[kWaves][kPartHStride];
  const int b = blockIdx.y;
  const ProbState &st = e.st[b];
  const int npts = st.npts;
  const int M = e.M;
  const PlaneSet pl = e.planes[b * e.nlev + level];
  const float *__restrict__ pt2d = e.pt2d + ((size_t)b * e.nlev + level) * 2 * M;
  const float *__restrict__ p3r = e.pt3d_ref + (size_t)b * 3 * M;
  float *T = e.T + (size_t)b * M * 64;
  float *Gx = e.Gx + (size_t)b * M * 64;
  float *Gy = e.Gy + (size_t)b * M * 64;
  float *coefb = e.coef + (size_t)b * M * kCoefStride;
  gconst_f32 pref = (gconst_f32)pl.ref, pdx = (gconst_f32)pl.dx, pdy = (gconst_f32)pl.dy;
  gconst_f32x4 ppack = (gconst_f32x4)pl.pack;
  const int sw = lc.sw;
  const int wl = (int)lc.swo, hl = (int)lc.sho, padl = (sw - wl) / 2;  // (OTF) the level's unpadded size, its padding

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int loff = (lane >> 3) * sw + (lane & 7);
  float *rec = sRec[wave];
  const float4 *rec4 = reinterpret_cast<const float4 *>(rec);

  float accH = 0.0f;  // lane j < 21: the wave's sum of H entry j (upper triangle, row-major)

  const int nchunks = (npts + cpw - 1) / cpw;
  for (int ch = xcd_band_block(blockIdx.x, gridDim.x) * kWaves + wave; ch < nchunks; ch += gridDim.x * kWaves) {
    const int i0 = ch * cpw;
    const int cnt = min(cpw, npts - i0);
    // ---- stage 1: one point per lane: visibility (odometer.cpp:273-282), sd coefficients (:313-326)
    const bool pv = lane < cnt;
    const int ip = i0 + (pv ? lane : 0);
    const float mx = pt2d[ip], my = pt2d[ip + M];
    const bool vis = pv && in_view(mx, my, lc.swo, lc.sho);
    const Taps tp = make_taps(vis ? mx : 1.0f, vis ? my : 1.0f, 4);
#ifndef ICTR_REF8_TOUCH
#define ICTR_REF8_TOUCH 6  // line touches per lane and chunk (0: none)
#endif
    // (ST) Touch every cache line of the chunk's windows ONCE before the taps ask for them. The taps of a window are four
    // loads over the same eleven rows, and neighbouring patches share lines: with a frame that is not in the L2 each of
    // them finds its line "miss pending" and the L1's tag pipeline stalls until the line arrives -- for every wave of
    // the CU, the stores included (TCP_PENDING_STALL_CYCLES: 278 us of a 448 us launch; 32 distinct 1080p pairs 448 /
    // 325 / 275 us per level, with the touches 342 / 282 / 255; profiles/r03_notes.md 10). Lanes 16 g .. 16 g + 15
    // own the chunk's group g: lane i takes row i of the group's bounding box, touch k the box's k-th line of that row,
    // so no two lanes of one load and no two loads of a group ask for the same line. Issued before the coefficients
    // are formed, waited for before the first window. Boxes of scattered points (more than 16 rows or kTouch lines
    // per row) are not touched.
    constexpr int kTouch = ST ? ICTR_REF8_TOUCH : 0;
    float touch[kTouch > 0 ? kTouch : 1];
    if constexpr (kTouch > 0) {
      const int kBig = 1 << 28;
      const int rmin = row_min_dpp(vis ? tp.row0 : kBig), rmax = -row_min_dpp(vis ? -tp.row0 : kBig);
      const int cmin = row_min_dpp(vis ? tp.col0 : kBig), cmax = -row_min_dpp(vis ? -tp.col0 : kBig);
      const int li = lane & 15, nrows = rmax - rmin + 11;
      const unsigned long long a_first = (unsigned long long)(pl.ref + (size_t)(rmin - 2 + li) * sw + (cmin - 2));
      const unsigned long long a_last = (unsigned long long)(pl.ref + (size_t)(rmin - 2 + li) * sw + (cmax + 8)) + 3;
      const unsigned long long a0 = a_first & ~127ull;
      const int nl = (int)(((a_last | 127ull) - a0 + 1) >> 7);
      const bool box = rmin < kBig && nrows <= 16 && li < nrows && nl <= kTouch;
#pragma unroll
      for (int k = 0; k < kTouch; ++k) {
        touch[k] = 0.0f;
        if (box && k < nl) touch[k] = *(gconst_f32)(a0 + 128ull * k);
      }
    }
    float cx[6], cy[6];
    float4 *c4 = reinterpret_cast<float4 *>(coefb + (size_t)ip * kCoefStride);
    if (vis) {
      sd_coefs(p3r[ip], p3r[ip + M], p3r[ip + 2 * M], lc.fx, lc.fy, cx, cy);
      c4[0] = make_float4(cx[0], cx[1], cx[2], cx[3]);
      c4[1] = make_float4(cx[4], cx[5], cy[0], cy[1]);
      c4[2] = make_float4(cy[2], cy[3], cy[4], cy[5]);
    } else {  // stale coefficients stay in force (odometer.cpp:304); zeros if the point was never seen
      const float4 a0 = c4[0], a1 = c4[1], a2 = c4[2];
      cx[0] = a0.x; cx[1] = a0.y; cx[2] = a0.z; cx[3] = a0.w; cx[4] = a1.x; cx[5] = a1.y;
      cy[0] = a1.z; cy[1] = a1.w; cy[2] = a2.x; cy[3] = a2.y; cy[4] = a2.z; cy[5] = a2.w;
    }
    const int base_v = tp.row0 * sw + tp.col0;
    const int vis_v = vis ? 1 : 0;
    const int tx_v = tp.col0 - padl, ty_v = tp.row0 - padl;  // (OTF) tap a of pixel (0,0) in unpadded image coordinates
    {
      float4 *r4 = reinterpret_cast<float4 *>(rec + lane * kRec);
      r4[0] = make_float4(tp.w0, tp.w1, tp.w2, tp.w3);
      r4[1] = make_float4(cx[0], cx[1], cx[2], cx[3]);
      r4[2] = make_float4(cx[4], cx[5], cy[0], cy[1]);
      r4[3] = make_float4(cy[2], cy[3], cy[4], cy[5]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- stage 2 (utilities.cpp:115-189); patches of a step are nsteps apart
    float sxx_v = 0.0f, sxy_v = 0.0f, syy_v = 0.0f;  // lane j: S of the chunk's point j (0 beyond the chunk)
    const int nsteps = (cnt + kU - 1) / kU;
    auto issue = [&](RefLoads<kU> &L, int sidx) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int jraw = sidx + u * nsteps;
        const int jj = min(jraw, cnt - 1);
        L.rec[u] = (jraw < cnt) ? jj : -1;
        const int base = rlane(base_v, jj);
        L.vis[u] = rlane(vis_v, jj);
        if (L.vis[u]) {  // wave-uniform
          if constexpr (OTF) {
            L.tx[u] = rlane(tx_v, jj);
            L.ty[u] = rlane(ty_v, jj);
            L.inner[u] = (L.tx[u] >= 2 && L.tx[u] + 7 <= wl - 2 && L.ty[u] >= 2 && L.ty[u] + 7 <= hl - 2) ? 1 : 0;
            L.o[u] = taps_issue_otf(pref + base, loff, sw);
          } else if constexpr (PK) {
            L.p4[u] = taps_issue4(ppack + base, loff, sw, lane);
          } else {
            L.r[u] = taps_issue(pref + base, loff, sw, lane);
            L.x[u] = taps_issue(pdx + base, loff, sw, lane);
            L.y[u] = taps_issue(pdy + base, loff, sw, lane);
          }
        }
      }
    };
    auto patch_sums = [&](float gx, float gy, int j) {  // the patch's three sums into lane j (the sums are wave-uniform)
      const float sxx = wave_sum_dpp(gx * gx), sxy = wave_sum_dpp(gx * gy), syy = wave_sum_dpp(gy * gy);
      const bool mine = lane == j;
      sxx_v = mine ? sxx : sxx_v;
      sxy_v = mine ? sxy : sxy_v;
      syy_v = mine ? syy : syy_v;
    };
    auto reduce = [&](const RefLoads<kU> &L) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (L.rec[u] < 0 || !L.vis[u]) continue;  // wave-uniform
        float gx, gy;
        {
          const float4 w = rec4[L.rec[u] * 4 + 0];
          float t;
          if constexpr (OTF) {
            taps_blend_otf(L.o[u], w.x, w.y, w.z, w.w, L.inner[u] != 0, L.tx[u] + (lane & 7), L.ty[u] + (lane >> 3), wl, hl,
                           t, gx, gy);
          } else if constexpr (PK) {
            t = taps_blend4(L.p4[u], 0, w.x, w.y, w.z, w.w, lane);
            gx = taps_blend4(L.p4[u], 1, w.x, w.y, w.z, w.w, lane);
            gy = taps_blend4(L.p4[u], 2, w.x, w.y, w.z, w.w, lane);
          } else {
            t = taps_blend(L.r[u], w.x, w.y, w.z, w.w, lane);
            gx = taps_blend(L.x[u], w.x, w.y, w.z, w.w, lane);
            gy = taps_blend(L.y[u], w.x, w.y, w.z, w.w, lane);
          }
          if constexpr (PN) t -= wave_sum(t) / 64.0f;  // utilities.cpp:187-188
          const size_t po = (size_t)(i0 + L.rec[u]) * 64;
          // 400 MB written once per level: do not let them push the pyramid planes out of L2
          __builtin_nontemporal_store(t, T + po + lane);
          __builtin_nontemporal_store(gx, Gx + po + lane);
          __builtin_nontemporal_store(gy, Gy + po + lane);
        }
        patch_sums(gx, gy, L.rec[u]);
      }
    };
    if constexpr (ST) {
      const int inner_v = (tx_v >= 2 && tx_v + 7 <= wl - 2 && ty_v >= 2 && ty_v + 7 <= hl - 2) ? 1 : 0;
      // the wave waits for its touched lines here (the other waves of the CU carry on) rather than let its taps find
      // them pending
      if constexpr (kTouch > 0) {
#pragma unroll
        for (int k = 0; k < kTouch; ++k) asm volatile("" ::"v"(touch[k]));
      }
      for (int g0 = 0; g0 < cnt; g0 += 16) {  // (wave-uniform trip count: cnt is)
        const unsigned vmask = (unsigned)(__builtin_amdgcn_ballot_w64(vis) >> g0) & 0xffffu;  // the group's visible points
#ifndef ICTR_REF8_D
#define ICTR_REF8_D 3  // windows in flight per wave (12 registers each)
#endif
        constexpr int kD = ICTR_REF8_D;
        TapLoadsOTF W3[kD];
        auto issue_s = [&](int j) {
#if defined(ICTR_REF8_ABL) && (ICTR_REF8_ABL & 2)  // measurement build: every window = the plane's first (L1 hits)
          if ((vmask >> j) & 1u) W3[j % kD] = taps_issue_otf(pref + 2 * sw + 2 + (rlane(base_v, g0 + j) & 7), loff, sw);
#else
          if ((vmask >> j) & 1u) W3[j % kD] = taps_issue_otf(pref + rlane(base_v, g0 + j), loff, sw);
#endif
        };
        res_static_for<0, kD>([&](auto jc) { issue_s(decltype(jc)::value); });
        TrAcc<16> accS, accX;
        tr_for_each_patch<16, 0>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          float gx = 0.0f, gy = 0.0f;
          if ((vmask >> j) & 1u) {  // wave-uniform
            const float4 w = rec4[(g0 + j) * 4 + 0];
            float t;
            taps_blend_otf(W3[j % kD], w.x, w.y, w.z, w.w, rlane(inner_v, g0 + j) != 0, rlane(tx_v, g0 + j) + (lane & 7),
                           rlane(ty_v, g0 + j) + (lane >> 3), wl, hl, t, gx, gy);
            const size_t po = (size_t)(i0 + g0 + j) * 64;
#if defined(ICTR_REF8_ABL) && (ICTR_REF8_ABL & 1)   // measurement build: no patch stores (the sums keep the blends alive)
            gx += t * 1e-30f;
#else
            __builtin_nontemporal_store(t, T + po + lane);
            __builtin_nontemporal_store(gx, Gx + po + lane);
            __builtin_nontemporal_store(gy, Gy + po + lane);
#endif
          }
          if constexpr (j + kD < 16) issue_s(j + kD);
          accS.template push<j>(gx * gx, gy * gy, lane);
          accX.template push<j>(gx * gy, 0.0f, lane);
        });
        // lane = point again: the group's sixteen points are the lanes g0 .. g0 + 15
        const int pp = lane & 15;
        const float sxx = lane_gather(accS.F, tr_lane_of<16>(pp, 0)), syy = lane_gather(accS.F, tr_lane_of<16>(pp, 1));
        const float sxy = lane_gather(accX.F, tr_lane_of<16>(pp, 0));
        const bool mine = (lane >> 4) == (g0 >> 4);
        sxx_v = mine ? sxx : sxx_v;
        sxy_v = mine ? sxy : sxy_v;
        syy_v = mine ? syy : syy_v;
      }
    } else {
    RefLoads<kU> A, B;
    issue(A, 0);
    for (int sidx = 0; sidx < nsteps; sidx += 2) {
      if (sidx + 1 < nsteps) issue(B, sidx + 1);
      reduce(A);
      if (sidx + 2 < nsteps) issue(A, sidx + 2);
      if (sidx + 1 < nsteps) reduce(B);
    }
    }
    // points out of the reference view at this level (rare): S of the stale patch, from the stored gradients
    for (unsigned long long stale = __builtin_amdgcn_ballot_w64(pv && !vis); stale; stale &= stale - 1) {
      const int j = __builtin_ctzll(stale);
      const size_t po = (size_t)(i0 + j) * 64;
      patch_sums((Gx + po)[lane], (Gy + po)[lane], j);
    }
    // ---- H += J^T S J, one point per lane again (its coefficients come back from the wave's records)
    {
      const float4 k0 = rec4[lane * 4 + 1], k1 = rec4[lane * 4 + 2], k2 = rec4[lane * 4 + 3];
      const float ax[6] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y}, ay[6] = {k1.z, k1.w, k2.x, k2.y, k2.z, k2.w};
      float uj[6], vj[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        uj[j] = __builtin_fmaf(ax[j], sxx_v, ay[j] * sxy_v);
        vj[j] = __builtin_fmaf(ax[j], sxy_v, ay[j] * syy_v);
      }
      int jk = 0;
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int k = j; k < 6; ++k, ++jk) {
          const float h = wave_sum_dpp(__builtin_fmaf(ax[k], uj[j], ay[k] * vj[j]));
          accH += lane == jk ? h : 0.0f;
          if (jk % 3 == 2) asm volatile("" : "+v"(accH));  // three reductions at a time (their row sums sit in SGPRs)
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  if (lane < kPartHStride) sW[wave][lane] = lane < kHUnique ? accH : 0.0f;
  __syncthreads();
  if (threadIdx.x < kHUnique) {
    const float v = (sW[0][threadIdx.x] + sW[1][threadIdx.x]) + (sW[2][threadIdx.x] + sW[3][threadIdx.x]);
    e.partH[((size_t)b * gridDim.x + blockIdx.x) * kPartHStride + threadIdx.x] = v;
  }
}

// ================================================================ P = 4 fast path (wave64 == four 4x4 patches)
// The reference's other standard patch size (run_odometer_test.m:140: "4 0 4 5 0.01 0 0"). Same two-stage,
// software-pipelined form as the 8x8 kernels; a wave-step handles FOUR consecutive patches, one per 16-lane group
// (lane = 16 * sub + 4 * row + col): the four patches' T/Gx/Gy rows are one contiguous 256-byte row, the window
// base and the LDS record index are per lane (gathered with ds_bpermute instead of v_readlane), validity and
// visibility are per-lane predicates instead of wave-uniform branches. Requires the packed reference planes.
struct TapLoads4x4 {  // one patch group per 16 lanes; a/b of the own row, top row for the lanes of patch row 0
  f32x2_a4 ab, top;
};
__device__ __forceinline__ TapLoads4x4 taps4_issue(gconst_f32 plane, int idx, int sw, bool ok, bool toprow) {
  TapLoads4x4 t;
  f32x2_a4 z = {0.0f, 0.0f};
  t.ab = z;
  t.top = z;
  if (ok) {
    t.ab = *reinterpret_cast<gconst_f32x2>(plane + (idx - 1));
    if (toprow) t.top = *reinterpret_cast<gconst_f32x2>(plane + (idx - sw - 1));
  }
  return t;
}
__device__ __forceinline__ float taps4_blend(const TapLoads4x4 &t, float w0, float w1, float w2, float w3, bool toprow) {
  const float a = t.ab.y, b = t.ab.x;
  const float cu = __shfl_up(a, 4, 64), du = __shfl_up(b, 4, 64);  // the lane one patch row up (same 16-lane group)
  const float c = toprow ? t.top.y : cu, d = toprow ? t.top.x : du;
  return w0 * a + w1 * b + w2 * c + w3 * d;
}
struct TapLoadsP4 {
  f32x4_t l, r, tl, tr;
};
__device__ __forceinline__ float tapsP4_blend(const TapLoadsP4 &t, int k, float w0, float w1, float w2, float w3, bool toprow) {
  const float a = t.r[k], b = t.l[k];
  const float cu = __shfl_up(a, 4, 64), du = __shfl_up(b, 4, 64);
  const float c = toprow ? t.tr[k] : cu, d = toprow ? t.tl[k] : du;
  return w0 * a + w1 * b + w2 * c + w3 * d;
}

template <int kU>
struct Iter4Loads {
  float t[kU], gx[kU], gy[kU];
  TapLoads4x4 cur[kU];
  int jj[kU];  // per lane: index of its patch inside the chunk, or -1
};

template <bool PN, bool WH, int kU = 2>
__global__ __launch_bounds__(kBlock) void k_iter4(EngineDev e, LevelCam lc, int level, int cpw) {
  __shared__ __attribute__((aligned(16))) float sRec[kWaves][64 * kRec];
  __shared__ float sW[kWaves][kPartBStride];
  __shared__ float sWH[WH ? kWaves : 1][kPartHStride];
  const int b = blockIdx.y;
  const ProbState &st = e.st[b];
  if (!st.active) return;
  const int npts = st.npts;
  const int M = e.M;
  const PlaneSet pl = e.planes[b * e.nlev + level];
  const float *__restrict__ p3 = e.pt3d + (size_t)b * 3 * M;
  const float *__restrict__ T = e.T + (size_t)b * M * 16;
  const float *__restrict__ Gx = e.Gx + (size_t)b * M * 16;
  const float *__restrict__ Gy = e.Gy + (size_t)b * M * 16;
  const float *__restrict__ coefb = e.coef + (size_t)b * M * kCoefStride;
  gconst_f32 cur = (gconst_f32)pl.cur;
  const int sw = lc.sw;

  float G[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) G[k] = st.G[k];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane >> 4, q = lane & 15;
  const bool toprow = q < 4;
  const int loff = (q >> 2) * sw + (q & 3);
  float *rec = sRec[wave];
  const float4 *rec4 = reinterpret_cast<const float4 *>(rec);

  float acc[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) acc[k] = 0.0f;
  float accH[WH ? kHUnique : 1];
#pragma unroll
  for (int j = 0; j < (WH ? kHUnique : 1); ++j) accH[j] = 0.0f;

  const int nchunks = (npts + cpw - 1) / cpw;
  for (int ch = xcd_band_block(blockIdx.x, gridDim.x) * kWaves + wave; ch < nchunks; ch += gridDim.x * kWaves) {
    const int i0 = ch * cpw;
    const int cnt = min(cpw, npts - i0);
    // ---- stage 1: lane j <-> point i0 + j (as k_iter8, half-size 2)
    const bool pv = lane < cnt;
    const int ip = i0 + (pv ? lane : 0);
    const float X = p3[ip], Y = p3[ip + M], Z = p3[ip + 2 * M];
    const float4 *c4 = reinterpret_cast<const float4 *>(coefb + (size_t)ip * kCoefStride);
    const float4 q0 = c4[0], q1 = c4[1], q2 = c4[2];
    const float tx = G[0] * X + G[1] * Y + G[2] * Z + G[3];
    const float ty = G[4] * X + G[5] * Y + G[6] * Z + G[7];
    const float tz = G[8] * X + G[9] * Y + G[10] * Z + G[11];
    const float mx = (tx / tz) * lc.fx + lc.cx;
    const float my = (ty / tz) * lc.fy + lc.cy;
    const bool vis = pv && in_view(mx, my, lc.swo, lc.sho);
    const Taps tp = make_taps(vis ? mx : 1.0f, vis ? my : 1.0f, 2);
    const int base_v = tp.row0 * sw + tp.col0;
    {
      float4 *r4 = reinterpret_cast<float4 *>(rec + lane * kRec);
      r4[0] = make_float4(tp.w0, tp.w1, tp.w2, tp.w3);
      r4[1] = make_float4(q0.x, q0.z, q0.w, q1.x);
      r4[2] = make_float4(q1.y, q1.w, q2.x, q2.y);
      r4[3] = make_float4(q2.z, q2.w, vis ? 1.0f : 0.0f, 0.0f);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- stage 2: wave-steps of four patches; kU wave-steps per pipeline step, nsteps apart
    const int nws = (cnt + 3) >> 2;
    const int nsteps = (nws + kU - 1) / kU;
    auto issue = [&](Iter4Loads<kU> &L, int sidx) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int ws = sidx + u * nsteps;
        const int j = 4 * ws + sub;
        const bool ok = (ws < nws) & (j < cnt);
        L.jj[u] = ok ? j : -1;
        const int base = __shfl(base_v, ok ? j : 0, 64);
        const size_t po = (size_t)(i0 + 4 * min(ws, nws - 1)) * 16;  // wave-uniform: 4 patches = one 256-byte row
        L.t[u] = L.gx[u] = L.gy[u] = 0.0f;
        if (ok) {
          L.t[u] = __builtin_nontemporal_load(T + po + lane);
          L.gx[u] = __builtin_nontemporal_load(Gx + po + lane);
          L.gy[u] = __builtin_nontemporal_load(Gy + po + lane);
        }
        L.cur[u] = taps4_issue(cur, base + loff, sw, ok, toprow);
      }
    };
    auto reduce = [&](const Iter4Loads<kU> &L) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const bool ok = L.jj[u] >= 0;
        const int ri = (ok ? L.jj[u] : 0) * 4;
        const float4 w = rec4[ri + 0], k0 = rec4[ri + 1], k1 = rec4[ri + 2], k2 = rec4[ri + 3];
        float inew = taps4_blend(L.cur[u], w.x, w.y, w.z, w.w, toprow);
        if constexpr (PN) inew -= group_sum(inew, 16) / 16.0f;  // utilities.cpp:111-112
        const float r = ok ? (L.t[u] - inew) * k2.z : 0.0f;  // pdiff (odometer.cpp:381); k2.z = 0 out of the new view
        const float gx = L.gx[u], gy = L.gy[u];                // zero for lanes without a patch
        {
#pragma clang fp contract(fast)
          acc[0] += (gx * k0.x) * r;
          acc[1] += (gy * k1.y) * r;
          acc[2] += (gx * k0.y + gy * k1.z) * r;
          acc[3] += (gx * k0.z + gy * k1.w) * r;
          acc[4] += (gx * k0.w + gy * k2.x) * r;
          acc[5] += (gx * k1.x + gy * k2.y) * r;
          if constexpr (WH) {
            float sd[6];
            sd[0] = gx * k0.x;
            sd[1] = gy * k1.y;
            sd[2] = gx * k0.y + gy * k1.z;
            sd[3] = gx * k0.z + gy * k1.w;
            sd[4] = gx * k0.w + gy * k2.x;
            sd[5] = gx * k1.x + gy * k2.y;
            int jk = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
              for (int c = a; c < 6; ++c) accH[jk++] += sd[a] * sd[c];
          }
        }
      }
    };
    Iter4Loads<kU> A, B;
    issue(A, 0);
    for (int sidx = 0; sidx < nsteps; sidx += 2) {
      if (sidx + 1 < nsteps) issue(B, sidx + 1);
      reduce(A);
      if (sidx + 2 < nsteps) issue(A, sidx + 2);
      if (sidx + 1 < nsteps) reduce(B);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float v = wave_sum(acc[k]);
    if (lane == 0) sW[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6)
    e.partb[((size_t)b * gridDim.x + blockIdx.x) * kPartBStride + threadIdx.x] =
        (sW[0][threadIdx.x] + sW[1][threadIdx.x]) + (sW[2][threadIdx.x] + sW[3][threadIdx.x]);
  if constexpr (WH) {
#pragma unroll
    for (int j = 0; j < kHUnique; ++j) {
      const float v = wave_sum(accH[j]);
      if (lane == 0) sWH[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < kHUnique)
      e.partH[((size_t)b * gridDim.x + blockIdx.x) * kPartHStride + threadIdx.x] =
          (sWH[0][threadIdx.x] + sWH[1][threadIdx.x]) + (sWH[2][threadIdx.x] + sWH[3][threadIdx.x]);
  }
}

template <int kU>
struct Ref4Loads {
  TapLoadsP4 p4[kU];
  int jj[kU];   // per lane: patch index inside the chunk, -1 = none
  int vis[kU];  // per lane: that patch is inside the reference view at this level
};

// steps 4-5 for 4x4 patches (H deferred to the first k_iter4<.., WH = true> launch; packed reference planes)
template <bool PN>
__global__ __launch_bounds__(kBlock) void k_ref4(EngineDev e, LevelCam lc, int level, int cpw) {
  constexpr int kU = 2;
  const int b = blockIdx.y;
  const ProbState &st = e.st[b];
  const int npts = st.npts;
  const int M = e.M;
  const PlaneSet pl = e.planes[b * e.nlev + level];
  const float *__restrict__ pt2d = e.pt2d + ((size_t)b * e.nlev + level) * 2 * M;
  const float *__restrict__ p3r = e.pt3d_ref + (size_t)b * 3 * M;
  float *T = e.T + (size_t)b * M * 16;
  float *Gx = e.Gx + (size_t)b * M * 16;
  float *Gy = e.Gy + (size_t)b * M * 16;
  float *coefb = e.coef + (size_t)b * M * kCoefStride;
  gconst_f32x4 ppack = (gconst_f32x4)pl.pack;
  const int sw = lc.sw;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane >> 4, q = lane & 15;
  const bool toprow = q < 4;
  const int loff = (q >> 2) * sw + (q & 3);

  const int nchunks = (npts + cpw - 1) / cpw;
  for (int ch = xcd_band_block(blockIdx.x, gridDim.x) * kWaves + wave; ch < nchunks; ch += gridDim.x * kWaves) {
    const int i0 = ch * cpw;
    const int cnt = min(cpw, npts - i0);
    // ---- stage 1: one point per lane: visibility (odometer.cpp:273-282), sd coefficients (:313-326)
    const bool pv = lane < cnt;
    const int ip = i0 + (pv ? lane : 0);
    const float mx = pt2d[ip], my = pt2d[ip + M];
    const bool vis = pv && in_view(mx, my, lc.swo, lc.sho);
    if (vis) {  // invisible points keep their stale coefficients (odometer.cpp:304): nothing to do for them
      float cx[6], cy[6];
      sd_coefs(p3r[ip], p3r[ip + M], p3r[ip + 2 * M], lc.fx, lc.fy, cx, cy);
      float4 *c4 = reinterpret_cast<float4 *>(coefb + (size_t)ip * kCoefStride);
      c4[0] = make_float4(cx[0], cx[1], cx[2], cx[3]);
      c4[1] = make_float4(cx[4], cx[5], cy[0], cy[1]);
      c4[2] = make_float4(cy[2], cy[3], cy[4], cy[5]);
    }
    const Taps tp = make_taps(vis ? mx : 1.0f, vis ? my : 1.0f, 2);
    const int base_v = tp.row0 * sw + tp.col0;
    const int vis_v = vis ? 1 : 0;
    const float w0 = tp.w0, w1 = tp.w1, w2 = tp.w2, w3 = tp.w3;

    // ---- stage 2: wave-steps of four patches (utilities.cpp:115-189)
    const int nws = (cnt + 3) >> 2;
    const int nsteps = (nws + kU - 1) / kU;
    auto issue = [&](Ref4Loads<kU> &L, int sidx) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int ws = sidx + u * nsteps;
        const int j = 4 * ws + sub;
        const bool ok = (ws < nws) & (j < cnt);
        const int js = ok ? j : 0;
        L.jj[u] = ok ? j : -1;
        const int v = __shfl(vis_v, js, 64);
        const int base = __shfl(base_v, js, 64);
        L.vis[u] = (ok && v) ? 1 : 0;
        f32x4_t z = {0.0f, 0.0f, 0.0f, 0.0f};
        L.p4[u].l = L.p4[u].r = L.p4[u].tl = L.p4[u].tr = z;
        if (L.vis[u]) {
          gconst_f32x4 pp = ppack + (base + loff);
          L.p4[u].l = pp[-1];
          L.p4[u].r = pp[0];
          if (toprow) {
            L.p4[u].tl = pp[-sw - 1];
            L.p4[u].tr = pp[-sw];
          }
        }
      }
    };
    auto reduce = [&](const Ref4Loads<kU> &L, int sidx) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int ws = sidx + u * nsteps;
        const int js = L.jj[u] >= 0 ? L.jj[u] : 0;
        const float a0 = __shfl(w0, js, 64), a1 = __shfl(w1, js, 64), a2 = __shfl(w2, js, 64), a3 = __shfl(w3, js, 64);
        float t = tapsP4_blend(L.p4[u], 0, a0, a1, a2, a3, toprow);
        const float gx = tapsP4_blend(L.p4[u], 1, a0, a1, a2, a3, toprow);
        const float gy = tapsP4_blend(L.p4[u], 2, a0, a1, a2, a3, toprow);
        if constexpr (PN) t -= group_sum(t, 16) / 16.0f;  // utilities.cpp:187-188
        if (L.vis[u]) {
          const size_t po = (size_t)(i0 + 4 * ws) * 16;
          __builtin_nontemporal_store(t, T + po + lane);
          __builtin_nontemporal_store(gx, Gx + po + lane);
          __builtin_nontemporal_store(gy, Gy + po + lane);
        }
      }
    };
    Ref4Loads<kU> A, B;
    issue(A, 0);
    for (int sidx = 0; sidx < nsteps; sidx += 2) {
      if (sidx + 1 < nsteps) issue(B, sidx + 1);
      reduce(A, sidx);
      if (sidx + 2 < nsteps) issue(A, sidx + 2);
      if (sidx + 1 < nsteps) reduce(B, sidx + 1);
    }
  }
}

// ---------------------------------------------------------------- util_getPatch(_grad) for callers (NCC scoring etc.)
__global__ __launch_bounds__(kBlock) void k_getpatch(const float *__restrict__ img, const float *__restrict__ dx,
                                                     const float *__restrict__ dy, const float *__restrict__ mids,
                                                     int K, int P, int sw, int dopatchnorm, float *out, float *out_dx,
                                                     float *out_dy) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n = P * P;
  for (int i = blockIdx.x * kWaves + wave; i < K; i += gridDim.x * kWaves) {
    const Taps tp = make_taps(mids[i], mids[i + K], P / 2);
    const int base = tp.row0 * sw + tp.col0;
    float s = 0.0f;
    for (int q = lane; q < n; q += 64) {
      const int idx = base + (q / P) * sw + (q % P);
      const float t = tap4(img, idx, sw, tp);
      out[(size_t)i * n + q] = t;
      s += t;
      if (dx) out_dx[(size_t)i * n + q] = tap4(dx, idx, sw, tp);
      if (dy) out_dy[(size_t)i * n + q] = tap4(dy, idx, sw, tp);
    }
    if (dopatchnorm) {
      const float mean = wave_sum(s) / (float)n;
      for (int q = lane; q < n; q += 64) out[(size_t)i * n + q] -= mean;
    }
  }
}

// ---------------------------------------------------------------- patch NCC of run_track_nposes (run_track_nposes.cpp:271-355)
// One wave per point: the mean-subtracted patches (util_getPatch with dopatchnorm forced on, :281) around the point's
// position in the backward-most, the reference and the forward-most frame, each scaled to unit norm, and the two
// correlations back-ref / ref-forward combined with the weights nBack^2 / nFwd^2. Only the K correlations leave the
// device. NV = patch values per lane (psz^2 <= 64 NV). Validity tests are the reference's strict ones (:290,:297,:305).
struct NccFrame {
  const float *img;  // padded plane of the frame at the scoring level
};
template <int NV>
__global__ __launch_bounds__(kBlock) void k_ncc(NccFrame fb, NccFrame fr, NccFrame ff, const float *__restrict__ mids,
                                                int K, int P, int sw, float swo, float sho, float w_back, float w_fwd,
                                                float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n = P * P;
  for (int i = blockIdx.x * kWaves + wave; i < K; i += gridDim.x * kWaves) {
    float pat[3][NV];
    float nrm[3];
    bool val[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      const float mx = mids[(2 * f) * K + i], my = mids[(2 * f + 1) * K + i];
      val[f] = (mx > 0.0f) & (my > 0.0f) & (mx < swo) & (my < sho);
      const float *img = f == 0 ? fb.img : (f == 1 ? fr.img : ff.img);
      const Taps tp = make_taps(val[f] ? mx : 1.0f, val[f] ? my : 1.0f, P / 2);
      const int base = tp.row0 * sw + tp.col0;
      float s = 0.0f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int q = lane + 64 * v;
        pat[f][v] = (q < n && val[f]) ? tap4(img, base + (q / P) * sw + (q % P), sw, tp) : 0.0f;
        s += pat[f][v];
      }
      const float mean = wave_sum(s) / (float)n;  // utilities.cpp:111-112
      float ss = 0.0f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int q = lane + 64 * v;
        pat[f][v] = q < n ? pat[f][v] - mean : 0.0f;
        ss += pat[f][v] * pat[f][v];
      }
      nrm[f] = sqrtf(wave_sum(ss));  // patch.norm()
    }
    float dbr = 0.0f, drf = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const float b = pat[0][v] / nrm[0], r = pat[1][v] / nrm[1], f = pat[2][v] / nrm[2];  // patch /= patch.norm()
      if (lane + 64 * v < n) {
        dbr += b * r;
        drf += r * f;
      }
    }
    dbr = wave_sum(dbr);
    drf = wave_sum(drf);
    if (lane == 0) {
      float corr = -1.0f;
      if (val[1]) {
        const float cbr = val[0] ? fmaxf(0.0f, dbr) : -1.0f, w0 = val[0] ? w_back : 0.0f;
        const float crf = val[2] ? fmaxf(0.0f, drf) : -1.0f, w1 = val[2] ? w_fwd : 0.0f;
        corr = fmaxf(0.0f, (cbr * w0 + crf * w1) / (w0 + w1));  // std::max(0.0f, NaN) == 0.0f, as fmaxf
      }
      out[i] = corr;
    }
  }
}
void launch_ncc(const float *img_b, const float *img_r, const float *img_f, const float *mids, int K, int P, int sw,
                float swo, float sho, float w_back, float w_fwd, float *out, hipStream_t s) {
  int gx = (K + kWaves - 1) / kWaves;
  gx = std::max(1, std::min(gx, kMaxGridX));
  const NccFrame fb{img_b}, fr{img_r}, ff{img_f};
  const int n = P * P;
  if (n <= 64)
    hipLaunchKernelGGL((k_ncc<1>), dim3(gx), dim3(kBlock), 0, s, fb, fr, ff, mids, K, P, sw, swo, sho, w_back, w_fwd, out);
  else if (n <= 256)
    hipLaunchKernelGGL((k_ncc<4>), dim3(gx), dim3(kBlock), 0, s, fb, fr, ff, mids, K, P, sw, swo, sho, w_back, w_fwd, out);
  else if (n <= 1024)
    hipLaunchKernelGGL((k_ncc<16>), dim3(gx), dim3(kBlock), 0, s, fb, fr, ff, mids, K, P, sw, swo, sho, w_back, w_fwd, out);
  else
    hipLaunchKernelGGL((k_ncc<64>), dim3(gx), dim3(kBlock), 0, s, fb, fr, ff, mids, K, P, sw, swo, sho, w_back, w_fwd, out);
}

// ---------------------------------------------------------------- Python flow-tracking surface on the device
// func_get_transf_position (misc_src/classoftrack.py:4-34): K sub-pixel points moved by a displacement field sampled
// bilinearly at the points, in float64 with NumPy's operation order (field value widened, times weight, summed left
// to right), so the result is bit-identical to the NumPy restatement and to the reference's goldens. A point whose four
// taps are not all inside the field comes back as NaN. F = float or double (the field's dtype).
template <typename F>
__global__ __launch_bounds__(kBlock) void k_flow_gather(const F *__restrict__ du, const F *__restrict__ dv, int H, int W,
                                                        const double *__restrict__ xy, int K, double *__restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= K) return;
  const double x = xy[2 * i], y = xy[2 * i + 1];
  const double flx = floor(x), fly = floor(y);
  const double nan = __builtin_nan("");
  double ox = nan, oy = nan;
  // NaN / out-of-range coordinates fail the range test (written positively), like the INT_MIN cast in NumPy
  if (flx >= 0.0 && fly >= 0.0 && flx + 1.0 < (double)W && fly + 1.0 < (double)H) {
    const int x0 = (int)flx, y0 = (int)fly, x1 = x0 + 1, y1 = y0 + 1;
    const double fx = x - flx, fy = y - fly;
    const double w0 = fx * fy, w1 = (1 - fx) * fy, w2 = fx * (1 - fy), w3 = (1 - fx) * (1 - fy);
    const size_t a = (size_t)y1 * W + x1, b = (size_t)y1 * W + x0, c = (size_t)y0 * W + x1, d = (size_t)y0 * W + x0;
    ox = x + ((double)du[a] * w0 + (double)du[b] * w1 + (double)du[c] * w2 + (double)du[d] * w3);
    oy = y;
    if (dv) oy = y + ((double)dv[a] * w0 + (double)dv[b] * w1 + (double)dv[c] * w2 + (double)dv[d] * w3);
  }
  out[2 * i] = ox;
  out[2 * i + 1] = oy;
}
void launch_flow_gather(const void *du, const void *dv, int is_f64, int H, int W, const double *xy, int K, double *out,
                        hipStream_t s) {
  const dim3 g((K + kBlock - 1) / kBlock), blk(kBlock);
  if (is_f64)
    hipLaunchKernelGGL((k_flow_gather<double>), g, blk, 0, s, (const double *)du, (const double *)dv, H, W, xy, K, out);
  else
    hipLaunchKernelGGL((k_flow_gather<float>), g, blk, 0, s, (const float *)du, (const float *)dv, H, W, xy, K, out);
}

// func_extract_bil_patch (misc_src/func_OF_util.py:87-129), batched: K points, raw (side x side x C) bilinear patches
// of an (H, W, C) float64 image, side = 2 (pz / 2) (Python-2 integer division), patch-constant weights on four
// integer-shifted windows in NumPy's operation order. One thread per output value; windows that leave the image are
// the caller's error (checked on the host).
__global__ __launch_bounds__(kBlock) void k_bil_patches(const double *__restrict__ img, int H, int W, int C,
                                                        const double *__restrict__ pts, int K, int half,
                                                        double *__restrict__ out) {
  const int side = 2 * half;
  const size_t per = (size_t)side * side * C;
  const size_t n = (size_t)K * per;
  for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < n; t += (size_t)gridDim.x * kBlock) {
    const int k = (int)(t / per);
    const size_t r = t - (size_t)k * per;
    const int c = (int)(r % C), px = (int)((r / C) % side), py = (int)(r / ((size_t)C * side));
    const double x = pts[2 * k], y = pts[2 * k + 1];
    const double flx = floor(x), fly = floor(y);
    const double fx = x - flx, fy = y - fly;
    const int x0 = (int)flx - half + px, y0 = (int)fly - half + py;
    const double a = img[((size_t)(y0 + 1) * W + (x0 + 1)) * C + c], b = img[((size_t)(y0 + 1) * W + x0) * C + c];
    const double cc = img[((size_t)y0 * W + (x0 + 1)) * C + c], d = img[((size_t)y0 * W + x0) * C + c];
    out[t] = a * (fx * fy) + b * ((1 - fx) * fy) + cc * (fx * (1 - fy)) + d * ((1 - fx) * (1 - fy));
  }
}
void launch_bil_patches(const double *img, int H, int W, int C, const double *pts, int K, int half, double *out,
                        hipStream_t s) {
  const size_t n = (size_t)K * 4 * half * half * C;
  const int g = (int)std::min<size_t>((n + kBlock - 1) / kBlock, 8192);
  hipLaunchKernelGGL(k_bil_patches, dim3(std::max(g, 1)), dim3(kBlock), 0, s, img, H, W, C, pts, K, half, out);
}

// ---------------------------------------------------------------- pyramid (utilities.cpp:14-52)
// level 0: copy the w x h image into the interior of the padded plane
__global__ __launch_bounds__(kBlock) void k_pyr_copy(const float *__restrict__ src, float *dst, int w, int h, int pad,
                                                     int sw) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * kWaves + (threadIdx.x >> 6);
  if (x < w && y < h) dst[(size_t)(y + pad) * sw + x + pad] = src[(size_t)y * w + x];
}

// level l from level l-1: cv::resize(.5,.5,INTER_LINEAR) == 2x2 box mean for even sizes; clamped bilinear otherwise
__global__ __launch_bounds__(kBlock) void k_pyr_down(const float *__restrict__ src, int pw, int ph, int psw,
                                                     float *dst, int w, int h, int pad, int sw) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * kWaves + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const float *s = src + (size_t)pad * psw + pad;  // interior origin of the previous level
  float v;
  if (pw == 2 * w && ph == 2 * h) {
    const float *r0 = s + (size_t)(2 * y) * psw + 2 * x;
    const float *r1 = r0 + psw;
    v = ((r0[0] + r1[0]) + (r0[1] + r1[1])) * 0.25f;
  } else {
    int y0 = 2 * y, y1 = y0 + 1, x0 = 2 * x, x1 = x0 + 1;
    y0 = min(y0, ph - 1);
    y1 = min(y1, ph - 1);
    x0 = min(x0, pw - 1);
    x1 = min(x1, pw - 1);
    const float top = s[(size_t)y0 * psw + x0] * 0.5f + s[(size_t)y0 * psw + x1] * 0.5f;
    const float bot = s[(size_t)y1 * psw + x0] * 0.5f + s[(size_t)y1 * psw + x1] * 0.5f;
    v = top * 0.5f + bot * 0.5f;
  }
  dst[(size_t)(y + pad) * sw + x + pad] = v;
}

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  if (i < 0) return -i;
  if (i >= n) return 2 * n - 2 - i;
  return i;
}

// replicate-pad the image in place (border threads read interior, write border) and write both gradient planes:
// cv::Sobel(ksize=1) = I(x+1)-I(x-1) with reflect-101, then zero padding
__global__ __launch_bounds__(kBlock) void k_pyr_finish(float *img, float *dx, float *dy, int w, int h, int pad, int sw,
                                                       int sh, int getgrad) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * kWaves + (threadIdx.x >> 6);
  if (x >= sw || y >= sh) return;
  const int ix = x - pad, iy = y - pad;
  const bool inside = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h);
  const float *I = img + (size_t)pad * sw + pad;
  const size_t o = (size_t)y * sw + x;
  if (!inside) {
    const int cxx = min(max(ix, 0), w - 1), cyy = min(max(iy, 0), h - 1);
    img[o] = I[(size_t)cyy * sw + cxx];
    if (getgrad) {
      dx[o] = 0.0f;
      dy[o] = 0.0f;
    }
  } else if (getgrad) {
    dx[o] = I[(size_t)iy * sw + reflect101(ix + 1, w)] - I[(size_t)iy * sw + reflect101(ix - 1, w)];
    dy[o] = I[(size_t)reflect101(iy + 1, h) * sw + ix] - I[(size_t)reflect101(iy - 1, h) * sw + ix];
  }
}

// One launch per level (the builder's default): every thread owns one pixel of the PADDED level and evaluates the level
// image where it needs it straight from the level above (level 0: from the input frame) -- its own value (clamped:
// replicate padding), and for interior pixels the four reflect-101 neighbours of the Sobel pair -- with exactly the
// arithmetic of k_pyr_copy / k_pyr_down / k_pyr_finish / k_pyr_pack, so the planes are bit-identical to the
// four-kernel form; the redundant evaluations are cache hits, the level is written once (img, dx, dy, packed texel).
template <bool FIRST>
__device__ __forceinline__ float pyr_level_value(const float *__restrict__ src, int pw, int ph, int psw, int pad, int w,
                                                 int h, int x, int y) {
  if constexpr (FIRST) return src[(size_t)y * w + x];
  const float *s = src + (size_t)pad * psw + pad;  // interior origin of the previous level
  if (pw == 2 * w && ph == 2 * h) {
    const float *r0 = s + (size_t)(2 * y) * psw + 2 * x;
    const float *r1 = r0 + psw;
    return ((r0[0] + r1[0]) + (r0[1] + r1[1])) * 0.25f;
  }
  int y0 = 2 * y, y1 = y0 + 1, x0 = 2 * x, x1 = x0 + 1;
  y0 = min(y0, ph - 1);
  y1 = min(y1, ph - 1);
  x0 = min(x0, pw - 1);
  x1 = min(x1, pw - 1);
  const float top = s[(size_t)y0 * psw + x0] * 0.5f + s[(size_t)y0 * psw + x1] * 0.5f;
  const float bot = s[(size_t)y1 * psw + x0] * 0.5f + s[(size_t)y1 * psw + x1] * 0.5f;
  return top * 0.5f + bot * 0.5f;
}
template <bool FIRST>
__global__ __launch_bounds__(kBlock) void k_pyr_level(const float *__restrict__ src, int pw, int ph, int psw, float *img,
                                                      float *dx, float *dy, f32x4_t *pack, int w, int h, int pad, int sw,
                                                      int sh, int getgrad) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * kWaves + (threadIdx.x >> 6);
  if (x >= sw || y >= sh) return;
  const int ix = x - pad, iy = y - pad;
  const bool inside = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h);
  const int cxx = min(max(ix, 0), w - 1), cyy = min(max(iy, 0), h - 1);
  const size_t o = (size_t)y * sw + x;
  const float v = pyr_level_value<FIRST>(src, pw, ph, psw, pad, w, h, cxx, cyy);
  img[o] = v;
  if (!getgrad) return;
  float gx = 0.0f, gy = 0.0f;
  if (inside) {
    gx = pyr_level_value<FIRST>(src, pw, ph, psw, pad, w, h, reflect101(ix + 1, w), iy) -
         pyr_level_value<FIRST>(src, pw, ph, psw, pad, w, h, reflect101(ix - 1, w), iy);
    gy = pyr_level_value<FIRST>(src, pw, ph, psw, pad, w, h, ix, reflect101(iy + 1, h)) -
         pyr_level_value<FIRST>(src, pw, ph, psw, pad, w, h, ix, reflect101(iy - 1, h));
  }
  dx[o] = gx;
  dy[o] = gy;
  const f32x4_t t = {v, gx, gy, 0.0f};
  pack[o] = t;
}

// ---------------------------------------------------------------- host-side launchers
static inline dim3 grid2d(int w, int h) { return dim3((w + 63) / 64, (h + kWaves - 1) / kWaves); }

void launch_pyr_copy(const float *src, float *dst, int w, int h, int pad, int sw, hipStream_t s) {
  hipLaunchKernelGGL(k_pyr_copy, grid2d(w, h), dim3(kBlock), 0, s, src, dst, w, h, pad, sw);
}
void launch_pyr_down(const float *src, int pw, int ph, int psw, float *dst, int w, int h, int pad, int sw,
                     hipStream_t s) {
  hipLaunchKernelGGL(k_pyr_down, grid2d(w, h), dim3(kBlock), 0, s, src, pw, ph, psw, dst, w, h, pad, sw);
}
// plain streaming read (16-byte non-temporal loads, 8 in flight per lane): the practical HBM read ceiling of the
// box, reported next to the vendor peak (ictr_stream_read_bandwidth)
__global__ __launch_bounds__(kBlock) void k_stream_read(const f32x4_t *__restrict__ src, size_t nvec, float *sink) {
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x, nth = (size_t)gridDim.x * kBlock;
  float acc = 0.0f;
  size_t i = tid;
  for (; i + 7 * nth < nvec; i += 8 * nth) {
    f32x4_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + i + u * nth);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
  }
  for (; i < nvec; i += nth) {
    const f32x4_t v = __builtin_nontemporal_load(src + i);
    acc += (v.x + v.y) + (v.z + v.w);
  }
  if (acc == 1.2345e30f) sink[tid] = acc;
}
void launch_stream_read(const float *src, size_t nfloats, float *sink, hipStream_t s) {
  hipLaunchKernelGGL(k_stream_read, dim3(8192), dim3(kBlock), 0, s, reinterpret_cast<const f32x4_t *>(src), nfloats / 4,
                     sink);
}

// interleave a finished level into {img, dx, dy, 0} texels (read by k_ref8's packed taps)
__global__ __launch_bounds__(kBlock) void k_pyr_pack(const float *__restrict__ img, const float *__restrict__ dx,
                                                     const float *__restrict__ dy, f32x4_t *pack, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    f32x4_t v = {img[i], dx[i], dy[i], 0.0f};
    pack[i] = v;
  }
}
void launch_pyr_pack(const float *img, const float *dx, const float *dy, float *pack, size_t n, hipStream_t s) {
  const int g = (int)std::min<size_t>((n + kBlock - 1) / kBlock, 8192);
  hipLaunchKernelGGL(k_pyr_pack, dim3(g), dim3(kBlock), 0, s, img, dx, dy, reinterpret_cast<f32x4_t *>(pack), n);
}
// src: the input frame (first, unpadded, stride w) or the previous level's padded image plane (pw x ph, stride psw)
void launch_pyr_level(const float *src, int first, int pw, int ph, int psw, float *img, float *dx, float *dy,
                      float *pack, int w, int h, int pad, int sw, int sh, int getgrad, hipStream_t s) {
  if (first)
    hipLaunchKernelGGL(k_pyr_level<true>, grid2d(sw, sh), dim3(kBlock), 0, s, src, pw, ph, psw, img, dx, dy,
                       reinterpret_cast<f32x4_t *>(pack), w, h, pad, sw, sh, getgrad);
  else
    hipLaunchKernelGGL(k_pyr_level<false>, grid2d(sw, sh), dim3(kBlock), 0, s, src, pw, ph, psw, img, dx, dy,
                       reinterpret_cast<f32x4_t *>(pack), w, h, pad, sw, sh, getgrad);
}
void launch_pyr_finish(float *img, float *dx, float *dy, int w, int h, int pad, int sw, int sh, int getgrad,
                       hipStream_t s) {
  hipLaunchKernelGGL(k_pyr_finish, grid2d(sw, sh), dim3(kBlock), 0, s, img, dx, dy, w, h, pad, sw, sh, getgrad);
}
void launch_getpatch(const float *img, const float *dx, const float *dy, const float *mids, int K, int P, int sw,
                     int dopatchnorm, float *out, float *out_dx, float *out_dy, hipStream_t s) {
  int gx = (K + kWaves - 1) / kWaves;
  if (gx > kMaxGridX) gx = kMaxGridX;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(k_getpatch, dim3(gx), dim3(kBlock), 0, s, img, dx, dy, mids, K, P, sw, dopatchnorm, out, out_dx,
                     out_dy);
}
void launch_project_generic(const float *pt3d, float *pt3d_rot, float *pt2d, int n, int M, const float *G,
                            LevelCam lc, hipStream_t s) {
  hipLaunchKernelGGL(k_project_generic, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pt3d, pt3d_rot, pt2d, n, M,
                     G, lc);
}
void launch_project_ref(const EngineDev &e, const LevelCam *cams, int maxpts, hipStream_t s) {
  AllCams ac;
  for (int l = 0; l < e.nlev && l < 16; ++l) ac.lc[l] = cams[l];
  hipLaunchKernelGGL(k_project_ref, dim3((maxpts + kBlock - 1) / kBlock, e.B), dim3(kBlock), 0, s, e, ac);
}
// Deferred H: on the wave64 fast paths the level's H partials are reduced and factored by the level's FIRST iteration
// tail (k_iter_tail / k_iter_finish with first_h), not by k_level_tail -- in sharded mode H then travels in the same
// 27-float message as the first b and the level phase needs no collective of its own. Where the partials come from:
// P = 8: the setup kernel k_ref8 (three sums per patch, H = sum J^T S J); P = 4: the first k_iter4<.., WH> launch.
// Variant bit 8 (256) lets k_level_tail reduce and factor H for P = 8 (cross-check in the tests).
// wave64 fast paths: 8x8 always, 4x4 when every reference pyramid carries the packed planes
static bool fast8(const EngineDev &e, int variant) { return e.P == 8 && !(variant & 2); }
static bool fast4(const EngineDev &e, int variant) { return e.P == 4 && e.packed && !(variant & 2); }
bool defer_h(const EngineDev &e, int variant) { return (fast8(e, variant) && !(variant & 256)) || fast4(e, variant); }
// (the host sets variant bit 1 whenever a robustness option is on, see engine_variant in ictr_host.hip)

// steps 4-6 of one level for every problem: accumulate kernel + per-problem tail
void launch_ref_level(const EngineDev &e, const LevelCam &lc, int level, int gridx, int variant, int cpw, int gridx8,
                      hipStream_t s) {
  const dim3 blk(kBlock);
  int nblk = gridx;
  const bool dh = defer_h(e, variant);
  if (fast8(e, variant)) {
    nblk = gridx8;
    const dim3 g8(gridx8, e.B);
    const bool pk = e.packed && !(variant & 4096);  // variant bit 12: three separate planes (A/B)
    // gradients on the fly from the image plane whenever the reference pyramids are builder-made (r03: 287 -> 242 us per
    // level-0 launch of 32 pairs; a must for image-only pyramids, e.otf == 2); variant bit 27 (134217728): read the
    // gradient planes instead (A/B, cross-check)
    const bool otf = e.otf == 2 || (e.otf == 1 && !(variant & (1 << 27)));
    // patches per pipeline step: two with the packed planes (measured r02: 1 -> 2 saves 50-120 us per level, 4 adds
    // nothing), one otherwise
    if (otf && e.dopatchnorm)
      hipLaunchKernelGGL((k_ref8<true, 1, false, true>), g8, blk, 0, s, e, lc, level, cpw);
    else if (otf && cpw % 16 == 0 && !(variant & (1 << 28)))  // variant bit 28 (268435456): the dynamic patch loop (A/B)
      hipLaunchKernelGGL((k_ref8<false, 2, false, true, true>), g8, blk, 0, s, e, lc, level, cpw);
    else if (otf)
      hipLaunchKernelGGL((k_ref8<false, 2, false, true>), g8, blk, 0, s, e, lc, level, cpw);
    else if (pk && e.dopatchnorm)
      hipLaunchKernelGGL((k_ref8<true, 1, true>), g8, blk, 0, s, e, lc, level, cpw);
    else if (pk)
      hipLaunchKernelGGL((k_ref8<false, 2, true>), g8, blk, 0, s, e, lc, level, cpw);
    else if (e.dopatchnorm)
      hipLaunchKernelGGL((k_ref8<true, 1, false>), g8, blk, 0, s, e, lc, level, cpw);
    else
      hipLaunchKernelGGL((k_ref8<false, 1, false>), g8, blk, 0, s, e, lc, level, cpw);
  } else if (fast4(e, variant)) {
    nblk = gridx8;
    if (e.dopatchnorm)
      hipLaunchKernelGGL((k_ref4<true>), dim3(gridx8, e.B), blk, 0, s, e, lc, level, cpw);
    else
      hipLaunchKernelGGL((k_ref4<false>), dim3(gridx8, e.B), blk, 0, s, e, lc, level, cpw);
  } else if (e.P == 4)
    hipLaunchKernelGGL(k_ref_level<4>, dim3(gridx, e.B), blk, 0, s, e, lc, level);
  else
    hipLaunchKernelGGL(k_ref_level<0>, dim3(gridx, e.B), blk, 0, s, e, lc, level);
  // variant bit 24 (set by the host's resident path only): no tail -- the resident-iteration launch that follows reduces
  // and factors H itself (its solver workgroups are idle while the workers load their templates)
  if (!(fast8(e, variant) && (variant & (1 << 24))))
    hipLaunchKernelGGL(k_level_tail, dim3(e.B), blk, 0, s, e, nblk, dh ? 1 : 0);
}
void launch_level_finish(const EngineDev &e, int variant, hipStream_t s) {
  hipLaunchKernelGGL(k_level_finish, dim3(e.B), dim3(64), 0, s, e, defer_h(e, variant) ? 1 : 0);
}
// steps 7-9a of one Gauss-Newton iteration for every problem (the accumulate kernel) ...
// first: the level's first iteration (P = 4 with deferred H: it also accumulates the H partials)
// ev0 / ev1 (optional, timing runs): HIP events that take the kernel's own start and end time stamps (hipExtLaunchKernelGGL:
// the dispatch's completion-signal times, what rocprofv3's kernel trace lists), not the time between two event packets
void launch_iter_main(const EngineDev &e, const LevelCam &lc, int level, int gridx, int variant, int cpw, int gridx8,
                      int first, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
  const dim3 blk(kBlock);
#define ICTR_LAUNCH(kern, grid, ...)                                                      \
  do {                                                                                    \
    if (ev0 && ev1)                                                                       \
      hipExtLaunchKernelGGL(kern, grid, blk, 0, s, ev0, ev1, 0, __VA_ARGS__);             \
    else                                                                                  \
      hipLaunchKernelGGL(kern, grid, blk, 0, s, __VA_ARGS__);                             \
  } while (0)
  if (fast8(e, variant)) {
    const dim3 g8(gridx8, e.B);
    // patches per pipeline step: 4 measured best (profiles/r01_notes.md); variant bits 4-5 = 2: two (A/B)
    if (e.dopatchnorm)
      ICTR_LAUNCH((k_iter8<true, 2>), g8, e, lc, level, cpw);
    else if (((variant >> 4) & 3) == 2)
      ICTR_LAUNCH((k_iter8<false, 2>), g8, e, lc, level, cpw);
    else
      ICTR_LAUNCH((k_iter8<false, 4>), g8, e, lc, level, cpw);
  } else if (fast4(e, variant)) {
    const dim3 g8(gridx8, e.B);
    if (first) {
      if (e.dopatchnorm)
        ICTR_LAUNCH((k_iter4<true, true>), g8, e, lc, level, cpw);
      else
        ICTR_LAUNCH((k_iter4<false, true>), g8, e, lc, level, cpw);
    } else if (e.dopatchnorm)
      ICTR_LAUNCH((k_iter4<true, false>), g8, e, lc, level, cpw);
    else
      ICTR_LAUNCH((k_iter4<false, false>), g8, e, lc, level, cpw);
  } else if (e.P == 4)
    ICTR_LAUNCH((k_iter<4>), dim3(gridx, e.B), e, lc, level);
  else
    ICTR_LAUNCH((k_iter<0>), dim3(gridx, e.B), e, lc, level);
#undef ICTR_LAUNCH
}
// ... and steps 9b-10 (one workgroup per problem)
void launch_iter_tail(const EngineDev &e, int level, int gridx, int variant, int gridx8, int first, hipStream_t s) {
  const int nblk = (fast8(e, variant) || fast4(e, variant)) ? gridx8 : gridx;
  hipLaunchKernelGGL(k_iter_tail, dim3(e.B), dim3(kBlock), 0, s, e, level, nblk, (first && defer_h(e, variant)) ? 1 : 0);
}
void launch_iter(const EngineDev &e, const LevelCam &lc, int level, int gridx, int variant, int cpw, int gridx8,
                 int first, hipStream_t s) {
  launch_iter_main(e, lc, level, gridx, variant, cpw, gridx8, first, s, nullptr, nullptr);
  launch_iter_tail(e, level, gridx, variant, gridx8, first, s);
}
void launch_iter_finish(const EngineDev &e, int level, int variant, int first, hipStream_t s) {
  hipLaunchKernelGGL(k_iter_finish, dim3(e.B), dim3(64), 0, s, e, level, (first && defer_h(e, variant)) ? 1 : 0);
}

}  // namespace ictr
