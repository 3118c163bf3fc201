// ictr_track1.hip -- the WHOLE coarse-to-fine tracking of a small problem in ONE launch (gfx950 / CDNA4).
//
// The reference's own operating point is 50-1000 points per frame pair (run_odometer_test.m:140,232: every 10th
// visible point; odometer.cpp:156-167 times 100 patches). At that size the per-iteration launch pairs of
// ictr_kernels.hip are pure dependent-launch latency (111 launches = 0.65 ms, slower than the CPU below ~150
// points). Here one workgroup owns one problem and runs odometer.cpp:257-426 from the first level's setup to the
// last pose update without leaving the CU:
//
//   per level   stage A  one POINT per thread: visibility at the reference pose (odometer.cpp:268-282), the 12
//                        steepest-descent coefficients (:313-326; stale ones kept for points out of view, :304),
//                        bilinear weights + window base (utilities.cpp:66-77) -> LDS point record
//               stage B  one PATCH per wave (four 4x4 patches per wave; any other size: lanes loop over the
//                        pixels): T/Gx/Gy gathered (utilities.cpp:115-189), stored to the global patch buffers
//                        (they persist across frames like pat_ref_all) and, when they fit, kept in LDS; the 21 H
//                        sums (odometer.cpp:428-472) per lane -> wave shuffle -> fixed-order f64 over the waves
//               thread 0 full-pivot LU of H once per level (odometer.cpp:509-515), loop state reset (:341-346)
//   per iteration stage 1  one point per thread: projection at the current pose (pose.cpp:384-391), ind_new
//                        (odometer.cpp:369-377), weights + base -> LDS record
//               stage 2  one patch per wave: current-frame window (utilities.cpp:55-113), residual, J^T r
//                        (odometer.cpp:381-404) in six per-lane accumulators across all the wave's patches
//               thread 0 fixed-order f64 sum of the wave partials, substitution with the level's LU factors, pose
//                        update, exp map, |dp|_1 and the loop condition (odometer.cpp:407-418)
//
// Three workgroup barriers per iteration, no global synchronisation, no host round trip; the pose, the LU factors
// and the per-point records never leave LDS. Independent problems (run_track_nposes' pose samples) are the grid.
// Arithmetic: identical expressions to the per-iteration kernels (same helpers, -ffp-contract=off), so patches,
// projections and coefficients are bit-identical to the CPU path; H and b differ from it by summation order only.
#include <string.h>

#include <algorithm>

#include "ictr_dev.h"
#include "ictr_devfn.h"
#include "se3_math.h"

namespace ictr {

constexpr int kT1MaxWaves = 8;  // any-size form: 512 threads, two waves per SIMD with up to 256 registers each
constexpr int kT8MaxWaves = 8;  // 8x8 form

// A tracking of one small frame pair is ~270 us of kernel behind ~40 us of dependent small operations (two state
// uploads, a fill, the projection kernel) and in front of a read-back copy. For batches whose upload fits the kernel
// argument segment the launch carries the upload itself (`blob`: the initial ProbState of every problem and the plane
// table, byte for byte what ictr_batch_begin would copy), workgroup b stores its problem's part to device memory for
// later readers, runs step 3 (k_project_ref's arithmetic) for its own points, and at the end writes the final state
// straight into the host's pinned mirror as well: one launch and one event per tracking.
constexpr int kTeamSlot = 32;     // granules per workgroup and exchange (21 of H or 6 of b)
constexpr int kTeamMax = 64;      // workgroups per problem
constexpr int kT1BlobWords = 704;  // 2816 B: 4 problems x (ProbState 496 B + 5 levels x 40 B)
struct T1Args {
  LevelCam lc[16];
  int npts_cap;  // record / LDS template capacity in points (>= every problem's npts)
  int dbg;       // ICTR_T1_PROF builds: ablation bits (env ICTR_T1_DBG); otherwise unused
  int fused_begin;         // 1: `blob` is valid and this launch does ictr_batch_begin's device part too; 2: the records
                           // and the plane table are in device memory already, the launch does step 3 (the projections)
  int st_words, pl_words;  // dwords per problem of the blob's two sections
  int cap_w;               // 8x8 form: LDS slots per wave (>= its points, rounded up to whole pipeline steps + prefetch)
  ProbState *host_st;      // pinned host mirror of the final states [B], or nullptr
  // team form (k_track1_p8<..., TEAM>): `team` workgroups share one problem, see "Teams" below
  int team, team_q;                // workgroups per problem; points per workgroup (the last one may own fewer)
  unsigned team_tag0;              // launch epoch << 12: granule tags of this launch are team_tag0 + exchange number
  unsigned long long team_limit;   // polling limit in wall_clock64 ticks (100 MHz)
  unsigned long long *team_mail;   // [B][2][team][kTeamSlot] granules {float bits, tag}
  int *team_err;                   // sticky flag (pinned host memory): an exchange timed out
  int team_mute;                   // debug (variant bit 25): part `team_mute - 1` of every problem never posts its values
                                   // (its peers' polls run into the limit: the time-out path's test); 0 = off
  __attribute__((aligned(8))) unsigned blob[kT1BlobWords];  // [ProbState x B][PlaneSet x B x nlev]
};

// Initial state / plane table of problem b: from the kernel arguments (fused begin) or from device memory. The source
// is chosen at run time, so these few loads are flat loads into vector registers; v_readfirstlane puts every value back
// into a scalar register (all lanes hold the same value), so plane pointers and the pose stay wave-uniform operands.
__device__ __forceinline__ int t1_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float t1_uni(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ const float *t1_uni(const float *p) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return reinterpret_cast<const float *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ const ProbState *t1_initial_state(const EngineDev &e, const T1Args &a, int b) {
  return a.fused_begin == 1 ? reinterpret_cast<const ProbState *>(a.blob + (size_t)b * a.st_words) : e.st + b;
}
__device__ __forceinline__ PlaneSet t1_planes(const EngineDev &e, const T1Args &a, int b, int level) {
  // (fused begin: read from the kernel arguments, not back from the table this workgroup has just stored: the scalar
  // cache is not coherent with vector stores of the same launch)
  const PlaneSet *tab = a.fused_begin == 1 ? reinterpret_cast<const PlaneSet *>(a.blob + (size_t)e.B * a.st_words)
                                           : e.planes;
  const PlaneSet v = tab[b * e.nlev + level];
  PlaneSet r;
  r.ref = t1_uni(v.ref);
  r.dx = t1_uni(v.dx);
  r.dy = t1_uni(v.dy);
  r.cur = t1_uni(v.cur);
  r.pack = t1_uni(v.pack);
  return r;
}
// ictr_batch_begin's device part for problem b: state + plane table to device memory, then step 3 for every level
// (pose.cpp:400-488 at lv_f, pose.cpp:307-397 below it; the camera-frame point is level independent)
// (team form: workgroup `part` of `team` projects its own points [part q, part q + q) only; part 0 stores the tables)
__device__ __forceinline__ void t1_fused_begin(const EngineDev &e, const T1Args &a, int b, int tid, int nthr,
                                               int part = 0, int q = 0x7fffffff) {
  const unsigned *bs = a.fused_begin == 1 ? a.blob + (size_t)b * a.st_words : reinterpret_cast<const unsigned *>(e.st + b);
  if (part == 0 && a.fused_begin == 1) {
    unsigned *ds = reinterpret_cast<unsigned *>(e.st + b);
    for (int i = tid; i < a.st_words; i += nthr) ds[i] = bs[i];
    const unsigned *bp = a.blob + (size_t)e.B * a.st_words + (size_t)b * a.pl_words;
    unsigned *dp = reinterpret_cast<unsigned *>(const_cast<PlaneSet *>(e.planes) + (size_t)b * e.nlev);
    for (int i = tid; i < a.pl_words; i += nthr) dp[i] = bp[i];
  }
  const ProbState *st = reinterpret_cast<const ProbState *>(bs);
  const int lo = (int)min((long long)part * q, (long long)st->npts);
  const int npts = (int)min((long long)st->npts, (long long)lo + q);
  float G[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) G[k] = st->G[k];
  const float *p3 = e.pt3d + (size_t)b * 3 * e.M;
  float *p3r = e.pt3d_ref + (size_t)b * 3 * e.M;
  for (int i = lo + tid; i < npts; i += nthr) {
    const float X = p3[i], Y = p3[i + e.M], Z = p3[i + 2 * e.M];
    const float tx = G[0] * X + G[1] * Y + G[2] * Z + G[3];
    const float ty = G[4] * X + G[5] * Y + G[6] * Z + G[7];
    const float tz = G[8] * X + G[9] * Y + G[10] * Z + G[11];
    p3r[i] = tx;
    p3r[i + e.M] = ty;
    p3r[i + 2 * e.M] = tz;
    for (int l = e.lv_l; l <= e.lv_f; ++l) {
      float *p2 = e.pt2d + ((size_t)b * e.nlev + l) * 2 * e.M;
      p2[i] = (tx / tz) * a.lc[l].fx + a.lc[l].cx;
      p2[i + e.M] = (ty / tz) * a.lc[l].fy + a.lc[l].cy;
    }
  }
  __syncthreads();  // the projections are read back by other threads of this workgroup
}

struct T1Rec {  // 16 floats per point
  float w0, w1, w2, w3;
  float cx0, cx2, cx3, cx4;
  float cx5, cy1, cy2, cy3;
  float cy4, cy5, vis;
  int base;
};
static_assert(sizeof(T1Rec) == 64, "record must be one 64-byte row");

#ifdef ICTR_T1_PROF  // diagnostic builds only: per-phase cycle counters of thread 0 (tools/t1prof.py)
#define T1_MARK(k)                          \
  if (tid == 0) {                           \
    t1_ = __builtin_readcyclecounter();     \
    tp_[k] += t1_ - t0_;                    \
    t0_ = t1_;                              \
  }
#else
#define T1_MARK(k)
#endif
// wave totals of N per-lane accumulators -> dst[0..N-1] (lane k stores total k; one store instruction)
#define T1_REDUCE_STORE(N, acc, dst)                   \
  {                                                    \
    float o_ = 0.0f;                                   \
    _Pragma("unroll") for (int k_ = 0; k_ < (N); ++k_) { \
      const float v_ = wave_sum_dpp((acc)[k_]);        \
      o_ = lane == k_ ? v_ : o_;                       \
    }                                                  \
    if (lane < (N)) (dst)[lane] = o_;                  \
  }

typedef const char __attribute__((address_space(1))) *gconst_bytes;

// final state of the problem (the host reads p, G and the iteration count); wave 0 only
__device__ __forceinline__ void t1_store_final(ProbState &dst, const WaveSolver &S, const float *G, int lane) {
  if (lane < 6) {
    dst.p[lane] = S.p;
    dst.b[lane] = S.b;
    dst.dp[lane] = S.dp;
  }
  if (lane < 36) dst.H[lane] = S.h;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) dst.G[k] = G[k];
    dst.normdp = S.normdp;
    dst.normdp_init = S.normdp_init;
    dst.it = S.it;
    dst.active = S.active;
    dst.total_iters = S.total_iters;
  }
}

// Any patch size (run-time e.P), every option; 8x8 patches without behaviour-changing options take k_track1_p8 below.
// TL: the level's T/Gx/Gy patches are kept in LDS (they fit); otherwise they are re-read from the global patch
// buffers, which the same workgroup wrote during the level setup (L2 hits).
template <bool TL>
__global__ __launch_bounds__(64 * kT1MaxWaves) void k_track1(EngineDev e, T1Args a) {
  extern __shared__ __attribute__((aligned(16))) float sDyn[];
  __shared__ float sPart[kT1MaxWaves][kPartHStride];  // per-wave partial sums (21 of H, or 6 of b)
  __shared__ float sG[12];                             // cpos_G of the current iteration
  __shared__ int sActive;                              // loop condition, decided by wave 0

  const int b = blockIdx.x;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = nthr >> 6;
  const int P = e.P;
  const int n = P * P;
  const int pszd2 = P / 2;
  const int M = e.M;
  const int ppw = (n <= 64 && (64 % n) == 0) ? 64 / n : 1;  // patches per wave
  const int sub = ppw > 1 ? lane / n : 0;
  const int q0 = ppw > 1 ? lane % n : lane;
  const int qstride = ppw > 1 ? n : 64;
  const int gwidth = ppw > 1 ? n : 64;

  T1Rec *rec = reinterpret_cast<T1Rec *>(sDyn);
  int *sBase = reinterpret_cast<int *>(sDyn + (size_t)a.npts_cap * 16);
  float *lT = sDyn + (size_t)a.npts_cap * 17;   // any-size form: three planes [npts_cap * n]
  float *lGx = lT + (size_t)a.npts_cap * n;
  float *lGy = lGx + (size_t)a.npts_cap * n;
  float *lTpl = lT;                              // 8x8 form: [point][T | Gx | Gy][64], one address + fixed offsets

  if (a.fused_begin) t1_fused_begin(e, a, b, tid, blockDim.x);
  const ProbState &gst = *t1_initial_state(e, a, b);
  const int npts = t1_uni(gst.npts);
  // Wave 0 is the solver: LU factors, pose and loop state stay in its registers for the whole tracking
  // (WaveSolver, ictr_devfn.h); the other waves see only cpos_G and the loop flag, through LDS.
  WaveSolver S;
  float G[12];
  S.p = lane < 6 ? gst.p[lane] : 0.0f;
  S.b = S.dp = S.h = 0.0f;
  S.total_iters = t1_uni(gst.total_iters);
  S.normdp = S.normdp_init = 1e-10f;
  S.it = 0;
  S.active = 0;
#pragma unroll
  for (int k = 0; k < 12; ++k) G[k] = t1_uni(gst.G[k]);
  if (tid < 12) sG[tid] = gst.G[tid];

  const float *__restrict__ p3 = e.pt3d + (size_t)b * 3 * M;
  const float *__restrict__ p3r = e.pt3d_ref + (size_t)b * 3 * M;
  float *gT = e.T + (size_t)b * M * n;
  float *gGx = e.Gx + (size_t)b * M * n;
  float *gGy = e.Gy + (size_t)b * M * n;
  float *coefb = e.coef + (size_t)b * M * kCoefStride;
  const int ngroups = (npts + ppw - 1) / ppw;
  const int mycnt = wave < npts ? (npts - wave + nwaves - 1) / nwaves : 0;  // 8x8: patches wave, wave + nwaves, ...

#ifdef ICTR_T1_PROF
  unsigned long long tp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = 0, t1_ = 0;
  if (tid == 0) t0_ = __builtin_readcyclecounter();
#endif
  for (int sl = e.lv_f; sl >= e.lv_l; --sl) {
    const LevelCam lc = a.lc[sl];
    const int sw = lc.sw;
    const PlaneSet pl = t1_planes(e, a, b, sl);
    // ---------------------------------------------------------------- level setup, stage A (one point per thread)
    {
      const float *pt2d = e.pt2d + ((size_t)b * e.nlev + sl) * 2 * M;
      for (int i = tid; i < npts; i += nthr) {
        const float mx = pt2d[i], my = pt2d[i + M];
        const bool vis = in_view(mx, my, lc.swo, lc.sho);
        float cx[6], cy[6];
        float4 *c4 = reinterpret_cast<float4 *>(coefb + (size_t)i * kCoefStride);
        if (vis) {
          sd_coefs(p3r[i], p3r[i + M], p3r[i + 2 * M], lc.fx, lc.fy, cx, cy);
          c4[0] = make_float4(cx[0], cx[1], cx[2], cx[3]);
          c4[1] = make_float4(cx[4], cx[5], cy[0], cy[1]);
          c4[2] = make_float4(cy[2], cy[3], cy[4], cy[5]);
        } else {  // stale coefficients stay in force (odometer.cpp:304); zeros if the point was never seen
          const float4 a0 = c4[0], a1 = c4[1], a2 = c4[2];
          cx[0] = a0.x; cx[1] = a0.y; cx[2] = a0.z; cx[3] = a0.w; cx[4] = a1.x; cx[5] = a1.y;
          cy[0] = a1.z; cy[1] = a1.w; cy[2] = a2.x; cy[3] = a2.y; cy[4] = a2.z; cy[5] = a2.w;
        }
        const Taps tp = make_taps(vis ? mx : 1.0f, vis ? my : 1.0f, pszd2);  // (1,1): a harmless in-plane window
        float4 *r4 = reinterpret_cast<float4 *>(&rec[i]);
        const int base = tp.row0 * sw + tp.col0;
        r4[0] = make_float4(tp.w0, tp.w1, tp.w2, tp.w3);
        r4[1] = make_float4(cx[0], cx[2], cx[3], cx[4]);
        r4[2] = make_float4(cx[5], cy[1], cy[2], cy[3]);
        r4[3] = make_float4(cy[4], cy[5], vis ? 1.0f : 0.0f, __builtin_bit_cast(float, base));
        sBase[i] = base;
      }
    }
    __syncthreads();
    T1_MARK(0)
    // ---------------------------------------------------------------- stage B (one patch per wave / 16-lane group)
    {
      float acc[kHUnique];
#pragma unroll
      for (int j = 0; j < kHUnique; ++j) acc[j] = 0.0f;
      for (int g = wave; g < ngroups; g += nwaves) {
        const int i = g * ppw + sub;
        const bool valid = i < npts;
        const T1Rec r = rec[valid ? i : 0];
        const bool vis = valid && r.vis != 0.0f;
        Taps tp;
        tp.w0 = r.w0; tp.w1 = r.w1; tp.w2 = r.w2; tp.w3 = r.w3;
        const int base = r.base;
        float cx[6], cy[6];
        cx[0] = valid ? r.cx0 : 0.0f; cx[1] = 0.0f; cx[2] = valid ? r.cx2 : 0.0f; cx[3] = valid ? r.cx3 : 0.0f;
        cx[4] = valid ? r.cx4 : 0.0f; cx[5] = valid ? r.cx5 : 0.0f;
        cy[0] = 0.0f; cy[1] = valid ? r.cy1 : 0.0f; cy[2] = valid ? r.cy2 : 0.0f; cy[3] = valid ? r.cy3 : 0.0f;
        cy[4] = valid ? r.cy4 : 0.0f; cy[5] = valid ? r.cy5 : 0.0f;
        float mean = 0.0f;
        if (e.dopatchnorm) {  // utilities.cpp:187-188 : intensity patch only
          float s = 0.0f;
          for (int q = q0; q < n; q += qstride)
            if (vis) s += tap4(pl.ref, base + (q / P) * sw + (q % P), sw, tp);
          s = group_sum(s, gwidth);
          mean = s / (float)n;
        }
        for (int q = q0; q < n; q += qstride) {
          float t = 0.0f, gx = 0.0f, gy = 0.0f;
          const size_t o = (size_t)i * n + q;
          if (vis) {
            const int idx = base + (q / P) * sw + (q % P);
            t = tap4(pl.ref, idx, sw, tp);
            if (e.dopatchnorm) t -= mean;
            gx = tap4(pl.dx, idx, sw, tp);
            gy = tap4(pl.dy, idx, sw, tp);
            gT[o] = t;
            gGx[o] = gx;
            gGy[o] = gy;
          } else if (valid) {
            if (e.robust & ICTR_ROBUST_CLEAN) {  // option: no stale contributions
              gGx[o] = 0.0f;
              gGy[o] = 0.0f;
            } else {
              gx = gGx[o];
              gy = gGy[o];
            }
            if (TL) t = gT[o];
          }
          if (TL && valid) {
            lT[o] = t;
            lGx[o] = gx;
            lGy[o] = gy;
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
      T1_REDUCE_STORE(kHUnique, acc, sPart[wave])
    }
    __syncthreads();
    T1_MARK(1)
    if (wave == 0) {  // odometer.cpp:457-471 + 509-515: waves added in a fixed order in f64, mirrored, factored once
      double hs = 0.0;
      if (lane < kHUnique)
        for (int w = 0; w < nwaves; ++w) hs += (double)sPart[w][lane];
      const int r = lane / 6, c = lane - 6 * r;
      const int lo = r < c ? r : c, hi = r < c ? c : r;
      const int j = lane < 36 ? lo * 6 - lo * (lo - 1) / 2 + (hi - lo) : 0;
      ws_factor(S, lane_gather((float)hs, j), lane);
      ws_level_reset(S, solve_opts(e));
      if (lane == 0) sActive = S.active;
    }
    __syncthreads();
    T1_MARK(2)
    // ---------------------------------------------------------------- Gauss-Newton iterations of this level
    const float *__restrict__ cur = pl.cur;
    while (sActive) {  // workgroup-uniform: read after a barrier, rewritten only between barriers
      // stage 1: projection at the current pose (pose.cpp:384-391), ind_new (odometer.cpp:369-377)
      {
        float Gc[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) Gc[k] = sG[k];
        for (int i = tid; i < npts; i += nthr) {
          const float X = p3[i], Y = p3[i + M], Z = p3[i + 2 * M];
          const float tx = Gc[0] * X + Gc[1] * Y + Gc[2] * Z + Gc[3];
          const float ty = Gc[4] * X + Gc[5] * Y + Gc[6] * Z + Gc[7];
          const float tz = Gc[8] * X + Gc[9] * Y + Gc[10] * Z + Gc[11];
          const float mx = (tx / tz) * lc.fx + lc.cx;
          const float my = (ty / tz) * lc.fy + lc.cy;
          const bool vis = in_view(mx, my, lc.swo, lc.sho);
          const Taps tp = make_taps(vis ? mx : 1.0f, vis ? my : 1.0f, pszd2);
          const int base = tp.row0 * sw + tp.col0;
          T1Rec &r = rec[i];
          *reinterpret_cast<float4 *>(&r) = make_float4(tp.w0, tp.w1, tp.w2, tp.w3);
          r.vis = vis ? 1.0f : 0.0f;
          r.base = base;
          sBase[i] = base;
        }
      }
      __syncthreads();
      T1_MARK(3)  // stage 1 + barrier
      // stage 2
      float acc[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] = 0.0f;
      for (int g = wave; g < ngroups; g += nwaves) {
        const int i = g * ppw + sub;
        const bool valid = i < npts;
        const T1Rec r = rec[valid ? i : 0];
        const bool vis = valid && r.vis != 0.0f;
        Taps tp;
        tp.w0 = r.w0; tp.w1 = r.w1; tp.w2 = r.w2; tp.w3 = r.w3;
        const int base = r.base;
        float cx[6], cy[6];
        cx[0] = r.cx0; cx[1] = 0.0f; cx[2] = r.cx2; cx[3] = r.cx3; cx[4] = r.cx4; cx[5] = r.cx5;
        cy[0] = 0.0f; cy[1] = r.cy1; cy[2] = r.cy2; cy[3] = r.cy3; cy[4] = r.cy4; cy[5] = r.cy5;
        float mean = 0.0f;
        if (e.dopatchnorm) {  // utilities.cpp:111-112
          float s = 0.0f;
          for (int q = q0; q < n; q += qstride)
            if (vis) s += tap4(cur, base + (q / P) * sw + (q % P), sw, tp);
          s = group_sum(s, gwidth);
          mean = s / (float)n;
        }
        for (int q = q0; q < n; q += qstride) {
          if (vis) {
            const size_t o = (size_t)i * n + q;
            float inew = tap4(cur, base + (q / P) * sw + (q % P), sw, tp);
            if (e.dopatchnorm) inew -= mean;
            float rr = (TL ? lT[o] : gT[o]) - inew;  // pdiff (odometer.cpp:381)
            if (e.robust & ICTR_ROBUST_HUBER) {
              const float ar = fabsf(rr);
              if (ar > e.huber_k) rr *= e.huber_k / ar;
            }
            float sd[6];
            sd_values(TL ? lGx[o] : gGx[o], TL ? lGy[o] : gGy[o], cx, cy, sd);
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k] += sd[k] * rr;  // sd*_proj summed (odometer.cpp:386-404)
          }
        }
      }
      T1_MARK(4)  // stage 2 of wave 0
      T1_REDUCE_STORE(6, acc, sPart[wave])
      __syncthreads();
      T1_MARK(5)  // wave reduction + waiting for the other waves
      if (wave == 0) {  // steps 9a (final sum, fixed order, f64) - 10, loop condition
        double bs = 0.0;
        if (lane < 6)
          for (int w = 0; w < nwaves; ++w) bs += (double)sPart[w][lane];
        ws_iterate(S, (float)bs, solve_opts(e), sl, b, lane, G);
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < 12; ++k) sG[k] = G[k];
          sActive = S.active;
        }
      }
      T1_MARK(6)  // final sum + solve + update
      __syncthreads();
    }
  }
  if (wave == 0) {  // final state back to the problem's record, and straight to the host's pinned mirror if there is one
    t1_store_final(e.st[b], S, G, lane);
    if (a.host_st) {
      t1_store_final(a.host_st[b], S, G, lane);
      if (lane == 0) a.host_st[b].npts = npts;
    }
  }
#ifdef ICTR_T1_PROF
  if (tid == 0)
    for (int k = 0; k < 8; ++k) e.partH[(size_t)b * 8 + k] = (float)tp_[k];
#endif
}

// ---------------------------------------------------------------- Teams: several workgroups per problem
// Between the one-workgroup range (<= ~200 points) and the sizes where the per-iteration kernels carry real work the
// tracker is a chain of ~111 dependent 2-us launches (0.5 ms whatever the point count), and a batch of mid-size
// problems (run_track_nposes with a few hundred points) keeps one CU busy per problem for 10 us per iteration. In the
// team form `team` workgroups (consecutive block ids) share one problem: workgroup `part` owns the points
// [part q, part q + q) for the whole tracking -- its own share of patches, records and partial sums, nothing of it is
// ever read by another workgroup -- and the only exchange is the one the multi-GPU form has: the 21
// sums of H once per level and the 6 of b once per iteration, all-gathered through a mailbox in device memory; every
// workgroup then adds the parts in part order and runs the same solver turn on the same bits, so the redundant poses and
// loop conditions stay in lockstep without a second hop. A value travels as ONE naturally aligned 8-byte granule
// {float bits, tag} written by one agent-scope store and polled on its tag (ictr_p2p.hip's protocol; no flag, no fence:
// MI355X_MICROARCH.md "Valid forms", R2); tags = launch epoch << 12 | exchange number, so nothing is cleared between
// launches; slots are double-buffered by the exchange number's parity (a workgroup can be at most one exchange ahead of
// its slowest peer, which still owes it that exchange's granules). Polling is bounded by a wall-clock limit: on a
// time-out the workgroup raises a sticky flag in pinned host memory and stops waiting for good, so every wave reaches
// the end of the kernel (the host reports the tracking as failed). Progress needs the team's workgroups resident
// together: they are consecutive block ids of an in-order dispatch, so the lowest unfinished problem always holds its CUs.
struct TeamCtx {
  unsigned long long *mail;  // this problem's mailbox [2][team][kTeamSlot]
  int team, part;
  unsigned tag0, seq;        // seq: exchanges done so far in this launch
  unsigned long long limit;
  int *err;
  int dead;                  // a poll timed out: never wait again
  int mute;                  // debug: this workgroup never posts (time-out test)
};
__device__ __forceinline__ double lane_gather64(double v, int src_lane) {
  const int lo = lane_gather(__double2loint(v), src_lane), hi = lane_gather(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
// Wave 0 only. Lane k < N passes the workgroup's k-th value; lanes k < N (of the first LPP) receive the team's total in
// f64. LPP = lanes per part in a polling sweep (a power of two >= N): lane (rr, k) reads value k of the parts rr,
// rr + 64/LPP, ... -- eight granule loads in flight per lane, so a team of up to 8 * 64/LPP parts costs ONE round trip
// -- and adds them in that order; the 64/LPP lane groups are then added in group order. The same fixed order in every
// workgroup: the same bits everywhere.
template <int N, int LPP>
__device__ __forceinline__ double team_allsum(TeamCtx &c, float v, int lane) {
  c.seq += 1;
  const unsigned tag = c.tag0 + c.seq;
  unsigned long long *slot = c.mail + (size_t)(c.seq & 1u) * c.team * kTeamSlot;
  if (lane < N && !c.mute)
    __hip_atomic_store(slot + c.part * kTeamSlot + lane,
                       ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  constexpr int PPP = 64 / LPP;  // parts per sweep
  constexpr int CH = 8;          // sweeps in flight
  const int k = lane & (LPP - 1), rr = lane / LPP;
  const unsigned long long empty = (unsigned long long)tag << 32;  // "arrived, value +0": lanes without a granule
  double acc = 0.0;
  for (int r0 = 0; r0 < c.team; r0 += CH * PPP) {
    unsigned long long g[CH];
    const unsigned long long *src[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int r = r0 + u * PPP + rr;
      const bool mine = k < N && r < c.team;
      src[u] = slot + (size_t)(mine ? r : 0) * kTeamSlot + (mine ? k : 0);
      g[u] = empty;
      if (mine) g[u] = __hip_atomic_load(src[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!c.dead) {
      bool started = false;
      unsigned long long t0 = 0;
      for (;;) {
        bool miss = false;
#pragma unroll
        for (int u = 0; u < CH; ++u) miss |= (unsigned)(g[u] >> 32) != tag;
        if (__builtin_amdgcn_ballot_w64(miss) == 0) break;  // wave-uniform
        if (!started) {
          t0 = wall_clock64();
          started = true;
        } else if (wall_clock64() - t0 > c.limit) {  // a peer never arrived: flag it, never wait again
          if (lane == 0) __hip_atomic_store(c.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          c.dead = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int u = 0; u < CH; ++u)
          if ((unsigned)(g[u] >> 32) != tag) g[u] = __hip_atomic_load(src[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) acc += (double)__builtin_bit_cast(float, (unsigned)(g[u] & 0xffffffffu));
  }
  double tot = acc;
#pragma unroll
  for (int j = 1; j < PPP; ++j) tot += lane_gather64(acc, j * LPP + k);  // (meaningful in the lanes of group 0)
  return tot;
}

// ================================================================ 8x8 patches: the lean form
// A wave owns the points wave, wave + nwaves, ... for the whole tracking, one POINT per lane in "stage 1" (chunks of
// 64) and one PATCH per step in "stage 2" (lane == pixel). What stage 1 computes for a point (bilinear weights,
// window base, visibility; at level setup also the 10 non-zero steepest-descent coefficients) stays in that lane's
// registers and reaches stage 2 through v_readlane: no LDS record, no barrier between the two stages and no LDS
// latency on the per-patch path. The four taps of a pixel are two 8-byte loads per lane (rows y and y-1; neighbouring
// lanes' requests coalesce in the L1), so the blend needs no cross-lane traffic. Per Gauss-Newton iteration the
// workgroup meets at two barriers: partial sums ready, new pose ready.
struct T8Pt {  // stage-1 results of the lane's point
  float w0, w1, w2, w3;
  int base;  // element index of the window's top-left texel (tap d of pixel 0): (row0 - 1) * sw + col0 - 1
  int vis;
};
struct T8Coef {  // the 10 steepest-descent coefficients that are not identically zero (odometer.cpp:313-326)
  float cx0, cx2, cx3, cx4, cx5, cy1, cy2, cy3, cy4, cy5;
};
struct T8Win {
  f32x2_a4 ab, cd;  // (x-1,y),(x,y) and (x-1,y-1),(x,y-1)
};
__device__ __forceinline__ T8Win t8_issue(gconst_f32 plane, int base, unsigned off_cd, unsigned off_ab) {
  gconst_bytes p = reinterpret_cast<gconst_bytes>(plane + base);  // wave-uniform base + 32-bit per-lane offsets
  T8Win w;
  w.cd = *reinterpret_cast<gconst_f32x2>(p + off_cd);
  w.ab = *reinterpret_cast<gconst_f32x2>(p + off_ab);
  return w;
}
// utilities.cpp:107: a=(col,row) b=(col-1,row) c=(col,row-1) d=(col-1,row-1); reference operand order, not contracted
__device__ __forceinline__ float t8_blend(const T8Win &w, float w0, float w1, float w2, float w3) {
  return w0 * w.ab.y + w1 * w.ab.x + w2 * w.cd.y + w3 * w.cd.x;
}
template <int kU> struct T8Loads {
  T8Win cur[kU];
  float t[kU], gx[kU], gy[kU];
  int k[kU];  // patch index inside the chunk, -1 = padding of a partial step (wave-uniform)
};
template <int kU> struct T8RefLoads {
  T8Win r[kU], x[kU], y[kU];  // visible patch: the three reference planes' windows
  int k[kU];                  // bit 8 set: out of the reference view at this level (stale patch re-used)
};
__device__ __forceinline__ float rl(float v, int k) {  // v_readlane with a wave-uniform run-time lane index
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}
__device__ __forceinline__ int rl(int v, int k) { return __builtin_amdgcn_readlane(v, k); }

// LEAN: built for 128 registers (four waves per SIMD: TWO workgroups share a CU), one patch per step in the level setup,
// ~30 spilled registers. A batch of more problems than the chip has CUs runs 20 % faster this way (500 x 60 points:
// 0.64 -> 0.50 ms); a batch that leaves CUs idle anyway is 5 % faster with the 206-register build (LEAN = false).
// Same operations in the same order: the two builds give the same bits.
// TEAM: a.team workgroups per problem ("Teams" above); false: one workgroup per problem, no exchange code at all.
template <bool TL, bool PN, bool LEAN, bool TEAM>
__global__ __launch_bounds__(64 * kT8MaxWaves, LEAN ? 4 : 2) void k_track1_p8(EngineDev e, T1Args a) {
  extern __shared__ __attribute__((aligned(16))) float sDyn[];
  __shared__ float sPart[kT8MaxWaves][kPartHStride];  // per-wave partial sums (21 of H, or 6 of b)
  // The problem's state (pose, LU factors of the level, loop condition) lives in LDS between the solver's turns: the
  // solver's ~30 registers are then live in wave 0's tail only, not across every wave's patch loops (the kernel as a
  // whole stays under 128 VGPRs, so two workgroups -- or sixteen waves -- share a CU). Same helpers, same arithmetic as
  // the per-iteration tails (k_level_tail / k_iter_tail), which keep the state in device memory.
  __shared__ ProbState sSt;

  const int team = TEAM ? a.team : 1;
  const int b = TEAM ? (int)blockIdx.x / team : (int)blockIdx.x;
  const int part = TEAM ? (int)blockIdx.x - b * team : 0;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int M = e.M;
  // Wave-major LDS layout: wave w owns the slots [w cap_w, (w + 1) cap_w); its j-th point (i = w + j nwaves) is slot
  // w cap_w + j. Per slot one 64-byte record [w1 w0 w3 w2][cx2 cx3 cx4 cx5][cy2 cy3 cy4 cy5][cx0 cy1 vis -] -- the
  // weights and the visibility flag rewritten by the point's own lane in every iteration's stage 1, the ten non-zero
  // steepest-descent coefficients once per level -- and (TL) the patch [T | Gx | Gy][64]. Stage 2 walks a wave's slots
  // with ONE running address each: four broadcast ds_read_b128 and three ds_read_b32 at immediate offsets per patch, no
  // index arithmetic. The slot behind a wave's last point stays zero (weights, flag, patch): the patch loop runs in
  // whole pipeline steps of kUi patches without a per-patch bounds check.
  const int cap_w = a.cap_w;
  float4 *lRec = reinterpret_cast<float4 *>(sDyn);                      // [nwaves * cap_w][4]
  float *lTpl = sDyn + (size_t)nwaves * cap_w * 16;                    // [nwaves * cap_w][T | Gx | Gy][64] when TL
  const int slot0 = wave * cap_w;

  if (a.fused_begin) {
    if constexpr (TEAM)
      t1_fused_begin(e, a, b, tid, blockDim.x, part, a.team_q);
    else
      t1_fused_begin(e, a, b, tid, blockDim.x);
  }
  const ProbState &gst = *t1_initial_state(e, a, b);
  // this workgroup's points: [lo, lo + npts) of the problem's (TEAM; else all of them). Every per-point array below is
  // addressed relative to lo.
  const int npts_all = t1_uni(gst.npts);
  const int lo = TEAM ? min(part * a.team_q, npts_all) : 0;
  const int npts = TEAM ? min(a.team_q, npts_all - lo) : npts_all;
  TeamCtx tc;
  if constexpr (TEAM) {
    tc.mail = a.team_mail + (size_t)b * 2 * team * kTeamSlot;
    tc.team = team;
    tc.part = part;
    tc.tag0 = a.team_tag0;
    tc.seq = 0;
    tc.limit = a.team_limit;
    tc.err = a.team_err;
    tc.dead = 0;
    tc.mute = (a.team_mute != 0 && part + 1 == a.team_mute) ? 1 : 0;
  }
  SolveOpts sopt = solve_opts(e);
  sopt.robust = 0;  // the host routes every behaviour-changing option to the any-size form: no compose / log code here
  {
    const unsigned *src = reinterpret_cast<const unsigned *>(&gst);
    unsigned *dst = reinterpret_cast<unsigned *>(&sSt);
    for (int i = tid; i < (int)(sizeof(ProbState) / 4); i += blockDim.x) dst[i] = src[i];
  }  // (first read by wave 0 behind the level setup's barrier)
  {  // records and patches start as zeros: what the padding slots stay for the whole tracking
    const int nz = nwaves * cap_w * (TL ? 16 + 192 : 16);
    for (int i = tid; i < nz; i += blockDim.x) sDyn[i] = 0.0f;
    __syncthreads();
  }

  const float *__restrict__ p3 = e.pt3d + (size_t)b * 3 * M + lo;
  const float *__restrict__ p3r = e.pt3d_ref + (size_t)b * 3 * M + lo;
  float *gT = e.T + ((size_t)b * M + lo) * 64;
  float *gGx = e.Gx + ((size_t)b * M + lo) * 64;
  float *gGy = e.Gy + ((size_t)b * M + lo) * 64;
  float *coefb = e.coef + ((size_t)b * M + lo) * kCoefStride;
  const int mycnt = wave < npts ? (npts - wave + nwaves - 1) / nwaves : 0;  // points wave, wave + nwaves, ...
  float X1 = 0.0f, Y1 = 0.0f, Z1 = 1.0f;  // the lane's point when the wave has a single chunk
  if (mycnt > 0 && mycnt <= 64) {
    const int i1 = wave + (lane < mycnt ? lane : 0) * nwaves;
    X1 = p3[i1], Y1 = p3[i1 + M], Z1 = p3[i1 + 2 * M];
  }
#ifdef ICTR_T1_PROF
  unsigned long long tp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = 0, t1_ = 0;
  if (tid == 0) t0_ = __builtin_readcyclecounter();
  const int dbg = a.dbg;
#endif

  for (int sl = e.lv_f; sl >= e.lv_l; --sl) {
    const LevelCam lc = a.lc[sl];
    const int sw = lc.sw;
    const PlaneSet pl = t1_planes(e, a, b, sl);
    const unsigned off_cd = (unsigned)((lane >> 3) * sw + (lane & 7)) * 4u;  // bytes from the window's top-left texel
    const unsigned off_ab = off_cd + (unsigned)sw * 4u;
    constexpr int kU = LEAN ? 1 : 2;  // patches per pipeline step of the level setup (2: +24 registers)
    constexpr int kUi = 2;  // ... of the iterations (4: 256 VGPRs + spills, no gain; measured r02)
    // ---------------------------------------------------------------- level setup (odometer.cpp:268-334, 428-472)
    {
      float acc[kHUnique];
#pragma unroll
      for (int j = 0; j < kHUnique; ++j) acc[j] = 0.0f;
      gconst_f32 pref = (gconst_f32)pl.ref, pdx = (gconst_f32)pl.dx, pdy = (gconst_f32)pl.dy;
      const float *pt2d = e.pt2d + ((size_t)b * e.nlev + sl) * 2 * M + lo;
      for (int c0 = 0; c0 < mycnt; c0 += 64) {
        const int cn = min(64, mycnt - c0);
        // stage A: lane k <-> point i = wave + (c0 + k) nwaves
        const bool pv = lane < cn;
        const int ip = wave + (c0 + (pv ? lane : 0)) * nwaves;
        const float mx = pt2d[ip], my = pt2d[ip + M];
        const bool vis = pv && in_view(mx, my, lc.swo, lc.sho);
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
        const Taps tp = make_taps(vis ? mx : 1.0f, vis ? my : 1.0f, 4);  // (1,1): a harmless in-plane window
        const int base_v = (tp.row0 - 1) * sw + tp.col0 - 1;
        const int vis_v = vis ? 1 : 0;
        float4 *const recs = lRec + (size_t)(slot0 + c0) * 4;  // this chunk's records
        if (pv) {  // the level's coefficients into the point's record; for stage B below also the REFERENCE window's
                   // weights (slot 0: every iteration's stage 1 rewrites it, and the flag in slot 3, for the current frame)
          float4 *r4 = recs + lane * 4;
          r4[0] = make_float4(tp.w0, tp.w1, tp.w2, tp.w3);
          r4[1] = make_float4(cx[2], cx[3], cx[4], cx[5]);
          r4[2] = make_float4(cy[2], cy[3], cy[4], cy[5]);
          r4[3] = make_float4(cx[0], cy[1], 0.0f, 0.0f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto issue = [&](T8RefLoads<kU> &L, int k) {
#pragma unroll
          for (int u = 0; u < kU; ++u) {
            const bool ok = k + u < cn;
            const int kk = ok ? k + u : k;
            const int v = rl(vis_v, kk);
            L.k[u] = ok ? (v ? kk : kk | 256) : -1;
            if (v) {  // wave-uniform
              const int base = rl(base_v, kk);
              L.r[u] = t8_issue(pref, base, off_cd, off_ab);
              L.x[u] = t8_issue(pdx, base, off_cd, off_ab);
              L.y[u] = t8_issue(pdy, base, off_cd, off_ab);
            }
          }
        };
        auto reduce = [&](const T8RefLoads<kU> &L) {
#pragma unroll
          for (int u = 0; u < kU; ++u) {
            if (L.k[u] < 0) continue;  // wave-uniform
            const int kk = L.k[u] & 255;
            const int i = wave + (c0 + kk) * nwaves;
            const int o = i * 64 + lane;
            float t, gx, gy;
            const float4 *r4 = recs + kk * 4;  // broadcast reads: one address, immediate offsets
            const float4 qx = r4[1], qy = r4[2], qz = r4[3];
            if (!(L.k[u] & 256)) {
              const float4 w = r4[0];
              t = t8_blend(L.r[u], w.x, w.y, w.z, w.w);
              gx = t8_blend(L.x[u], w.x, w.y, w.z, w.w);
              gy = t8_blend(L.y[u], w.x, w.y, w.z, w.w);
              if constexpr (PN) t -= wave_sum(t) / 64.0f;  // utilities.cpp:187-188 (same order as k_ref8)
              gT[o] = t;
              gGx[o] = gx;
              gGy[o] = gy;
            } else {  // out of the reference view: the stale patch stays in force (odometer.cpp:304)
              t = TL ? gT[o] : 0.0f;
              gx = gGx[o];
              gy = gGy[o];
            }
            if constexpr (TL) {
              float *d = lTpl + (slot0 + c0 + kk) * 192 + lane;
              d[0] = t;
              d[64] = gx;
              d[128] = gy;
            }
            {
              // H = sum sd_j sd_k is compared to tolerance only, so multiply-adds may be fused -- but fused HERE, by
              // hand, not wherever a build's scheduling happens to allow it: both register budgets of this kernel
              // (LEAN or not) then produce the same bits
              float sd[6];
              sd[0] = gx * qz.x;
              sd[1] = gy * qz.y;
              sd[2] = __builtin_fmaf(gx, qx.x, gy * qy.x);
              sd[3] = __builtin_fmaf(gx, qx.y, gy * qy.y);
              sd[4] = __builtin_fmaf(gx, qx.z, gy * qy.z);
              sd[5] = __builtin_fmaf(gx, qx.w, gy * qy.w);
              int jk = 0;
#pragma unroll
              for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int q = j; q < 6; ++q, ++jk) acc[jk] = __builtin_fmaf(sd[j], sd[q], acc[jk]);
            }
          }
        };
        T8RefLoads<kU> A, B;
        issue(A, 0);
        for (int k = 0; k < cn; k += 2 * kU) {
          if (k + kU < cn) issue(B, k + kU);
          reduce(A);
          if (k + 2 * kU < cn) issue(A, k + 2 * kU);
          if (k + kU < cn) reduce(B);
        }
      }
      T1_MARK(0)  // setup: stage A + patches of wave 0
      T1_REDUCE_STORE(kHUnique, acc, sPart[wave])
    }
    __syncthreads();
    T1_MARK(1)  // H wave reduction + barrier
    if (wave == 0) {  // odometer.cpp:457-471 + 509-515: waves added in a fixed order in f64, mirrored, factored once
      double hs = 0.0;
      if (lane < kHUnique)
        for (int w = 0; w < nwaves; ++w) hs += (double)sPart[w][lane];
      const int r = lane / 6, c = lane - 6 * r;
      const int lo = r < c ? r : c, hi = r < c ? c : r;
      const int j = lane < 36 ? lo * 6 - lo * (lo - 1) / 2 + (hi - lo) : 0;
      float hv = (float)hs;
      if constexpr (TEAM) hv = (float)team_allsum<kHUnique, 32>(tc, hv, lane);
      WaveSolver S;  // wave 0 is the solver (ictr_devfn.h)
      ws_factor(S, lane_gather(hv, j), lane);
      ws_store_factor(S, sSt, lane);
      if (lane == 0) level_reset(sSt, e);  // odometer.cpp:341-346
    }
    __syncthreads();
    T1_MARK(2)  // H sum + LU

    // ---------------------------------------------------------------- Gauss-Newton iterations of this level
    // the current frame through a buffer descriptor: wave-uniform window base in an SGPR offset, per-lane constant
    // byte offsets in a VGPR -> no vector address arithmetic per load
    const __amdgpu_buffer_rsrc_t rcur =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pl.cur), 0, 0x7fffffff, 0x00020000);
    const bool single = mycnt <= 64;
    while (sSt.active) {  // workgroup-uniform: read after a barrier, rewritten only between barriers
      float acc[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] = 0.0f;
      float Gc[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) Gc[k] = sSt.G[k];
      for (int c0 = 0; c0 < mycnt; c0 += 64) {
        const int cn = min(64, mycnt - c0);
        // stage 1: projection at the current pose (pose.cpp:384-391), ind_new (odometer.cpp:369-377)
        const bool pv = lane < cn;
        const int ip = wave + (c0 + (pv ? lane : 0)) * nwaves;
        // a wave with at most 64 points (up to 512 points per problem) keeps each point's X, Y, Z in its lane's
        // registers for the whole tracking: no memory round trip in front of the projection
        float X = X1, Y = Y1, Z = Z1;
        if (!single) X = p3[ip], Y = p3[ip + M], Z = p3[ip + 2 * M];
        const float tx = Gc[0] * X + Gc[1] * Y + Gc[2] * Z + Gc[3];
        const float ty = Gc[4] * X + Gc[5] * Y + Gc[6] * Z + Gc[7];
        const float tz = Gc[8] * X + Gc[9] * Y + Gc[10] * Z + Gc[11];
        const float mx = (tx / tz) * lc.fx + lc.cx;
        const float my = (ty / tz) * lc.fy + lc.cy;
        const bool vis = pv && in_view(mx, my, lc.swo, lc.sho);
        const Taps tp = make_taps(vis ? mx : 1.0f, vis ? my : 1.0f, 4);  // (1,1): a harmless in-plane window
        const int base_v = ((tp.row0 - 1) * sw + tp.col0 - 1) * 4;  // bytes: the buffer load's scalar offset
        float4 *const recs = lRec + (size_t)(slot0 + c0) * 4;  // this chunk's records
        if (pv) {
          recs[lane * 4] = make_float4(tp.w1, tp.w0, tp.w3, tp.w2);
          reinterpret_cast<float *>(recs + lane * 4 + 3)[2] = vis ? 1.0f : 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        T1_MARK(3)  // stage 1
        // stage 2: kUi patches per step, software-pipelined. A step is never split: the slot behind an odd chunk's last
        // point is padding (zero record, zero patch -> an exact zero contribution; its lane's window is the harmless
        // in-plane one of stage 1), so no patch checks bounds
        const float *const tpls = lTpl + (size_t)(slot0 + c0) * 192 + lane;
        auto issue = [&](T8Loads<kUi> &L, int k) {
#pragma unroll
          for (int u = 0; u < kUi; ++u) {
            const int soff = rl(base_v, k + u);
#ifdef ICTR_T1_PROF
            if (dbg & 2) {  // ablation: no current-frame loads
              L.cur[u].cd = f32x2_a4{1.0f + lane, 2.0f};
              L.cur[u].ab = f32x2_a4{3.0f, 4.0f + k};
            } else
#endif
            {
              L.cur[u].cd = __builtin_bit_cast(f32x2_a4, __builtin_amdgcn_raw_buffer_load_b64(rcur, (int)off_cd, soff, 0));
              L.cur[u].ab = __builtin_bit_cast(f32x2_a4, __builtin_amdgcn_raw_buffer_load_b64(rcur, (int)off_ab, soff, 0));
            }
            if constexpr (TL) {
              const float *tpl = tpls + (k + u) * 192;
              L.t[u] = tpl[0];
              L.gx[u] = tpl[64];
              L.gy[u] = tpl[128];
            } else {  // patches from device memory: a padding slot re-reads the chunk's last patch (its record is zero)
              const int o = (wave + (c0 + min(k + u, cn - 1)) * nwaves) * 64 + lane;
              L.t[u] = gT[o];
              L.gx[u] = gGx[o];
              L.gy[u] = gGy[o];
            }
          }
        };
        auto reduce = [&](const T8Loads<kUi> &L, int k) {
          float4 wv[kUi], qx[kUi], qy[kUi], qz[kUi];
#pragma unroll
          for (int u = 0; u < kUi; ++u) {  // the patches' scalars: broadcast LDS reads, all in flight together
            const float4 *r4 = recs + (k + u) * 4;
            wv[u] = r4[0];
            qx[u] = r4[1];  // cx2 cx3 cx4 cx5
            qy[u] = r4[2];  // cy2 cy3 cy4 cy5
            qz[u] = r4[3];  // cx0 cy1 vis -
          }
#pragma unroll
          for (int u = 0; u < kUi; ++u) {
            // utilities.cpp:107 in the reference's operand order, not contracted: ((w0 a + w1 b) + w2 c) + w3 d
            float inew = wv[u].y * L.cur[u].ab.y + wv[u].x * L.cur[u].ab.x + wv[u].w * L.cur[u].cd.y + wv[u].z * L.cur[u].cd.x;
            if constexpr (PN) inew -= wave_sum_dpp(inew) / 64.0f;  // utilities.cpp:111-112
            const float r = (L.t[u] - inew) * qz[u].z;  // pdiff (odometer.cpp:381); 0 out of the new view
            {  // the J^T r sums are compared to tolerance only: explicit multiply-adds (see the level setup)
              const float gr = L.gx[u] * r, hr = L.gy[u] * r;
              acc[0] = __builtin_fmaf(gr, qz[u].x, acc[0]);                                  // sd1 = Gx cx0
              acc[1] = __builtin_fmaf(hr, qz[u].y, acc[1]);                                  // sd2 = Gy cy1
              acc[2] = __builtin_fmaf(gr, qx[u].x, __builtin_fmaf(hr, qy[u].x, acc[2]));     // sd3..sd6 = Gx cxk + Gy cyk
              acc[3] = __builtin_fmaf(gr, qx[u].y, __builtin_fmaf(hr, qy[u].y, acc[3]));     // (odometer.cpp:319-326)
              acc[4] = __builtin_fmaf(gr, qx[u].z, __builtin_fmaf(hr, qy[u].z, acc[4]));
              acc[5] = __builtin_fmaf(gr, qx[u].w, __builtin_fmaf(hr, qy[u].w, acc[5]));
            }
          }
        };
        T8Loads<kUi> A, B;
        issue(A, 0);
        for (int k = 0; k < cn; k += 2 * kUi) {  // step-level guards only (scalar compare + branch per kUi patches)
          if (k + kUi < cn) issue(B, k + kUi);
          reduce(A, k);
          if (k + 2 * kUi < cn) issue(A, k + 2 * kUi);
          if (k + kUi < cn) reduce(B, k + kUi);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // X, Y, Z and the records of the next chunk
      }
      T1_MARK(4)  // stage 2 of wave 0
      T1_REDUCE_STORE(6, acc, sPart[wave])
      __syncthreads();
      T1_MARK(5)  // wave reduction + barrier
      if (wave == 0) {  // steps 9a (final sum, fixed order, f64) - 10, loop condition
        double bs = 0.0;
        if (lane < 6)
          for (int w = 0; w < nwaves; ++w) bs += (double)sPart[w][lane];
        float bv = (float)bs;
        if constexpr (TEAM) {
          // b is a sum of signed terms: a share's partial can be much larger than the total, so it travels as a
          // (hi, lo) pair of floats (lanes 0-5 / 6-11) and the team's total is rounded ONCE -- the sum of all waves'
          // partials in f64, as accurate as a single workgroup's
          const float lo = (float)(bs - (double)bv);
          const float pv = lane < 6 ? bv : lane_gather(lo, lane - 6);
          const double t = team_allsum<12, 16>(tc, pv, lane);
          bv = (float)(t + lane_gather64(t, lane + 6));
        }
        WaveSolver S;
        float G[12];
        ws_load_state(S, sSt, lane, G);
        ws_load_factor(S, sSt, lane);
        ws_iterate(S, bv, sopt, sl, TEAM ? b + part : b, lane, G);  // (trace: problem 0's part 0 only)
        ws_store_state(S, sSt, lane, G);
      }
      T1_MARK(6)  // final sum + solve + update
      __syncthreads();
      T1_MARK(7)
    }
  }
#ifdef ICTR_T1_PROF
  if (tid == 0) {
    for (int k = 0; k < 8; ++k) e.partH[(size_t)b * 8 + k] = (float)tp_[k];
  }
#endif
  if (wave == 0 && part == 0) {  // final state back to the problem's record, and straight to the host's pinned mirror
    const unsigned *src = reinterpret_cast<const unsigned *>(&sSt);
    unsigned *dst = reinterpret_cast<unsigned *>(e.st + b);
    unsigned *hst = reinterpret_cast<unsigned *>(a.host_st ? a.host_st + b : nullptr);
    for (int i = lane; i < (int)(sizeof(ProbState) / 4); i += 64) {
      const unsigned v = src[i];
      dst[i] = v;
      if (hst) hst[i] = v;
    }
  }
}

// ---------------------------------------------------------------- host-side launcher
// LDS a workgroup needs beyond the static part; tmpl_lds is switched off when the templates do not fit.
// slots per wave of the 8x8 form: the wave's share of the points, rounded up to whole pipeline steps of two patches
static int track1_cap_w(int npts_cap, int waves) { return ((npts_cap + waves - 1) / waves + 1) & ~1; }
size_t track1_plan(int npts_cap, int n, int p8, int waves, int *tmpl_lds) {
  const size_t slots = p8 ? (size_t)waves * track1_cap_w(npts_cap, waves) : (size_t)npts_cap;
  const size_t recb = slots * (p8 ? 64 : 68);  // 8x8: one 64-byte record per slot; else 64-byte record + base
  const size_t tmplb = slots * n * 3 * sizeof(float);
  const size_t budget = 140 * 1024;  // of the CU's 160 KB; the static arrays (sRecW, partial sums, state) take ~18 KB
  *tmpl_lds = (recb + tmplb <= budget) ? 1 : 0;
  return recb + (*tmpl_lds ? tmplb : 0);
}

template <typename K>
static hipError_t launch_t1(K kernel, size_t *granted, const EngineDev &e, const T1Args &a, int waves, size_t lds,
                            hipStream_t s, int team = 1) {
  if (lds > *granted) {  // dynamic LDS above 64 KB must be granted per function
    const size_t want = 141 * 1024;
    hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
    if (rc != hipSuccess) return rc;
    *granted = want;
  }
  hipLaunchKernelGGL(kernel, dim3(e.B * team), dim3(64 * waves), lds, s, e, a);
  return hipGetLastError();
}

// bytes of initial state + plane table a launch can carry in its arguments (fused begin, see T1Args)
size_t track1_blob_bytes(void) { return sizeof(unsigned) * kT1BlobWords; }

// Points per workgroup of the team form for a problem of `maxpts` points, and with it the team size -- a function of
// the point count alone, so a problem's sums (and bits) do not depend on what else shares its launch.
int track1_team_q(int maxpts, int target) {
  const int team = std::min(kTeamMax, std::max(1, (maxpts + target - 1) / target));
  return ((maxpts + team - 1) / team + 7) & ~7;  // whole rows of eight waves
}
int track1_team_size(int maxpts, int target) {
  const int q = track1_team_q(maxpts, target);
  return std::max(1, (maxpts + q - 1) / q);
}
size_t track1_team_mail_bytes(int B, int team) { return sizeof(unsigned long long) * (size_t)B * 2 * team * kTeamSlot; }

// blob (may be NULL): [ProbState x B][PlaneSet x B x nlev] for the fused begin; host_st (may be NULL): pinned mirror;
// project_here (without a blob): the records and the plane table have been uploaded, the launch projects (step 3) itself;
// tm (may be NULL): team form -- tm->team workgroups of tm->q points per problem, mailbox, tag epoch, error flag
hipError_t launch_track1(const EngineDev &e, const LevelCam *cams, int maxpts, int waves, const void *blob,
                         ProbState *host_st, hipStream_t s, const T1Team *tm, bool project_here) {
  T1Args a;
  for (int l = 0; l < 16; ++l) a.lc[l] = cams[l < e.nlev ? l : 0];
  const bool p8 = e.P == 8 && !e.robust;  // the lean 8x8 form; behaviour-changing options run in the any-size form
  const int team = (tm && p8 && tm->team > 1) ? tm->team : 1;
  a.team = team;
  a.team_q = team > 1 ? tm->q : 0x7fffffff;
  a.team_tag0 = team > 1 ? tm->tag0 : 0;
  a.team_limit = team > 1 ? tm->limit : 0;
  a.team_mail = team > 1 ? tm->mail : nullptr;
  a.team_err = team > 1 ? tm->err : nullptr;
  a.team_mute = team > 1 ? tm->mute : 0;
  if (team > 1) {
    if (team > kTeamMax || !tm->mail || !tm->err || (long long)tm->q * team < maxpts) return hipErrorInvalidValue;
    maxpts = std::min(maxpts, tm->q);  // LDS records and patches: this workgroup's share only
  }
  a.npts_cap = (std::max(maxpts, 1) + 3) & ~3;  // keeps the LDS template arrays 16-byte aligned
  a.dbg = 0;
  a.cap_w = 0;
  a.st_words = (int)(sizeof(ProbState) / 4);
  a.pl_words = (int)(sizeof(PlaneSet) / 4) * e.nlev;
  a.host_st = host_st;
  a.fused_begin = project_here ? 2 : 0;
  if (blob) {
    const size_t bytes = (size_t)e.B * 4 * (a.st_words + a.pl_words);
    if (bytes > sizeof(a.blob)) return hipErrorInvalidValue;
    memcpy(a.blob, blob, bytes);
    a.fused_begin = 1;
  }
#ifdef ICTR_T1_PROF
  if (const char *d = getenv("ICTR_T1_DBG")) a.dbg = atoi(d);
#endif
  int tl = 0;
  waves = std::min(std::max(waves, 1), p8 ? kT8MaxWaves : kT1MaxWaves);
  const size_t lds = track1_plan(a.npts_cap, e.n, p8 ? 1 : 0, waves, &tl);
  if (p8) a.cap_w = track1_cap_w(a.npts_cap, waves);
  static size_t g[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (p8) {
    // more workgroups than CUs, and two of them fit one CU's LDS: the 128-register build (see k_track1_p8)
    int cus = 256;
    {
      static const int n_cu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1)
          v = 256;
        return v;
      }();
      cus = n_cu;
    }
    const bool lean = (long long)e.B * team > cus && lds <= 60 * 1024;
    if (team > 1) {  // several workgroups per problem ("Teams"); the templates of a share always fit the LDS
      if (!tl) return hipErrorInvalidValue;
      static size_t gt[4] = {0, 0, 0, 0};
      if (lean)
        return e.dopatchnorm ? launch_t1(&k_track1_p8<true, true, true, true>, &gt[0], e, a, waves, lds, s, team)
                             : launch_t1(&k_track1_p8<true, false, true, true>, &gt[1], e, a, waves, lds, s, team);
      return e.dopatchnorm ? launch_t1(&k_track1_p8<true, true, false, true>, &gt[2], e, a, waves, lds, s, team)
                           : launch_t1(&k_track1_p8<true, false, false, true>, &gt[3], e, a, waves, lds, s, team);
    }
    if (lean) {
      if (e.dopatchnorm)
        return tl ? launch_t1(&k_track1_p8<true, true, true, false>, &g[6], e, a, waves, lds, s)
                  : launch_t1(&k_track1_p8<false, true, true, false>, &g[7], e, a, waves, lds, s);
      return tl ? launch_t1(&k_track1_p8<true, false, true, false>, &g[8], e, a, waves, lds, s)
                : launch_t1(&k_track1_p8<false, false, true, false>, &g[9], e, a, waves, lds, s);
    }
    if (e.dopatchnorm)
      return tl ? launch_t1(&k_track1_p8<true, true, false, false>, &g[0], e, a, waves, lds, s)
                : launch_t1(&k_track1_p8<false, true, false, false>, &g[1], e, a, waves, lds, s);
    return tl ? launch_t1(&k_track1_p8<true, false, false, false>, &g[2], e, a, waves, lds, s)
              : launch_t1(&k_track1_p8<false, false, false, false>, &g[3], e, a, waves, lds, s);
  }
  return tl ? launch_t1(&k_track1<true>, &g[4], e, a, waves, lds, s)
            : launch_t1(&k_track1<false>, &g[5], e, a, waves, lds, s);
}

}  // namespace ictr
