// ictr_devfn.h -- device helpers shared by the kernel translation units (ictr_kernels.hip, ictr_track1.hip):
// bilinear tap selection (utilities.cpp:66-107), visibility (odometer.cpp:273-276), steepest-descent coefficients
// (odometer.cpp:313-326), the per-level LU factorisation and the per-iteration solve + pose update + loop condition
// (odometer.cpp:341-346, 407-418, 509-515; pose.cpp:116-129). Everything keeps the reference's operand order; the
// translation units are compiled with -ffp-contract=off.
#pragma once

#include <type_traits>

#include "ictr_dev.h"
#include "se3_math.h"

namespace ictr {

// ---------------------------------------------------------------- small device helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
// sum over aligned groups of `width` lanes (width = power of two <= 64)
__device__ __forceinline__ float group_sum(float v, int width) {
  for (int m = width >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

struct Taps {
  float w0, w1, w2, w3;
  int col0, row0;
};

// utilities.cpp:66-77 : patch-constant bilinear weights, ceil(x+1e-5f) tap selection
__device__ __forceinline__ Taps make_taps(float mx, float my, int pszd2) {
  Taps t;
  const int p0 = (int)ceilf(mx + .00001f);
  const int p1 = (int)ceilf(my + .00001f);
  const int p2 = (int)floorf(mx);
  const int p3 = (int)floorf(my);
  const float r0 = mx - (float)p2;
  const float r1 = my - (float)p3;
  t.w0 = r0 * r1;
  t.w1 = (1 - r0) * r1;
  t.w2 = r0 * (1 - r1);
  t.w3 = (1 - r0) * (1 - r1);
  t.col0 = p0 + pszd2;
  t.row0 = p1 + pszd2;
  return t;
}

// utilities.cpp:107 : a=(col,row) b=(col-1,row) c=(col,row-1) d=(col-1,row-1)
__device__ __forceinline__ float tap4(const float *__restrict__ img, int idx, int sw, const Taps &t) {
  const float a = img[idx], b = img[idx - 1], c = img[idx - sw], d = img[idx - sw - 1];
  return t.w0 * a + t.w1 * b + t.w2 * c + t.w3 * d;
}

__device__ __forceinline__ bool in_view(float mx, float my, float swo, float sho) {
  // odometer.cpp:273-276 rejects (x<0)|(y<0)|(x>swo)|(y>sho); written positively so NaN is "outside"
  return (mx >= 0.0f) & (my >= 0.0f) & (mx <= swo) & (my <= sho);
}

// odometer.cpp:313-326 : per-point steepest-descent coefficients; the "1.0 +" terms are f64, narrowed
__device__ __forceinline__ void sd_coefs(float X, float Y, float Z, float fx, float fy, float *cx, float *cy) {
  const float zsq = Z * Z;
  cx[0] = fx / Z;
  cy[0] = 0.0f;
  cx[1] = 0.0f;
  cy[1] = fy / Z;
  cx[2] = -X / zsq * fx;
  cy[2] = -Y / zsq * fy;
  cx[3] = -X * Y / zsq * fx;
  cy[3] = (float)((-(1.0 + (double)(Y * Y / zsq))) * (double)fy);
  cx[4] = (float)((1.0 + (double)(X * X / zsq)) * (double)fx);
  cy[4] = X * Y / zsq * fy;
  cx[5] = -Y / Z * fx;
  cy[5] = X / Z * fy;
}

__device__ __forceinline__ void sd_values(float gx, float gy, const float *cx, const float *cy, float *sd) {
  sd[0] = gx * cx[0];
  sd[1] = gy * cy[1];
#pragma unroll
  for (int k = 2; k < 6; ++k) sd[k] = gx * cx[k] + gy * cy[k];
}

__device__ __forceinline__ void level_reset(ProbState &st, const EngineDev &e) {
  // odometer.cpp:341-346
  st.normdp_init = 1e-10f;
  st.normdp = 1e-10f;
  st.it = 0;
  st.active = ((0 < e.maxiter) & ((st.normdp / st.normdp_init) > e.ratio)) ? 1 : 0;
}

// ---------------------------------------------------------------- 8x8 patches: wave64 == patch, lane == pixel
typedef const float __attribute__((address_space(1))) *gconst_f32;  // plane pointers come out of a table in memory:
                                                                    // tell the compiler they are global, not flat
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef const f32x2_a4 __attribute__((address_space(1))) *gconst_f32x2;

__device__ __forceinline__ int rlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rlane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

constexpr int kRec = 16;  // floats per point record in LDS: [w0 w1 w2 w3][cx0 cx2 cx3 cx4][cx5 cy1 cy2 cy3][cy4 cy5 vis -]

// The four bilinear taps of a lane's pixel (lanes = 8x8 pixels, row-major) with every cache line of the 9x9 window
// requested once: one 8-byte load per lane gives (b,a) of its own row; lanes 0..7 also load the row above; all
// other lanes take (d,c) from the lane one row up (ds_bpermute) when the values are consumed.
struct TapLoads {
  f32x2_a4 ab, top;
};
__device__ __forceinline__ TapLoads taps_issue(gconst_f32 plane_at_base, int loff, int sw, int lane) {
  TapLoads t;
  t.ab = *reinterpret_cast<gconst_f32x2>(plane_at_base + (loff - 1));
  f32x2_a4 q = {0.0f, 0.0f};
  if (lane < 8) q = *reinterpret_cast<gconst_f32x2>(plane_at_base + (loff - sw - 1));
  t.top = q;
  return t;
}
// utilities.cpp:107 with the reference's operand order, never contracted: template and current patch must round
// identically so that identical frames give a residual of exactly zero (identity KAT)
__device__ __forceinline__ float taps_blend(const TapLoads &t, float w0, float w1, float w2, float w3, int lane) {
  const float a = t.ab.y, b = t.ab.x;
  const float cu = __shfl_up(a, 8, 64), du = __shfl_up(b, 8, 64);
  const float c = lane < 8 ? t.top.y : cu, d = lane < 8 ? t.top.x : du;
  return w0 * a + w1 * b + w2 * c + w3 * d;
}

// ---------------------------------------------------------------- wave-parallel 6x6 solver state (one wave64)
// The per-iteration tail of the Gauss-Newton loop (final sum -> fullPivLu solve -> pose update -> exp map -> loop
// condition, odometer.cpp:407-418,509-515) is a serial dependency chain on the critical path of every iteration. Run
// by one thread on LDS / private arrays it costs ~14 k cycles (a lone wave issues one dependent instruction every
// 4-8 cycles and every runtime-indexed array access is an LDS round trip). Here one wave does it in registers:
//   * the LU factors live in lanes 0-5 (lane r holds row r), the 6x6 matrix being factored in lanes 0-35 (lane =
//     6 r + c); pivot search = two DPP wave reductions, row/column exchange = one ds_bpermute, elimination = two;
//   * substitution: lane i owns c[i]; c[i] is broadcast with v_readlane as soon as it is final;
//   * the arithmetic per matrix / vector entry is EXACTLY that of lu_factor_ws / lu_apply_ws (same operations in the
//     same order on the same operands; Eigen FullPivLU's pivot order, rank threshold and zero-filled free variables),
//     so the results are bit-identical to the serial code (tests/test_gpu_parity.py compares the two).
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ int dpp_mov(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
// Sum over the 64 lanes, the same value in every lane; fixed order: xor 1, xor 2 inside a quad (quad_perm), the
// mirrored half-rows and rows (row_half_mirror, row_mirror), then the four rows by v_readlane. ~11 instructions
// against 6 dependent ds_bpermute round trips for the shuffle butterfly (wave_sum): used where sums are compared to
// tolerance only (H, b); the patch means keep wave_sum / group_sum.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return (rlane(v, 0) + rlane(v, 16)) + (rlane(v, 32) + rlane(v, 48));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return fmaxf(fmaxf(rlane(v, 0), rlane(v, 16)), fmaxf(rlane(v, 32), rlane(v, 48)));
}
__device__ __forceinline__ int wave_min_dpp(int v) {
  v = min(v, dpp_mov<0xB1>(v));
  v = min(v, dpp_mov<0x4E>(v));
  v = min(v, dpp_mov<0x141>(v));
  v = min(v, dpp_mov<0x140>(v));
  return min(min(rlane(v, 0), rlane(v, 16)), min(rlane(v, 32), rlane(v, 48)));
}
__device__ __forceinline__ int row_min_dpp(int v) {  // the minimum over the lane's DPP row (16 lanes), in all of them
  v = min(v, dpp_mov<0xB1>(v));
  v = min(v, dpp_mov<0x4E>(v));
  v = min(v, dpp_mov<0x141>(v));
  return min(v, dpp_mov<0x140>(v));
}
__device__ __forceinline__ float lane_gather(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int lane_gather(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ int rlane_dyn(int v, int uniform_lane) {
  return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(uniform_lane));
}

// ---------------------------------------------------------------- transposing wave reduction
// merge(u, v) on lane bit k: the lanes with bit k = 0 end up with u summed over the lane pair {l, l ^ (1 << k)}, the
// lanes with bit k = 1 with v summed over the same pair. Six levels of merges (one per lane bit) take 64 per-lane
// values to ONE register in which every lane holds the complete 64-lane sum of one of the values.
//   bits 2, 3: two DPP adds, each writing half of the banks (row_shl / row_shr by 4 or 8 with a bank mask);
//   bits 4, 5: v_permlane16_swap / v_permlane32_swap (gfx950) + one add;  bits 0, 1: quad permutes + select.
// (inline DPP: the s_nop covers the "VALU write -> DPP read" hazard, which the compiler cannot see inside asm)
__device__ __forceinline__ float tr_merge_b2(float u, float v) {
  float w;
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5" : "=v"(w) : "v"(u));
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(w) : "v"(v));
  return w;
}
__device__ __forceinline__ float tr_merge_b3(float u, float v) {
  float w;
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:8 row_mask:0xf bank_mask:0x3" : "=v"(w) : "v"(u));
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xc" : "+v"(w) : "v"(v));
  return w;
}
// (inline asm: with ROCm 7.2's hipcc the two results of __builtin_amdgcn_permlane16/32_swap come back as the SAME register
// -- "v_permlane16_swap v25, v26; v_add_f32 v6, v25, v25" -- found by tests/test_gpu_parity.py's reduction test; the
// s_nops cover the VALU-write -> permlane-swap hazard on both sides, which the compiler cannot see inside asm)
__device__ __forceinline__ float tr_merge_b4(float u, float v) {  // rows (16 lanes): [u0 u1 u2 u3],[v0..] -> [u0+u1, v0+v1, u2+u3, v2+v3]
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(u), "+v"(v));  // odd rows of u <-> even rows of v
  return u + v;
}
__device__ __forceinline__ float tr_merge_b5(float u, float v) {  // halves: -> [u_lo + u_hi, v_lo + v_hi]
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(u), "+v"(v));  // upper half of u <-> lower half of v
  return u + v;
}
__device__ __forceinline__ float tr_merge_b0(float u, float v, int lane) {
  const float t = u + dpp_mov<0xB1>(u), s = v + dpp_mov<0xB1>(v);
  return (lane & 1) ? s : t;
}
__device__ __forceinline__ float tr_merge_b1(float u, float v, int lane) {
  const float t = u + dpp_mov<0x4E>(u), s = v + dpp_mov<0x4E>(v);
  return (lane & 2) ? s : t;
}
// level L = 1..5 of the reduction tree over a wave's 32 patches (level 0 = the patch's own (A, B) pair on lane bit 2)
template <int L>
__device__ __forceinline__ float tr_merge_level(float u, float v, int lane) {
  if constexpr (L == 1) return tr_merge_b3(u, v);
  if constexpr (L == 2) return tr_merge_b4(u, v);
  if constexpr (L == 3) return tr_merge_b5(u, v);
  if constexpr (L == 4) return tr_merge_b0(u, v, lane);
  return tr_merge_b1(u, v, lane);
}
// which patch of the wave and which of its two sums (0: A = sum Gx r, 1: B = sum Gy r) lane l holds at the end. NP = 32:
// all six lane bits are consumed by merges; NP = 16: the last bit (1) is summed plainly, so lane pairs {l, l ^ 2} hold the
// same sum and only the lanes with bit 1 clear ("primary") may contribute it
template <int NP>
__device__ __forceinline__ int tr_patch_of_lane(int l) {
  const int p = ((l >> 3) & 1) | (((l >> 4) & 1) << 1) | (((l >> 5) & 1) << 2) | ((l & 1) << 3);
  return NP == 32 ? (p | (((l >> 1) & 1) << 4)) : p;
}
__device__ __forceinline__ int tr_kind_of_lane(int l) { return (l >> 2) & 1; }
template <int NP>
__device__ __forceinline__ bool tr_primary_lane(int l) { return NP == 32 ? true : ((l & 2) == 0); }
// the binary counter of pending registers: push<J>(A, B) takes patch J's pair of per-lane values (J = 0..NP-1 in order);
// after push<NP-1> `F` holds, in lane l, the 64-lane sum of (A if kind == 0 else B) of patch tr_patch_of_lane<NP>(l)
template <int NP>
struct TrAcc {
  static_assert(NP == 32 || NP == 16, "patches per wave");
  float p1, p2, p3, p4, p5, F;
  template <int J>
  __device__ __forceinline__ void push(float A, float B, int lane) {
    float m = tr_merge_b2(A, B);
    if constexpr ((J & 1) == 0) {
      p1 = m;
    } else {
      m = tr_merge_level<1>(p1, m, lane);
      if constexpr ((J & 2) == 0) {
        p2 = m;
      } else {
        m = tr_merge_level<2>(p2, m, lane);
        if constexpr ((J & 4) == 0) {
          p3 = m;
        } else {
          m = tr_merge_level<3>(p3, m, lane);
          if constexpr ((J & 8) == 0) {
            p4 = m;
          } else {
            m = tr_merge_level<4>(p4, m, lane);
            if constexpr (NP == 16) {
              F = m + dpp_mov<0x4E>(m);  // lane bit 1: plain sum
            } else if constexpr ((J & 16) == 0) {
              p5 = m;
            } else {
              F = tr_merge_level<5>(p5, m, lane);
            }
          }
        }
      }
    }
  }
};
template <int NP, int J, class Fn>
__device__ __forceinline__ void tr_for_each_patch(Fn &&fn) {  // fn(integral_constant<J>) for J = 0..NP-1, in order
  if constexpr (J < NP) {
    fn(std::integral_constant<int, J>{});
    tr_for_each_patch<NP, J + 1>(fn);
  }
}

// inverse of tr_patch_of_lane / tr_kind_of_lane: the (primary) lane that holds sum `kind` of patch p
template <int NP>
__device__ __forceinline__ int tr_lane_of(int p, int kind) {
  return ((p >> 3) & 1) | ((NP == 32 ? ((p >> 4) & 1) : 0) << 1) | (kind << 2) | ((p & 1) << 3) | (((p >> 1) & 1) << 4) |
         (((p >> 2) & 1) << 5);
}


struct WaveSolver {  // every member is a per-lane register of the solving wave
  float lu[6];       // lane r < 6: row r of the LU factors
  float diag;        // lane r < 6: LU[r][r]
  float h;           // lane l < 36: H[l / 6][l % 6] of the current level (trace, state write-back)
  int rowmap, colmap;  // lane i < 6: c[i] = b[rowmap], x[i] = c[colmap] (the composed transpositions)
  int rank, nonzero;   // uniform
  float p;             // lane i < 6: cpos_p[i]
  float b, dp;         // lane i < 6: sumsd[i], delta_p[i] of the last iteration
  float normdp, normdp_init;  // uniform
  int it, total_iters, active;
#ifdef ICTR_T1_PROF
  unsigned long long tm[4];  // diagnostic builds: cycles in apply / pose update + exp / rest
#endif
};

// lu_factor_ws<6>, lane-parallel. a: lane l < 36 holds H[l/6][l%6].
__device__ __forceinline__ void ws_factor(WaveSolver &s, float a, int lane) {
  const int r = lane / 6, c = lane - 6 * r;
  const bool in = lane < 36;
  s.h = a;
  int idr = lane, idc = lane;  // lane i < 6: images of i under the row / column transpositions applied so far
  int nonzero = 6;
  float maxpiv = 0.0f;
  int colsw[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) colsw[k] = k;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (nonzero == 6) {  // wave-uniform
      const float v = (in && r >= k && c >= k) ? fabsf(a) : -1.0f;
      const float m = wave_max_dpp(v);
      int sidx = wave_min_dpp((v == m) ? c * 6 + r : (1 << 20));  // first maximum of the column-major scan
      if (sidx >= 36 || !(m >= 0.0f)) sidx = k * 6 + k;            // NaN candidates: the serial code keeps (k,k)
      const int bc = sidx / 6, br = sidx - 6 * bc;
      if (m == 0.0f) {
        nonzero = k;
      } else {
        if (m > maxpiv) maxpiv = m;
        {  // rowsw[k] = br: c[k] <-> c[br] composes into rowmap
          const int vk = rlane(idr, k), vr = rlane_dyn(idr, br);
          idr = lane == k ? vr : (lane == br ? vk : idr);
        }
        colsw[k] = bc;
        const int sr = r == k ? br : (r == br ? k : r);
        const int sc = c == k ? bc : (c == bc ? k : c);
        a = lane_gather(a, sr * 6 + sc);
        if (k < 5) {
          const float piv = rlane(a, k * 6 + k);
          const float q = a / piv;
          if (in && c == k && r > k) a = q;
          const float l = lane_gather(a, r * 6 + k), u = lane_gather(a, k * 6 + c);
          if (in && r > k && c > k) a = a - l * u;
        }
      }
    }
  }
  // x[k] <-> x[colsw[k]] for k = 5..0 composes into colmap
#pragma unroll
  for (int k = 5; k >= 0; --k) {
    const int q = colsw[k];
    const int vk = rlane(idc, k), vq = rlane_dyn(idc, q);
    idc = lane == k ? vq : (lane == q ? vk : idc);
  }
  const float d = lane_gather(a, lane * 7);  // lane r < 6: LU[r][r]
  int rank = 0;
  if (nonzero > 0) {
    const float thr = maxpiv * (1.1920929e-07f * 6);
#pragma unroll
    for (int i = 0; i < 6; ++i) rank += (i < nonzero && fabsf(rlane(d, i)) > thr) ? 1 : 0;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) s.lu[j] = lane_gather(a, lane * 6 + j);
  s.diag = d;
  s.rowmap = idr;
  s.colmap = idc;
  s.rank = rank;
  s.nonzero = nonzero;
}

// lu_apply_ws<6>, lane-parallel: lane i < 6 passes b[i] and receives x[i]
__device__ __forceinline__ float ws_apply(const WaveSolver &s, float bi, int lane) {
  if (s.nonzero == 0) return 0.0f;
  float c = lane_gather(bi, s.rowmap);
#pragma unroll
  for (int i = 0; i < 5; ++i) {  // c[r] -= c[i] * A[r][i], r > i
    const float ci = rlane(c, i);
    const float t = c - ci * s.lu[i];
    if (lane > i) c = t;
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    if (i < s.rank) {  // uniform
      const float q = c / s.diag;
      if (lane == i) c = q;
      const float ci = rlane(c, i);
      const float t = c - ci * s.lu[i];
      if (lane < i) c = t;
    }
  }
  c = (lane < s.rank) ? c : 0.0f;
  return lane_gather(c, s.colmap);
}

// what the solver needs of the engine's parameters (passed by value: 8 dwords instead of the whole EngineDev)
struct SolveOpts {
  int maxiter, robust;
  float ratio;
  DevTrace trace;
};
__device__ __forceinline__ SolveOpts solve_opts(const EngineDev &e) {
  SolveOpts o;
  o.maxiter = e.maxiter;
  o.robust = e.robust;
  o.ratio = e.ratio;
  o.trace = e.trace;
  return o;
}
__device__ __forceinline__ void ws_level_reset(WaveSolver &s, const SolveOpts &e) {  // odometer.cpp:341-346
  s.normdp_init = 1e-10f;
  s.normdp = 1e-10f;
  s.it = 0;
  s.active = ((0 < e.maxiter) & ((s.normdp / s.normdp_init) > e.ratio)) ? 1 : 0;
}

// Steps 9b + 10 + loop condition (solve_and_update, wave form). bi: lane i < 6 holds sumsd[i]. G (uniform, 12
// registers): cpos_G, current on entry, updated on return. The arithmetic is solve_and_update's.
__device__ __forceinline__ void ws_iterate(WaveSolver &s, float bi, const SolveOpts &e, int level, int prob, int lane,
                                           float *G) {
#ifdef ICTR_T1_PROF
  unsigned long long c0_ = __builtin_readcyclecounter(), c1_;
#define WS_MARK(k) c1_ = __builtin_readcyclecounter(); s.tm[k] += c1_ - c0_; c0_ = c1_;
#else
#define WS_MARK(k)
#endif
  s.b = bi;
  const float dpi = ws_apply(s, bi, lane);
  s.dp = dpi;
  WS_MARK(0)
  float dp[6], p[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) dp[k] = rlane(dpi, k);
  if (e.robust & ICTR_ROBUST_COMPOSE) {  // option: left-compositional update G <- exp(dp) G, p = log(G)
    float D[12], Go[12], Gn[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) Go[k] = G[k];
    se3_exp<float>(D, dp);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        Gn[r * 4 + c] = D[r * 4 + 0] * Go[c] + D[r * 4 + 1] * Go[4 + c] + D[r * 4 + 2] * Go[8 + c] + (c == 3 ? D[r * 4 + 3] : 0.0f);
    }
    se3_log<float>(p, Gn);
    float pl = p[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) pl = lane == k ? p[k] : pl;
    s.p = pl;
  } else {
    s.p = s.p + dpi;  // pose.cpp:118-123 additive update
#pragma unroll
    for (int k = 0; k < 6; ++k) p[k] = rlane(s.p, k);
  }
  WS_MARK(1)
  se3_exp<float>(G, p);
  WS_MARK(2)
  // delta_p.lpNorm<1>() : Eigen's unrolled redux tree for 6 coefficients
  const float nd = (fabsf(dp[0]) + (fabsf(dp[1]) + fabsf(dp[2]))) + (fabsf(dp[3]) + (fabsf(dp[4]) + fabsf(dp[5])));
  s.normdp = nd;
  if (s.it == 0) s.normdp_init = nd;
  if (e.trace.rec != nullptr && prob == 0) {
    const int c = *e.trace.count;
    if (c < e.trace.capacity) {
      ictr_trace_rec &r = e.trace.rec[c];
      if (lane == 0) {
        r.level = level;
        r.iter = s.it;
      }
      if (lane < 36) r.H[lane] = s.h;
      if (lane < 6) {
        r.b[lane] = bi;
        r.dp[lane] = dpi;
        r.p[lane] = s.p;
      }
    }
    if (lane == 0) *e.trace.count = c + 1;
  }
  s.it += 1;
  s.total_iters += 1;
  s.active = ((s.it < e.maxiter) & ((s.normdp / s.normdp_init) > e.ratio)) ? 1 : 0;
  WS_MARK(3)
}

// ---- the solver state in the problem's device record (between the launches of the per-iteration form)
// ProbState.LU holds the factors row-major, ProbState.piv the COMPOSED permutations (rowmap[6] | colmap[6]).
__device__ __forceinline__ int h_unique_index(int lane) {  // lane l < 36 -> index of H[l/6][l%6] among the 21 sums
  const int r = lane / 6, c = lane - 6 * r;
  const int lo = r < c ? r : c, hi = r < c ? c : r;
  return lane < 36 ? lo * 6 - lo * (lo - 1) / 2 + (hi - lo) : 0;
}
__device__ __forceinline__ void ws_store_factor(const WaveSolver &s, ProbState &st, int lane) {
  if (lane < 36) st.H[lane] = s.h;
  if (lane < 6) {
#pragma unroll
    for (int j = 0; j < 6; ++j) st.LU[lane * 6 + j] = s.lu[j];
    st.piv[lane] = s.rowmap;
    st.piv[6 + lane] = s.colmap;
  }
  if (lane == 0) {
    st.luinfo[0] = s.nonzero;
    st.luinfo[1] = s.rank;
  }
}
__device__ __forceinline__ void ws_load_factor(WaveSolver &s, const ProbState &st, int lane) {
  const int r = lane < 6 ? lane : 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) s.lu[j] = st.LU[r * 6 + j];
  s.diag = st.LU[r * 7];
  s.rowmap = st.piv[r];
  s.colmap = st.piv[6 + r];
  s.h = lane < 36 ? st.H[lane] : 0.0f;
  s.nonzero = st.luinfo[0];
  s.rank = st.luinfo[1];
}
__device__ __forceinline__ void ws_load_state(WaveSolver &s, const ProbState &st, int lane, float *G) {
  s.p = lane < 6 ? st.p[lane] : 0.0f;
  s.b = s.dp = 0.0f;
  s.normdp = st.normdp;
  s.normdp_init = st.normdp_init;
  s.it = st.it;
  s.total_iters = st.total_iters;
  s.active = st.active;
#pragma unroll
  for (int k = 0; k < 12; ++k) G[k] = st.G[k];
}
__device__ __forceinline__ void ws_store_state(const WaveSolver &s, ProbState &st, int lane, const float *G) {
  if (lane < 6) {
    st.p[lane] = s.p;
    st.b[lane] = s.b;
    st.dp[lane] = s.dp;
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) st.G[k] = G[k];
    st.normdp = s.normdp;
    st.normdp_init = s.normdp_init;
    st.it = s.it;
    st.active = s.active;
    st.total_iters = s.total_iters;
  }
}

}  // namespace ictr
