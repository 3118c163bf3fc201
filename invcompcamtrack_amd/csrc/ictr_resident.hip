// ictr_resident.hip -- all Gauss-Newton iterations of a pyramid level with the templates RESIDENT on the chip
// (gfx950 / CDNA4; 8x8 patches, large problems: thousands to tens of thousands of points per frame pair).
//
// The per-iteration kernel k_iter8 streams T, Gx, Gy of every patch from HBM in every iteration: 12 of its 16 bytes
// per pixel, ten times per level, and it runs at the HBM roofline doing so (profiles/r02_notes.md). But a frame pair's
// templates are only 25 MB per level -- the chip has 128 MB of vector registers and 40 MB of LDS. Here a frame pair is
// shared by `parts` worker workgroups (128 points each: sixteen patches per wave; Gx and Gy of a patch = two registers
// of its wave, lane = pixel; T in LDS), which load their templates ONCE per level and keep them for all iterations of
// odometer.cpp:344-418; an iteration then reads only the current frame's windows (cache-resident) and exchanges 12
// numbers per workgroup:
//
//   every workgroup   stage 1  one point per lane (lanes 0-15 of a wave): projection at the current pose
//                              (pose.cpp:384-391), ind_new (odometer.cpp:369-377), bilinear weights + window offset
//                     stage 2  sixteen patches per wave from registers: current-frame window (utilities.cpp:55-113),
//                              residual, J^T r (odometer.cpp:381-404) in six per-lane accumulators
//                     gather   the workgroup's six sums (as hi/lo float pairs) -> the pair's mailbox
//   the pair's solver workgroup (no templates: its registers never compete with the resident ones)
//                     all eight waves poll the mailbox (32 workers each, one round trip), fixed-order f64 sum, then
//                     ONE wave: substitution with the level's LU factors, pose update, exp map, loop condition
//                     (odometer.cpp:407-418, 509-515; WaveSolver, ictr_devfn.h) -> broadcast of cpos_G + loop flag
//   every worker      polls the broadcast, next iteration
//
// Two hops per iteration instead of two kernel boundaries and 33 MB of HBM traffic per pair. An iteration is a serial
// chain of ~9.4 us for ONE pair (tools/resprof.py), so `slots` pairs are in flight at once (two workgroups per CU, 128
// registers each) and the chains of different pairs overlap on the same SIMDs (two in flight: +10 % per chain); every
// slot walks through its share of the batch's pairs. Measured: one 1080p pair 0.42 ms against 0.72 with the streaming
// kernels; the default for up to 8 pairs per engine (ictr_host.hip, resident_plan).
// The mailbox protocol is the one of the team form (ictr_track1.hip "Teams"): 8-byte granules {float bits, tag},
// tags = launch epoch << 12 | exchange number, double-buffered by parity, bounded polling with a sticky error flag.
// H is accumulated by the level's setup launch (k_ref8<.., WH = true, PK = true>: per-workgroup partials) and reduced +
// factored by the pair's solver workgroup at the start of the pair (what k_level_tail does in the other launch forms),
// the templates and the (possibly stale) coefficients come from the buffers that launch wrote: patches, coefficients and
// projections are bit-identical to the other launch forms, b differs by summation order only.
#include "ictr_dev.h"
#include "ictr_devfn.h"
#include "se3_math.h"

namespace ictr {

constexpr int kResWaves = 8;          // waves per workgroup
constexpr int kResPPW = 16;           // patches (points) per wave
constexpr int kResQ = kResWaves * kResPPW;  // points per workgroup
constexpr int kResSlot = 16;          // granules per workgroup in the gather box (12 used)
constexpr int kGxL = 0;               // patches per wave whose Gx lives in LDS instead of a register (experiments)

struct ResArgs {
  LevelCam lc;
  int level;
  int parts, slots;          // worker workgroups per frame pair; pairs in flight (grid = slots * (parts + 1))
  int nblk;                  // workgroups per problem of the level's setup launch (their H partials: e.partH)
  unsigned tag0;             // launch epoch << 12
  unsigned long long limit;  // polling limit, wall_clock64 ticks (100 MHz)
  unsigned long long *mail;  // per slot: gather box [2][parts][kResSlot], then broadcast box [2][16]
  int *err;                  // sticky time-out flag (pinned host memory as the device sees it)
};

__device__ __forceinline__ size_t res_slot_granules(int parts) { return (size_t)2 * parts * kResSlot + 2 * 16; }

__device__ __forceinline__ double res_gather64(double v, int src_lane) {
  const int lo = lane_gather(__double2loint(v), src_lane), hi = lane_gather(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

struct ResPoll {
  unsigned long long limit;
  int *err;
  int dead;
};
// poll up to eight granules per lane until every tag matches; lanes / entries without a granule pass nullptr
// (entries written out one by one: everything stays in registers)
#define RES_EACH8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
__device__ __forceinline__ void res_poll(ResPoll &pc, const unsigned long long *(&src)[8], unsigned long long (&g)[8],
                                         unsigned tag, int lane) {
  const unsigned long long empty = (unsigned long long)tag << 32;
#define RES_LOAD(u) g[u] = src[u] ? __hip_atomic_load(src[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : empty;
  RES_EACH8(RES_LOAD)
  if (pc.dead) return;
  bool started = false;
  unsigned long long t0 = 0;
  for (;;) {
    bool miss = false;
#define RES_MISS(u) miss |= (unsigned)(g[u] >> 32) != tag;
    RES_EACH8(RES_MISS)
    if (__builtin_amdgcn_ballot_w64(miss) == 0) break;  // wave-uniform
    if (!started) {
      t0 = wall_clock64();
      started = true;
    } else if (wall_clock64() - t0 > pc.limit) {  // a peer never arrived: flag it, never wait again
      if (lane == 0) __hip_atomic_store(pc.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      pc.dead = 1;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
#define RES_RELOAD(u) \
  if ((unsigned)(g[u] >> 32) != tag) g[u] = __hip_atomic_load(src[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    RES_EACH8(RES_RELOAD)
  }
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
#ifdef ICTR_RES_PROF  // diagnostic builds only: per-phase cycle counters of one wave (tools/resprof.py)
#define RES_MARK(k)                        \
  {                                        \
    t1_ = __builtin_readcyclecounter();    \
    tp_[k] += t1_ - t0_;                   \
    t0_ = t1_;                             \
  }
#else
#define RES_MARK(k)
#endif
struct ResWin {
  f32x2_a4 ab, cd;  // (x-1,y),(x,y) and (x-1,y-1),(x,y-1)
};

// Register budget: 128 per wave = two workgroups per CU. Gx and Gy of a wave's sixteen patches live in registers (32),
// T in LDS (32 KB per workgroup). An iteration is a latency chain and the chains of the pairs in flight overlap almost
// perfectly (two in flight cost 10 % per chain), so pairs in flight = throughput -- but a CU cannot hold a third pair's
// share: a build for 80 registers (three workgroups per CU; T and five patches' Gx in LDS, kGxL = 5, two windows in
// flight) spills in the solver and in the patch loop and was SLOWER (7.1 against 6.5 ms per 32 pairs,
// profiles/r02_notes.md).
__global__ __launch_bounds__(64 * kResWaves, 4) void k_level_resident(EngineDev e, ResArgs a) {
  __shared__ __attribute__((aligned(16))) float4 sRec[kResWaves][kResPPW * 4];  // per point [w1 w0 w3 w2][cx2..5][cy2..5][cx0 cy1 vis -]
  __shared__ float sT[kResWaves][(kResPPW + kGxL) * 64];  // T of the wave's patches, then Gx of the first kGxL (lane = pixel)
  __shared__ float sPart[kResWaves][8];
  __shared__ double sRed[kResWaves][8];
  __shared__ float sG[16];  // cpos_G of the current iteration, [12] = loop flag (bits)
  __shared__ ProbState sSt; // solver workgroup: the problem's state between the solver's turns

  // a pair's workgroups: `parts` workers (128 points each) + ONE solver workgroup (index parts) that holds no templates,
  // so the solver's registers and the sixteen resident patches never compete
  const int parts = a.parts;
  const int group = parts + 1;
  const int slot = (int)blockIdx.x / group;
  const int part = (int)blockIdx.x - slot * group;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = e.M;
  unsigned long long *gbox = a.mail + (size_t)slot * res_slot_granules(parts);
  unsigned long long *bbox = gbox + (size_t)2 * parts * kResSlot;
  ResPoll pc;
  pc.limit = a.limit;
  pc.err = a.err;
  pc.dead = 0;
  unsigned seq = 0;
#ifdef ICTR_RES_PROF
  unsigned long long tp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_readcyclecounter(), t1_ = 0;
#endif

  if (part == parts) {
    // ================================================================ the pair's solver workgroup
    SolveOpts sopt = solve_opts(e);
    sopt.robust = 0;
    __shared__ double sRedH[64 * kResWaves / 32][32];
    __shared__ float sH[32];
    for (int b = slot; b < e.B; b += a.slots) {
      const ProbState &gst = e.st[b];
      // loop condition of odometer.cpp:341-346 at the start of a level: normdp / normdp_init = 1 (every workgroup of the
      // pair evaluates it for itself; maxiter >= 1 is the host's condition for this form)
      int active = ((0 < e.maxiter) & (1.0f > e.ratio)) ? 1 : 0;
      {
        const unsigned *src = reinterpret_cast<const unsigned *>(&gst);
        unsigned *dst = reinterpret_cast<unsigned *>(&sSt);
        for (int i = tid; i < (int)(sizeof(ProbState) / 4); i += blockDim.x) dst[i] = src[i];
      }
      // ---- what k_level_tail does in the other launch forms: fixed-order f64 sum of the setup launch's H partials
      // (16 slices x 32 components, then the slices in order), full-pivot LU once per level, loop state reset --
      // while the pair's workers load their templates
      {
        const int j = tid & 31, sl = tid >> 5;
        double sacc = 0.0;
        const float *ph = e.partH + (size_t)b * a.nblk * kPartHStride + j;
        if (j < kHUnique) {
#pragma unroll 8
          for (int k = sl; k < a.nblk; k += 64 * kResWaves / 32) sacc += (double)ph[(size_t)k * kPartHStride];
        }
        sRedH[sl][j] = sacc;
      }
      __syncthreads();
      if (tid < kHUnique) {
        double sacc = 0.0;
#pragma unroll
        for (int sl = 0; sl < 64 * kResWaves / 32; ++sl) sacc += sRedH[sl][tid];
        sH[tid] = (float)sacc;
      }
      __syncthreads();
      if (wave == 0) {
        WaveSolver S;
        ws_factor(S, sH[h_unique_index(lane)], lane);
        ws_store_factor(S, sSt, lane);
        if (lane == 0) level_reset(sSt, e);
      }
      __syncthreads();
      if (!active) {  // nothing to iterate: the state (H, factors, reset loop state) still goes back
        if (wave == 0) {
          const unsigned *src = reinterpret_cast<const unsigned *>(&sSt);
          unsigned *dst = reinterpret_cast<unsigned *>(e.st + b);
          for (int i = lane; i < (int)(sizeof(ProbState) / 4); i += 64) dst[i] = src[i];
        }
        __syncthreads();
        continue;
      }
      while (active) {
        seq += 1;
        const unsigned tag = a.tag0 + seq;
        const unsigned long long *gslot = gbox + (size_t)(seq & 1u) * parts * kResSlot;
        unsigned long long *bslot = bbox + (size_t)(seq & 1u) * 16;
        // every wave sums a share of the workers: lane (rr, k) reads value k of the workers rr, rr + 4, ... of the
        // share, eight granule loads in flight per lane (32 workers per round trip and wave)
        RES_MARK(0)  // solver: loop overhead / barrier behind the previous broadcast
        const int k = lane & 15, rr = lane >> 4;
        const int per_wave = (parts + kResWaves - 1) / kResWaves;
        const int p_lo = wave * per_wave, p_hi = min(parts, p_lo + per_wave);
        double accd = 0.0;
        for (int r0 = p_lo; r0 < p_hi; r0 += 32) {
          const unsigned long long *src[8];
          unsigned long long g[8];
#define RES_SRC(u)                                                                  \
  {                                                                                 \
    const int r = r0 + u * 4 + rr;                                                  \
    src[u] = (k < 12 && r < p_hi) ? gslot + (size_t)r * kResSlot + k : nullptr;    \
  }
          RES_EACH8(RES_SRC)
          res_poll(pc, src, g, tag, lane);
#define RES_ACC(u) accd += (double)__builtin_bit_cast(float, (unsigned)(g[u] & 0xffffffffu));
          RES_EACH8(RES_ACC)
        }
        double tot = accd;
#pragma unroll
        for (int j = 1; j < 4; ++j) tot += res_gather64(accd, j * 16 + k);
        tot += res_gather64(tot, (lane + 6) & 63);  // hi + lo (meaningful in lanes 0-5)
        RES_MARK(1)  // solver: waiting for / summing the workers' granules
        if (lane < 6) sRed[wave][lane] = tot;
        __syncthreads();
        RES_MARK(2)  // solver: barrier
        if (wave == 0) {
          double bsum = 0.0;
          if (lane < 6)
            for (int w = 0; w < kResWaves; ++w) bsum += sRed[w][lane];
          WaveSolver S;
          float G[12];
          ws_load_state(S, sSt, lane, G);
          ws_load_factor(S, sSt, lane);
          ws_iterate(S, (float)bsum, sopt, a.level, b, lane, G);
          ws_store_state(S, sSt, lane, G);
          float gv = 0.0f;
#pragma unroll
          for (int q = 0; q < 12; ++q) gv = lane == q ? G[q] : gv;
          const int act = pc.dead ? 0 : S.active;  // a time-out ends the pair (the host reports the failure)
          if (lane == 12) gv = __builtin_bit_cast(float, act);
          if (lane < 13)
            __hip_atomic_store(bslot + lane,
                               ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned, gv),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (lane == 0) sG[12] = __builtin_bit_cast(float, act);
          RES_MARK(3)  // solver: solve + broadcast
        }
        __syncthreads();
        active = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sG[12]));
      }
      if (wave == 0) {  // final state of the level back to the problem's record
        const unsigned *src = reinterpret_cast<const unsigned *>(&sSt);
        unsigned *dst = reinterpret_cast<unsigned *>(e.st + b);
        for (int i = lane; i < (int)(sizeof(ProbState) / 4); i += 64) dst[i] = src[i];
      }
      __syncthreads();
    }
#ifdef ICTR_RES_PROF
    if (slot == 0 && tid == 0)
      for (int k = 0; k < 8; ++k) e.partH[8 + k] = (float)tp_[k];
#endif
    return;
  }

  // ================================================================ a worker workgroup: 128 points of the pair
  const LevelCam lc = a.lc;
  const int sw = lc.sw;
  const unsigned off_cd = (unsigned)((lane >> 3) * sw + (lane & 7)) * 4u;  // bytes from the window's top-left texel
  const unsigned off_ab = off_cd + (unsigned)sw * 4u;
  float4 *const recs = sRec[wave];
  for (int b = slot; b < e.B; b += a.slots) {
    const ProbState &gst = e.st[b];
    const int npts = gst.npts;
    int active = ((0 < e.maxiter) & (1.0f > e.ratio)) ? 1 : 0;  // as the solver workgroup evaluates it (odometer.cpp:341-346)
    if (!active) continue;
    const int i0 = part * kResQ + wave * kResPPW;
    const int cnt = min(kResPPW, max(0, npts - i0));  // this wave's points (wave-uniform)
    const PlaneSet pl = e.planes[b * e.nlev + a.level];
    const __amdgpu_buffer_rsrc_t rcur =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pl.cur), 0, 0x7fffffff, 0x00020000);
    // ---- templates of this wave's patches into registers (written by the level's setup launch; stale ones included)
    float Gx[kResPPW - kGxL], Gy[kResPPW];
    float *const tw = sT[wave] + lane;
    {
      const float *gT = e.T + ((size_t)b * M + i0) * 64 + lane;
      const float *gGx = e.Gx + ((size_t)b * M + i0) * 64 + lane;
      const float *gGy = e.Gy + ((size_t)b * M + i0) * 64 + lane;
#pragma unroll
      for (int j = 0; j < kResPPW; ++j) {
        float t = 0.0f, gx = 0.0f;
        Gy[j] = 0.0f;
        if (j < cnt) {
          t = __builtin_nontemporal_load(gT + j * 64);
          gx = __builtin_nontemporal_load(gGx + j * 64);
          Gy[j] = __builtin_nontemporal_load(gGy + j * 64);
        }
        tw[j * 64] = t;
        if (j < kGxL)
          tw[(kResPPW + j) * 64] = gx;
        else
          Gx[j - kGxL] = gx;
      }
    }
    // ---- this lane's point (lanes 0-15): X, Y, Z stay in registers, the level's coefficients go into the record
    const bool pv = lane < cnt;
    float X = 0.0f, Y = 0.0f, Z = 1.0f;
    if (lane < kResPPW) {
      float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
      if (pv) {
        const int ip = i0 + lane;
        const float *p3 = e.pt3d + (size_t)b * 3 * M;
        X = p3[ip], Y = p3[ip + M], Z = p3[ip + 2 * M];
        const float4 *c4 = reinterpret_cast<const float4 *>(e.coef + ((size_t)b * M + ip) * kCoefStride);
        q0 = c4[0], q1 = c4[1], q2 = c4[2];  // cx0..3 | cx4 cx5 cy0 cy1 | cy2..5
      }
      float4 *r4 = recs + lane * 4;
      r4[0] = make_float4(0.f, 0.f, 0.f, 0.f);
      r4[1] = make_float4(q0.z, q0.w, q1.x, q1.y);  // cx2 cx3 cx4 cx5
      r4[2] = make_float4(q2.x, q2.y, q2.z, q2.w);  // cy2 cy3 cy4 cy5
      r4[3] = make_float4(q0.x, q1.w, 0.0f, 0.0f);  // cx0 cy1 vis -
    }
    if (tid < 12) sG[tid] = gst.G[tid];
    __syncthreads();

    while (active) {  // uniform over the pair's workgroups: every one of them follows the same broadcast
      RES_MARK(0)  // worker: barrier behind the broadcast
      float Gc[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) Gc[k] = sG[k];
      // ---- stage 1 (pose.cpp:384-391, odometer.cpp:369-377)
      int base_v = 0;
      {
        const float tx = Gc[0] * X + Gc[1] * Y + Gc[2] * Z + Gc[3];
        const float ty = Gc[4] * X + Gc[5] * Y + Gc[6] * Z + Gc[7];
        const float tz = Gc[8] * X + Gc[9] * Y + Gc[10] * Z + Gc[11];
        const float mx = (tx / tz) * lc.fx + lc.cx;
        const float my = (ty / tz) * lc.fy + lc.cy;
        const bool vis = pv && in_view(mx, my, lc.swo, lc.sho);
        const Taps tp = make_taps(vis ? mx : 1.0f, vis ? my : 1.0f, 4);  // (1,1): a harmless in-plane window
        base_v = ((tp.row0 - 1) * sw + tp.col0 - 1) * 4;                 // bytes: the buffer load's scalar offset
        if (lane < kResPPW) {
          recs[lane * 4] = make_float4(tp.w1, tp.w0, tp.w3, tp.w2);
          reinterpret_cast<float *>(recs + lane * 4 + 3)[2] = vis ? 1.0f : 0.0f;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      RES_MARK(1)  // worker: stage 1
      // ---- stage 2: sixteen patches from registers, the windows of four patches in flight
      f32x2_t acc01 = {0.0f, 0.0f}, acc23 = {0.0f, 0.0f}, acc45 = {0.0f, 0.0f};
      constexpr int kD = 4;  // windows in flight (registers: 4 each)
      ResWin W[kD];
      auto issue = [&](int j) {
        const int soff = rlane(base_v, j);
        W[j % kD].cd = __builtin_bit_cast(f32x2_a4, __builtin_amdgcn_raw_buffer_load_b64(rcur, (int)off_cd, soff, 0));
        W[j % kD].ab = __builtin_bit_cast(f32x2_a4, __builtin_amdgcn_raw_buffer_load_b64(rcur, (int)off_ab, soff, 0));
      };
#pragma unroll
      for (int j = 0; j < kD; ++j) issue(j);
#pragma unroll
      for (int j = 0; j < kResPPW; ++j) {
        if ((j & 3) == 0 && j >= cnt) break;  // wave-uniform: whole groups of four behind the wave's last point
        // (compiler barrier tied to the sums: the next patch's record reads stay behind this patch's arithmetic)
        asm volatile("" : "+v"(acc01), "+v"(acc23), "+v"(acc45) : : "memory");
        const float4 *r4 = recs + j * 4;
        const float4 wv = r4[0], qx = r4[1], qy = r4[2], qz = r4[3];
        const ResWin w = W[j % kD];
        if (j + kD < kResPPW) issue(j + kD);
        // utilities.cpp:107 in the reference's operand order, not contracted: ((w0 a + w1 b) + w2 c) + w3 d
        const float inew = wv.y * w.ab.y + wv.x * w.ab.x + wv.w * w.cd.y + wv.z * w.cd.x;
        const float r = (tw[j * 64] - inew) * qz.z;  // pdiff (odometer.cpp:381); 0 out of the new view and for padding
        const float gxj = j < kGxL ? tw[(kResPPW + j) * 64] : Gx[j < kGxL ? 0 : j - kGxL];
        const f32x2_t g2 = {gxj * r, Gy[j] * r}, gr2 = {g2.x, g2.x}, hr2 = {g2.y, g2.y};
        acc01 = __builtin_elementwise_fma(g2, (f32x2_t){qz.x, qz.y}, acc01);  // sd1 = Gx cx0, sd2 = Gy cy1
        acc23 = __builtin_elementwise_fma(gr2, (f32x2_t){qx.x, qx.y},         // sd3..sd6 = Gx cxk + Gy cyk
                                          __builtin_elementwise_fma(hr2, (f32x2_t){qy.x, qy.y}, acc23));
        acc45 = __builtin_elementwise_fma(gr2, (f32x2_t){qx.z, qx.w},         // (odometer.cpp:319-326)
                                          __builtin_elementwise_fma(hr2, (f32x2_t){qy.z, qy.w}, acc45));
      }
      RES_MARK(2)  // worker: stage 2
      {
        const float accs[6] = {acc01.x, acc01.y, acc23.x, acc23.y, acc45.x, acc45.y};
        float o = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const float v = wave_sum_dpp(accs[k]);
          o = lane == k ? v : o;
        }
        if (lane < 6) sPart[wave][lane] = o;
      }
      __syncthreads();
      RES_MARK(3)  // worker: wave reduction + barrier
      // ---- gather: the workgroup's six sums as (hi, lo) float pairs -> the pair's mailbox; then the broadcast
      seq += 1;
      const unsigned tag = a.tag0 + seq;
      unsigned long long *gslot = gbox + (size_t)(seq & 1u) * parts * kResSlot;
      const unsigned long long *bslot = bbox + (size_t)(seq & 1u) * 16;
      if (wave == 0) {
        double bs = 0.0;
        if (lane < 6)
          for (int w = 0; w < kResWaves; ++w) bs += (double)sPart[w][lane];
        const float hi = (float)bs;
        const float lo = (float)(bs - (double)hi);
        const float pvv = lane < 6 ? hi : lane_gather(lo, lane - 6);
        if (lane < 12)
          __hip_atomic_store(gslot + (size_t)part * kResSlot + lane,
                             ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned, pvv),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long *src[8] = {lane < 13 ? bslot + lane : nullptr, nullptr, nullptr, nullptr,
                                            nullptr, nullptr, nullptr, nullptr};
        unsigned long long g[8];
        res_poll(pc, src, g, tag, lane);
        // a time-out ends the pair in this workgroup (bounded time; the host reports the tracking as failed)
        if (lane < 13) sG[lane] = (lane == 12 && pc.dead) ? 0.0f : __builtin_bit_cast(float, (unsigned)(g[0] & 0xffffffffu));
        RES_MARK(4)  // worker (wave 0): gather store + waiting for the broadcast
      }
      __syncthreads();
      active = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sG[12]));
    }
    RES_MARK(5)  // worker: pair prologue (template loads) -- charged at the next pair's end
    __syncthreads();  // sG and the records are rewritten for the next pair
  }
#ifdef ICTR_RES_PROF
  if (blockIdx.x == 0 && tid == 0)
    for (int k = 0; k < 8; ++k) e.partH[k] = (float)tp_[k];
#endif
}

// ---------------------------------------------------------------- host side
size_t resident_mail_bytes(int parts, int slots) {
  return sizeof(unsigned long long) * (size_t)slots * ((size_t)2 * parts * kResSlot + 2 * 16);
}
int resident_points_per_workgroup(void) { return kResQ; }
// workgroups of this kernel that one CU holds at once (0: the kernel cannot run)
int resident_blocks_per_cu(void) {
  static const int n = [] {
    int v = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, reinterpret_cast<const void *>(&k_level_resident),
                                                     64 * kResWaves, 0) != hipSuccess) {
      (void)hipGetLastError();
      v = 0;
    }
    return v;
  }();
  return n;
}
hipError_t launch_level_resident(const EngineDev &e, const LevelCam &lc, int level, int parts, int slots, int nblk,
                                 unsigned tag0, unsigned long long limit, unsigned long long *mail, int *err,
                                 hipStream_t s) {
  ResArgs a;
  a.nblk = nblk;
  a.lc = lc;
  a.level = level;
  a.parts = parts;
  a.slots = slots;
  a.tag0 = tag0;
  a.limit = limit;
  a.mail = mail;
  a.err = err;
  hipLaunchKernelGGL(k_level_resident, dim3((parts + 1) * slots), dim3(64 * kResWaves), 0, s, e, a);
  return hipGetLastError();
}

}  // namespace ictr
