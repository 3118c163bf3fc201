// ictr_resident.hip -- all Gauss-Newton iterations of a pyramid level with the templates RESIDENT on the chip
// (gfx950 / CDNA4; 8x8 patches, large problems: thousands to tens of thousands of points per frame pair).
//
// The per-iteration kernel k_iter8 streams T, Gx, Gy of every patch from HBM in every iteration: 12 of its 16 bytes
// per pixel, ten times per level, and it runs at the HBM roofline doing so. But a frame pair's templates are only 25 MB
// per level -- the chip has 128 MB of vector registers and 40 MB of LDS. Here a frame pair is shared by `parts` worker
// workgroups that load their templates ONCE per level and keep them for all iterations of odometer.cpp:344-418; an
// iteration then reads only the current frame's windows (cache-resident) and exchanges six numbers per workgroup.
//
// r03 form (what bounds this kernel is the number of frame pairs in flight -- an iteration is a latency chain of two
// hops through memory -- so the design maximises the templates a CU holds):
//   * a worker workgroup = 4 waves, a wave keeps THIRTY-TWO patches: Gx and Gy of a patch are two registers of the wave
//     (lane = pixel; 64 of its 128 registers), T sits in LDS (32 KB per workgroup); four workgroups per CU = 512 points
//     = 384 KB of templates per CU, FOUR 1080p frame pairs in flight on the chip (r02: two);
//   * J is constant over a patch (odometer.cpp:313-326), so b_k = sum_px sd_k r = sum_patches [cx_k A + cy_k B] with the
//     two per-patch sums A = sum Gx r, B = sum Gy r: per pixel the loop does the bilinear blend, the residual and two
//     products; the 64 per-lane values (A, B of 32 patches) are summed over the 64 lanes by a TRANSPOSING reduction
//     (each merge halves the registers and doubles the lanes summed: two instructions per value) that leaves lane l
//     with the complete sum of one (patch, A|B); the lane then applies ITS point's six coefficients -- no per-patch
//     coefficient broadcast, no per-pixel multiply-adds into six sums;
//   * the workgroup's six partial sums travel as six float granules (the precision of the streaming form's float
//     partials); the pair's SOLVER workgroup polls them in one round trip (256 lanes x 8 granules), adds in a fixed
//     order in f64, one wave runs the solver turn (WaveSolver, ictr_devfn.h: substitution with the level's LU factors,
//     pose update, exp map, loop condition; odometer.cpp:407-418, 509-515) and broadcasts cpos_G + the loop flag.
// The mailbox protocol is the one of the team form (ictr_track1.hip "Teams"): 8-byte granules {float bits, tag},
// tags = launch epoch << 12 | exchange number, double-buffered by parity, bounded polling with a sticky error flag.
// H partials come from the level's setup launch (k_ref8: three sums per patch) and are reduced + factored by the pair's
// solver workgroup at the start of the pair (what k_level_tail does in the other launch forms); templates and (possibly
// stale) coefficients come from the buffers that launch wrote: patches, coefficients and projections are bit-identical
// to the other launch forms, b differs by summation order only.
#include <string.h>

#include <type_traits>

#include "ictr_dev.h"
#include "ictr_devfn.h"
#include "se3_math.h"

namespace ictr {

constexpr int kResWaves = 4;                // waves per workgroup
constexpr int kResThreads = 64 * kResWaves;
// patches (points) per wave: template parameter NP of the kernel, 32 (large batches: four pairs in flight) or 16 (one
// or two pairs: twice the workgroups, half the patch loop); points per workgroup = kResWaves * NP
constexpr int kResSlot = 8;                 // granules per worker workgroup in the gather box (6 used)
#ifndef ICTR_RES_AUX
#define ICTR_RES_AUX 2  // template loads of the pair prologue: slc (streamed)
#endif
#ifndef ICTR_RES_D
#define ICTR_RES_D 4
#endif
constexpr int kResD = ICTR_RES_D;           // current-frame windows in flight per wave (4 registers each)

struct ResArgs {
  LevelCam lc;
  int level;
  int parts, slots;          // worker workgroups per frame pair; pairs in flight (grid = slots * (parts + 1))
  int nblk;                  // workgroups per problem of the level's setup launch (their H partials: e.partH)
  int dbg_mute;              // debug (variant bit 25): worker `dbg_mute - 1` never posts its sums (time-out test); 0 = off
  int prof_slot;             // ICTR_RES_PROF builds: the slot whose worker 0 / solver report their cycle counters
  int stagger;               // slot s starts s * stagger ticks (100 MHz) late: the pairs in flight on a CU leave lockstep
  ResXchg x;                 // x.world > 1: the solver workgroups sum H and b over the ranks (one-hop mailbox exchange)
  int prio_mode;             // wave priorities (s_setprio): 0 none; 1 by slot; 2 by slot, rotating with the slot's pair count
  unsigned tag0;             // launch epoch << 12
  unsigned long long limit;  // polling limit, wall_clock64 ticks (100 MHz)
  unsigned long long *mail;  // per slot: gather box [2][parts][kResSlot], broadcast box [2][16], H box [parts][kResHSlot]
  int *err;                  // sticky time-out flag (pinned host memory as the device sees it)
};

constexpr int kResHSlot = 24;               // granules per worker workgroup in the H box (21 used; fused setup only)
__host__ __device__ __forceinline__ size_t res_slot_granules(int parts) {
  return (size_t)2 * parts * kResSlot + 2 * 16 + (size_t)parts * kResHSlot;
}

struct ResPoll {
  unsigned long long limit;
  int *err;
  int dead;
};
// poll up to eight granules per lane until every tag matches; lanes / entries without a granule pass nullptr
// (entries written out one by one: everything stays in registers). TWO sets of requests are kept in flight, half a
// round trip apart: a granule is seen at most half a memory round trip after it became visible (one set: a whole one).
#define RES_EACH8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
__device__ __forceinline__ void res_poll(ResPoll &pc, const unsigned long long *(&src)[8], unsigned long long (&g)[8],
                                         unsigned tag, int lane) {
  const unsigned long long empty = (unsigned long long)tag << 32;
#define RES_LOAD(u) g[u] = src[u] ? __hip_atomic_load(src[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : empty;
  RES_EACH8(RES_LOAD)
  if (pc.dead) return;
  unsigned long long h[8];
  __builtin_amdgcn_s_sleep(4);
#define RES_LOADH(u) h[u] = src[u] ? __hip_atomic_load(src[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : empty;
  RES_EACH8(RES_LOADH)
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
    // the older set's answers replace what is still missing; that set is requested again for what is missing then
#define RES_ROTATE(u)                                                                                         \
  if ((unsigned)(g[u] >> 32) != tag) {                                                                        \
    g[u] = h[u];                                                                                              \
    h[u] = __hip_atomic_load(src[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                             \
  }
    RES_EACH8(RES_ROTATE)
  }
}
// one granule per lane (the workers' wait for the broadcast): three requests in flight
__device__ __forceinline__ unsigned long long res_poll1(ResPoll &pc, const unsigned long long *src, unsigned tag, int lane) {
  const unsigned long long empty = (unsigned long long)tag << 32;
  unsigned long long g = src ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : empty;
  if (pc.dead) return g;
  __builtin_amdgcn_s_sleep(3);
  unsigned long long h1 = src ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : empty;
  __builtin_amdgcn_s_sleep(3);
  unsigned long long h2 = src ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : empty;
  bool started = false;
  unsigned long long t0 = 0;
  while (__builtin_amdgcn_ballot_w64((unsigned)(g >> 32) != tag) != 0) {
    if (!started) {
      t0 = wall_clock64();
      started = true;
    } else if (wall_clock64() - t0 > pc.limit) {
      if (lane == 0) __hip_atomic_store(pc.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      pc.dead = 1;
      break;
    }
    if ((unsigned)(g >> 32) != tag) {
      g = h1;
      h1 = h2;
      h2 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  return g;
}

// (the transposing wave reduction -- tr_merge_*, TrAcc, tr_for_each_patch -- lives in ictr_devfn.h: k_ref8 uses it too)
// inspection (tests/test_gpu_parity.py): the reduction alone on caller data, vals[lane][2 patch + kind]
template <int NP>
__global__ __launch_bounds__(64) void k_debug_transpose_reduce(const float *__restrict__ vals, float *__restrict__ out,
                                                               int *__restrict__ patch_of_lane, int *__restrict__ kind_of_lane) {
  const int lane = threadIdx.x;
  TrAcc<NP> acc;
  tr_for_each_patch<NP, 0>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    acc.template push<j>(vals[lane * 64 + 2 * j], vals[lane * 64 + 2 * j + 1], lane);
  });
  out[lane] = acc.F;
  patch_of_lane[lane] = tr_patch_of_lane<NP>(lane);
  kind_of_lane[lane] = tr_kind_of_lane(lane);
}
hipError_t launch_debug_transpose_reduce(const float *vals, float *out, int *pl, int *kl, int np, hipStream_t s) {
  if (np == 16)
    hipLaunchKernelGGL(k_debug_transpose_reduce<16>, dim3(1), dim3(64), 0, s, vals, out, pl, kl);
  else
    hipLaunchKernelGGL(k_debug_transpose_reduce<32>, dim3(1), dim3(64), 0, s, vals, out, pl, kl);
  return hipGetLastError();
}

// Sharded resident form: lanes [0, n) of the solver's wave 0 each hold one local value; it goes as one granule {value,
// tag} into slot [parity][rank][pair * kXchgPerPair + off + lane] of EVERY rank's mailbox (system-scope stores over the
// point-to-point links), then the lane polls its own mailbox for the same granule of every rank and adds them in rank
// order in f64 -- the same bits on every rank, so the redundant solves stay in lockstep (the protocol of ictr_p2p.hip,
// inside the launch: no kernel boundary, no host, no communicator). Tags live in the upper half of the 32-bit space
// (the p2p object's own self-test uses the small sequence numbers).
__device__ __forceinline__ double res_xchg_sum(const ResXchg &x, ResPoll &pc, int pair, unsigned xs, int off, int n, float v,
                                               int lane) {
  const unsigned tag = 0x80000000u | xs;
  const size_t par = (size_t)(xs & 1u) * x.world;
  const size_t idx = (size_t)pair * kXchgPerPair + off + lane;
  if (lane < n) {
    const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v);
    for (int r = 0; r < x.world; ++r)
      __hip_atomic_store(x.peer[r] + (par + x.rank) * x.cap + idx, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // the peers' granules: eight ranks' requests in flight at a time (one fabric round trip for a node of eight), re-polled
  // until every tag matches; added in rank order
  double sum = 0.0;
  bool started = false;
  unsigned long long t0 = 0;
  const unsigned long long empty = (unsigned long long)tag << 32;
  for (int r0 = 0; r0 < x.world; r0 += 8) {
    unsigned long long g[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      g[u] = (lane < n && r0 + u < x.world)
                 ? __hip_atomic_load(x.local + (par + r0 + u) * x.cap + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                 : empty;
    while (!pc.dead) {
      bool miss = false;
#pragma unroll
      for (int u = 0; u < 8; ++u) miss |= (unsigned)(g[u] >> 32) != tag;
      if (__builtin_amdgcn_ballot_w64(miss) == 0) break;  // wave-uniform
      if (!started) {
        t0 = wall_clock64();
        started = true;
      } else if (wall_clock64() - t0 > pc.limit) {  // a peer rank never arrived: flag it, never wait again
        if (lane == 0) __hip_atomic_store(pc.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        pc.dead = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if ((unsigned)(g[u] >> 32) != tag)
          g[u] = __hip_atomic_load(x.local + (par + r0 + u) * x.cap + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += (double)__builtin_bit_cast(float, (unsigned)(g[u] & 0xffffffffu));
  }
  return sum;
}

// sum of a double over aligned groups of eight lanes (every lane of the group gets it)
__device__ __forceinline__ double res_sum8(double v) {
#define RES_DPP64(ctrl)                                                                              \
  {                                                                                                  \
    const int lo = dpp_mov<ctrl>(__double2loint(v)), hi = dpp_mov<ctrl>(__double2hiint(v));          \
    v += __hiloint2double(hi, lo);                                                                   \
  }
  RES_DPP64(0xB1)   // quad_perm [1,0,3,2]
  RES_DPP64(0x4E)   // quad_perm [2,3,0,1]
  RES_DPP64(0x141)  // row_half_mirror: the other quad of the group (its lanes all hold that quad's sum)
  return v;
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
#ifdef ICTR_RES_PROF  // diagnostic builds only: per-phase cycle counters of one wave (tools/resprof.py)
#define RES_MARK(k)                        \
  {                                        \
    t1_ = __builtin_readcyclecounter();    \
    tp_[k] += t1_ - t0_;                   \
    t0_ = t1_;                             \
  }
// time line (tools/restrace.py): 100 MHz wall-clock stamps of wave 0 of the workers that share CU `blockIdx % 256 == 0`
// (rows 0..7: one per slot) and of every slot's solver workgroup (rows 8..15), four stamps per iteration
constexpr int kTrIters = 128;
__device__ unsigned long long g_res_trace[16][kTrIters][4];
#define RES_STAMP(row, it, k) \
  if ((row) >= 0 && (it) < kTrIters && tid == 0) g_res_trace[row][it][k] = wall_clock64();
#else
#define RES_MARK(k)
#define RES_STAMP(row, it, k)
#endif
struct ResWin {
  f32x2_a4 ab, cd;  // (x-1,y),(x,y) and (x-1,y-1),(x,y-1)
};

__device__ __forceinline__ void res_setprio(int p) {  // wave-uniform p in 0..3 (the instruction takes an immediate)
  if (p == 0) __builtin_amdgcn_s_setprio(0);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(3);
}

// Register budget: 128 per wave = four 4-wave workgroups per CU (launch bounds), 64 of them Gx / Gy of the wave's patches.
// fused setup: after patch j is consumed, the windows of the patches up to res_upto(j) have been requested (j = -1: before
// the first). The depth shrinks as the template registers fill up (two registers per finished patch, twelve per window)
template <int NP>
constexpr int res_upto(int j) {
  if (j < 0) return (NP < 7 ? NP : 7) - 1;
  int d = (90 - 2 * j) / 12;
  d = d > 7 ? 7 : (d < 2 ? 2 : d);
  const int u = j + d;
  return u > NP - 1 ? NP - 1 : u;
}
template <int FROM, int TO, class Fn>
__device__ __forceinline__ void res_issue_range(Fn &&fn) {  // fn(k) for k = FROM .. TO (compile-time bounds)
  if constexpr (FROM <= TO) {
    fn(FROM);
    res_issue_range<FROM + 1, TO>(fn);
  }
}
// FUSED: the level's setup (steps 4-6: odometer.cpp:268-334, 428-472) happens in the pair's prologue -- every wave
// gathers and blends the reference patches of its own points straight into the registers / LDS they stay in, sums
// S = (sum Gx^2, sum Gx Gy, sum Gy^2) per patch and posts the workgroup's part of H = sum J^T S J to the pair's solver
// workgroup; T / Gx / Gy / coefficients are still written through to the batch's buffers (the state that later levels
// and later trackings fall back to for points out of view, odometer.cpp:304; what read_buffer shows), but nothing is
// read back from them except such stale patches. No setup launch, no k_level_tail.
template <int NP, bool FUSED>
__global__ __launch_bounds__(kResThreads, 4) void k_level_resident(EngineDev e, ResArgs a) {
  constexpr int kResPPW = NP, kResQ = kResWaves * NP;
  __shared__ __attribute__((aligned(16))) float4 sRecW[kResWaves][kResPPW];  // per point: bilinear weights [w1 w0 w3 w2]
  __shared__ float sT[kResWaves][kResPPW * 64];  // T of the wave's patches (lane = pixel)
  __shared__ float sPart[kResWaves][8];
  __shared__ float sPartH[kResWaves][32];  // FUSED: the waves' parts of H (21 used)
  __shared__ double sRed[kResWaves][8];
  __shared__ float sG[16];   // cpos_G of the current iteration, [12] = loop flag (bits)
  __shared__ ProbState sSt;  // solver workgroup: the problem's state between the solver's turns

  const int parts = a.parts;
  const int group = parts + 1;  // a pair's workgroups: `parts` workers (128 points each) + ONE solver workgroup (index parts)
  const int slot = (int)blockIdx.x / group;
  const int part = (int)blockIdx.x - slot * group;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = e.M;
  unsigned long long *gbox = a.mail + (size_t)slot * res_slot_granules(parts);
  unsigned long long *bbox = gbox + (size_t)2 * parts * kResSlot;
  unsigned long long *hbox = bbox + 2 * 16;
  ResPoll pc;
  pc.limit = a.limit;
  pc.err = a.err;
  pc.dead = 0;
  unsigned seq = 0;
#ifdef ICTR_RES_PROF
  const int tr_row = part == parts ? (slot < 8 ? 8 + slot : -1) : ((part == 0 && slot < 8) ? slot : -1);
  int tr_it = 0;
  unsigned long long tp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_readcyclecounter(), t1_ = 0;
#endif

  if (a.stagger > 0 && slot > 0) {
    const unsigned long long t0s = wall_clock64(), dts = (unsigned long long)slot * (unsigned)a.stagger;
    while (wall_clock64() - t0s < dts) __builtin_amdgcn_s_sleep(8);
  }
  if (part == parts) {
    if (a.prio_mode) __builtin_amdgcn_s_setprio(3);
    // ================================================================ the pair's solver workgroup
    SolveOpts sopt = solve_opts(e);
    sopt.robust = 0;
    __shared__ double sRedH[kResThreads / 32][32];
    __shared__ float sH[32];
    for (int b = slot, round = 0; b < e.B; b += a.slots, ++round) {
      const ProbState &gst = e.st[b];
      // loop condition of odometer.cpp:341-346 at the start of a level: normdp / normdp_init = 1 (every workgroup of the
      // pair evaluates it for itself; maxiter >= 1 is the host's condition for this form)
      int active = ((0 < e.maxiter) & (1.0f > e.ratio)) ? 1 : 0;
      {
        const unsigned *src = reinterpret_cast<const unsigned *>(&gst);
        unsigned *dst = reinterpret_cast<unsigned *>(&sSt);
        for (int i = tid; i < (int)(sizeof(ProbState) / 4); i += kResThreads) dst[i] = src[i];
      }
      // ---- what k_level_tail does in the other launch forms: fixed-order f64 sum of the H partials (8 slices x 32
      // components, then the slices in order), full-pivot LU once per level, loop state reset. Unfused: the setup
      // launch's partials (e.partH), while the pair's workers load their templates; FUSED: the workers' own partials out
      // of the pair's H box (granules tagged with the slot's pair count), while they run their first stage 2
      {
        const int j = tid & 31, sl = tid >> 5;
        double sacc = 0.0;
        if constexpr (FUSED) {
          const unsigned tagh = a.tag0 + (unsigned)round + 1u;
          for (int r0 = 0; r0 < parts; r0 += 8 * (kResThreads / 32)) {
            const unsigned long long *src[8];
            unsigned long long g[8];
#define RES_HSRC(u)                                                                        \
  {                                                                                        \
    const int r = r0 + u * (kResThreads / 32) + sl;                                        \
    src[u] = (j < kHUnique && r < parts) ? hbox + (size_t)r * kResHSlot + j : nullptr;     \
  }
            RES_EACH8(RES_HSRC)
            res_poll(pc, src, g, tagh, lane);
#define RES_HACC(u) sacc += (double)__builtin_bit_cast(float, (unsigned)(g[u] & 0xffffffffu));
            RES_EACH8(RES_HACC)
          }
        } else {
          const float *ph = e.partH + (size_t)b * a.nblk * kPartHStride + j;
          if (j < kHUnique) {
#pragma unroll 8
            for (int k = sl; k < a.nblk; k += kResThreads / 32) sacc += (double)ph[(size_t)k * kPartHStride];
          }
        }
        sRedH[sl][j] = sacc;
      }
      __syncthreads();
      if (tid < kHUnique) {
        double sacc = 0.0;
#pragma unroll
        for (int sl = 0; sl < kResThreads / 32; ++sl) sacc += sRedH[sl][tid];
        sH[tid] = (float)sacc;
      }
      __syncthreads();
      unsigned xs = 0;  // (sharded resident form) this pair's exchange count, wave 0 only
      if (wave == 0) {
        float hl = lane < kHUnique ? sH[lane] : 0.0f;
        if (a.x.world > 1) {  // H of the level = the sum over the ranks' point shards
          xs = a.x.xseq[b] + 1u;
          hl = (float)res_xchg_sum(a.x, pc, b, xs, 12, kHUnique, hl, lane);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (every lane has read its sH entry)
          __builtin_amdgcn_wave_barrier();
          if (lane < kHUnique) sH[lane] = hl;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        WaveSolver S;
        ws_factor(S, sH[h_unique_index(lane)], lane);
        ws_store_factor(S, sSt, lane);
        if (lane == 0) level_reset(sSt, e);
      }
      __syncthreads();
      while (active) {
        seq += 1;
        const unsigned tag = a.tag0 + seq;
        const unsigned long long *gslot = gbox + (size_t)(seq & 1u) * parts * kResSlot;
        unsigned long long *bslot = bbox + (size_t)(seq & 1u) * 16;
        // lane (k, rr) = 8 k + rr (k < 6) reads value k of the workers ((8 wave + u) 8 + rr), u = 0..7: eight granule
        // loads in flight per lane, 256 workers per round trip; sums in a fixed order: u, rounds, rr, waves
        RES_MARK(0)  // solver: loop overhead / barrier behind the previous broadcast
        RES_STAMP(tr_row, tr_it, 0)
        const int k = lane >> 3, rr = lane & 7;
        double accd = 0.0;
        for (int r0 = 0; r0 < parts; r0 += 64 * kResWaves) {
          const unsigned long long *src[8];
          unsigned long long g[8];
#define RES_SRC(u)                                                                   \
  {                                                                                  \
    const int r = r0 + (wave * 8 + u) * 8 + rr;                                      \
    src[u] = (k < 6 && r < parts) ? gslot + (size_t)r * kResSlot + k : nullptr;      \
  }
          RES_EACH8(RES_SRC)
          res_poll(pc, src, g, tag, lane);
#define RES_ACC(u) accd += (double)__builtin_bit_cast(float, (unsigned)(g[u] & 0xffffffffu));
          RES_EACH8(RES_ACC)
        }
        const double tot = res_sum8(accd);
        RES_MARK(1)  // solver: waiting for / summing the workers' granules
        RES_STAMP(tr_row, tr_it, 1)
        if (rr == 0 && k < 6) sRed[wave][k] = tot;
        __syncthreads();
        RES_MARK(2)  // solver: barrier
        RES_STAMP(tr_row, tr_it, 2)
        if (wave == 0) {
          double bsum = 0.0;
          if (lane < 6)
            for (int w = 0; w < kResWaves; ++w) bsum += sRed[w][lane];
          if (a.x.world > 1) {
            // b is a sum of signed terms: a shard's partial can be much larger than the total, so it travels as a (hi, lo)
            // pair of floats (lanes 0-5 / 6-11) and the ranks' total is rounded once (as the team form does)
            xs += 1u;
            const float hi = (float)bsum;
            const float lo = (float)(bsum - (double)hi);
            const float pv = lane < 6 ? hi : lane_gather(lo, lane - 6);
            const double t = res_xchg_sum(a.x, pc, b, xs, 0, 12, pv, lane);
            const int l6 = lane + 6;
            bsum = t + __hiloint2double(lane_gather(__double2hiint(t), l6), lane_gather(__double2loint(t), l6));
          }
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
          RES_STAMP(tr_row, tr_it, 3)
        }
#ifdef ICTR_RES_PROF
        tr_it += 1;
#endif
        __syncthreads();
        active = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sG[12]));
      }
      if (wave == 0) {  // final state of the level (H, factors, pose, loop state) back to the problem's record
        if (a.x.world > 1 && lane == 0) a.x.xseq[b] = xs;
        const unsigned *src = reinterpret_cast<const unsigned *>(&sSt);
        unsigned *dst = reinterpret_cast<unsigned *>(e.st + b);
        for (int i = lane; i < (int)(sizeof(ProbState) / 4); i += 64) dst[i] = src[i];
      }
      __syncthreads();
    }
#ifdef ICTR_RES_PROF
    if (slot == a.prof_slot && tid == 0)
      for (int k = 0; k < 8; ++k) e.partH[8 + k] = (float)tp_[k];
#endif
    return;
  }

  // ================================================================ a worker workgroup: 128 points of the pair
  const LevelCam lc = a.lc;
  const int sw = lc.sw;
  const unsigned off_cd = (unsigned)((lane >> 3) * sw + (lane & 7)) * 4u;  // bytes from the window's top-left texel
  const unsigned off_ab = off_cd + (unsigned)sw * 4u;
  float4 *const recs = sRecW[wave];
  float *const tw = sT[wave] + lane;
  const int my_patch = tr_patch_of_lane<NP>(lane);  // the (patch, A|B) whose complete sum this lane holds after stage 2
  const int my_kind = tr_kind_of_lane(lane);
  for (int b = slot, round = 0; b < e.B; b += a.slots, ++round) {
    if (a.prio_mode) res_setprio((a.prio_mode == 2 ? slot + round : slot) & 3);
    const ProbState &gst = e.st[b];
    const int npts = gst.npts;
    int active = ((0 < e.maxiter) & (1.0f > e.ratio)) ? 1 : 0;  // as the solver workgroup evaluates it (odometer.cpp:341-346)
    if (!active) continue;
    const int i0 = part * kResQ + wave * kResPPW;
    const int cnt = min(kResPPW, max(0, npts - i0));  // this wave's points (wave-uniform)
    // (the plane pointer is wave-uniform, but it comes out of a table in memory: say so, or every buffer load below is
    // wrapped in a waterfall loop over its descriptor)
    const float *cur_plane;
    {
      const unsigned long long pc64 = reinterpret_cast<unsigned long long>(e.planes[b * e.nlev + a.level].cur);
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pc64), hi = __builtin_amdgcn_readfirstlane((unsigned)(pc64 >> 32));
      cur_plane = reinterpret_cast<const float *>(((unsigned long long)hi << 32) | lo);
    }
    const __amdgpu_buffer_rsrc_t rcur =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(cur_plane), 0, 0x7fffffff, 0x00020000);
    f32x2_t Gxy[kResPPW];  // {Gx, Gy} of the lane's pixel, patch by patch: one register pair, one packed multiply by r
    const float *gT = e.T + ((size_t)b * M + i0) * 64 + lane;
    const float *gGx = e.Gx + ((size_t)b * M + i0) * 64 + lane;
    const float *gGy = e.Gy + ((size_t)b * M + i0) * 64 + lane;
    const bool pv = lane < cnt;
    float c6[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (FUSED) {
      // ---- setup, stage 1 (lane l < NP <-> point i0 + l): visibility in the reference view (odometer.cpp:273-282),
      // steepest-descent coefficients (:313-326; a point out of view keeps its stale line, :304), bilinear weights and
      // window base of the reference patch (utilities.cpp:66-77)
      const PlaneSet *plp = e.planes + (b * e.nlev + a.level);
      auto uniform_plane = [](const float *q) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float *>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff,
                                                 0x00020000);
      };
      const __amdgpu_buffer_rsrc_t rref = uniform_plane(plp->ref), rdx = uniform_plane(plp->dx), rdy = uniform_plane(plp->dy);
      const float *pt2 = e.pt2d + ((size_t)b * e.nlev + a.level) * 2 * M;
      const float *p3r = e.pt3d_ref + (size_t)b * 3 * M;
      const int ip = i0 + lane;
      float mx = 0.0f, my = 0.0f;
      if (pv) {
        mx = pt2[ip];
        my = pt2[ip + M];
      }
      const bool vis = pv && in_view(mx, my, lc.swo, lc.sho);
      float cx[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, cy[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      if (pv) {
        float4 *c4 = reinterpret_cast<float4 *>(e.coef + ((size_t)b * M + ip) * kCoefStride);
        if (vis) {
          sd_coefs(p3r[ip], p3r[ip + M], p3r[ip + 2 * M], lc.fx, lc.fy, cx, cy);
          c4[0] = make_float4(cx[0], cx[1], cx[2], cx[3]);
          c4[1] = make_float4(cx[4], cx[5], cy[0], cy[1]);
          c4[2] = make_float4(cy[2], cy[3], cy[4], cy[5]);
        } else {
          const float4 a0 = c4[0], a1 = c4[1], a2 = c4[2];
          cx[0] = a0.x; cx[1] = a0.y; cx[2] = a0.z; cx[3] = a0.w; cx[4] = a1.x; cx[5] = a1.y;
          cy[0] = a1.z; cy[1] = a1.w; cy[2] = a2.x; cy[3] = a2.y; cy[4] = a2.z; cy[5] = a2.w;
        }
      }
      const unsigned vmask = (unsigned)__builtin_amdgcn_ballot_w64(vis);          // points are lanes 0 .. NP-1 <= 31
      const unsigned smask = (unsigned)__builtin_amdgcn_ballot_w64(pv && !vis);   // stale patches stay in force
      int base_r;
      {
        const Taps tp = make_taps(vis ? mx : 1.0f, vis ? my : 1.0f, 4);  // (1,1): a harmless in-plane window
        base_r = ((tp.row0 - 1) * sw + tp.col0 - 1) * 4;
        if (lane < kResPPW) recs[lane] = make_float4(tp.w1, tp.w0, tp.w3, tp.w2);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // ---- setup, stage 2 (utilities.cpp:115-189; lane = pixel): the three planes' windows. What bounds this phase
      // is the number of windows a wave has in flight (first-touch latency of the reference planes), and early in the
      // loop most template registers are still free: seven patches in flight at the start, two at the end (res_depth).
      // Blends in the reference's operand order, products rounded one by one (bit-identical to k_ref8 / the CPU path)
      struct RefWin {
        f32x2_a4 ab[3], cd[3];
      };
      RefWin RW[8];
      auto issue_ref = [&](int j) {
#if defined(ICTR_RES_ABL) && ICTR_RES_ABL == 4  // timing ablation: every reference window = the plane's first
        const int soff = 0;
#else
        const int soff = rlane(base_r, j);
#endif
        RefWin &w = RW[j & 7];
        w.cd[0] = __builtin_bit_cast(f32x2_a4, __builtin_amdgcn_raw_buffer_load_b64(rref, (int)off_cd, soff, 0));
        w.ab[0] = __builtin_bit_cast(f32x2_a4, __builtin_amdgcn_raw_buffer_load_b64(rref, (int)off_ab, soff, 0));
        w.cd[1] = __builtin_bit_cast(f32x2_a4, __builtin_amdgcn_raw_buffer_load_b64(rdx, (int)off_cd, soff, 0));
        w.ab[1] = __builtin_bit_cast(f32x2_a4, __builtin_amdgcn_raw_buffer_load_b64(rdx, (int)off_ab, soff, 0));
        w.cd[2] = __builtin_bit_cast(f32x2_a4, __builtin_amdgcn_raw_buffer_load_b64(rdy, (int)off_cd, soff, 0));
        w.ab[2] = __builtin_bit_cast(f32x2_a4, __builtin_amdgcn_raw_buffer_load_b64(rdy, (int)off_ab, soff, 0));
      };
      tr_for_each_patch<res_upto<NP>(-1) + 1, 0>([&](auto jc) { issue_ref(decltype(jc)::value); });
      tr_for_each_patch<NP, 0>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const float4 wv = recs[j];
        const RefWin w = RW[j & 7];
        float v3[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const f32x2_t pab = f32x2_t{wv.x, wv.y} * f32x2_t{w.ab[k].x, w.ab[k].y};
          const f32x2_t pcd = f32x2_t{wv.z, wv.w} * f32x2_t{w.cd[k].x, w.cd[k].y};
          v3[k] = ((pab.y + pab.x) + pcd.y) + pcd.x;
        }
        asm volatile("" ::: "memory");  // (the next windows are requested only now: their registers are this patch's)
        res_issue_range<res_upto<NP>(j - 1) + 1, res_upto<NP>(j)>(issue_ref);
        const bool vj = (vmask >> j) & 1u;  // wave-uniform
        const float t = vj ? v3[0] : 0.0f;
        Gxy[j] = f32x2_t{vj ? v3[1] : 0.0f, vj ? v3[2] : 0.0f};
        tw[j * 64] = t;
#if defined(ICTR_RES_ABL) && ICTR_RES_ABL == 3  // timing ablation: no write-through
        if (false) {
#else
        if (vj) {  // written through: the state a point falls back to when it leaves the reference view
#endif
          __builtin_nontemporal_store(t, const_cast<float *>(gT) + j * 64);
          __builtin_nontemporal_store(Gxy[j].x, const_cast<float *>(gGx) + j * 64);
          __builtin_nontemporal_store(Gxy[j].y, const_cast<float *>(gGy) + j * 64);
        }
      });
      if (smask != 0u) {  // rare: points out of the reference view at this level keep their stale patches
        tr_for_each_patch<NP, 0>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if ((smask >> j) & 1u) {
            const float t = __builtin_nontemporal_load(gT + j * 64);
            Gxy[j].x = __builtin_nontemporal_load(gGx + j * 64);
            Gxy[j].y = __builtin_nontemporal_load(gGy + j * 64);
            tw[j * 64] = t;
          }
        });
      }
      // ---- S = (sum Gx^2, sum Gx Gy, sum Gy^2) of every patch by the transposing reduction: lane (patch, kind) of accS
      // holds Sxx | Syy, lane (pair, kind) of accX the Sxy of patch pair | pair + 16; then lane = point again
      TrAcc<NP> accS;
      tr_for_each_patch<NP, 0>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const f32x2_t q = Gxy[j] * Gxy[j];
        accS.template push<j>(q.x, q.y, lane);
      });
      TrAcc<16> accX;
      tr_for_each_patch<16, 0>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (NP == 32)
          accX.template push<j>(Gxy[j].x * Gxy[j].y, Gxy[j + 16].x * Gxy[j + 16].y, lane);
        else
          accX.template push<j>(Gxy[j].x * Gxy[j].y, 0.0f, lane);
      });
      {
        const int pp = lane & (NP - 1);
        const float sxx = lane_gather(accS.F, tr_lane_of<NP>(pp, 0)), syy = lane_gather(accS.F, tr_lane_of<NP>(pp, 1));
        const float sxy = lane_gather(accX.F, tr_lane_of<16>(pp & 15, NP == 32 ? (pp >> 4) : 0));
        // H += J^T S J (odometer.cpp:428-472 with sd_k = Gx cx_k + Gy cy_k), lane = point; coefficients are zero beyond
        // the wave's points. Same arithmetic as k_ref8.
        float uj[6], vj[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          uj[j] = __builtin_fmaf(cx[j], sxx, cy[j] * sxy);
          vj[j] = __builtin_fmaf(cx[j], sxy, cy[j] * syy);
        }
        const bool cntl = lane < NP;
        float accH = 0.0f;
        int jk = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int k = j; k < 6; ++k, ++jk) {
            const float h = wave_sum_dpp(cntl ? __builtin_fmaf(cx[k], uj[j], cy[k] * vj[j]) : 0.0f);
            accH = lane == jk ? h : accH;
          }
        if (lane < 32) sPartH[wave][lane] = lane < kHUnique ? accH : 0.0f;
      }
      // every lane: the six coefficients of ITS (patch, A|B)
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float vx = lane_gather(cx[k], my_patch), vy = lane_gather(cy[k], my_patch);
        c6[k] = (my_patch < cnt && tr_primary_lane<NP>(lane)) ? (my_kind ? vy : vx) : 0.0f;
      }
    } else {
      // ---- templates of this wave's patches (written by the level's setup launch; stale ones included): Gx, Gy into
      // registers, T into LDS. Buffer loads bounded by the wave's point count (patches beyond it read as zeros: no
      // branches), Gx / Gy of all patches requested at once straight into the registers they stay in (one memory round
      // trip for the whole wave), T eight patches at a time (it only passes through)
      const int nbytes = __builtin_amdgcn_readfirstlane(cnt * 256);
      auto bounded = [&](const float *q) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(q - lane);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float *>(((unsigned long long)hi << 32) | lo), 0, nbytes,
                                                 0x00020000);
      };
      const __amdgpu_buffer_rsrc_t rT = bounded(gT), rGx = bounded(gGx), rGy = bounded(gGy);
      const int loff = lane * 4;
#pragma unroll
      for (int j = 0; j < kResPPW; ++j) {
        Gxy[j].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rGx, loff, j * 256, ICTR_RES_AUX));  // slc: streamed
        Gxy[j].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rGy, loff, j * 256, ICTR_RES_AUX));
      }
#pragma unroll
      for (int j0 = 0; j0 < kResPPW; j0 += 8) {
        float t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t8[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rT, loff, (j0 + u) * 256, ICTR_RES_AUX));
#pragma unroll
        for (int u = 0; u < 8; ++u) tw[(j0 + u) * 64] = t8[u];
      }
      // every lane: the six coefficients of ITS (patch, A|B): cx_k of point my_patch for an A lane, cy_k for a B lane
      // (odometer.cpp:313-326; zeros beyond the wave's points)
      if (my_patch < cnt && tr_primary_lane<NP>(lane)) {
        const float *cl = e.coef + ((size_t)b * M + i0 + my_patch) * kCoefStride + (my_kind ? 6 : 0);
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = cl[k];
      }
    }
    // ---- lane l < NP: point i0 + l (X, Y, Z stay in registers)
    float X = 0.0f, Y = 0.0f, Z = 1.0f;
    if (pv) {
      const float *p3 = e.pt3d + (size_t)b * 3 * M;
      X = p3[i0 + lane], Y = p3[i0 + lane + M], Z = p3[i0 + lane + 2 * M];
    }
    if (tid < 12) sG[tid] = gst.G[tid];
    __syncthreads();
    if constexpr (FUSED) {  // the workgroup's part of H -> the pair's H box (the solver workgroup sums and factors it)
      if (wave == 0 && lane < kHUnique) {
        const float hv = (sPartH[0][lane] + sPartH[1][lane]) + (sPartH[2][lane] + sPartH[3][lane]);
        __hip_atomic_store(hbox + (size_t)part * kResHSlot + lane,
                           ((unsigned long long)(a.tag0 + (unsigned)round + 1u) << 32) |
                               (unsigned long long)__builtin_bit_cast(unsigned, hv),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }

    while (active) {  // uniform over the pair's workgroups: every one of them follows the same broadcast
      RES_MARK(0)  // worker: barrier behind the broadcast
      RES_STAMP(tr_row, tr_it, 0)
      // ---- stage 1 (pose.cpp:384-391, odometer.cpp:369-377): lane l < 32 <-> point i0 + l
      int base_v;
      float vis_f;
      {
        float Gc[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) Gc[k] = sG[k];
        const float tx = Gc[0] * X + Gc[1] * Y + Gc[2] * Z + Gc[3];
        const float ty = Gc[4] * X + Gc[5] * Y + Gc[6] * Z + Gc[7];
        const float tz = Gc[8] * X + Gc[9] * Y + Gc[10] * Z + Gc[11];
        const float mx = (tx / tz) * lc.fx + lc.cx;
        const float my = (ty / tz) * lc.fy + lc.cy;
        const bool vis = pv && in_view(mx, my, lc.swo, lc.sho);
        const Taps tp = make_taps(vis ? mx : 1.0f, vis ? my : 1.0f, 4);  // (1,1): a harmless in-plane window
        base_v = ((tp.row0 - 1) * sw + tp.col0 - 1) * 4;                 // bytes: the buffer load's scalar offset
        vis_f = vis ? 1.0f : 0.0f;
        if (lane < kResPPW) recs[lane] = make_float4(tp.w1, tp.w0, tp.w3, tp.w2);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      RES_MARK(1)  // worker: stage 1
      // ---- stage 2: thirty-two patches from registers, kResD windows in flight. Per pixel: blend (utilities.cpp:107,
      // reference operand order, not contracted), residual (odometer.cpp:381), Gx r and Gy r; the per-patch sums by the
      // transposing reduction: a patch's (A, B) pair merges on lane bit 2, then a binary counter of pending registers
      ResWin W[kResD];
      auto issue = [&](int j) {
#if defined(ICTR_RES_ABL) && ICTR_RES_ABL == 2  // timing ablations (diagnostic builds, wrong results): no window loads in the loop
        if (j >= kResD) return;
#endif
#if defined(ICTR_RES_ABL) && ICTR_RES_ABL == 1  // ... every window = the plane's first (cache-resident)
        const int soff = 0;
#else
        const int soff = rlane(base_v, j);
#endif
        W[j % kResD].cd = __builtin_bit_cast(f32x2_a4, __builtin_amdgcn_raw_buffer_load_b64(rcur, (int)off_cd, soff, 0));
        W[j % kResD].ab = __builtin_bit_cast(f32x2_a4, __builtin_amdgcn_raw_buffer_load_b64(rcur, (int)off_ab, soff, 0));
      };
#pragma unroll
      for (int j = 0; j < kResD; ++j) issue(j);
      TrAcc<NP> acc;
      float4 wv_n = recs[0];  // the next patch's weights and T: one patch ahead of the arithmetic (LDS latency hidden)
      float t_n = tw[0];
      tr_for_each_patch<NP, 0>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const float4 wv = wv_n;
        const float t = t_n;
        asm volatile("" ::: "memory");  // (the LDS reads of later patches stay behind this point: two patches' worth live)
        if constexpr (j + 1 < kResPPW) {
          wv_n = recs[j + 1];
          t_n = tw[(j + 1) * 64];
        }
        const ResWin w = W[j % kResD];
        // w0 a + w1 b + w2 c + w3 d in the reference's order; the four products as two packed multiplies (each product
        // rounded on its own, like the scalar form)
        const f32x2_t pab = f32x2_t{wv.x, wv.y} * f32x2_t{w.ab.x, w.ab.y}, pcd = f32x2_t{wv.z, wv.w} * f32x2_t{w.cd.x, w.cd.y};
        const float inew = ((pab.y + pab.x) + pcd.y) + pcd.x;
        if constexpr (j + kResD < kResPPW) issue(j + kResD);
        const float r = t - inew;  // pdiff (odometer.cpp:381); visibility is applied to the patch sums below
        const f32x2_t gr = Gxy[j] * f32x2_t{r, r};
        acc.template push<j>(gr.x, gr.y, lane);
      });
      const float F = acc.F;
      RES_MARK(2)  // worker: stage 2
      RES_STAMP(tr_row, tr_it, 1)
      // ---- lane = (patch, A|B): its point's visibility and coefficients, then the six sums over the wave
      {
        const float val = F * lane_gather(vis_f, my_patch);  // 0 out of the new view (ind_new) and beyond the wave's points
        float o = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const float v = wave_sum_dpp(c6[k] * val);  // b_k = sum_patches cx_k A + cy_k B (odometer.cpp:386-404)
          o = lane == k ? v : o;
        }
        if (lane < 6) sPart[wave][lane] = o;
      }
      __syncthreads();
      RES_MARK(3)  // worker: wave reduction + barrier
      RES_STAMP(tr_row, tr_it, 2)
      // ---- gather: the workgroup's six sums -> the pair's mailbox; then the broadcast
      seq += 1;
      const unsigned tag = a.tag0 + seq;
      unsigned long long *gslot = gbox + (size_t)(seq & 1u) * parts * kResSlot;
      const unsigned long long *bslot = bbox + (size_t)(seq & 1u) * 16;
      if (wave == 0) {
        int lane_v = lane;  // (opaque copy: the two mailbox addresses are formed here, not kept in registers over the loop)
        asm volatile("" : "+v"(lane_v));
        float bs = 0.0f;
        if (lane < 6) bs = (sPart[0][lane] + sPart[1][lane]) + (sPart[2][lane] + sPart[3][lane]);
        if (lane < 6 && part + 1 != a.dbg_mute)
          __hip_atomic_store(gslot + (size_t)part * kResSlot + lane_v,
                             ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned, bs),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long g = res_poll1(pc, lane < 13 ? bslot + lane_v : nullptr, tag, lane);
        // a time-out ends the pair in this workgroup (bounded time; the host reports the tracking as failed)
        if (lane < 13) sG[lane] = (lane == 12 && pc.dead) ? 0.0f : __builtin_bit_cast(float, (unsigned)(g & 0xffffffffu));
        RES_MARK(4)  // worker (wave 0): gather store + waiting for the broadcast
        RES_STAMP(tr_row, tr_it, 3)
      }
#ifdef ICTR_RES_PROF
      tr_it += 1;
#endif
      __syncthreads();
      active = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sG[12]));
    }
    RES_MARK(5)  // worker: pair prologue (template loads) -- charged at the next pair's end
    __syncthreads();  // sG and the records are rewritten for the next pair
  }
#ifdef ICTR_RES_PROF
  if (slot == a.prof_slot && part == 0 && tid == 0)
    for (int k = 0; k < 8; ++k) e.partH[k] = (float)tp_[k];
#endif
}

// ---------------------------------------------------------------- host side
#ifdef ICTR_RES_PROF
extern "C" int ictr_prof_res_trace(unsigned long long *out, int clear) {  // out[16][128][4]
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_res_trace), sizeof(g_res_trace)) != hipSuccess) return 1;
  if (clear) {
    static unsigned long long z[16][kTrIters][4];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_res_trace), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
size_t resident_mail_bytes(int parts, int slots) {
  return sizeof(unsigned long long) * (size_t)slots * res_slot_granules(parts);
}
int resident_points_per_workgroup(int np) { return kResWaves * np; }
template <int NP, bool FUSED>
static int res_occupancy() {
  int v = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, reinterpret_cast<const void *>(&k_level_resident<NP, FUSED>),
                                                   kResThreads, 0) != hipSuccess) {
    (void)hipGetLastError();
    v = 0;
  }
  return v;
}
// workgroups of this kernel that one CU holds at once (0: the kernel cannot run)
int resident_blocks_per_cu(int np, int fused) {
  static const int n[4] = {res_occupancy<16, false>(), res_occupancy<32, false>(), res_occupancy<16, true>(),
                           res_occupancy<32, true>()};
  return n[(np == 32 ? 1 : 0) + (fused ? 2 : 0)];
}
hipError_t launch_level_resident(const EngineDev &e, const LevelCam &lc, int level, int np, int fused, int parts, int slots,
                                 int nblk, unsigned tag0, unsigned long long limit, unsigned long long *mail, int *err,
                                 int dbg_mute, int prio_mode, const ResXchg *xchg, hipStream_t s) {
  ResArgs a;
  if (xchg)
    a.x = *xchg;
  else
    memset(&a.x, 0, sizeof(a.x));
  a.prio_mode = prio_mode;
  {
    const char *ps = getenv("ICTR_RESIDENT_STAGGER");
    a.stagger = ps ? atoi(ps) : 0;
  }
  {
    const char *ps = getenv("ICTR_RES_PROF_SLOT");
    a.prof_slot = ps ? atoi(ps) : 0;
  }
  a.nblk = nblk;
  a.lc = lc;
  a.level = level;
  a.parts = parts;
  a.slots = slots;
  a.dbg_mute = dbg_mute;
  a.tag0 = tag0;
  a.limit = limit;
  a.mail = mail;
  a.err = err;
  const dim3 grid((parts + 1) * slots), blk(kResThreads);
  if (np == 32 && fused)
    hipLaunchKernelGGL((k_level_resident<32, true>), grid, blk, 0, s, e, a);
  else if (np == 32)
    hipLaunchKernelGGL((k_level_resident<32, false>), grid, blk, 0, s, e, a);
  else if (fused)
    hipLaunchKernelGGL((k_level_resident<16, true>), grid, blk, 0, s, e, a);
  else
    hipLaunchKernelGGL((k_level_resident<16, false>), grid, blk, 0, s, e, a);
  return hipGetLastError();
}

}  // namespace ictr
