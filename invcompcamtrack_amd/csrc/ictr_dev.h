// ictr_dev.h -- structures shared by the HIP kernels and the host-side ABI implementation.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ictr.h"

namespace ictr {

constexpr int kBlock = 256;        // threads per workgroup = 4 wave64
constexpr int kWaves = kBlock / 64;
constexpr int kCoefStride = 16;    // 12 steepest-descent coefficients per point, padded to one 64-B line
constexpr int kHUnique = 21;       // upper triangle of the 6x6 normal matrix
constexpr int kRedStride = 27;     // per problem: 21 (H) + 6 (b) in the reduction buffer
constexpr int kPartHStride = 32;   // per block partial of H (21 padded to 32)
constexpr int kPartBStride = 8;    // per block partial of b (6 padded)
constexpr int kMaxGridX = 2048;    // 256 CUs x 8 resident workgroups: cap, then grid-stride

// per pyramid level camera constants (camera.cpp:31-42), passed by value to the kernels
struct LevelCam {
  float fx, fy, cx, cy, swo, sho;
  int sw;  // padded row stride, (int)getsw(level) as the reference passes it (odometer.cpp:286)
};

// device pointers of the four planes one problem reads at one level
struct PlaneSet {
  const float *ref, *dx, *dy, *cur;
  const float *pack;  // optional: the reference level as one interleaved plane {img, dx, dy, 0} per pixel (k_ref8)
};

// per-problem state living in device memory for the whole coarse-to-fine loop
struct ProbState {
  float p[6];    // cpos_p  (pose.h:54)
  float G[12];   // cpos_G  (pose.h:53)
  float H[36];   // Hes     (odometer.h:62)
  float LU[36];  // full-pivot LU factors of H, computed once per level (H is constant across its iterations)
  int piv[12];   // 6 row + 6 column transpositions
  int luinfo[2]; // non-zero pivots, rank
  float b[6];    // sumsd
  float dp[6];   // delta_p
  float normdp;
  float normdp_init;
  int it;           // iteration counter of the current level
  int active;       // loop condition of odometer.cpp:344-346, evaluated on the device
  int total_iters;  // executed GN iterations, all levels
  int npts;         // nopoints of this problem
  unsigned reserved_;
  int pad_;
};

// team form of the one-launch tracker (ictr_track1.hip, "Teams"): several workgroups per problem
struct T1Team {
  int team, q;               // workgroups per problem, points per workgroup
  unsigned tag0;             // launch epoch << 12
  unsigned long long limit;  // polling limit, wall_clock64 ticks (100 MHz)
  unsigned long long *mail;  // [B][2][team][32] granules
  int *err;                  // sticky time-out flag (pinned host memory as the device sees it)
  int mute;                  // debug (variant bit 25): part `mute - 1` never posts its values (time-out test); 0 = off
};

// Cross-GPU exchange inside the resident-iteration launch (ictr_resident.hip, "sharded resident form"): the mailboxes of
// an ictr_p2p object (ictr_p2p.hip: one per rank, hipIpc-mapped into every peer) as the kernel sees them. world == 1: off.
constexpr int kXchgMaxWorld = 16;
constexpr int kXchgPerPair = 64;  // granules per frame pair and rank: [0, 12) b as (hi, lo) pairs, [12, 33) H
struct ResXchg {
  unsigned long long *peer[kXchgMaxWorld];  // every rank's mailbox as mapped into this process (own rank: the local one)
  unsigned long long *local;
  int rank, world;
  long long cap;     // granules per (parity, rank) slot of a mailbox
  unsigned *xseq;    // [B] exchanges done so far per frame pair (device memory; every rank counts alike)
};

struct DevTrace {
  ictr_trace_rec *rec;
  int *count;
  int capacity;
};

// everything a kernel needs, passed by value (kernarg segment)
struct EngineDev {
  int B, M, P, n, nlev;  // nlev = lv_f+1
  int lv_f, lv_l, maxiter;
  float ratio;
  int dopatchnorm;
  int sharded;  // 1: accumulate kernels stop after writing rank-local sums to red[]
  int packed;   // 1: every problem's reference pyramid carries the interleaved {img, dx, dy, 0} planes
  int otf;      // every reference pyramid is builder-made (its gradients ARE the central differences of its image plane):
                // 1 = the 8x8 setup kernel may form them on the fly from the image plane; 2 = it must (some reference
                // pyramid is image-only: no dx / dy / packed planes exist); 0 = caller-supplied gradient planes
  // behaviour-changing options, all off by default; any of them routes P = 8 through the
  // any-size kernels. ICTR_ROBUST_CLEAN: a point outside the reference view at a level contributes nothing (its
  // stale patches / gradients are zeroed instead of reused, cf. odometer.cpp:304); ICTR_ROBUST_COMPOSE: G <- exp(dp) G
  // instead of p += dp (pose.cpp:118-123); ICTR_ROBUST_HUBER: residuals weighted min(1, k / |r|) in J^T r.
  int robust;
  float huber_k;
  float *pt3d;      // [B][3M]  X..Y..Z..
  float *pt3d_ref;  // [B][3M]  camera-frame points at the reference pose
  float *pt2d;      // [B][nlev][2M]
  float *T, *Gx, *Gy;  // [B][M*n] patch-major, the reference's pat_ref_all / _dx_all / _dy_all
  float *coef;         // [B][M][16]
  ProbState *st;       // [B]
  const PlaneSet *planes;  // [B][nlev]
  float *partH;  // [B][gridx][32]
  float *partb;  // [B][gridx][8]
  float *red;    // [B][27]
  DevTrace trace;
};

// per-patch translation IC-LK (ictr_patchflow.hip)
struct PFLevel {
  const float *a, *ax, *ay, *b;  // frame A image + gradients, frame B image (padded planes)
  int sw;
  float swo, sho, scale;  // unpadded size, 0.5^level
};
struct PFArgs {
  PFLevel lv[16];
  int lv_f, lv_l, P, maxiter, K;
  float eps2;        // stop when |dp|^2 < eps2
  float min_det;     // relative conditioning threshold of the 2x2 system
  const float *pts;  // SoA x[K] y[K] at level 0
  float *out;        // SoA x'[K] y'[K] at level 0 (NaN when lost)
  int *status;       // 1 tracked, 0 lost
  int *iters;        // executed iterations (all levels)
};

}  // namespace ictr
