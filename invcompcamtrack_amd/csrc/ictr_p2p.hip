// ictr_p2p.hip -- one-shot peer-to-peer all-reduce of the per-problem 27-float records.
//
// The sharded Gauss-Newton loop exchanges B x 27 floats per iteration (H 21 + b 6 per problem). For a message of a
// few KB a ring / tree collective is pure latency: 2 (N - 1) hops over point-to-point xGMI links. Every GPU of a
// node has a direct link to every other one, so the latency-optimal exchange is ONE hop: each rank stores its
// record straight into a mailbox slot in every peer's memory (mapped with hipIpc), then adds up the N slots of its
// own mailbox in rank order. No communicator, no proxy thread, no second stream: one small kernel enqueued on the
// compute stream between the tail and the finish kernels.
//
// Protocol (no fences, no flags): a value travels as one naturally aligned 8-byte granule {float bits, tag} written by
// ONE system-scope store; the tag is the exchange's sequence number, so a reader polls the granule itself until the
// tag matches -- a granule is either old or complete (MI355X_MICROARCH.md, "Valid forms", R2). Mailboxes are
// double-buffered by the parity of the sequence number: a peer can be at most one exchange ahead (it needs this
// rank's granules of exchange k+1 before it can finish k+1 and start k+2), so slot k & 1 is never overwritten while
// it is still being read. Every rank adds the slots in the same order (rank 0, 1, ...): identical bits everywhere, the
// redundant solves stay in lockstep. Polling is bounded by a wall-clock limit; a timeout raises a sticky error flag
// instead of hanging the GPU.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>

#include "ictr_dev.h"

extern int ictr_fail_(int code, const char *fmt, ...);

namespace ictr {

constexpr int kP2PMaxWorld = 16;

struct P2PArgs {
  uint64_t *peer[kP2PMaxWorld];  // every rank's mailbox as mapped into this process (own rank: the local pointer)
  uint64_t *local;
  int rank, world;
  int64_t cap;                // granules per slot
  int *err;                   // sticky device flag: an exchange timed out
  unsigned long long limit;   // polling limit in wall_clock64 ticks (100 MHz)
};

__global__ __launch_bounds__(256) void k_p2p_allreduce(P2PArgs a, float *buf, int count, unsigned seq) {
  const size_t par = (size_t)(seq & 1u) * a.world;
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    const uint64_t g = ((uint64_t)seq << 32) | (uint64_t)__builtin_bit_cast(unsigned, buf[i]);
    for (int r = 0; r < a.world; ++r)  // my record into slot [rank] of every mailbox, mine included
      __hip_atomic_store(a.peer[r] + (par + a.rank) * a.cap + i, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const unsigned long long t0 = wall_clock64();
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    float sum = 0.0f;
    for (int r = 0; r < a.world; ++r) {
      const uint64_t *src = a.local + (par + r) * a.cap + i;
      uint64_t g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      while ((unsigned)(g >> 32) != seq) {
        if (wall_clock64() - t0 > a.limit) {  // a peer never arrived: give up, flag it, leave the kernel
          atomicExch(a.err, 1);
          g = (uint64_t)seq << 32;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
        g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      sum += __builtin_bit_cast(float, (unsigned)(g & 0xffffffffu));  // rank order: the same bits on every rank
    }
    buf[i] = sum;
  }
}

}  // namespace ictr

using namespace ictr;

struct ictr_p2p {
  int rank = 0, world = 1;
  int64_t cap = 0;
  uint64_t *mail = nullptr;
  uint64_t *peer[kP2PMaxWorld] = {};
  bool opened[kP2PMaxWorld] = {};
  int *d_err = nullptr;
  unsigned seq = 0;
  bool connected = false;
  double timeout_s = 2.0;
};

#define P2PCHK(expr)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess) return ictr_fail_(ICTR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

extern "C" int ictr_p2p_create(ictr_p2p **out, int rank, int world, int64_t count) {
  if (!out || world < 1 || world > kP2PMaxWorld || rank < 0 || rank >= world || count < 1 || count > (1 << 24))
    return ictr_fail_(ICTR_ERR_INVALID, "p2p_create: bad arguments (world 1..%d)", kP2PMaxWorld);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return ictr_fail_(ICTR_ERR_NO_DEVICE, "no usable HIP device: the tracker has no CPU fallback");
  ictr_p2p *p = new ictr_p2p;
  p->rank = rank;
  p->world = world;
  p->cap = (count + 31) / 32 * 32;
  const size_t bytes = sizeof(uint64_t) * 2 * (size_t)world * p->cap;
  // mailbox memory that remote stores and local polls see coherently: uncached (fine-grained) device memory
  hipError_t e = hipExtMallocWithFlags((void **)&p->mail, bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) e = hipExtMallocWithFlags((void **)&p->mail, bytes, hipDeviceMallocFinegrained);
  if (e == hipSuccess) e = hipMemset(p->mail, 0, bytes);  // tag 0 = "nothing yet"; sequence numbers start at 1
  if (e == hipSuccess) e = hipMalloc((void **)&p->d_err, sizeof(int));
  if (e == hipSuccess) e = hipMemset(p->d_err, 0, sizeof(int));
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    if (p->mail) (void)hipFree(p->mail);
    if (p->d_err) (void)hipFree(p->d_err);
    delete p;
    return ictr_fail_(ICTR_ERR_HIP, "p2p_create: mailbox allocation failed: %s", hipGetErrorString(e));
  }
  if (const char *t = getenv("ICTR_P2P_TIMEOUT_S")) p->timeout_s = std::max(0.01, atof(t));
  p->peer[rank] = p->mail;
  *out = p;
  return ICTR_OK;
}

extern "C" int ictr_p2p_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

// the handle other processes open to reach this rank's mailbox (hipIpcMemHandle_t, ictr_p2p_handle_bytes() bytes)
extern "C" int ictr_p2p_local_handle(ictr_p2p *p, void *handle_out) {
  if (!p || !handle_out) return ictr_fail_(ICTR_ERR_INVALID, "p2p_local_handle: NULL argument");
  hipIpcMemHandle_t h;
  P2PCHK(hipIpcGetMemHandle(&h, p->mail));
  memcpy(handle_out, &h, sizeof(h));
  return ICTR_OK;
}

// all_handles: world handles in rank order (this rank's own entry is ignored)
extern "C" int ictr_p2p_connect(ictr_p2p *p, const void *all_handles) {
  if (!p || !all_handles) return ictr_fail_(ICTR_ERR_INVALID, "p2p_connect: NULL argument");
  for (int r = 0; r < p->world; ++r) {
    if (r == p->rank || p->opened[r]) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, (const char *)all_handles + (size_t)r * sizeof(h), sizeof(h));
    void *ptr = nullptr;
    P2PCHK(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
    p->peer[r] = (uint64_t *)ptr;
    p->opened[r] = true;
  }
  p->connected = true;
  return ICTR_OK;
}

// in-place sum of dev_buf[0..count) over all ranks, enqueued on `hip_stream`; every rank must call it the same
// number of times with the same count
extern "C" int ictr_p2p_allreduce(ictr_p2p *p, float *dev_buf, int64_t count, void *hip_stream) {
  if (!p || !dev_buf || count < 1 || count > p->cap) return ictr_fail_(ICTR_ERR_INVALID, "p2p_allreduce: bad arguments");
  if (!p->connected && p->world > 1) return ictr_fail_(ICTR_ERR_STATE, "p2p_allreduce before p2p_connect");
  P2PArgs a;
  memset(&a, 0, sizeof(a));
  for (int r = 0; r < p->world; ++r) a.peer[r] = p->peer[r];
  a.local = p->mail;
  a.rank = p->rank;
  a.world = p->world;
  a.cap = p->cap;
  a.err = p->d_err;
  a.limit = (unsigned long long)(p->timeout_s * 1e8);  // wall_clock64 ticks at 100 MHz
  p->seq += 1;
  if (p->seq == 0) p->seq = 1;  // tag 0 is reserved for "empty"
  hipLaunchKernelGGL(k_p2p_allreduce, dim3(1), dim3(256), 0, (hipStream_t)hip_stream, a, dev_buf, (int)count, p->seq);
  P2PCHK(hipGetLastError());
  return ICTR_OK;
}

// internal (ictr_host.hip): the mailboxes as a kernel argument of the resident-iteration launch (sharded resident form)
extern "C" int ictr_p2p_fill_xchg_(const ictr_p2p *p, ictr::ResXchg *x) {
  if (!p || !x || (!p->connected && p->world > 1)) return 1;
  memset(x, 0, sizeof(*x));
  for (int r = 0; r < p->world; ++r) x->peer[r] = (unsigned long long *)p->peer[r];
  x->local = (unsigned long long *)p->mail;
  x->rank = p->rank;
  x->world = p->world;
  x->cap = p->cap;
  return 0;
}

// 0: every exchange so far completed; 1: one timed out (a peer did not arrive). Synchronises the device.
extern "C" int ictr_p2p_error(ictr_p2p *p) {
  if (!p) return 1;
  int e = 1;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpy(&e, p->d_err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  return e;
}

extern "C" void ictr_p2p_destroy(ictr_p2p *p) {
  if (!p) return;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < p->world; ++r)
    if (p->opened[r] && p->peer[r]) (void)hipIpcCloseMemHandle(p->peer[r]);
  if (p->mail) (void)hipFree(p->mail);
  if (p->d_err) (void)hipFree(p->d_err);
  delete p;
}
