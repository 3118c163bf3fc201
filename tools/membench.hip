// tools/membench.hip -- micro-benchmark that sized the load shapes of k_iter8v (not part of the product library).
// Streams N MB with different per-lane widths / loads in flight / chunking, and a patch-window gather.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int W, int U>  // W floats per lane per load, U loads in flight
__global__ __launch_bounds__(256) void k_stream(const float* __restrict__ src, float* out, size_t n) {
  typedef float vec __attribute__((ext_vector_type(W)));
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * 256;
  const vec* s = reinterpret_cast<const vec*>(src);
  const size_t nv = n / W;
  float acc = 0.f;
  for (size_t i = tid; i + (U - 1) * nthreads < nv; i += U * nthreads) {
    vec v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = s[i + u * nthreads];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < W; ++k) acc += v[u][k];
  }
  if (acc == 123.456f) out[tid] = acc;
}

// each wave owns contiguous chunks of `chunk` floats (like 64 patches x 64 px), three streams
template <int W>
__global__ __launch_bounds__(256) void k_stream3_chunk(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ c, float* out, size_t n, int U) {
  typedef float vec __attribute__((ext_vector_type(W)));
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
  const size_t chunk = 64 * 64;  // floats
  float acc = 0.f;
  for (size_t ch = wave; (ch + 1) * chunk <= n; ch += nwaves) {
    const vec* pa = reinterpret_cast<const vec*>(a + ch * chunk);
    const vec* pb = reinterpret_cast<const vec*>(b + ch * chunk);
    const vec* pc = reinterpret_cast<const vec*>(c + ch * chunk);
    const int steps = chunk / (64 * W);
    for (int s0 = 0; s0 < steps; s0 += 4) {
      vec va[4], vb[4], vc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        va[u] = pa[(s0 + u) * 64 + lane];
        vb[u] = pb[(s0 + u) * 64 + lane];
        vc[u] = pc[(s0 + u) * 64 + lane];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < W; ++k) acc += va[u][k] * vb[u][k] + vc[u][k];
    }
  }
  if (acc == 123.456f) out[wave] = acc;
}

// store-only, copy, and "read 1, write 3" (the shape of k_ref8: one plane window in, three patch streams out)
template <int W, bool NT>
__global__ __launch_bounds__(256) void k_write(float* dst, size_t n, float v) {
  typedef float vec __attribute__((ext_vector_type(W)));
  vec* d = reinterpret_cast<vec*>(dst);
  const size_t nv = n / W;
  vec x;
  for (int k = 0; k < W; ++k) x[k] = v;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(x, d + i); else d[i] = x;
  }
}
template <int W, bool NT>
__global__ __launch_bounds__(256) void k_copy(const float* __restrict__ src, float* dst, size_t n) {
  typedef float vec __attribute__((ext_vector_type(W)));
  const vec* s = reinterpret_cast<const vec*>(src);
  vec* d = reinterpret_cast<vec*>(dst);
  const size_t nv = n / W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
    const vec x = NT ? __builtin_nontemporal_load(s + i) : s[i];
    if (NT) __builtin_nontemporal_store(x, d + i); else d[i] = x;
  }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_r1w3(const float* __restrict__ src, float* d0, float* d1, float* d2, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float x = src[i];
    if (NT) {
      __builtin_nontemporal_store(x, d0 + i); __builtin_nontemporal_store(x + 1.f, d1 + i); __builtin_nontemporal_store(x + 2.f, d2 + i);
    } else {
      d0[i] = x; d1[i] = x + 1.f; d2[i] = x + 2.f;
    }
  }
}

int main(int argc, char** argv) {
  const size_t n = (size_t)128 << 20;  // 128 Mi floats = 512 MB per array
  float *a, *b, *c, *out;
  CHK(hipMalloc(&a, n * 4)); CHK(hipMalloc(&b, n * 4)); CHK(hipMalloc(&c, n * 4)); CHK(hipMalloc(&out, 64 << 20));
  CHK(hipMemset(a, 0, n * 4)); CHK(hipMemset(b, 0, n * 4)); CHK(hipMemset(c, 0, n * 4));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  auto time = [&](auto launch, double bytes, const char* name) {
    launch(); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.1f us  %7.2f TB/s\n", name, ms / 5 * 1e3, bytes / (ms / 5 * 1e-3) / 1e12);
  };
  if (argc > 1) {  // write-side tests only
    float* d; CHK(hipMalloc(&d, n * 4));
    for (int g : {4096, 16384}) {
      char nm[128];
      snprintf(nm, sizeof nm, "write W=1 temporal grid %d", g); time([&] { k_write<1, false><<<g, 256>>>(a, n, 1.f); }, n * 4.0, nm);
      snprintf(nm, sizeof nm, "write W=1 NT grid %d", g);       time([&] { k_write<1, true><<<g, 256>>>(a, n, 1.f); }, n * 4.0, nm);
      snprintf(nm, sizeof nm, "write W=4 temporal grid %d", g); time([&] { k_write<4, false><<<g, 256>>>(a, n, 1.f); }, n * 4.0, nm);
      snprintf(nm, sizeof nm, "write W=4 NT grid %d", g);       time([&] { k_write<4, true><<<g, 256>>>(a, n, 1.f); }, n * 4.0, nm);
      snprintf(nm, sizeof nm, "copy W=1 temporal grid %d (r+w bytes)", g); time([&] { k_copy<1, false><<<g, 256>>>(a, b, n); }, n * 8.0, nm);
      snprintf(nm, sizeof nm, "copy W=4 NT grid %d (r+w bytes)", g);       time([&] { k_copy<4, true><<<g, 256>>>(a, b, n); }, n * 8.0, nm);
      snprintf(nm, sizeof nm, "read1 write3 temporal grid %d (r+w bytes)", g); time([&] { k_r1w3<false><<<g, 256>>>(a, b, c, d, n); }, n * 16.0, nm);
      snprintf(nm, sizeof nm, "read1 write3 NT grid %d (r+w bytes)", g);       time([&] { k_r1w3<true><<<g, 256>>>(a, b, c, d, n); }, n * 16.0, nm);
    }
    return 0;
  }
  const int grids[] = {2048, 4096, 8192};
  for (int g : grids) {
    char nm[128];
    snprintf(nm, sizeof nm, "stream W=1 U=8  grid %d", g);  time([&] { k_stream<1, 8><<<g, 256>>>(a, out, n); }, n * 4.0, nm);
    snprintf(nm, sizeof nm, "stream W=1 U=16 grid %d", g);  time([&] { k_stream<1, 16><<<g, 256>>>(a, out, n); }, n * 4.0, nm);
    snprintf(nm, sizeof nm, "stream W=2 U=8  grid %d", g);  time([&] { k_stream<2, 8><<<g, 256>>>(a, out, n); }, n * 4.0, nm);
    snprintf(nm, sizeof nm, "stream W=4 U=4  grid %d", g);  time([&] { k_stream<4, 4><<<g, 256>>>(a, out, n); }, n * 4.0, nm);
    snprintf(nm, sizeof nm, "stream W=4 U=8  grid %d", g);  time([&] { k_stream<4, 8><<<g, 256>>>(a, out, n); }, n * 4.0, nm);
    snprintf(nm, sizeof nm, "3-stream chunked W=1 grid %d", g); time([&] { k_stream3_chunk<1><<<g, 256>>>(a, b, c, out, n / 4, 4); }, 3 * n * 1.0, nm);
    snprintf(nm, sizeof nm, "3-stream chunked W=4 grid %d", g); time([&] { k_stream3_chunk<4><<<g, 256>>>(a, b, c, out, n / 4, 4); }, 3 * n * 1.0, nm);
  }
  return 0;
}
