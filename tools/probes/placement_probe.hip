// Where does the dispatcher put the workgroups of a launch shaped like k_level_resident (256 threads, 38 KB of LDS:
// four per CU, 1020 workgroups all resident at once)?  hipcc --offload-arch=gfx950 -O2 placement_probe.hip -o placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned *out, int spin_ticks) {
  __shared__ float pad[38 * 256];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  if (pad[(threadIdx.x * 7) & 255] < 0) out[0] = 0;
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1020;
  unsigned *d;
  hipMalloc(&d, n * 8);
  std::vector<unsigned> h(2 * n);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d, 5000);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  }
  for (int i = 0; i < n; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7, simd = (hw >> 4) & 3, wv = hw & 0xf;
    printf("%d xcc %u se %u sh %u cu %u simd %u wave %u\n", i, xcc, se, sh, cu, simd, wv);
  }
  return 0;
}
