// L1 (TCP) throughput of per-lane loads on gfx950: cycles per wave64 load instruction and CU for 4 / 8 / 12 / 16-byte loads when the
// lanes of a wavefront read (a) 64 consecutive elements, (b) one element each from 64 different 128-byte lines of an L1-resident
// table, (c) all the same element.  8 waves per SIMD, 8 independent loads in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 l1_rate.hip -o l1_rate && ./l1_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ __launch_bounds__(256) void k_load(const T* __restrict__ tab, int mode, int nelem_mask, unsigned* out) {
  const int lane = threadIdx.x & 63;
  unsigned idx;
  if (mode == 0) idx = lane;                          // consecutive elements
  else if (mode == 1) idx = lane * (128 / sizeof(T)); // one element per 128-byte line
  else idx = 0;                                       // all lanes the same element
  unsigned acc = 0;
  unsigned step = (mode == 1 ? 7u : 1u) * 64u;
  for (int i = 0; i < 2048; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const T v = tab[(idx + u * step) & nelem_mask];
      acc += ((const unsigned*)&v)[0];
    }
    idx += 8 * step + 1;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
struct f3 { float x, y, z; };
template <typename T>
void run(const char* name, int cus, int clk_khz) {
  const int bytes = 16384;  // L1-resident table (32 KB L1 per CU)
  const int nelem = bytes / 16 * 16 / (int)sizeof(T);
  int pow2 = 1;
  while (pow2 * 2 <= nelem) pow2 *= 2;
  T* tab;
  unsigned* out;
  hipMalloc(&tab, bytes + 64);
  hipMemset(tab, 1, bytes + 64);
  const int blocks = cus * 8;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const char* modes[3] = {"consecutive", "64 lines", "same element"};
  for (int m = 0; m < 3; ++m) {
    hipLaunchKernelGGL(k_load<T>, dim3(blocks), dim3(256), 0, 0, tab, m, pow2 - 1, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_load<T>, dim3(blocks), dim3(256), 0, 0, tab, m, pow2 - 1, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double loads_per_cu = 32.0 * 2048 * 8;  // 32 waves per CU x 2048 x 8 wave-level load instructions
    std::printf("%-10s %-14s %8.3f ms  %6.1f cycles per wave load and CU\n", name, modes[m], best, best * 1e-3 * clk_khz * 1e3 / loads_per_cu);
  }
  hipFree(tab), hipFree(out);
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  std::printf("# %d CUs, nominal %d MHz; table 16 KB (L1-resident), 32 waves per CU, 8 loads in flight per lane\n", p.multiProcessorCount, clk_khz / 1000);
  run<unsigned>("dword", p.multiProcessorCount, clk_khz);
  run<uint2>("dwordx2", p.multiProcessorCount, clk_khz);
  run<f3>("dwordx3", p.multiProcessorCount, clk_khz);
  run<uint4>("dwordx4", p.multiProcessorCount, clk_khz);
  return 0;
}
