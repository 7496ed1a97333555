// TEST INFRASTRUCTURE (tests/test_emu_kernels_cpu.py): does the CPU model of tests/emu SEE the reuse of a pinned staging buffer before its stream
// was synchronised?  ("streams are synchronous and in order: the staging ring ... is invisible to it".)  Prints three words:
//   staged_then_synced   ok        -- write slot, copy, synchronise, rewrite slot: the device holds the first value
//   reused_before_sync   corrupted -- write slot, copy, rewrite slot, THEN synchronise: the device holds the second value (a device may
//                                     perform the copy that late; round 4's synchronous model answered "ok" here)
//   pageable_source      ok        -- the same reuse with pageable host memory is legal (the runtime stages it inside the call)
#include <cstdio>
#include <vector>

#include "hip/hip_runtime.h"

__global__ void k_read(const int* p, int* out) { *out = *p; }

int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  int *dev = nullptr, *out = nullptr, *pin = nullptr;
  hipMalloc(&dev, 256), hipMalloc(&out, 256), hipHostMalloc(&pin, 256);
  auto device_value = [&]() {
    hipLaunchKernelGGL(k_read, dim3(1), dim3(1), 0, s, dev, out);
    int v = 0;
    hipMemcpy(&v, out, sizeof v, hipMemcpyDeviceToHost);
    return v;
  };
  *pin = 1;
  hipMemcpyAsync(dev, pin, sizeof(int), hipMemcpyHostToDevice, s);
  hipStreamSynchronize(s);
  *pin = 2;
  std::printf("staged_then_synced %s\n", device_value() == 1 ? "ok" : "corrupted");
  *pin = 1;
  hipMemcpyAsync(dev, pin, sizeof(int), hipMemcpyHostToDevice, s);
  *pin = 2;
  hipStreamSynchronize(s);
  std::printf("reused_before_sync %s\n", device_value() == 1 ? "ok" : "corrupted");
  std::vector<int> pageable(1, 1);
  hipMemcpyAsync(dev, pageable.data(), sizeof(int), hipMemcpyHostToDevice, s);
  pageable[0] = 2;
  hipStreamSynchronize(s);
  std::printf("pageable_source %s\n", device_value() == 1 ? "ok" : "corrupted");
  return 0;
}
