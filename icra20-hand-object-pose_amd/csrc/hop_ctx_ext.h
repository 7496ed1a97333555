// hop_ctx_ext.h -- internal glue between hop_ctx.hip (owner of hop_ctx) and the row modules compiled as separate
// translation units (hop_physics.hip): device buffers, the stream, the error string, the resident hypothesis set and
// one owned extension object per module.  Not part of the C-ABI.
#ifndef HOP_CTX_EXT_H_
#define HOP_CTX_EXT_H_

#include <hip/hip_runtime.h>

#include <string>

#include "../../include/hop.h"

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  template <class T>
  T* as() const { return static_cast<T*>(p); }
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      e = hipMalloc(&p, bytes);
      want = bytes;
    }
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct PinnedBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

struct HopExt {
  virtual ~HopExt() {}
};
constexpr int HOP_EXT_SLOTS = 4;
constexpr int HOP_EXT_PHYSICS = 0;
constexpr int HOP_EXT_NORMALS = 1;
constexpr int HOP_EXT_RENDER = 2;

struct HopHypView {
  float* pose;   // n x 16 row-major
  float* score;  // n
  int* id;       // n
  int n;
};

hipStream_t hop_ctx_stream(hop_ctx* c);
int hop_ctx_device(const hop_ctx* c);
void hop_ctx_set_error(hop_ctx* c, const std::string& msg);
HopExt*& hop_ctx_ext(hop_ctx* c, int slot);
HopHypView hop_ctx_hyp(hop_ctx* c);
// Large transfers between the caller's (pageable) arrays and the device go through pinned staging buffers of the context: the runtime's
// own path for pageable copies of a megabyte and more pins and unpins the caller's pages per call -- 15 ms for the 2 x 3.7 MB of an
// organised 640 x 480 cloud, against 0.6 ms staged (measured from host/app/run_real_all: HOP_APP_TIMING).  Below 256 KB both are plain
// hipMemcpyAsync calls.  hop_ctx_h2d is asynchronous on the context's stream (the staging area is recycled at the context's
// synchronisation points); hop_ctx_d2h returns with the data in `dst` (it synchronises the stream when it stages).
hipError_t hop_ctx_h2d(hop_ctx* c, void* dst_device, const void* src_host, size_t bytes);
hipError_t hop_ctx_d2h(hop_ctx* c, void* dst_host, const void* src_device, size_t bytes);
void hop_ctx_hyp_set_count(hop_ctx* c, int n);

// hop_normals.hip: integral-image normals of an organised device cloud (planes of H*W floats each)
int hop_normals_ii_device(hop_ctx* c, const float* x, const float* y, const float* z, int H, int W, float max_depth_change_factor, float smoothing_size,
                          int depth_dependent, float* nx, float* ny, float* nz);

#endif  // HOP_CTX_EXT_H_
