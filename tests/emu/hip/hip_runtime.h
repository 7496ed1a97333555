// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE.  A functional model of the HIP execution model on the CPU, just large enough to compile
// the PRODUCT's kernel sources (icra20-hand-object-pose_amd/csrc/*.hip, unchanged) with g++ and run them: 64-lane wavefronts in lockstep at every
// cross-lane operation, workgroups with __syncthreads and LDS, grids, the handful of runtime calls the host side makes.
//
// Why it exists: GPU access to this repository was closed from outside the build for all of rounds 3 (end) and 4.  Kernels written in that time
// could not run where they were written; this model lets the SAME kernel text be executed and compared with the oracle (tests/test_emu_*.py).
// It checks what the source computes under HIP's semantics -- indices, masks, queues, reductions, operation order -- not timing, not the
// compiler's code generation for gfx950, not hardware approximations (rsq / sqrt estimates are modelled as correctly rounded).
//
// It is NOT a backend: libhop.so is never built from it, api.lib() never loads it, nothing outside tests/ refers to it.  The product still
// fails loudly without a GPU.
//
// Model:
//   * a launch runs its workgroups one after the other on the calling thread; each work-item is a fiber (ucontext) with its own stack;
//   * a work-item runs until it reaches a collective (__syncthreads, __ballot, __shfl*, wave barrier, readfirstlane) or returns;
//   * when every live lane of a wavefront waits, the lanes waiting at the same call site form the active set of that operation (the lowest
//     site first when there are several: lanes that left a loop early wait further down the text), its result is computed and they continue;
//   * __syncthreads releases when every live work-item of the workgroup waits in one;
//   * atomics, global and "device" memory are plain memory of the process; streams are in order and synchronous; __shared__ is `static`.
#ifndef HOP_EMU_HIP_RUNTIME_H_
#define HOP_EMU_HIP_RUNTIME_H_

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>

#ifndef HOP_EMU
#define HOP_EMU 1
#endif
#ifndef __HIPCC__
#define __HIPCC__ 1  // the product headers select their __host__ __device__ forms on it
#endif

// ---- qualifiers ------------------------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static
#define amdgpu_waves_per_eu(...) unused  // (inside __attribute__((...)): an occupancy hint has no meaning here)
#define amdgpu_flat_work_group_size(...) unused

// ---- index types ------------------------------------------------------------------------------------------------------------------------
struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 {
  unsigned x, y, z;
};
extern emu_uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
constexpr int warpSize = 64;

// ---- vector types ------------------------------------------------------------------------------------------------------------------------
#define EMU_VEC2(T, N, A) \
  struct alignas(A) N {   \
    T x, y;               \
  };                      \
  inline N make_##N(T x, T y) { return N{x, y}; }
#define EMU_VEC3(T, N)  \
  struct N {            \
    T x, y, z;          \
  };                    \
  inline N make_##N(T x, T y, T z) { return N{x, y, z}; }
#define EMU_VEC4(T, N, A) \
  struct alignas(A) N {   \
    T x, y, z, w;         \
  };                      \
  inline N make_##N(T x, T y, T z, T w) { return N{x, y, z, w}; }
EMU_VEC2(float, float2, 8)
EMU_VEC3(float, float3)
EMU_VEC4(float, float4, 16)
EMU_VEC2(int, int2, 8)
EMU_VEC3(int, int3)
EMU_VEC4(int, int4, 16)
EMU_VEC2(unsigned, uint2, 8)
EMU_VEC3(unsigned, uint3)
EMU_VEC4(unsigned, uint4, 16)
EMU_VEC2(double, double2, 16)
EMU_VEC2(short, short2, 4)
EMU_VEC2(unsigned short, ushort2, 4)
EMU_VEC4(unsigned char, uchar4, 4)

// ---- the scheduler's side of the collectives (tests/emu/emu_runtime.cpp) ------------------------------------------------------------------
namespace emu {
enum Kind { K_NONE = 0, K_SYNC, K_BALLOT, K_SHFL, K_WAVE_BARRIER, K_FIRSTLANE, K_MFMA_I8_16X16X64 };
// blocks the calling work-item until the operation has been resolved; payload in / result out through 8 bytes
unsigned long long collective(Kind kind, int site, unsigned long long payload, int arg, int arg2);
int site_of(const char* file, int line);  // call sites ordered by (file registration order, line)
void launch(dim3 grid, dim3 block, const std::function<void()>& body);

template <class T>
inline unsigned long long to_bits(T v) {
  static_assert(sizeof(T) <= 8, "shuffles move at most 8 bytes");
  unsigned long long b = 0;
  std::memcpy(&b, &v, sizeof(T));
  return b;
}
template <class T>
inline T from_bits(unsigned long long b) {
  T v;
  std::memcpy(&v, &b, sizeof(T));
  return v;
}
// mode 0: absolute source lane, 1: xor mask, 2: lane + delta (down), 3: lane - delta (up); width as in HIP
template <class T>
inline T shfl(int site, T v, int a, int mode, int width = 64) {
  return from_bits<T>(collective(K_SHFL, site, to_bits(v), a, mode | (width << 8)));
}
}  // namespace emu

#define EMU_SITE() ([]() { static const int s_ = emu::site_of(__FILE__, __LINE__); return s_; }())
#define __syncthreads() ((void)emu::collective(emu::K_SYNC, EMU_SITE(), 0, 0, 0))
#define __ballot(...) (emu::collective(emu::K_BALLOT, EMU_SITE(), (__VA_ARGS__) ? 1ull : 0ull, 0, 0))  // (variadic: template argument lists carry commas)
#define __any(...) (__ballot(__VA_ARGS__) != 0ull)
#define __shfl(v, ...) emu::shfl(EMU_SITE(), (v), emu_arg0(__VA_ARGS__), 0, emu_arg1_or_64(__VA_ARGS__))
#define __shfl_xor(v, ...) emu::shfl(EMU_SITE(), (v), emu_arg0(__VA_ARGS__), 1, emu_arg1_or_64(__VA_ARGS__))
#define __shfl_down(v, ...) emu::shfl(EMU_SITE(), (v), emu_arg0(__VA_ARGS__), 2, emu_arg1_or_64(__VA_ARGS__))
#define __shfl_up(v, ...) emu::shfl(EMU_SITE(), (v), emu_arg0(__VA_ARGS__), 3, emu_arg1_or_64(__VA_ARGS__))
inline int emu_arg0(int a) { return a; }
inline int emu_arg0(int a, int) { return a; }
inline int emu_arg1_or_64(int) { return 64; }
inline int emu_arg1_or_64(int, int w) { return w; }
#define __builtin_amdgcn_wave_barrier() ((void)emu::collective(emu::K_WAVE_BARRIER, EMU_SITE(), 0, 0, 0))
#define __builtin_amdgcn_fence(...) ((void)0)
template <class T>
inline T emu_readfirstlane(int site, T v) {
  return emu::from_bits<T>(emu::collective(emu::K_FIRSTLANE, site, emu::to_bits(v), 0, 0));
}
#define __builtin_amdgcn_readfirstlane(x) emu_readfirstlane(EMU_SITE(), (x))

// ---- scalar builtins ------------------------------------------------------------------------------------------------------------------------
inline float emu_fmed3(float a, float b, float c) { return std::fmax(std::fmin(a, b), std::fmin(std::fmax(a, b), c)); }
#define __builtin_amdgcn_fmed3f(a, b, c) emu_fmed3((a), (b), (c))
// ---- gfx950 instructions the product names (builtins, and inline asm through chevrons.py -> emu_asm::) ----------------------------------------
// Functional models from the instruction descriptions of the CDNA ISA; tests/test_gpu_zzzz_dev_selftest.py runs the same inputs through the real
// instructions on a device (hop_dev_selftest) and compares with these semantics restated in numpy.
typedef short emu_v2i16 __attribute__((vector_size(4)));
typedef int emu_v4i32 __attribute__((vector_size(16)));
inline emu_v2i16 emu_cvt_pk_i16(int a, int b) {  // v_cvt_pk_i16_i32: each source saturated to 16 bits
  auto sat = [](int v) { return (short)std::min(32767, std::max(-32768, v)); };
  return emu_v2i16{sat(a), sat(b)};
}
#define __builtin_amdgcn_cvt_pk_i16(a, b) emu_cvt_pk_i16((a), (b))
inline int emu_sdot2(emu_v2i16 a, emu_v2i16 b, int c, bool clamp) {  // v_dot2_i32_i16: c + a.x b.x + a.y b.y (wrapping unless clamp)
  const long long r = (long long)c + (long long)a[0] * b[0] + (long long)a[1] * b[1];
  if (clamp) return (int)std::min<long long>(2147483647ll, std::max<long long>(-2147483648ll, r));
  return (int)(unsigned)(unsigned long long)r;
}
#define __builtin_amdgcn_sdot2(a, b, c, clamp) emu_sdot2((a), (b), (c), (clamp))
// v_perm_b32 D, S0, S1, S2: byte k of D = byte sel_k of the pool {S0 (bytes 7..4), S1 (bytes 3..0)}; selectors 8..11 replicate a sign bit
// (of S1's halves / S0's halves ...), 12 = 0x00, >= 13 = 0xFF.  The product uses selectors 0..7 only; the others abort here.
inline unsigned emu_perm(unsigned s0, unsigned s1, unsigned sel) {
  const unsigned long long pool = ((unsigned long long)s0 << 32) | s1;
  unsigned d = 0;
  for (int k = 0; k < 4; ++k) {
    const unsigned c = (sel >> (8 * k)) & 0xFFu;
    unsigned b;
    if (c <= 7) b = (unsigned)(pool >> (8 * c)) & 0xFFu;
    else if (c == 12) b = 0;
    else if (c >= 13) b = 0xFFu;
    else {
      std::fprintf(stderr, "emu: v_perm_b32 selector %u is not modelled\n", c);
      std::abort();
    }
    d |= b << (8 * k);
  }
  return d;
}
#define __builtin_amdgcn_perm(s0, s1, sel) emu_perm((s0), (s1), (sel))
namespace emu_asm {
inline void v_med3_u32(unsigned& d, unsigned a, unsigned b, unsigned c) { d = std::max(std::min(a, b), std::min(std::max(a, b), c)); }
inline int sext24(int a) { return (int)((unsigned)a << 8) >> 8; }
template <class D, class A, class B, class Cc>
inline void v_mad_i32_i24(D& d, A a, B b, Cc c) {  // D = sext24(a) * sext24(b) + c, low 32 bits
  d = (D)((unsigned)sext24((int)a) * (unsigned)sext24((int)b) + (unsigned)c);
}
template <class D, class A, class B, class Cc>
inline void v_mad_i32_i16(D& d, A a, B b, Cc c) {  // D = sext16(a) * sext16(b) + c (low halves of the sources), low 32 bits
  d = (D)((unsigned)((int)(short)(unsigned short)(unsigned)a * (int)(short)(unsigned short)(unsigned)b) + (unsigned)c);
}
}  // namespace emu_asm
// v_mfma_i32_16x16x64_i8: D[16][16] = C + A[16][64] B[64][16], 8-bit signed inputs, 32-bit wrapping sums.  Operand layout (CDNA3/4 MFMA
// register maps; K = 64 laid out as four blocks of 16 consecutive k, one block per 16-lane group, 16 bytes = 4 VGPRs per lane):
//   A[i][k]: lane (i + 16 (k / 16)), byte k % 16        B[k][j]: lane (j + 16 (k / 16)), byte k % 16
//   C / D[i][j]: lane (j + 16 (i / 4)), register i % 4
// The instruction ignores EXEC; the model insists that all 64 lanes arrive together (a kernel that issues it under divergence is wrong).
struct emu_mfma_regs {
  emu_v4i32 a, b, c, d;
};
inline emu_v4i32 emu_mfma_i32_16x16x64_i8(int site, emu_v4i32 a, emu_v4i32 b, emu_v4i32 c) {
  emu_mfma_regs r{a, b, c, c};
  (void)emu::collective(emu::K_MFMA_I8_16X16X64, site, (unsigned long long)(uintptr_t)&r, 0, 0);
  return r.d;
}
#define __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, cbsz, abid, blgp) emu_mfma_i32_16x16x64_i8(EMU_SITE(), (a), (b), (c))

// (only the -DHOP_ICP_COUNT statistics build names these: "the first active lane" is modelled as lane 0 of the wavefront)
#define __builtin_amdgcn_read_exec_lo() 0u
#define __builtin_amdgcn_read_exec_hi() 0u
#define __builtin_amdgcn_mbcnt_lo(mask, v) ((unsigned)(threadIdx.x & 63u))
#define __builtin_amdgcn_mbcnt_hi(mask, v) (v)
// The hardware's square-root / reciprocal instructions are ESTIMATES (v_sqrt_f32 / v_rsq_f32: 1 ulp; v_rsq_f64 / v_rcp_f64: about single
// precision).  Modelled as correctly rounded by default; EMU_ULP=<k> perturbs every result -- f32 by an integer in [-k, k] units in the last
// place, f64 by up to k 2^-24 relative, both a fixed function of the operand's bits -- so that a kernel which silently relies on a correctly
// rounded estimate changes its answers (the kernels that use estimates re-decide near-ties with exact operations, or refine them).
inline int emu_ulp_k() {
  static const int k = std::getenv("EMU_ULP") ? std::atoi(std::getenv("EMU_ULP")) : 0;
  return k;
}
inline unsigned emu_mix(unsigned long long v) {
  v ^= v >> 33, v *= 0xff51afd7ed558ccdull, v ^= v >> 33, v *= 0xc4ceb9fe1a85ec53ull, v ^= v >> 33;
  return (unsigned)v;
}
inline float emu_est_f32(float exact, float operand) {
  const int k = emu_ulp_k();
  if (k == 0 || !std::isfinite(exact) || exact == 0.f) return exact;
  unsigned ob, eb;
  std::memcpy(&ob, &operand, 4), std::memcpy(&eb, &exact, 4);
  eb += (unsigned)((int)(emu_mix(ob) % (unsigned)(2 * k + 1)) - k);
  float r;
  std::memcpy(&r, &eb, 4);
  return std::isfinite(r) ? r : exact;
}
inline double emu_est_f64(double exact, double operand) {
  const int k = emu_ulp_k();
  if (k == 0 || !std::isfinite(exact) || exact == 0.0) return exact;
  unsigned long long ob;
  std::memcpy(&ob, &operand, 8);
  const double u = (double)(emu_mix(ob) % 2001u) / 1000.0 - 1.0;  // [-1, 1]
  return exact * (1.0 + u * (double)k * 5.9604644775390625e-08);
}
#define __builtin_amdgcn_sqrtf(x) emu_est_f32(std::sqrt((float)(x)), (float)(x))
#define __builtin_amdgcn_rsqf(x) emu_est_f32(1.0f / std::sqrt((float)(x)), (float)(x))
#define __builtin_amdgcn_rsq(x) emu_est_f64(1.0 / std::sqrt((double)(x)), (double)(x))
#define __builtin_amdgcn_rcp(x) emu_est_f64(1.0 / (double)(x), (double)(x))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
inline int __float_as_int(float f) {
  int i;
  std::memcpy(&i, &f, 4);
  return i;
}
inline unsigned __float_as_uint(float f) {
  unsigned i;
  std::memcpy(&i, &f, 4);
  return i;
}
inline float __int_as_float(int i) {
  float f;
  std::memcpy(&f, &i, 4);
  return f;
}
inline float __uint_as_float(unsigned i) {
  float f;
  std::memcpy(&f, &i, 4);
  return f;
}
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
inline int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll((unsigned long long)v); }
inline int emu_sext24(int a) { return (int)((unsigned)a << 8) >> 8; }
inline int __mul24(int a, int b) { return (int)((unsigned)emu_sext24(a) * (unsigned)emu_sext24(b)); }
inline float __fdividef(float a, float b) { return a / b; }
inline float __frcp_rn(float a) { return 1.0f / a; }
inline float __fsqrt_rn(float a) { return std::sqrt(a); }
inline float rsqrtf(float a) { return 1.0f / std::sqrt(a); }
using std::max;
using std::min;

// ---- atomics (work-items run one at a time) ------------------------------------------------------------------------------------------------
template <class T, class U>
inline T atomicAdd(T* p, U v) {
  const T o = *p;
  *p = (T)(o + (T)v);
  return o;
}
template <class T, class U>
inline T atomicSub(T* p, U v) {
  const T o = *p;
  *p = (T)(o - (T)v);
  return o;
}
template <class T, class U>
inline T atomicMin(T* p, U v) {
  const T o = *p;
  if ((T)v < o) *p = (T)v;
  return o;
}
template <class T, class U>
inline T atomicMax(T* p, U v) {
  const T o = *p;
  if ((T)v > o) *p = (T)v;
  return o;
}
template <class T, class U>
inline T atomicOr(T* p, U v) {
  const T o = *p;
  *p = (T)(o | (T)v);
  return o;
}
template <class T, class U>
inline T atomicAnd(T* p, U v) {
  const T o = *p;
  *p = (T)(o & (T)v);
  return o;
}
template <class T, class U>
inline T atomicExch(T* p, U v) {
  const T o = *p;
  *p = (T)v;
  return o;
}
template <class T, class U, class V>
inline T atomicCAS(T* p, U cmp, V v) {
  const T o = *p;
  if (o == (T)cmp) *p = (T)v;
  return o;
}
inline void __threadfence() {}
inline void __threadfence_block() {}

// ---- the runtime calls the host side makes ------------------------------------------------------------------------------------------------
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101;
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
typedef void* hipDeviceptr_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
constexpr unsigned hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipHostRegisterDefault = 0;
const char* hipGetErrorString(hipError_t e);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
hipError_t emu_malloc(void** p, size_t bytes, int kind);  // kind 0: device memory, 1: pinned host memory
template <class T>
inline hipError_t hipMalloc(T** p, size_t bytes) {
  return emu_malloc(reinterpret_cast<void**>(p), bytes, 0);
}
template <class T>
inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned = 0) {
  return emu_malloc(reinterpret_cast<void**>(p), bytes, 1);
}
hipError_t hipFree(void* p);
hipError_t hipHostFree(void* p);
hipError_t hipHostRegister(void* p, size_t bytes, unsigned flags);
hipError_t hipHostUnregister(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s = nullptr);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s = nullptr);
hipError_t hipMemset(void* dst, int value, size_t bytes);
hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int value, size_t count, hipStream_t s = nullptr);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
#define HIP_SYMBOL(x) (&(x))
inline hipError_t hipMemcpyToSymbol(void* sym, const void* src, size_t bytes, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice) {
  std::memcpy(static_cast<char*>(sym) + off, src, bytes);
  return hipSuccess;
}
inline hipError_t hipMemcpyFromSymbol(void* dst, const void* sym, size_t bytes, size_t off = 0, hipMemcpyKind = hipMemcpyDeviceToHost) {
  std::memcpy(dst, static_cast<const char*>(sym) + off, bytes);
  return hipSuccess;
}
// Streams are in order, but NOT synchronous: a hipMemcpyAsync is queued and performed as LATE as the stream order allows (before the next
// kernel / memset / event / synchronisation of its stream, or any device-wide synchronisation) -- the worst schedule a device may legally
// choose.  A caller that reuses a staging buffer before its stream was synchronised therefore corrupts its own data here, visibly.
// Only copies between device memory and PINNED host memory (hipHostMalloc / hipHostRegister) are queued: with pageable host memory the runtime
// stages the data before the call returns (a caller may reuse such a buffer at once), and the model copies at the call, after the stream's
// queued copies.  EMU_ASYNC=0: every copy happens at the call (round 4's model).
extern "C" void emu_stream_flush(hipStream_t s);  // performs the queued copies of s (nullptr: of every stream); also used by tests/emu/fake_rccl.cpp
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  (emu_stream_flush(stream), emu::launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); }))

#endif  // HOP_EMU_HIP_RUNTIME_H_
