// tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE: the scheduler and the runtime calls of the functional HIP model (hip/hip_runtime.h).
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "hip/hip_runtime.h"

emu_uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

// A context switch without the signal-mask system calls of swapcontext (a launch of a few hundred thousand small workgroups makes tens of
// millions of them): callee-saved registers on the old stack, stack pointers exchanged.  x86-64 System V.
extern "C" void emu_swap(void** save_sp, void* load_sp);
__asm__(
    ".text\n"
    ".globl emu_swap\n"
    ".type emu_swap,@function\n"
    "emu_swap:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size emu_swap,.-emu_swap\n");

namespace emu {
namespace {
constexpr size_t STACK_BYTES = 512 << 10;
enum State { ST_READY = 0, ST_BLOCKED, ST_DONE };
struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  State state = ST_DONE;
  Kind kind = K_NONE;
  int site = 0, arg = 0, arg2 = 0;
  unsigned long long payload = 0, result = 0;
  emu_uint3 tid;
};
std::vector<Fiber> g_fibers;
void* g_sched_sp = nullptr;
int g_current = -1;
const std::function<void()>* g_body = nullptr;
std::recursive_mutex g_launch_lock;  // the model runs one launch at a time (host threads of the product may launch concurrently)
std::vector<std::string> g_files;
std::mutex g_site_lock;
bool g_strict = getenv("EMU_STRICT_CONVERGENCE") != nullptr;
// EMU_ORDER=reverse | shuffle:<seed>: the order in which the workgroups of a launch run and in which the runnable work-items of a workgroup
// are resumed.  The hardware promises neither; a result that changes with it (slots handed out by atomics that are not re-ordered later, a
// read of another wavefront's LDS without a barrier) is a result that depends on scheduling.
int g_order_mode = 0;  // 0 ascending, 1 reverse, 2 shuffled
unsigned long long g_order_state = 0x9E3779B97F4A7C15ull;
struct OrderInit {
  OrderInit() {
    const char* e = getenv("EMU_ORDER");
    if (!e) return;
    if (!std::strcmp(e, "reverse")) g_order_mode = 1;
    else if (!std::strncmp(e, "shuffle", 7)) {
      g_order_mode = 2;
      if (e[7] == ':') g_order_state ^= std::strtoull(e + 8, nullptr, 10) * 0xD1342543DE82EF95ull;
    }
  }
} g_order_init;
unsigned long long next_rand() {  // xorshift64*
  g_order_state ^= g_order_state >> 12, g_order_state ^= g_order_state << 25, g_order_state ^= g_order_state >> 27;
  return g_order_state * 0x2545F4914F6CDD1Dull;
}
void make_order(std::vector<int>& o, int n) {
  o.resize(n);
  for (int i = 0; i < n; ++i) o[i] = g_order_mode == 1 ? n - 1 - i : i;
  if (g_order_mode == 2)
    for (int i = n - 1; i > 0; --i) std::swap(o[i], o[(int)(next_rand() % (unsigned long long)(i + 1))]);
}

void fiber_entry() {
  (*g_body)();
  g_fibers[g_current].state = ST_DONE;
  emu_swap(&g_fibers[g_current].sp, g_sched_sp);
  std::abort();  // (a finished work-item is never resumed)
}
void prepare_fiber(Fiber& f) {
  // stack as emu_swap expects to find it: six callee-saved registers, then the address it returns to; above that the slot a call would
  // have pushed, so that the entry function sees the alignment of an ordinary call
  uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + STACK_BYTES) & ~(uintptr_t)15;
  void** sp = reinterpret_cast<void**>(top);
  *--sp = nullptr;
  *--sp = reinterpret_cast<void*>(&fiber_entry);
  for (int i = 0; i < 6; ++i) *--sp = nullptr;
  f.sp = sp;
}
void run_fiber(int t) {
  g_current = t;
  threadIdx = g_fibers[t].tid;
  emu_swap(&g_sched_sp, g_fibers[t].sp);
  g_current = -1;
}
[[noreturn]] void die(const char* what, int nthreads) {
  std::fprintf(stderr, "emu: %s\n", what);
  std::map<std::pair<int, int>, int> hist;
  for (int t = 0; t < nthreads; ++t)
    if (g_fibers[t].state == ST_BLOCKED) hist[{(int)g_fibers[t].kind, g_fibers[t].site}]++;
  for (auto& kv : hist)
    std::fprintf(stderr, "  %d work-items wait in kind %d at %s:%d\n", kv.second, kv.first.first, g_files[kv.first.second >> 20].c_str(), kv.first.second & 0xFFFFF);
  std::abort();
}
}  // namespace

int site_of(const char* file, int line) {
  std::lock_guard<std::mutex> lk(g_site_lock);
  size_t f = 0;
  for (; f < g_files.size(); ++f)
    if (g_files[f] == file) break;
  if (f == g_files.size()) g_files.push_back(file);
  return (int)(f << 20) | line;
}

unsigned long long collective(Kind kind, int site, unsigned long long payload, int arg, int arg2) {
  if (g_current < 0) {
    std::fprintf(stderr, "emu: collective outside a kernel\n");
    std::abort();
  }
  Fiber& f = g_fibers[g_current];
  f.kind = kind, f.site = site, f.payload = payload, f.arg = arg, f.arg2 = arg2;
  f.state = ST_BLOCKED;
  emu_swap(&f.sp, g_sched_sp);
  threadIdx = f.tid;
  return f.result;
}

// resolves one wavefront's waiting lanes; true if any were released
static bool resolve_wave(int t0, int t1) {
  int site = -1;
  Kind kind = K_NONE;
  int n_sites = 0, last_site = -1;
  for (int t = t0; t < t1; ++t) {
    Fiber& f = g_fibers[t];
    if (f.state == ST_READY) return false;  // (cannot happen after a run phase)
    if (f.state != ST_BLOCKED || f.kind == K_SYNC) continue;
    if (f.site != last_site) ++n_sites, last_site = f.site;
    if (site < 0 || f.site < site) site = f.site, kind = f.kind;
  }
  if (site < 0) return false;
  if (g_strict && n_sites > 1) die("EMU_STRICT_CONVERGENCE: a wavefront waits at more than one call site", t1);
  unsigned long long mask = 0;
  int first = -1;
  for (int t = t0; t < t1; ++t) {
    Fiber& f = g_fibers[t];
    if (f.state == ST_BLOCKED && f.kind != K_SYNC && f.site == site) {
      mask |= 1ull << (t - t0);
      if (first < 0) first = t;
    }
  }
  if (kind == K_MFMA_I8_16X16X64) {
    if (t1 - t0 != 64 || mask != ~0ull) die("v_mfma issued with inactive lanes (the instruction ignores EXEC: every lane's registers take part)", t1);
    emu_mfma_regs* R[64];
    for (int l = 0; l < 64; ++l) R[l] = reinterpret_cast<emu_mfma_regs*>((uintptr_t)g_fibers[t0 + l].payload);
    auto byte_of = [](const emu_v4i32& v, int k) { return (int)(signed char)(((unsigned)v[k >> 2] >> (8 * (k & 3))) & 0xFFu); };
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        unsigned acc = (unsigned)R[j + 16 * (i / 4)]->c[i % 4];
        for (int k = 0; k < 64; ++k) acc += (unsigned)(byte_of(R[i + 16 * (k / 16)]->a, k % 16) * byte_of(R[j + 16 * (k / 16)]->b, k % 16));
        R[j + 16 * (i / 4)]->d[i % 4] = (int)acc;
      }
  }
  unsigned long long ballot = 0;
  if (kind == K_BALLOT)
    for (int t = t0; t < t1; ++t)
      if ((mask >> (t - t0)) & 1ull && g_fibers[t].payload) ballot |= 1ull << (t - t0);
  for (int t = t0; t < t1; ++t) {
    if (!((mask >> (t - t0)) & 1ull)) continue;
    Fiber& f = g_fibers[t];
    const int lane = t - t0;
    switch (kind) {
      case K_BALLOT: f.result = ballot; break;
      case K_WAVE_BARRIER: f.result = 0; break;
      case K_FIRSTLANE: f.result = g_fibers[first].payload; break;
      case K_SHFL: {
        const int mode = f.arg2 & 0xFF, width = std::max(1, f.arg2 >> 8);
        const int seg = lane / width * width;
        int src;
        if (mode == 0) src = seg + (f.arg & (width - 1));
        else if (mode == 1) src = lane ^ f.arg;
        else if (mode == 2) src = lane + f.arg;
        else src = lane - f.arg;
        const bool in_seg = src >= seg && src < seg + width && src < 64 && src >= 0;
        // a source lane outside the segment or not taking part returns the caller's own value (HIP: own value / undefined)
        f.result = (in_seg && ((mask >> src) & 1ull)) ? g_fibers[t0 + src].payload : f.payload;
        break;
      }
      default: f.result = 0;
    }
  }
  for (int t = t0; t < t1; ++t)
    if ((mask >> (t - t0)) & 1ull) g_fibers[t].state = ST_READY;
  return true;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  std::lock_guard<std::recursive_mutex> lk(g_launch_lock);
  const int nthreads = (int)(block.x * block.y * block.z);
  if (nthreads <= 0 || grid.x * grid.y * grid.z == 0) return;
  if ((int)g_fibers.size() < nthreads) g_fibers.resize(nthreads);
  for (int t = 0; t < nthreads; ++t)
    if (!g_fibers[t].stack) g_fibers[t].stack = static_cast<char*>(std::malloc(STACK_BYTES));
  const emu_uint3 save_t = threadIdx, save_b = blockIdx;
  const dim3 save_bd = blockDim, save_gd = gridDim;
  const std::function<void()>* save_body = g_body;
  g_body = &body;
  blockDim = block, gridDim = grid;
  std::vector<int> block_order, thread_order;
  make_order(block_order, (int)(grid.x * grid.y * grid.z));
  for (int bi : block_order) {
        const unsigned bx = (unsigned)bi % grid.x, by = (unsigned)bi / grid.x % grid.y, bz = (unsigned)bi / (grid.x * grid.y);
        blockIdx = emu_uint3{bx, by, bz};
        for (int t = 0; t < nthreads; ++t) {
          Fiber& f = g_fibers[t];
          f.tid = emu_uint3{(unsigned)t % block.x, (unsigned)t / block.x % block.y, (unsigned)t / (block.x * block.y)};
          f.state = ST_READY, f.kind = K_NONE;
          prepare_fiber(f);
        }
        int live = nthreads;
        while (live > 0) {
          bool progressed = false;
          make_order(thread_order, nthreads);
          for (int t : thread_order)
            if (g_fibers[t].state == ST_READY) {
              run_fiber(t);
              progressed = true;
            }
          live = 0;
          for (int t = 0; t < nthreads; ++t) live += g_fibers[t].state != ST_DONE;
          if (live == 0) break;
          bool released = false;
          for (int t0 = 0; t0 < nthreads; t0 += 64) released |= resolve_wave(t0, std::min(nthreads, t0 + 64));
          if (!released) {
            bool all_sync = true;
            for (int t = 0; t < nthreads; ++t)
              if (g_fibers[t].state == ST_BLOCKED && g_fibers[t].kind != K_SYNC) all_sync = false;
            if (!all_sync) die("deadlock: work-items wait in a wavefront operation that cannot complete", nthreads);
            for (int t = 0; t < nthreads; ++t)
              if (g_fibers[t].state == ST_BLOCKED) g_fibers[t].state = ST_READY, g_fibers[t].result = 0;
            released = true;
          }
          (void)progressed;
        }
      }
  g_body = save_body;
  threadIdx = save_t, blockIdx = save_b, blockDim = save_bd, gridDim = save_gd;
}
}  // namespace emu

// ---- runtime ---------------------------------------------------------------------------------------------------------------------------
struct emu_stream {
  int id;
};
namespace {
struct PendingCopy {
  emu_stream* s;
  void* dst;
  const void* src;
  size_t bytes;
};
std::mutex g_pending_lock;
std::vector<PendingCopy> g_pending;  // in enqueue order (all streams)
const bool g_async = !(getenv("EMU_ASYNC") && std::atoi(getenv("EMU_ASYNC")) == 0);
}  // namespace
extern "C" void emu_stream_flush(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_pending_lock);
  size_t w = 0;
  for (size_t r = 0; r < g_pending.size(); ++r) {
    PendingCopy& c = g_pending[r];
    if (s == nullptr || c.s == s || c.s == nullptr) std::memmove(c.dst, c.src, c.bytes);  // (the null stream orders with every stream)
    else g_pending[w++] = c;
  }
  g_pending.resize(w);
}
struct emu_event {
  std::chrono::steady_clock::time_point t;
};
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : e == hipErrorInvalidDevice ? "invalid device ordinal" : "invalid value"; }
hipError_t hipGetDeviceCount(int* n) {
  *n = 1;
  return hipSuccess;
}
static thread_local int g_device = 0;
hipError_t hipSetDevice(int d) {
  if (d != 0) return hipErrorInvalidDevice;
  g_device = d;
  return hipSuccess;
}
hipError_t hipGetDevice(int* d) {
  *d = g_device;
  return hipSuccess;
}
hipError_t hipDeviceSynchronize() {
  emu_stream_flush(nullptr);
  return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
namespace {
std::mutex g_range_lock;
std::map<uintptr_t, std::pair<size_t, int>> g_ranges;  // start -> (bytes, kind: 0 device, 1 pinned host)
void range_add(void* p, size_t bytes, int kind) {
  std::lock_guard<std::mutex> lk(g_range_lock);
  g_ranges[(uintptr_t)p] = {bytes, kind};
}
void range_del(void* p) {
  std::lock_guard<std::mutex> lk(g_range_lock);
  g_ranges.erase((uintptr_t)p);
}
bool is_pageable_host(const void* p) {  // neither device memory nor pinned host memory
  std::lock_guard<std::mutex> lk(g_range_lock);
  auto it = g_ranges.upper_bound((uintptr_t)p);
  if (it == g_ranges.begin()) return true;
  --it;
  return !((uintptr_t)p < it->first + it->second.first);
}
}  // namespace
hipError_t emu_malloc(void** p, size_t bytes, int kind) {
  const size_t sz = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
  *p = std::aligned_alloc(256, sz);
  if (!*p) return hipErrorOutOfMemory;
  std::memset(*p, 0xCD, std::min<size_t>(bytes, 1 << 20));  // (device memory is not zeroed: make reads of unwritten memory visible)
  range_add(*p, sz, kind);
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  emu_stream_flush(nullptr);  // (hipFree synchronises the device)
  range_del(p);
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostFree(void* p) {
  emu_stream_flush(nullptr);
  range_del(p);
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostRegister(void* p, size_t bytes, unsigned) {
  range_add(p, bytes, 1);
  return hipSuccess;
}
hipError_t hipHostUnregister(void* p) {
  emu_stream_flush(nullptr);
  range_del(p);
  return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
  emu_stream_flush(nullptr);  // (a synchronous copy on the null stream waits for the blocking streams; the model orders it after everything)
  std::memmove(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s) {
  if (!g_async || is_pageable_host(src) || is_pageable_host(dst)) {  // pageable host memory: staged / waited for inside the call
    emu_stream_flush(s);
    std::memmove(dst, src, bytes);
    return hipSuccess;
  }
  std::lock_guard<std::mutex> lk(g_pending_lock);
  if (bytes) g_pending.push_back(PendingCopy{s, dst, src, bytes});
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s) {
  emu_stream_flush(s);
  std::memset(dst, value, bytes);
  return hipSuccess;
}
hipError_t hipMemset(void* dst, int value, size_t bytes) {
  emu_stream_flush(nullptr);
  std::memset(dst, value, bytes);
  return hipSuccess;
}
hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int value, size_t count, hipStream_t s) {
  emu_stream_flush(s);
  int* p = static_cast<int*>(dst);
  for (size_t i = 0; i < count; ++i) p[i] = value;
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = new emu_stream{1};
  return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* s) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamDestroy(hipStream_t s) {
  emu_stream_flush(s);
  delete s;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
  emu_stream_flush(s);
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) {
  *e = new emu_event{std::chrono::steady_clock::now()};
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  emu_stream_flush(s);
  e->t = std::chrono::steady_clock::now();
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
