// hop_comm.hip -- the one cross-GPU exchange of the path: every rank's top-k table (k rows of 72 bytes)
// all-gathered over RCCL / xGMI and merged identically on every rank -- "all-reduce with a top-k operator", which RCCL
// does not provide natively.  One exchange per frame, latency-bound (9 KB per rank).
//
// The reference has no multi-GPU path (shared-memory OpenMP only); this is the hypothesis-parallel extension north_star
// asks for.  RCCL is loaded at run time (dlopen librccl.so) so that the single-GPU library has no hard dependency on it.
// The unique id of the communicator is produced by rank 0 (hop_comm_unique_id) and handed to the other ranks by the
// launcher (bench.py broadcasts it through torch.distributed; any out-of-band channel does).
#include <dlfcn.h>
#include <link.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hop.h"
#include "hop_math.h"

namespace {

struct NcclId {
  char internal[128];
};
typedef void* NcclComm;
struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommCount)(NcclComm, int*) = nullptr;
  std::string error;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy the process already holds (PyTorch ships its own librccl.so) is taken first: one RCCL per process
    std::string loaded;
    dl_iterate_phdr(
        [](struct dl_phdr_info* info, size_t, void* out) -> int {
          if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl.so")) {
            *static_cast<std::string*>(out) = info->dlpi_name;
            return 1;
          }
          return 0;
        },
        &loaded);
    if (!loaded.empty()) r.handle = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      if (r.handle) break;
      r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.handle) {
      r.error = std::string("dlopen librccl.so: ") + (dlerror() ? dlerror() : "not found");
      return;
    }
    r.GetUniqueId = reinterpret_cast<int (*)(NcclId*)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<int (*)(NcclComm*, int, NcclId, int)>(dlsym(r.handle, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<int (*)(NcclComm)>(dlsym(r.handle, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, NcclComm, hipStream_t)>(dlsym(r.handle, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(r.handle, "ncclGetErrorString"));
    r.CommCount = reinterpret_cast<int (*)(NcclComm, int*)>(dlsym(r.handle, "ncclCommCount"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather;
    if (!r.ok) r.error = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
  });
  return r;
}

}  // namespace

struct hop_comm {
  NcclComm comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  float* send = nullptr;
  float* recv = nullptr;
  float* merged = nullptr;   // device: the k merged rows
  int k_cap = 0;
  float* fsend = nullptr;    // frames all-gather staging (device)
  float* frecv = nullptr;
  size_t f_cap = 0;
  double exchange_us_sum = 0;  // wall time of the top-k exchanges (pack excluded), for the bench line
  long exchange_count = 0;
  std::string last_error;
};

namespace {
// keeps the calling thread's current HIP device as it found it (a PyTorch caller may have another one current)
struct DeviceGuard {
  int prev = -1;
  bool ok;
  explicit DeviceGuard(int dev) {
    (void)hipGetDevice(&prev);
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
// errors that have no communicator to hang on (hop_comm_unique_id, hop_comm_create): one string per process, written and read under a
// lock (several host threads may create communicators at once); the reader gets a thread-local copy, so the pointer it hands out stays
// valid while another thread writes
std::mutex& process_error_lock() {
  static std::mutex m;
  return m;
}
std::string& process_error_unlocked() {
  static std::string e;
  return e;
}
void set_process_error(const std::string& e) {
  std::lock_guard<std::mutex> lk(process_error_lock());
  process_error_unlocked() = e;
}
const char* process_error_cstr() {
  static thread_local std::string copy;
  std::lock_guard<std::mutex> lk(process_error_lock());
  copy = process_error_unlocked();
  return copy.c_str();
}
std::string nccl_err(const char* what, int rc) {
  Rccl& r = rccl();
  return std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "error") + " (" + std::to_string(rc) + ")";
}

// a HIP call inside an exchange: on failure the communicator's error string says which call and why, and the stream is drained so
// that nothing queued before the failure still runs when the caller reacts
#define COMM_HIP(c, call)                                                                  \
  do {                                                                                     \
    const hipError_t e__ = (call);                                                         \
    if (e__ != hipSuccess) {                                                               \
      (c)->last_error = std::string(#call) + ": " + hipGetErrorString(e__);                \
      (void)hipStreamSynchronize((c)->stream);                                             \
      return HOP_E_HIP;                                                                    \
    }                                                                                      \
  } while (0)

// HypoCompare key of a row: descending score, then ascending id, as one ascending 64-bit key; empty rows (id < 0) last
__device__ __forceinline__ unsigned long long row_key(const float* row) {
  const int id = __float_as_int(row[1]);
  if (id < 0) return ~0ull;
  return ((unsigned long long)(~hop::score_order_key(row[0])) << 32) | (unsigned)id;
}
// merge of the gathered tables on the device: one workgroup sorts the n_rows (<= 4096) keys with their row numbers (bitonic, LDS)
// and writes the k best rows -- the order hop_topk_merge produces on the host (stable_sort by the same comparator; ids are unique)
constexpr int MERGE_MAX = 4096;
__global__ __launch_bounds__(1024) void k_topk_merge(const float* __restrict__ rows, int n_rows, int k, float* __restrict__ out) {
  __shared__ unsigned long long key[MERGE_MAX];
  __shared__ unsigned short idx[MERGE_MAX];
  int n2 = 1;
  while (n2 < n_rows) n2 <<= 1;
  for (int i = threadIdx.x; i < n2; i += blockDim.x) {
    key[i] = i < n_rows ? row_key(rows + (size_t)i * HOP_TOPK_ROW_FLOATS) : ~0ull;
    idx[i] = (unsigned short)i;
  }
  __syncthreads();
  for (int size = 2; size <= n2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        const int j = i ^ stride;
        if (j > i) {
          const bool up = (i & size) == 0;
          const unsigned long long a = key[i], b = key[j];
          const bool swap = up ? (a > b || (a == b && idx[i] > idx[j])) : (a < b || (a == b && idx[i] < idx[j]));
          if (swap) {
            key[i] = b, key[j] = a;
            const unsigned short t = idx[i];
            idx[i] = idx[j], idx[j] = t;
          }
        }
      }
      __syncthreads();
    }
  for (int t = threadIdx.x; t < k * HOP_TOPK_ROW_FLOATS; t += blockDim.x) {
    const int r = t / HOP_TOPK_ROW_FLOATS, q = t % HOP_TOPK_ROW_FLOATS;
    float v;
    if (r < n_rows && key[r] != ~0ull) v = rows[(size_t)idx[r] * HOP_TOPK_ROW_FLOATS + q];
    else v = q == 0 ? -3.402823466e+38f : q == 1 ? __int_as_float(-1) : 0.f;
    out[t] = v;
  }
}
}  // namespace

extern "C" {

int hop_comm_unique_id(unsigned char id_out[HOP_COMM_ID_BYTES]) {
  if (!id_out) return HOP_E_INVALID;
  Rccl& r = rccl();
  if (!r.ok) {
    set_process_error(r.error);
    return HOP_E_COMM;
  }
  NcclId id;
  if (const int rc = r.GetUniqueId(&id)) {
    set_process_error(nccl_err("ncclGetUniqueId", rc));
    return HOP_E_COMM;
  }
  std::memcpy(id_out, id.internal, HOP_COMM_ID_BYTES);
  return HOP_OK;
}

int hop_comm_create(int device, const unsigned char id[HOP_COMM_ID_BYTES], int rank, int world, hop_comm** out) {
  if (!id || !out || world <= 0 || rank < 0 || rank >= world) return HOP_E_INVALID;
  *out = nullptr;
  Rccl& r = rccl();
  if (!r.ok) {
    set_process_error(r.error);
    return HOP_E_COMM;
  }
  DeviceGuard dg(device);
  if (!dg.ok) return HOP_E_NO_DEVICE;
  hop_comm* c = new hop_comm;
  c->rank = rank, c->world = world, c->device = device;
  NcclId nid;
  std::memcpy(nid.internal, id, HOP_COMM_ID_BYTES);
  const int rc = r.CommInitRank(&c->comm, world, nid, rank);
  hipError_t he = hipSuccess;
  if (rc != 0 || (he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
    set_process_error(rc != 0 ? nccl_err("ncclCommInitRank", rc) : std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(he));
    if (c->comm) r.CommDestroy(c->comm);
    delete c;
    return HOP_E_COMM;
  }
  *out = c;
  return HOP_OK;
}

void hop_comm_destroy(hop_comm* c) {
  if (!c) return;
  DeviceGuard dg(c->device);
  // everything this communicator queued has completed before it, its stream and its buffers go
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) rccl().CommDestroy(c->comm);
  if (c->send) (void)hipFree(c->send);
  if (c->recv) (void)hipFree(c->recv);
  if (c->merged) (void)hipFree(c->merged);
  if (c->fsend) (void)hipFree(c->fsend);
  if (c->frecv) (void)hipFree(c->frecv);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* hop_comm_last_error(const hop_comm* c) {
  if (c) return c->last_error.c_str();
  const char* e = process_error_cstr();
  return *e ? e : rccl().error.c_str();
}

// ranks of the communicator as RCCL reports them (ncclCommCount), and the mean wall time of the exchanges so far
int hop_comm_info(hop_comm* c, int* rccl_ranks_out, double* mean_exchange_us_out, long* exchanges_out) {
  if (!c) return HOP_E_INVALID;
  int n = c->world;
  if (rccl().CommCount && rccl().CommCount(c->comm, &n) != 0) n = -1;
  if (rccl_ranks_out) *rccl_ranks_out = n;
  if (mean_exchange_us_out) *mean_exchange_us_out = c->exchange_count ? c->exchange_us_sum / (double)c->exchange_count : 0.0;
  if (exchanges_out) *exchanges_out = c->exchange_count;
  return HOP_OK;
}

static int ensure_topk_buffers(hop_comm* c, int k) {
  const size_t row_floats = (size_t)k * HOP_TOPK_ROW_FLOATS;
  if (k > c->k_cap) {
    if (c->send) (void)hipFree(c->send);
    if (c->recv) (void)hipFree(c->recv);
    if (c->merged) (void)hipFree(c->merged);
    c->send = c->recv = c->merged = nullptr;
    if (hipMalloc(&c->send, sizeof(float) * row_floats) != hipSuccess || hipMalloc(&c->recv, sizeof(float) * row_floats * c->world) != hipSuccess ||
        hipMalloc(&c->merged, sizeof(float) * row_floats) != hipSuccess) {
      c->last_error = "hipMalloc of the exchange buffers failed";
      return HOP_E_ALLOC;
    }
    c->k_cap = k;
  }
  return HOP_OK;
}

// The exchange with the table on the device from end to end: rows_dev is this rank's table in DEVICE memory (hop_topk_pack_device,
// completed), ncclAllGather straight from it, k_topk_merge on the gathered rows (world * k <= 4096; beyond that the host merge), and only the k merged
// rows (9 KB) come back to the host -- no download of the hypothesis set, no host sort, no upload.
int hop_topk_allgather_device(hop_comm* c, const float* rows_dev, int k, float* merged_out, int* n_rows_out) {
  if (!c || !rows_dev || !merged_out || k <= 0) return HOP_E_INVALID;
  Rccl& r = rccl();
  DeviceGuard dg(c->device);
  if (!dg.ok) {
    c->last_error = "hipSetDevice failed for the communicator's device";
    return HOP_E_HIP;
  }
  if (const int rc = ensure_topk_buffers(c, k)) return rc;
  const size_t row_floats = (size_t)k * HOP_TOPK_ROW_FLOATS;
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = r.AllGather(rows_dev, c->recv, row_floats, 7 /* ncclFloat32 */, c->comm, c->stream);
  if (rc != 0) {
    c->last_error = nccl_err("ncclAllGather", rc);
    (void)hipStreamSynchronize(c->stream);
    return HOP_E_COMM;
  }
  const int n_rows = k * c->world;
  int status = HOP_OK;
  if (n_rows <= MERGE_MAX) {
    hipLaunchKernelGGL(k_topk_merge, dim3(1), dim3(1024), 0, c->stream, c->recv, n_rows, k, c->merged);
    COMM_HIP(c, hipMemcpyAsync(merged_out, c->merged, sizeof(float) * row_floats, hipMemcpyDeviceToHost, c->stream));
    COMM_HIP(c, hipStreamSynchronize(c->stream));
    if (n_rows_out) {
      int m = 0;
      for (int q = 0; q < k; ++q) {
        int id;
        std::memcpy(&id, &merged_out[(size_t)q * HOP_TOPK_ROW_FLOATS + 1], 4);
        m += id >= 0;
      }
      *n_rows_out = m;
    }
  } else {
    std::vector<float> all(row_floats * c->world);
    COMM_HIP(c, hipMemcpyAsync(all.data(), c->recv, sizeof(float) * row_floats * c->world, hipMemcpyDeviceToHost, c->stream));
    COMM_HIP(c, hipStreamSynchronize(c->stream));
    status = hop_topk_merge(all.data(), c->world, k, merged_out, n_rows_out);
  }
  c->exchange_us_sum += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  c->exchange_count += 1;
  return status;
}

// BASELINE configs[3] / SURVEY 8(e) "C4 frames": every rank hands in the poses of the frames it processed -- n_local rows of
// HOP_FRAME_ROW_FLOATS floats (frame index as a float, then the 4 x 4 pose, row-major; run_real_all.cpp:256-262 writes this matrix per
// frame) -- padded to rows_per_rank; ONE ncclAllGather at the end of the shard; all_out receives world * rows_per_rank rows in rank
// order (padding rows carry frame index -1).
int hop_frames_allgather(hop_comm* c, const float* rows_local, int n_local, int rows_per_rank, float* all_out) {
  if (!c || !all_out || n_local < 0 || rows_per_rank <= 0 || n_local > rows_per_rank || (n_local > 0 && !rows_local)) return HOP_E_INVALID;
  Rccl& r = rccl();
  DeviceGuard dg(c->device);
  if (!dg.ok) {
    c->last_error = "hipSetDevice failed for the communicator's device";
    return HOP_E_HIP;
  }
  const size_t nf = (size_t)rows_per_rank * HOP_FRAME_ROW_FLOATS;
  if (nf > c->f_cap) {
    if (c->fsend) (void)hipFree(c->fsend);
    if (c->frecv) (void)hipFree(c->frecv);
    c->fsend = c->frecv = nullptr;
    if (hipMalloc(&c->fsend, sizeof(float) * nf) != hipSuccess || hipMalloc(&c->frecv, sizeof(float) * nf * c->world) != hipSuccess) {
      c->last_error = "hipMalloc of the frame-table buffers failed";
      return HOP_E_ALLOC;
    }
    c->f_cap = nf;
  }
  std::vector<float> pad(nf, 0.f);
  for (int q = 0; q < rows_per_rank; ++q) pad[(size_t)q * HOP_FRAME_ROW_FLOATS] = -1.f;
  if (n_local > 0) std::memcpy(pad.data(), rows_local, sizeof(float) * (size_t)n_local * HOP_FRAME_ROW_FLOATS);
  COMM_HIP(c, hipMemcpyAsync(c->fsend, pad.data(), sizeof(float) * nf, hipMemcpyHostToDevice, c->stream));
  const int rc = r.AllGather(c->fsend, c->frecv, nf, 7 /* ncclFloat32 */, c->comm, c->stream);
  if (rc != 0) {
    c->last_error = nccl_err("ncclAllGather", rc);
    (void)hipStreamSynchronize(c->stream);
    return HOP_E_COMM;
  }
  COMM_HIP(c, hipMemcpyAsync(all_out, c->frecv, sizeof(float) * nf * c->world, hipMemcpyDeviceToHost, c->stream));
  COMM_HIP(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

// rows_in: this rank's table (hop_topk_pack, k rows of HOP_TOPK_ROW_FLOATS); merged_out: the k best rows of all ranks in
// HypoCompare order (hop_topk_merge), identical on every rank.  Every rank must call it the same number of times, in the
// same order.
int hop_topk_allgather(hop_comm* c, const float* rows_in, int k, float* merged_out, int* n_rows_out) {
  if (!c || !rows_in || !merged_out || k <= 0) return HOP_E_INVALID;
  Rccl& r = rccl();
  DeviceGuard dg(c->device);
  if (!dg.ok) {
    c->last_error = "hipSetDevice failed for the communicator's device";
    return HOP_E_HIP;
  }
  if (const int rc = ensure_topk_buffers(c, k)) return rc;
  const size_t row_floats = (size_t)k * HOP_TOPK_ROW_FLOATS;
  std::vector<float> all(row_floats * c->world);
  const auto t0 = std::chrono::steady_clock::now();
  COMM_HIP(c, hipMemcpyAsync(c->send, rows_in, sizeof(float) * row_floats, hipMemcpyHostToDevice, c->stream));
  const int rc = r.AllGather(c->send, c->recv, row_floats, 7 /* ncclFloat32 */, c->comm, c->stream);
  if (rc != 0) {
    c->last_error = nccl_err("ncclAllGather", rc);
    (void)hipStreamSynchronize(c->stream);
    return HOP_E_COMM;
  }
  COMM_HIP(c, hipMemcpyAsync(all.data(), c->recv, sizeof(float) * row_floats * c->world, hipMemcpyDeviceToHost, c->stream));
  COMM_HIP(c, hipStreamSynchronize(c->stream));
  c->exchange_us_sum += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  c->exchange_count += 1;
  return hop_topk_merge(all.data(), c->world, k, merged_out, n_rows_out);
}

}  // extern "C"
