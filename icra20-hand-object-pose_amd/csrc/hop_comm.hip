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

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hop.h"

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
  int k_cap = 0;
  std::string last_error;
};

extern "C" {

int hop_comm_unique_id(unsigned char id_out[HOP_COMM_ID_BYTES]) {
  if (!id_out) return HOP_E_INVALID;
  Rccl& r = rccl();
  if (!r.ok) return HOP_E_COMM;
  NcclId id;
  if (r.GetUniqueId(&id) != 0) return HOP_E_COMM;
  std::memcpy(id_out, id.internal, HOP_COMM_ID_BYTES);
  return HOP_OK;
}

int hop_comm_create(int device, const unsigned char id[HOP_COMM_ID_BYTES], int rank, int world, hop_comm** out) {
  if (!id || !out || world <= 0 || rank < 0 || rank >= world) return HOP_E_INVALID;
  *out = nullptr;
  Rccl& r = rccl();
  if (!r.ok) return HOP_E_COMM;
  if (hipSetDevice(device) != hipSuccess) return HOP_E_NO_DEVICE;
  hop_comm* c = new hop_comm;
  c->rank = rank, c->world = world, c->device = device;
  NcclId nid;
  std::memcpy(nid.internal, id, HOP_COMM_ID_BYTES);
  const int rc = r.CommInitRank(&c->comm, world, nid, rank);
  if (rc != 0 || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    if (c->comm) r.CommDestroy(c->comm);
    delete c;
    return HOP_E_COMM;
  }
  *out = c;
  return HOP_OK;
}

void hop_comm_destroy(hop_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm) rccl().CommDestroy(c->comm);
  if (c->send) (void)hipFree(c->send);
  if (c->recv) (void)hipFree(c->recv);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* hop_comm_last_error(const hop_comm* c) { return c ? c->last_error.c_str() : rccl().error.c_str(); }

// rows_in: this rank's table (hop_topk_pack, k rows of HOP_TOPK_ROW_FLOATS); merged_out: the k best rows of all ranks in
// HypoCompare order (hop_topk_merge), identical on every rank.  Every rank must call it the same number of times, in the
// same order.
int hop_topk_allgather(hop_comm* c, const float* rows_in, int k, float* merged_out, int* n_rows_out) {
  if (!c || !rows_in || !merged_out || k <= 0) return HOP_E_INVALID;
  Rccl& r = rccl();
  if (hipSetDevice(c->device) != hipSuccess) return HOP_E_HIP;
  const size_t row_floats = (size_t)k * HOP_TOPK_ROW_FLOATS;
  if (k > c->k_cap) {
    if (c->send) (void)hipFree(c->send);
    if (c->recv) (void)hipFree(c->recv);
    c->send = c->recv = nullptr;
    if (hipMalloc(&c->send, sizeof(float) * row_floats) != hipSuccess || hipMalloc(&c->recv, sizeof(float) * row_floats * c->world) != hipSuccess) return HOP_E_ALLOC;
    c->k_cap = k;
  }
  std::vector<float> all(row_floats * c->world);
  if (hipMemcpyAsync(c->send, rows_in, sizeof(float) * row_floats, hipMemcpyHostToDevice, c->stream) != hipSuccess) return HOP_E_HIP;
  const int rc = r.AllGather(c->send, c->recv, row_floats, 7 /* ncclFloat32 */, c->comm, c->stream);
  if (rc != 0) {
    c->last_error = std::string("ncclAllGather: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error");
    return HOP_E_COMM;
  }
  if (hipMemcpyAsync(all.data(), c->recv, sizeof(float) * row_floats * c->world, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return HOP_E_HIP;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return HOP_E_HIP;
  return hop_topk_merge(all.data(), c->world, k, merged_out, n_rows_out);
}

}  // extern "C"
