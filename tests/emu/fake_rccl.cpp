// tests/emu/fake_rccl.cpp -- TEST INFRASTRUCTURE, only meaningful together with the CPU model of the HIP execution model (tests/emu): the six RCCL
// entry points csrc/hop_comm.hip binds by dlopen("librccl.so"), implemented between PROCESSES OF ONE MACHINE through files in /dev/shm, so
// that the product's exchange code -- id rendezvous, hop_comm_create, hop_topk_allgather_device + its merge kernel, hop_frames_allgather, the
// C++ dataset driver's end-of-shard gather -- runs with MORE THAN ONE RANK where no GPU (let alone two) is available.  "Device" buffers
// are host memory in the model, so a collective is a file written per rank and read by the others.  Not RCCL: no topology, no streams, no
// performance; the call sequence, the buffer arithmetic and the rank bookkeeping of the caller are what it exercises.
#include <dlfcn.h>
#include <dirent.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
struct Id {
  char internal[128];
};
struct Comm {
  std::string dir;
  int nranks, rank;
  long seq;
};
const double WAIT_S = getenv("FAKE_RCCL_WAIT_S") ? atof(getenv("FAKE_RCCL_WAIT_S")) : 120.0;
bool exists(const std::string& p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0;
}
bool wait_for(const std::string& p) {
  const auto t0 = std::chrono::steady_clock::now();
  while (!exists(p)) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > WAIT_S) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
  }
  return true;
}
bool write_file(const std::string& p, const void* data, size_t bytes) {
  const std::string tmp = p + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = bytes == 0 || std::fwrite(data, 1, bytes, f) == bytes;
  std::fclose(f);
  return ok && std::rename(tmp.c_str(), p.c_str()) == 0;  // (a reader never sees half a file)
}
bool read_file(const std::string& p, void* data, size_t bytes) {
  FILE* f = std::fopen(p.c_str(), "rb");
  if (!f) return false;
  const bool ok = bytes == 0 || std::fread(data, 1, bytes, f) == bytes;
  std::fclose(f);
  return ok;
}
size_t type_bytes(int dtype) {  // ncclDataType_t: 0 int8, 1 uint8, 2 int32, 3 uint32, 4 int64, 5 uint64, 6 half, 7 float, 8 double, 9 bfloat16
  switch (dtype) {
    case 0: case 1: return 1;
    case 6: case 9: return 2;
    case 2: case 3: case 7: return 4;
    default: return 8;
  }
}
}  // namespace

extern "C" {
int ncclGetUniqueId(Id* id) {
  int fd = open("/dev/urandom", O_RDONLY);
  if (fd < 0 || read(fd, id->internal, sizeof id->internal) != (ssize_t)sizeof id->internal) return 2;  // ncclSystemError
  close(fd);
  return 0;
}
int ncclCommInitRank(void** comm, int nranks, Id id, int rank) {
  if (!comm || nranks <= 0 || rank < 0 || rank >= nranks) return 4;  // ncclInvalidArgument
  char hex[33];
  for (int i = 0; i < 16; ++i) std::snprintf(hex + 2 * i, 3, "%02x", (unsigned char)id.internal[i]);
  Comm* c = new Comm{std::string("/dev/shm/hop_fake_rccl_") + hex, nranks, rank, 0};
  mkdir(c->dir.c_str(), 0700);
  if (!write_file(c->dir + "/init.r" + std::to_string(rank), "", 0)) return 2;
  for (int r = 0; r < nranks; ++r)
    if (!wait_for(c->dir + "/init.r" + std::to_string(r))) return 2;
  *comm = c;
  return 0;
}
int ncclCommCount(void* comm, int* n) {
  *n = static_cast<Comm*>(comm)->nranks;
  return 0;
}
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, void* stream) {
  // the collective is ordered on `stream`: copies the model still holds queued there (tests/emu/emu_runtime.cpp) happen first
  if (auto flush = reinterpret_cast<void (*)(void*)>(dlsym(RTLD_DEFAULT, "emu_stream_flush"))) flush(stream);
  Comm* c = static_cast<Comm*>(comm);
  const size_t bytes = count * type_bytes(dtype);
  const std::string base = c->dir + "/ag" + std::to_string(++c->seq) + ".r";
  std::vector<char> mine(static_cast<const char*>(send), static_cast<const char*>(send) + bytes);  // (send may alias a slice of recv)
  if (!write_file(base + std::to_string(c->rank), mine.data(), bytes)) return 2;
  for (int r = 0; r < c->nranks; ++r) {
    if (!wait_for(base + std::to_string(r))) return 2;
    if (!read_file(base + std::to_string(r), static_cast<char*>(recv) + (size_t)r * bytes, bytes)) return 2;
  }
  return 0;
}
int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  // the last rank to leave removes the directory (every rank has read everything it waited for by the time it gets here)
  write_file(c->dir + "/bye.r" + std::to_string(c->rank), "", 0);
  bool all = true;
  for (int r = 0; r < c->nranks; ++r) all = all && exists(c->dir + "/bye.r" + std::to_string(r));
  if (all) {
    if (DIR* d = opendir(c->dir.c_str())) {
      while (dirent* e = readdir(d))
        if (e->d_name[0] != '.') unlink((c->dir + "/" + e->d_name).c_str());
      closedir(d);
    }
    rmdir(c->dir.c_str());
  }
  delete c;
  return 0;
}
const char* ncclGetErrorString(int rc) { return rc == 0 ? "no error" : rc == 2 ? "system error (fake RCCL of tests/emu: a rank did not arrive?)" : "invalid argument"; }
}
