// tools/select_bench.cpp -- CPU-only harness for the host side of the generator (hop_select.h):
// builds the PPF membership matrix on the CPU (OpenMP; the GPU kernel k_ppf_matrix computes the same bits),
// then runs the sequential base selection and times it.  Development tool, not part of the product or the tests.
//   g++ -O2 -std=c++17 -fopenmp -ffp-contract=off tools/select_bench.cpp -o /tmp/select_bench
//   /tmp/select_bench /tmp/sel_dump.bin 2048 [fast]
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>
#include "../icra20-hand-object-pose_amd/csrc/hop_select.h"

using namespace hop;

static std::vector<float> readf(std::ifstream& f, size_t n) {
  std::vector<float> v(n);
  f.read(reinterpret_cast<char*>(v.data()), sizeof(float) * n);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  std::ifstream f(argv[1], std::ios::binary);
  int32_t hdr[3];
  f.read(reinterpret_cast<char*>(hdr), 12);
  const int N = hdr[0], Mq = hdr[1], nkeys = hdr[2];
  auto pxyz = readf(f, 3 * (size_t)N), pnrm = readf(f, 3 * (size_t)N), conf = readf(f, N);
  auto qxyz = readf(f, 3 * (size_t)Mq), qnrm = readf(f, 3 * (size_t)Mq);
  std::vector<int32_t> keys(4 * (size_t)nkeys);
  f.read(reinterpret_cast<char*>(keys.data()), sizeof(int32_t) * keys.size());
  const int trials = atoi(argv[2]);
  const bool fast = argc > 3 && std::string(argv[3]) == "fast";

  GenState gen;
  load_cloud_host(gen.scene_h, pxyz.data(), pnrm.data(), N, true);
  gen.scene_conf = conf;
  load_cloud_host(gen.model_h[0], qxyz.data(), qnrm.data(), Mq, true);
  hop_s4pcs_opts o;
  o.sample_size = 100, o.overlap = 0.2f, o.delta = 0.003f, o.dispersion = 0.5f, o.success_quadrilaterals = trials;
  o.max_time_seconds = 0, o.n_trials = trials, o.random_seed = 5489u, o.max_normal_difference = -1, o.max_color_distance = -1, o.verify_mode = 0;
  GenHost G(&gen, o);
  G.init_clouds();
  std::vector<unsigned> bitmap;
  int dist_bins = 0;
  build_key_bitmap(keys.data(), nkeys, bitmap, dist_bins);
  const int W = (N + 63) / 64;
  std::vector<unsigned long long> M((size_t)N * W, 0ull);
  const std::string cache = "/tmp/sel_matrix_" + std::to_string(N) + ".bin";
  {
    std::ifstream c(cache, std::ios::binary);
    if (c) c.read(reinterpret_cast<char*>(M.data()), sizeof(unsigned long long) * M.size());
    if (!c) {
      std::vector<V3> pos(N), nn(N);
      for (int i = 0; i < N; ++i) {
        pos[i] = v3(gen.gp_h.x[i], gen.gp_h.y[i], gen.gp_h.z[i]);
        nn[i] = vnormalized(vnormalized(v3(gen.gp_h.nx[i], gen.gp_h.ny[i], gen.gp_h.nz[i])));
      }
      const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic, 16)
      for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
          if (i != j && ppf_member_host(pos[i], nn[i], pos[j], nn[j], bitmap, dist_bins)) M[(size_t)i * W + (j >> 6)] |= 1ull << (j & 63);
      std::printf("matrix on CPU: %.1f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
      std::ofstream oc(cache, std::ios::binary);
      oc.write(reinterpret_cast<const char*>(M.data()), sizeof(unsigned long long) * M.size());
    }
  }
  G.M = M.data();
  G.W = W;
  G.use_fast = fast;
  long long popc = 0;
  for (auto w : M) popc += __builtin_popcountll(w);
  std::printf("N=%d density=%.3f\n", N, (double)popc / ((double)N * N));
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long long h = 1469598103934665603ull;
  int nsel = 0;
  for (int t = 0; t < trials; ++t) {
    float i1 = 0, i2 = 0;
    int ids[4];
    if (G.SelectQuadrilateral(i1, i2, ids)) {
      ++nsel;
      for (int k = 0; k < 4; ++k) h = (h ^ (unsigned)ids[k]) * 1099511628211ull;
      h = (h ^ f2u(i1)) * 1099511628211ull;
      h = (h ^ f2u(i2)) * 1099511628211ull;
    }
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("%s: %d trials, %d bases, %.1f ms total, %.1f us/trial, checksum %016llx\n", fast ? "fast" : "literal", trials, nsel, 1e3 * dt,
              1e6 * dt / trials, h);
  G.print_profile();
  return 0;
}
