// CPU check of the PRODUCT's host side of the generator (icra20-hand-object-pose_amd/csrc/hop_select.h: MatchBase::init's sampling and
// centring, the PPF key membership k_ppf_matrix evaluates on the device -- here through ppf_member_host, the same expression on the host --,
// SelectRandomTriangle / SelectQuadrilateral) on the inputs of a reference-built golden vector: prints the sampled-Q count, the centroids,
// the diameter and the successful bases (ids, invariants) of the reference's 30 trials for tests/test_select_cpu.py to compare with the
// reference build's own trace (tests/golden/s4pcs_*.npz).  Input: the binary layout of tools/select_dump.py.
//   select_golden <dump.bin> <sample_size> <fast 0|1>
#include <cstdio>
#include <fstream>
#include <string>
#include "../../icra20-hand-object-pose_amd/csrc/hop_select.h"

using namespace hop;

static std::vector<float> readf(std::ifstream& f, size_t n) {
  std::vector<float> v(n);
  f.read(reinterpret_cast<char*>(v.data()), sizeof(float) * n);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  std::ifstream f(argv[1], std::ios::binary);
  int32_t hdr[3];
  f.read(reinterpret_cast<char*>(hdr), 12);
  const int N = hdr[0], Mq = hdr[1], nkeys = hdr[2];
  auto pxyz = readf(f, 3 * (size_t)N), pnrm = readf(f, 3 * (size_t)N), conf = readf(f, N);
  auto qxyz = readf(f, 3 * (size_t)Mq), qnrm = readf(f, 3 * (size_t)Mq);
  std::vector<int32_t> keys(4 * (size_t)nkeys);
  f.read(reinterpret_cast<char*>(keys.data()), sizeof(int32_t) * keys.size());
  if (!f) return 3;
  GenState gen;
  load_cloud_host(gen.scene_h, pxyz.data(), pnrm.data(), N, true);
  gen.scene_conf = conf;
  load_cloud_host(gen.model_h[0], qxyz.data(), qnrm.data(), Mq, true);
  hop_s4pcs_opts o;
  o.sample_size = atoi(argv[2]), o.overlap = 0.2f, o.delta = 0.003f, o.dispersion = 0.5f, o.success_quadrilaterals = 1 << 20;
  o.max_time_seconds = 0, o.n_trials = 30, o.random_seed = 5489u, o.max_normal_difference = -1, o.max_color_distance = -1, o.verify_mode = 0;
  GenHost G(&gen, o);
  G.init_clouds();
  // MatchBase::init's state (sampling.h:67-144, matchBase.hpp:380-462): sampled, centred Q; centroids; diameter -- as bit patterns
  std::printf("state %d %08x %08x %08x %08x %08x %08x %08x\n", gen.gq_h.n, f2u(gen.centroid_p[0]), f2u(gen.centroid_p[1]), f2u(gen.centroid_p[2]),
              f2u(gen.centroid_q[0]), f2u(gen.centroid_q[1]), f2u(gen.centroid_q[2]), f2u(gen.diameter));
  for (int k = 0; k < gen.gq_h.n; ++k)
    std::printf("q %08x %08x %08x %08x %08x %08x\n", f2u(gen.gq_h.x[k]), f2u(gen.gq_h.y[k]), f2u(gen.gq_h.z[k]), f2u(gen.gq_h.nx[k]), f2u(gen.gq_h.ny[k]),
                f2u(gen.gq_h.nz[k]));
  std::vector<unsigned> bitmap;
  int dist_bins = 0;
  build_key_bitmap(keys.data(), nkeys, bitmap, dist_bins);
  const int W = (N + 63) / 64;
  std::vector<unsigned long long> M((size_t)N * W, 0ull);
  std::vector<V3> pos(N), nn(N);
  for (int i = 0; i < N; ++i) {
    pos[i] = v3(gen.gp_h.x[i], gen.gp_h.y[i], gen.gp_h.z[i]);
    nn[i] = vnormalized(vnormalized(v3(gen.gp_h.nx[i], gen.gp_h.ny[i], gen.gp_h.nz[i])));
  }
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j)
      if (i != j && ppf_member_host(pos[i], nn[i], pos[j], nn[j], bitmap, dist_bins)) M[(size_t)i * W + (j >> 6)] |= 1ull << (j & 63);
  G.M = M.data();
  G.W = W;
  G.use_fast = atoi(argv[3]) != 0;
  for (int t = 0; t < 30; ++t) {
    float i1 = 0, i2 = 0;
    int ids[4];
    if (G.SelectQuadrilateral(i1, i2, ids)) std::printf("base %d %d %d %d %08x %08x\n", ids[0], ids[1], ids[2], ids[3], f2u(i1), f2u(i2));
  }
  return 0;
}
