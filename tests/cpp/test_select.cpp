// CPU unit test of the host-side base selection (icra20-hand-object-pose_amd/csrc/hop_select.h):
//  1. exact_discrete_index() returns what libstdc++'s std::discrete_distribution returns for the same engine state;
//  2. the fast selection (partial sums + exactness guard + bit-mask 4th-point scan, AVX2 where available) picks the
//     same bases with the same invariants as the literal restatement of the reference, on random clouds and
//     random key-membership matrices, also when every draw is forced through the exact fallback.
// Build/run: see tests/test_select_cpu.py.
#include <cstdio>
#include <random>
#include "../../icra20-hand-object-pose_amd/csrc/hop_select.h"

using namespace hop;

static int fails = 0;
#define CHECK(c)                                              \
  do {                                                        \
    if (!(c)) {                                               \
      std::printf("CHECK failed: %s (line %d)\n", #c, __LINE__); \
      ++fails;                                                \
    }                                                         \
  } while (0)

static void test_exact_index() {
  std::mt19937 rng(7);
  for (int rep = 0; rep < 300; ++rep) {
    const int n = 2 + (int)(rng() % 400);
    std::vector<float> w(n);
    for (auto& v : w) {
      const unsigned k = rng() % 8;
      v = (k == 0) ? 0.f : (k < 4 ? 1.0f : 0.85f) * std::ldexp(1.0f, -(int)(rng() % 6));
    }
    w[rng() % n] = 1.0f;  // at least one positive
    std::mt19937 e1(rep), e2(rep);
    for (int d = 0; d < 50; ++d) {
      std::discrete_distribution<> dist(w.begin(), w.end());
      const int a = dist(e1);
      const double u = std::generate_canonical<double, std::numeric_limits<double>::digits>(e2);
      const int b = exact_discrete_index(w.data(), n, u);
      CHECK(a == b);
    }
    CHECK(e1() == e2());  // both consumed the same amount of engine state
  }
}

struct Problem {
  GenState gen;
  std::vector<unsigned long long> M;
  int W = 0;
};

static void make_problem(Problem& p, int n, unsigned seed, double density) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> U(-0.04f, 0.04f);
  std::vector<float> xyz(3 * (size_t)n), nrm(3 * (size_t)n), conf(n);
  for (int i = 0; i < n; ++i) {
    xyz[i] = U(rng), xyz[n + i] = U(rng), xyz[2 * (size_t)n + i] = 0.3f * U(rng);
    nrm[i] = U(rng), nrm[n + i] = U(rng), nrm[2 * (size_t)n + i] = 0.05f + std::fabs(U(rng));
    conf[i] = (rng() % 5 == 0) ? 0.85f : 1.0f;
  }
  load_cloud_host(p.gen.scene_h, xyz.data(), nrm.data(), n, true);
  p.gen.scene_conf = conf;
  const int m = 150;
  std::vector<float> q(3 * (size_t)m), qn(3 * (size_t)m);
  for (int i = 0; i < m; ++i) {
    q[i] = U(rng), q[m + i] = U(rng), q[2 * (size_t)m + i] = U(rng);
    qn[i] = 0, qn[m + i] = 0, qn[2 * (size_t)m + i] = 1;
  }
  load_cloud_host(p.gen.model_h[0], q.data(), qn.data(), m, true);
  p.W = (n + 63) / 64;
  p.M.assign((size_t)n * p.W, 0ull);
  std::bernoulli_distribution B(density);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      if (i != j && B(rng)) p.M[(size_t)i * p.W + (j >> 6)] |= 1ull << (j & 63);
}

static void run(Problem& p, bool fast, double tol, int trials, std::vector<int>& ids_out, std::vector<float>& inv_out, long long* fallbacks) {
  hop_s4pcs_opts o;
  o.sample_size = 60, o.overlap = 0.2f, o.delta = 0.003f, o.dispersion = 0.5f, o.success_quadrilaterals = trials, o.max_time_seconds = 0;
  o.n_trials = trials, o.random_seed = 5489u, o.max_normal_difference = -1, o.max_color_distance = -1, o.verify_mode = 0;
  GenHost G(&p.gen, o);
  G.init_clouds();
  G.M = p.M.data(), G.W = p.W;
  G.use_fast = fast;
  G.guard_tol = tol;
  ids_out.clear(), inv_out.clear();
  for (int t = 0; t < trials; ++t) {
    float i1 = 0, i2 = 0;
    int ids[4] = {-1, -1, -1, -1};
    const bool ok = G.SelectQuadrilateral(i1, i2, ids);
    ids_out.push_back(ok ? 1 : 0);
    for (int k = 0; k < 4; ++k) ids_out.push_back(ids[k]);
    inv_out.push_back(i1), inv_out.push_back(i2);
  }
  if (fallbacks) *fallbacks = G.n_fallbacks;
}

static void test_fast_equals_literal() {
  const struct { int n; double density; int trials; } cases[] = {{70, 0.5, 60}, {333, 0.3, 80}, {1500, 0.7, 60}, {4100, 0.05, 40}, {129, 0.02, 30}};
  unsigned seed = 100;
  for (const auto& cs : cases) {
    Problem p;
    make_problem(p, cs.n, seed++, cs.density);
    std::vector<int> a, b, c;
    std::vector<float> ia, ib, ic;
    long long fb_norm = 0, fb_forced = 0;
    run(p, false, 1e-10, cs.trials, a, ia, nullptr);
    run(p, true, 1e-10, cs.trials, b, ib, &fb_norm);
    run(p, true, 2.0, cs.trials, c, ic, &fb_forced);  // tolerance > 1: every draw goes through the exact routine
    CHECK(a == b);
    CHECK(a == c);
    CHECK(ia.size() == ib.size() && std::memcmp(ia.data(), ib.data(), sizeof(float) * ia.size()) == 0);
    CHECK(ia.size() == ic.size() && std::memcmp(ia.data(), ic.data(), sizeof(float) * ia.size()) == 0);
    CHECK(fb_forced > 0);
    int nsel = 0;
    for (size_t t = 0; t < a.size(); t += 5) nsel += a[t];
    std::printf("n=%d density=%.2f: %d/%d bases, fallbacks normal %lld forced %lld\n", cs.n, cs.density, nsel, cs.trials, fb_norm, fb_forced);
  }
}

// angle-bin thresholds (k_ppf_matrix fast path) against the literal acosf route, on random cosines, on every float
// around the bin boundaries cos(5 + 10 k deg), and on invalid inputs
static void test_angle_thresholds() {
  float thr[32];
  CHECK(build_angle_thresholds(thr));
  for (int k = 0; k + 1 < 18; ++k) CHECK(thr[k] > thr[k + 1]);
  std::mt19937 e(123);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  for (int i = 0; i < 4000000; ++i) {
    const float c = U(e);
    int a = -1, b = -2;
    const bool oa = ppf_angle_bin(c, &a), ob = ppf_angle_bin_thr(c, thr, &b);
    if (oa != ob || a != b) {
      CHECK(false);
      break;
    }
  }
  for (int k = 0; k < 18; ++k) {
    float c = (float)std::cos((5.0 + 10.0 * k) * M_PI / 180.0);
    for (int d = 0; d < 3000; ++d) c = std::nextafter(c, -2.f);
    for (int d = 0; d < 6000; ++d, c = std::nextafter(c, 2.f)) {
      int a = -1, b = -2;
      const bool oa = ppf_angle_bin(c, &a), ob = ppf_angle_bin_thr(c, thr, &b);
      if (oa != ob || a != b) {
        CHECK(false);
        break;
      }
    }
  }
  int b;
  CHECK(!ppf_angle_bin_thr(1.5f, thr, &b) && !ppf_angle_bin_thr(NAN, thr, &b));
  int a;
  CHECK(ppf_angle_bin_thr(1.f, thr, &b) && ppf_angle_bin(1.f, &a) && a == b && b == 0);
  CHECK(ppf_angle_bin_thr(-1.f, thr, &b) && ppf_angle_bin(-1.f, &a) && a == b && b == 180);
}

int main() {
  test_exact_index();
  test_fast_equals_literal();
  test_angle_thresholds();
  std::printf(fails ? "FAILED (%d)\n" : "OK\n", fails);
  return fails ? 1 : 0;
}
