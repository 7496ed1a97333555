// CPU build of the PRODUCT's minimiser text (icra20-hand-object-pose_amd/csrc/hop_lm_core.h -- the header k_icp_lm7_solve runs on the GPU), driven
// the way the kernel drives it.  tests/test_lm_core_cpu.py compares what it returns with the oracle's independent statement of the same
// algorithm (oracle/hop_oracle.cpp lm_point_to_plane_moments) bit for bit: the arithmetic of nn_mode 7's solve is checked without a GPU.
//   lm_core_host <cases.bin> [6] -> one line per case: x[6] as hex floats, status, nfev, iter; then "counts <general lmpar2> <register, par = 0> <register, iterated>"
//   cases.bin: int32 n, then n x (169 doubles M row-major, 3 doubles c)
//   "6": the fast-arithmetic machine of nn_mode 6 (LmDev6, lm6_eval<true>: on the host its reciprocal estimates are IEEE operations) -- the
//   same state machine through its other branches; no oracle states it, the test checks what it does with a NaN evaluation
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace hop_host {
long g_count[8];  // the header's statistics hooks: [2] general (pivoted) lmpar2 calls, [4] register lmpar2 with par = 0, [5] with the secular iteration
#define LM_COUNT(slot, v) (hop_host::g_count[slot] += (v))
constexpr float LM_SQRT_EPS_F = 3.4526698300124393e-04f;  // (csrc/hop_device.h)
#include "../../icra20-hand-object-pose_amd/csrc/hop_lm_core.h"

// the loop of k_icp_lm7_solve / k_icp_lm6_solve (csrc/hop_icp_lm.hip) on one moment matrix
template <class LS, bool FAST>
void solve(const double* M169, const double* c, float x_out[6], int st_out[3]) {
  double M[91];
  for (int i = 0; i < 13; ++i)
    for (int j = 0; j <= i; ++j) M[i * (i + 1) / 2 + j] = M169[13 * i + j];
  LS s;
  for (int j = 0; j < 6; ++j) s.x[j] = s.xc[j] = 0.f, s.p[j] = 0.f;
  s.phase = 0, s.status = -1, s.iter = 0, s.nfev = 0;
  s.par = s.delta = s.xnorm = s.fnorm = s.gnorm = s.pnorm = 0.0;
  double cand[28];
  for (int guard = 0; guard < 420; ++guard) {
    lm6_eval<FAST, 1>(M, c, s.xc, cand);
    if (!lm_advance(s, cand)) break;
  }
  for (int j = 0; j < 6; ++j) x_out[j] = s.x[j];
  st_out[0] = s.status, st_out[1] = s.nfev, st_out[2] = s.iter;
}
}  // namespace hop_host

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const bool mode6 = argc > 2 && argv[2][0] == '6';
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t n = 0;
  if (std::fread(&n, 4, 1, f) != 1) return 3;
  std::vector<double> buf(172);
  for (int k = 0; k < n; ++k) {
    if (std::fread(buf.data(), sizeof(double), 172, f) != 172) return 3;
    float x[6];
    int st[3];
    if (mode6) hop_host::solve<hop_host::LmDev6, true>(buf.data(), buf.data() + 169, x, st);
    else hop_host::solve<hop_host::LmDev7, false>(buf.data(), buf.data() + 169, x, st);
    for (int j = 0; j < 6; ++j) {
      uint32_t u;
      std::memcpy(&u, &x[j], 4);
      std::printf("%08x ", u);
    }
    std::printf("%d %d %d\n", st[0], st[1], st[2]);
  }
  std::fclose(f);
  std::printf("counts %ld %ld %ld\n", hop_host::g_count[2], hop_host::g_count[4], hop_host::g_count[5]);
  return 0;
}
