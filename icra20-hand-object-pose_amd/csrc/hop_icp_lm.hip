// refineByICP with the reference's own minimiser (nn_mode 5).
//
// Utils::runICP (src/perception/src/Utils.cpp:200-216) installs pcl::registration::TransformationEstimationPointToPlane, which
// is a TransformationEstimationLM: per ICP iteration PCL 1.9 minimises sum_i ((warp(x) p_i - q_i) . n_i)^2 over
// x = (t, quaternion xyz) from x = 0 with Eigen::LevenbergMarquardt<Eigen::NumericalDiff<Functor>, float>
// (registration/impl/transformation_estimation_lm.hpp:146-197).  The algorithm below follows Eigen's
// unsupported/Eigen/src/NonLinearOptimization/{LevenbergMarquardt.h:208-355 minimizeOneStep, lmpar.h:163-293 lmpar2} and
// NumericalDiff/NumericalDiff.h:64-122 (forward differences, h = sqrt(eps) |x_j| or sqrt(eps)) as vendored in the reference
// (src/OpenGR_4pcs/3rdparty/Eigen/unsupported):
//   * k_icp_lm_pass: one pass over the correspondences of a hypothesis per function evaluation the minimiser asks for.  For
//     the parameter vector the minimiser wants evaluated it forms the residual f_i and the six forward-difference Jacobian
//     entries (f_i(x + h_j e_j) - f_i(x)) / h_j in float with the operation order of PCL's warp_point_rigid_6d.h /
//     transformation_estimation_point_to_plane.h on Eigen's SSE2 packet reductions -- every f_i and J_ij is the float the
//     reference's build evaluates (pinned: oracle/ref_icp_driver.cpp) -- and adds sum f^2, J^T J, J^T f in double;
//   * k_icp_lm_solve: one lane per hypothesis advances the minimiser's state machine on those 28 sums: Eigen's m x 6
//     Householder QR only ever enters through R^T R = P^T J^T J P and Q^T f = R^-T P^T J^T f, so the Gauss-Newton
//     direction, |D p|, |J p|, the scaled gradient and lmpar's secular iteration are evaluated on the 6 x 6 normal equations
//     in double with the same control flow and the same float constants.  When the minimiser stops, the same lane does the
//     ICP bookkeeping of icp.hpp / default_convergence_criteria.hpp (increment history, final = T * final, stop rules).
// No moved copy of the source and no stored correspondences beyond the 4-byte list position: a pass recomputes the moved source
// point from the increment history and the target from the model list entry (L2-resident).
//
// The result equals the CPU restatement (oracle lm_*; tests/test_gpu_icp_lm.py) and differs from Eigen's own float run by the
// rounding of its float reductions only -- measured to be what two builds of the reference differ by (profiles/r03_icp_lm_deltas.json).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "hop_device.h"
#include "hop_math.h"

namespace hop {
// -DHOP_LM_COUNT (tools/build_variant.sh): statistics of the minimiser -- [0] hypotheses solved, [1] evaluations, [2] general lmpar
// calls (out of line), [3] runs ended by maxfev, [4] fast lmpar calls
__device__ unsigned long long g_lm_count[8];
#ifdef HOP_LM_COUNT
#define LM_COUNT(slot, v) atomicAdd(&g_lm_count[slot], (unsigned long long)(v))
#else
#define LM_COUNT(slot, v) do { } while (0)
#endif
void lm_counters_read(unsigned long long* out8, bool reset) {
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lm_count), sizeof(unsigned long long) * 8);
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lm_count), z, sizeof(z));
  }
}
namespace {

__device__ __forceinline__ double lm_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float sum4_sse2(float a0, float a1, float a2, float a3) { return (a0 + a2) + (a1 + a3); }  // Eigen predux<Packet4f>

// WarpPointRigid6D::setParam (warp_point_rigid_6d.h:77-95): Quaternionf(0, x3, x4, x5), w = sqrt(1 - q.dot(q)), normalize, toRotationMatrix
__device__ void lm_warp6(const float x[6], float T[12]) {
  float qx = x[3], qy = x[4], qz = x[5], qw = 0.f;
  const float d = sum4_sse2(qx * qx, qy * qy, qz * qz, qw * qw);
  qw = sqrtf(1 - d);
  const float nn = sqrtf(sum4_sse2(qx * qx, qy * qy, qz * qz, qw * qw));
  qx = qx / nn, qy = qy / nn, qz = qz / nn, qw = qw / nn;
  const float tx = 2.f * qx, ty = 2.f * qy, tz = 2.f * qz;
  const float twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  T[0] = 1.f - (tyy + tzz), T[1] = txy - twz, T[2] = txz + twy, T[3] = x[0];
  T[4] = txy + twz, T[5] = 1.f - (txx + tzz), T[6] = tyz - twx, T[7] = x[1];
  T[8] = txz - twy, T[9] = tyz + twx, T[10] = 1.f - (txx + tyy), T[11] = x[2];
}
// warpPoint + TransformationEstimationPointToPlane::computeDistance
__device__ __forceinline__ float lm_residual(const float* __restrict__ T, V3 p, V3 q, V3 n) {
  const V3 w = m4_point(T, p);
  const float dx = w.x - q.x, dy = w.y - q.y, dz = w.z - q.z;
  return sum4_sse2(dx * n.x, dy * n.y, dz * n.z, 0.f);
}

// the seven warps of a pass: at xc and at xc + h_j e_j (NumericalDiff::df, Forward)
__device__ void lm_prepare_pass(LmDev& s) {
  lm_warp6(s.xc, s.W[0]);
  for (int j = 0; j < 6; ++j) {
    float xx[6];
    for (int k = 0; k < 6; ++k) xx[k] = s.xc[k];
    float h = LM_SQRT_EPS_F * fabsf(s.xc[j]);
    if (h == 0.f) h = LM_SQRT_EPS_F;
    xx[j] += h;
    s.h[j] = h;
    lm_warp6(xx, s.W[1 + j]);
  }
}

// ---- 6 x 6 linear algebra in double ---------------------------------------------------------------------------------------------
struct PivChol {
  double R[6][6];
  int perm[6], rank;
};
// Cholesky with diagonal pivoting: R^T R = P^T A P in the pivot order of a column-pivoted QR of J; rank by ColPivHouseholderQR::rank()
__device__ void piv_chol(const double A[6][6], PivChol& c) {
  double S[6][6];
  for (int i = 0; i < 6; ++i) {
    c.perm[i] = i;
    for (int j = 0; j < 6; ++j) S[i][j] = A[i][j], c.R[i][j] = 0;
  }
  double maxpiv = 0;
  int k = 0;
  for (; k < 6; ++k) {
    int best = k;
    for (int j = k + 1; j < 6; ++j)
      if (S[j][j] > S[best][best]) best = j;
    if (!(S[best][best] > 0)) break;
    if (best != k) {
      for (int i = 0; i < 6; ++i) {
        const double t = S[i][k];
        S[i][k] = S[i][best], S[i][best] = t;
      }
      for (int j = 0; j < 6; ++j) {
        const double t = S[k][j];
        S[k][j] = S[best][j], S[best][j] = t;
      }
      for (int i = 0; i < k; ++i) {
        const double t = c.R[i][k];
        c.R[i][k] = c.R[i][best], c.R[i][best] = t;
      }
      const int t = c.perm[k];
      c.perm[k] = c.perm[best], c.perm[best] = t;
    }
    const double d = sqrt(S[k][k]);
    c.R[k][k] = d;
    maxpiv = fmax(maxpiv, d);
    for (int j = k + 1; j < 6; ++j) c.R[k][j] = S[k][j] / d;
    for (int i = k + 1; i < 6; ++i)
      for (int j = k + 1; j < 6; ++j) S[i][j] -= c.R[k][i] * c.R[k][j];
  }
  c.rank = 0;
  const double thr = maxpiv * 6.0 * (double)FLT_EPSILON;
  for (int i = 0; i < k; ++i)
    if (c.R[i][i] > thr) c.rank++;
    else break;
}
__device__ void piv_chol_solve(const PivChol& c, const double g[6], double x[6]) {
  double y[6] = {0, 0, 0, 0, 0, 0};
  const int r = c.rank;
  for (int i = 0; i < r; ++i) {
    double s = g[c.perm[i]];
    for (int k = 0; k < i; ++k) s -= c.R[k][i] * y[k];
    y[i] = s / c.R[i][i];
  }
  double z[6] = {0, 0, 0, 0, 0, 0};
  for (int i = r - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < r; ++k) s -= c.R[i][k] * z[k];
    z[i] = s / c.R[i][i];
  }
  for (int i = 0; i < 6; ++i) x[c.perm[i]] = z[i];
}
__device__ bool spd_solve6(const double M[6][6], const double b[6], double x[6]) {
  double L[6][6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) L[i][j] = 0.0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = M[i][j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 1e-300)) return false;
        L[i][i] = sqrt(s);
      } else
        L[i][j] = s / L[j][j];
    }
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
    y[i] = s / L[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
  return true;
}
__device__ __forceinline__ double norm6(const double v[6]) {
  double s = 0;
  for (int i = 0; i < 6; ++i) s += v[i] * v[i];
  return sqrt(s);
}
__device__ void unpack_sym(const double a21[21], double A[6][6]) {
  int k = 0;
#pragma unroll
  for (int u = 0; u < 6; ++u)
#pragma unroll
    for (int v = 0; v <= u; ++v) {
      A[u][v] = a21[k];
      A[v][u] = a21[k];
      ++k;
    }
}

// internal::lmpar2 (lmpar.h:163-293) on the normal equations
__device__ __noinline__ void lm_par(const double A[6][6], const double g[6], const double diag[6], double delta, double& par, double x[6]) {
  const double dwarf = (double)FLT_MIN, p1 = (double)0.1f;
  PivChol c;
  piv_chol(A, c);
  piv_chol_solve(c, g, x);
  int iter = 0;
  double wa2[6];
  for (int j = 0; j < 6; ++j) wa2[j] = diag[j] * x[j];
  double dxnorm = norm6(wa2);
  double fp = dxnorm - delta;
  if (fp <= p1 * delta) {
    par = 0;
    return;
  }
  double parl = 0;
  if (c.rank == 6) {
    double w[6], u[6];
    for (int j = 0; j < 6; ++j) w[j] = diag[j] * wa2[j] / dxnorm;
    if (spd_solve6(A, w, u)) {
      double t2 = 0;
      for (int j = 0; j < 6; ++j) t2 += w[j] * u[j];
      const double temp = sqrt(t2);
      parl = fp / delta / temp / temp;
    }
  }
  double wa1[6];
  for (int j = 0; j < 6; ++j) wa1[j] = g[j] / diag[j];
  const double gnorm = norm6(wa1);
  double paru = gnorm / delta;
  if (paru == 0) paru = dwarf / fmin(delta, p1);
  par = fmax(par, parl);
  par = fmin(par, paru);
  if (par == 0) par = gnorm / dxnorm;
  while (true) {
    ++iter;
    if (par == 0) par = fmax(dwarf, (double)0.001f * paru);
    double M[6][6];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) M[i][j] = A[i][j] + (i == j ? par * diag[i] * diag[i] : 0.0);
    spd_solve6(M, g, x);
    for (int j = 0; j < 6; ++j) wa2[j] = diag[j] * x[j];
    dxnorm = norm6(wa2);
    double temp = fp;
    fp = dxnorm - delta;
    if (fabs(fp) <= p1 * delta || (parl == 0 && fp <= temp && temp < 0) || iter == 10) break;
    double w[6], u[6];
    for (int j = 0; j < 6; ++j) w[j] = diag[j] * (wa2[j] / dxnorm);
    spd_solve6(M, w, u);
    double t2 = 0;
    for (int j = 0; j < 6; ++j) t2 += w[j] * u[j];
    temp = sqrt(t2);
    const double parc = fp / delta / temp / temp;
    if (fp > 0) parl = fmax(parl, par);
    if (fp < 0) paru = fmin(paru, par);
    par = fmax(parl, par + parc);
  }
  if (iter == 0) par = 0;
}

// 1 / sqrt(x) to double precision from the hardware estimate and two Newton steps (nn_mode 6 only: its arithmetic is "exact" up to
// 1e-15, no operation order to preserve): the state machine spends most of its instructions in IEEE divisions and square roots
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  y = y * fma(-hx * y, y, 1.5);
  return y;
}
__device__ __forceinline__ double rcp_nr(double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  return r;
}

// nn_mode 6: lmpar2 (lmpar.h:163-293) in registers for a comfortably full-rank Jacobian (unpivoted Cholesky, smallest pivot above 2e-5
// of the largest: ColPivHouseholderQR::rank()'s threshold is 7e-7 of it) -- every index static, no scratch memory, reciprocal
// square roots instead of divisions.  The triangular solves against R become solves against the Cholesky factor of A = J^T J,
// qrsolv's problem [R; sqrt(par) D] the factor of A + par D^2: the same numbers lm_par computes.  false: rank-deficient or
// ill-conditioned -- the caller takes the general, pivoted routine.
__device__ __forceinline__ bool chol6_regs(const double* __restrict__ a21, const double* __restrict__ diag, double par, double* __restrict__ L,
                                           double* __restrict__ Linv) {
  double lmin = 1e300, lmax = 0;
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double v = a21[i * (i + 1) / 2 + j];
      if (i == j) v = fma(par * diag[i], diag[i], v);
#pragma unroll
      for (int k = 0; k < j; ++k) v -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
      if (i == j) {
        ok = ok && v > 1e-290;
        const double r = rsqrt_nr(fmax(v, 1e-290)), d = v * r;
        L[i * (i + 1) / 2 + i] = d;
        Linv[i] = r;
        lmin = fmin(lmin, d), lmax = fmax(lmax, d);
      } else
        L[i * (i + 1) / 2 + j] = v * Linv[j];
    }
  return ok && lmin > 2e-5 * lmax;
}
__device__ __forceinline__ void chol6_fwd(const double* __restrict__ L, const double* __restrict__ Linv, const double* __restrict__ b, double* __restrict__ y) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double v = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= L[i * (i + 1) / 2 + k] * y[k];
    y[i] = v * Linv[i];
  }
}
__device__ __forceinline__ void chol6_bwd(const double* __restrict__ L, const double* __restrict__ Linv, const double* __restrict__ y, double* __restrict__ x) {
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double v = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) v -= L[k * (k + 1) / 2 + i] * x[k];
    x[i] = v * Linv[i];
  }
}
__device__ __forceinline__ double dnorm6(const double* __restrict__ diag, const double* __restrict__ x, double* __restrict__ wa2) {
  double q = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    wa2[j] = diag[j] * x[j];
    q = fma(wa2[j], wa2[j], q);
  }
  return q > 1e-290 ? q * rsqrt_nr(q) : 0.0;
}
__device__ __forceinline__ bool lm_par_fast(const double* __restrict__ a21, const double* __restrict__ g, const double* __restrict__ diag, double delta,
                                            double& par_io, double* __restrict__ x) {
  const double dwarf = (double)FLT_MIN, p1 = (double)0.1f;
  double L[21], Linv[6], y[6], xs[6], wa2[6], w[6];
  if (!chol6_regs(a21, diag, 0.0, L, Linv)) return false;
  chol6_fwd(L, Linv, g, y);
  chol6_bwd(L, Linv, y, xs);
  double dxnorm = dnorm6(diag, xs, wa2);
  double fp = dxnorm - delta;
  if (fp <= p1 * delta) {
#pragma unroll
    for (int j = 0; j < 6; ++j) x[j] = xs[j];
    par_io = 0;
    LM_COUNT(4, 1);
    return true;
  }
  LM_COUNT(5, 1);
  const double dinv = rcp_nr(delta);
  // parl = fp / delta / |R^-T D (D x) / |D x||^2 (the Jacobian has full rank here)
  {
    const double ninv = rcp_nr(dxnorm);
#pragma unroll
    for (int j = 0; j < 6; ++j) w[j] = diag[j] * wa2[j] * ninv;
    chol6_fwd(L, Linv, w, y);
  }
  double t2 = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) t2 = fma(y[j], y[j], t2);
  double parl = fp * dinv * rcp_nr(t2);
  double gq = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double v = g[j] * rcp_nr(diag[j]);
    gq = fma(v, v, gq);
  }
  const double gnorm = gq > 1e-290 ? gq * rsqrt_nr(gq) : 0.0;
  double paru = gnorm * dinv;
  if (paru == 0) paru = dwarf / fmin(delta, p1);
  double par = fmin(fmax(par_io, parl), paru);
  if (par == 0) par = gnorm * rcp_nr(dxnorm);
  for (int iter = 1;; ++iter) {
    if (par == 0) par = fmax(dwarf, (double)0.001f * paru);
    if (!chol6_regs(a21, diag, par, L, Linv)) return false;  // (A + par D^2 is better conditioned than A: does not happen)
    chol6_fwd(L, Linv, g, y);
    chol6_bwd(L, Linv, y, xs);
    dxnorm = dnorm6(diag, xs, wa2);
    const double temp = fp;
    fp = dxnorm - delta;
    if (fabs(fp) <= p1 * delta || (parl == 0 && fp <= temp && temp < 0) || iter == 10) break;
    const double ninv = rcp_nr(dxnorm);
#pragma unroll
    for (int j = 0; j < 6; ++j) w[j] = diag[j] * (wa2[j] * ninv);
    chol6_fwd(L, Linv, w, y);
    double t3 = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) t3 = fma(y[j], y[j], t3);
    const double parc = fp * dinv * rcp_nr(t3);
    if (fp > 0) parl = fmax(parl, par);
    if (fp < 0) paru = fmin(paru, par);
    par = fmax(parl, par + parc);
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) x[j] = xs[j];
  par_io = par;
  return true;
}

// ---- LevenbergMarquardt::minimize as a state machine around the passes ------------------------------------------------------------
__device__ __forceinline__ double lm_scaled_norm(const double diag[6], const float v[6]) {
  double q = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) q += (diag[j] * (double)v[j]) * (diag[j] * (double)v[j]);
  return sqrt(q);
}
// do { lmpar; candidate } of minimizeOneStep (LevenbergMarquardt.h:262-275)
template <class LS>
__device__ void lm_inner(LS& s) {
  double xs[6];
  if (!(LS::fast_lmpar && lm_par_fast(s.A, s.g, s.diag, s.delta, s.par, xs))) {
    // (copies: the out-of-line call must not expose the state struct's address, or all of it lives in scratch memory)
    double A[6][6], gg[6], dd[6], xo[6], par = s.par;
    LM_COUNT(2, 1);
    unpack_sym(s.A, A);
#pragma unroll
    for (int j = 0; j < 6; ++j) gg[j] = s.g[j], dd[j] = s.diag[j];
    lm_par(A, gg, dd, s.delta, par, xo);
    s.par = par;
#pragma unroll
    for (int j = 0; j < 6; ++j) xs[j] = xo[j];
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    s.p[j] = -(float)xs[j];
    s.xc[j] = s.x[j] + s.p[j];
  }
  s.pnorm = lm_scaled_norm(s.diag, s.p);
  if (s.iter == 1) s.delta = fmin(s.delta, s.pnorm);
  s.phase = 1;
}
// head of minimizeOneStep (LevenbergMarquardt.h:219-260); false = finished
template <class LS>
__device__ bool lm_outer(LS& s) {
  s.nfev += 7;
  double wa2[6], wa2inv[6];
#pragma unroll
  for (int j = 0, k = 0; j < 6; ++j) {
    k += j;  // index of the diagonal element (j, j) in the packed lower triangle: j (j + 1) / 2 + j
    if (LS::fast_lmpar) {
      const double a = s.A[k + j], r = a > 1e-290 ? rsqrt_nr(a) : 0.0;
      wa2[j] = a * r, wa2inv[j] = r;
    } else
      wa2[j] = sqrt(s.A[k + j]);
  }
  if (s.iter == 1) {
#pragma unroll
    for (int j = 0; j < 6; ++j) s.diag[j] = wa2[j] == 0 ? 1.0 : wa2[j];
    s.xnorm = lm_scaled_norm(s.diag, s.x);
    s.delta = 100.0 * s.xnorm;
    if (s.delta == 0) s.delta = 100.0;
  }
  s.gnorm = 0;
  if (s.fnorm != 0) {
    if (LS::fast_lmpar) {
      const double finv = rcp_nr(s.fnorm);
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (wa2[j] != 0) s.gnorm = fmax(s.gnorm, fabs(s.g[j] * finv * wa2inv[j]));
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (wa2[j] != 0) s.gnorm = fmax(s.gnorm, fabs(s.g[j] / s.fnorm / wa2[j]));
    }
  }
  if (s.gnorm <= 0) {
    s.status = 4;  // CosinusTooSmall
    return false;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) s.diag[j] = fmax(s.diag[j], wa2[j]);
  lm_inner(s);
  return true;
}
// consumes the sums of the pass at s.xc (cand: 21 + 6 + 1); true = another pass at the new s.xc
template <class LS>
__device__ bool lm_advance(LS& s, const double* cand) {
  if (s.phase == 0) {  // minimizeInit
#pragma unroll
    for (int k = 0; k < 21; ++k) s.A[k] = cand[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) s.g[k] = cand[21 + k];
    s.ff = cand[27];
    s.nfev = 1;
    s.fnorm = sqrt(s.ff);
    s.par = 0;
    s.iter = 1;
    return lm_outer(s);
  }
  const double ftol = (double)LM_SQRT_EPS_F, xtol = (double)LM_SQRT_EPS_F, eps = (double)FLT_EPSILON;
  const double p1 = (double)0.1f, p25 = 0.25, p5 = 0.5, p75 = 0.75, p0001 = (double)1e-4f;
  ++s.nfev;
  double fnorm1, actred = -1, temp1, temp2;
  double A[6][6];
  unpack_sym(s.A, A);
  double jp2 = 0;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) jp2 += (double)s.p[a] * A[a][b] * (double)s.p[b];
  if (LS::fast_lmpar) {
    // the same quantities without the square roots that are squared again: (|J p| / |f|)^2 = p^T A p / |f|^2
    const double c27 = cand[27];
    fnorm1 = c27 > 1e-290 ? c27 * rsqrt_nr(c27) : 0.0;
    const double finv = rcp_nr(s.fnorm), r1 = fnorm1 * finv;
    if (p1 * fnorm1 < s.fnorm) actred = 1.0 - r1 * r1;
    temp1 = fmax(jp2, 0.0) * finv * finv;
    temp2 = s.par * (s.pnorm * finv) * (s.pnorm * finv);
  } else {
    fnorm1 = sqrt(cand[27]);
    if (p1 * fnorm1 < s.fnorm) actred = 1.0 - (fnorm1 / s.fnorm) * (fnorm1 / s.fnorm);
    const double t1r = sqrt(fmax(jp2, 0.0)) / s.fnorm, t2r = sqrt(s.par) * s.pnorm / s.fnorm;
    temp1 = t1r * t1r, temp2 = t2r * t2r;
  }
  const double prered = temp1 + temp2 / p5, dirder = -(temp1 + temp2);
  double ratio = 0;
  if (prered != 0) ratio = LS::fast_lmpar ? actred * rcp_nr(prered) : actred / prered;
  if (ratio <= p25) {
    double temp = p5;
    if (actred < 0) temp = p5 * dirder / (dirder + p5 * actred);
    if (p1 * fnorm1 >= s.fnorm || temp < p1) temp = p1;
    s.delta = temp * fmin(s.delta, s.pnorm / p1);
    s.par /= temp;
  } else if (!(s.par != 0 && ratio < p75)) {
    s.delta = s.pnorm / p5;
    s.par = p5 * s.par;
  }
  if (ratio >= p0001) {
#pragma unroll
    for (int j = 0; j < 6; ++j) s.x[j] = s.xc[j];
#pragma unroll
    for (int k = 0; k < 21; ++k) s.A[k] = cand[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) s.g[k] = cand[21 + k];
    s.ff = cand[27];
    s.xnorm = lm_scaled_norm(s.diag, s.x);
    s.fnorm = fnorm1;
    ++s.iter;
  }
  const bool small_red = fabs(actred) <= ftol && prered <= ftol && p5 * ratio <= 1.0;
  const bool small_err = s.delta <= xtol * s.xnorm;
  if (small_red && small_err) s.status = 3;
  else if (small_red) s.status = 1;
  else if (small_err) s.status = 2;
  else if (s.nfev >= 400) s.status = 5;
  else if (fabs(actred) <= eps && prered <= eps && p5 * ratio <= 1.0) s.status = 6;
  else if (s.delta <= eps * s.xnorm) s.status = 7;
  else if (s.gnorm <= eps) s.status = 8;
  if (s.status != -1) return false;
  if (ratio < p0001) {
    lm_inner(s);
    return true;
  }
  return lm_outer(s);
}

// after estimateRigidTransformation (st.T_inc holds transformation_): transformCloud / final_transformation_ = transformation_ *
// final_transformation_ / ++nr_iterations_ (icp.hpp) and DefaultConvergenceCriteria::hasConverged with the thresholds ICP installs
__device__ void icp_iteration_bookkeeping(const IcpArgs& a, IcpState& st, int hl, double mse_sum, int cnt) {
  const float* T = st.T_inc;
  for (int i = 0; i < 12; ++i) a.hist[((size_t)hl * a.max_iter + st.iterations) * 12 + i] = T[i];
  M4 Tm = m4_identity(), F;
  for (int i = 0; i < 12; ++i) Tm.m[i] = T[i];
  for (int i = 0; i < 16; ++i) F.m[i] = st.final_tf[i];
  F = m4_mul(Tm, F);
  for (int i = 0; i < 16; ++i) st.final_tf[i] = F.m[i];
  st.iterations += 1;
  const double mse = mse_sum / (double)cnt;
  bool stop = false;
  if (st.iterations >= a.max_iter) stop = true;
  else {
    // criterion 2 with ICP's thresholds (rotation 1.0 - transformation_epsilon_ = 1, translation 0): only an identity increment
    const double cos_angle = 0.5 * (double)(T[0] + T[5] + T[10] - 1);
    const double translation_sqr = (double)(T[3] * T[3] + T[7] * T[7] + T[11] * T[11]);
    if (cos_angle >= 1.0 && translation_sqr <= 0.0) stop = true;
    else if (fabs(mse - st.mse_prev) < 1e-6) stop = true;  // Utils.cpp:208; the relative criterion is overwritten by ICP's default (never fires)
  }
  st.mse_prev = mse;
  if (stop) {
    st.active = 0;
    st.converged = 1;
  }
}

// ---- nn_mode 6: every function evaluation of the minimiser from the 13 x 13 moment matrix of the correspondences --------------------------
// w(x) = (R(x) - I [9], t + (R(x) - I) c [3], 1) in double; R from the quaternion as WarpPointRigid6D::setParam forms it
__device__ void lm6_w(const float x[6], const double c[3], double w[13]) {
  const double qx = (double)x[3], qy = (double)x[4], qz = (double)x[5];
  const double qw2 = 1.0 - (qx * qx + qy * qy + qz * qz);  // (the quaternion's norm is 1 in exact arithmetic: normalize() is the identity)
  const double qw = qw2 > 1e-290 ? qw2 * rsqrt_nr(qw2) : sqrt(qw2);
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  w[0] = -(tyy + tzz), w[1] = txy - twz, w[2] = txz + twy;
  w[3] = txy + twz, w[4] = -(txx + tzz), w[5] = tyz - twx;
  w[6] = txz - twy, w[7] = tyz + twx, w[8] = -(txx + tyy);
#pragma unroll
  for (int a = 0; a < 3; ++a) w[9 + a] = (double)x[a] + (w[3 * a] * c[0] + w[3 * a + 1] * c[1] + w[3 * a + 2] * c[2]);
  w[12] = 1.0;
}
// Sums of the pass at xc from the moment matrix (one lane per hypothesis, M in LDS as sym-packed [91][64] doubles):
// cand = { J^T J packed lower (21), J^T f (6), |f|^2 } with NumericalDiff's forward differences.
// d_j = (w(xc + h_j e_j) - w(xc)) / h_j is the Jacobian column as a functional on u.  For the translation parameters it is
// s_j e_{9+j} (s_j = the float step actually taken / h_j), so only the three rotation columns and w itself need a product with M.
__device__ __forceinline__ int sym13(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
__device__ void lm6_eval(const double* __restrict__ Mlds /* + lane, stride 64 */, const double c[3], const float xc[6], double* __restrict__ cand) {
  double w0[13], d[3][12], st[3];
  lm6_w(xc, c, w0);
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float xx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) xx[k] = xc[k];
    float h = LM_SQRT_EPS_F * fabsf(xc[j]);
    if (h == 0.f) h = LM_SQRT_EPS_F;
    xx[j] += h;  // the float sum the reference forms; J = (f(xx) - f(x)) / h with the nominal h
    const double hinv = 1.0 / (double)h;
    if (j < 3) st[j] = ((double)xx[j] - (double)xc[j]) * hinv;
    else {
      double wj[13];
      lm6_w(xx, c, wj);
#pragma unroll
      for (int k = 0; k < 12; ++k) d[j - 3][k] = (wj[k] - w0[k]) * hinv;
    }
  }
  double ff = 0, gr[3] = {0, 0, 0}, Arr[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, gt[3], Atr[3][3], Att[6];
#pragma unroll
  for (int i = 0; i < 13; ++i) {
    double row[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) row[k] = Mlds[sym13(i, k) * 64];
    double y0 = 0, z[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 13; ++k) y0 += row[k] * w0[k];
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int k = 0; k < 12; ++k) z[v] += row[k] * d[v][k];
    ff += w0[i] * y0;
    if (i < 12) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        gr[u] += d[u][i] * y0;
#pragma unroll
        for (int v = 0; v <= u; ++v) Arr[u][v] += d[u][i] * z[v];
      }
    }
    if (i >= 9 && i < 12) {
      const int t = i - 9;
      gt[t] = st[t] * y0;
#pragma unroll
      for (int v = 0; v < 3; ++v) Atr[t][v] = st[t] * z[v];  // (J^T J)[3 + v][t]
#pragma unroll
      for (int t2 = 0; t2 <= t; ++t2) Att[t * (t + 1) / 2 + t2] = st[t] * st[t2] * row[9 + t2];
    }
  }
  // parameter order (tx, ty, tz, qx, qy, qz): packed lower triangle of J^T J, then J^T f, then |f|^2
  int k = 0;
#pragma unroll
  for (int u = 0; u < 6; ++u)
#pragma unroll
    for (int v = 0; v <= u; ++v) {
      double a;
      if (u < 3) a = Att[u * (u + 1) / 2 + v];
      else if (v < 3) a = Atr[v][u - 3];
      else a = Arr[u - 3][v - 3];
      cand[k++] = a;
    }
#pragma unroll
  for (int u = 0; u < 6; ++u) cand[21 + u] = u < 3 ? gt[u] : gr[u - 3];
  cand[27] = fmax(ff, 0.0);
}

}  // namespace

// nn_mode 6: one lane per hypothesis -- the whole minimisation of an ICP iteration from the moment sums of k_icp_fusedq_mom.
// (The minimiser's state machine is a chain of ~2 000 dependent double operations per evaluation and the slowest of the 64
// hypotheses of a wavefront sets its time; a few hundred wavefronts, which other frames' kernels overlap.)
__global__ __launch_bounds__(64) void k_icp_lm6_solve(IcpArgs a, int hb, int nblocks) {
  __shared__ double Msh[91 * 64];
  const int lane = threadIdx.x;
  const int hl = blockIdx.x * 64 + lane;
  if (hl >= hb) return;
  IcpState& st = a.state[hl];
  if (!st.active) return;
  double S[ICP_NMOM + 1];
#pragma unroll
  for (int k = 0; k <= ICP_NMOM; ++k) S[k] = 0.0;
  for (int blk = 0; blk < nblocks; ++blk) {
    const double* __restrict__ pp = a.partial + ((size_t)hl * nblocks + blk) * ICP_NMOM_STRIDE;
#pragma unroll
    for (int k = 0; k <= ICP_NMOM; ++k) S[k] += pp[k];
  }
  const int cnt = (int)S[ICP_NMOM];
  if (cnt < 3) {
    st.active = 0, st.converged = 0;
    return;
  }
  if (cnt >= 4) {
    // the 13 x 13 moment matrix of u = (n_a p_b [9], n_a [3], r0), lower triangle packed, this lane's column of the LDS table
    double* __restrict__ M = Msh + lane;
#pragma unroll
    for (int i = 0; i < 13; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        // index of the unordered pair (a, c) among (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
        auto sym = [](int a_, int c_) { const int lo = a_ < c_ ? a_ : c_, hi = a_ < c_ ? c_ : a_; return lo * 3 - lo * (lo - 1) / 2 + (hi - lo); };
        double v;
        if (i < 9 && j < 9) v = S[sym(i / 3, j / 3) * 6 + sym(i % 3, j % 3)];
        else if (i < 12 && j < 9) v = S[36 + sym(i - 9, j / 3) * 3 + j % 3];
        else if (i < 12) v = S[54 + sym(i - 9, j - 9)];
        else if (j < 9) v = S[60 + j];
        else if (j < 12) v = S[69 + (j - 9)];
        else v = S[72];
        M[(i * (i + 1) / 2 + j) * 64] = v;
      }
    const float* pose = a.pose + (size_t)(a.h0 + hl) * 16;
    const double c[3] = {(double)pose[3], (double)pose[7], (double)pose[11]};
    LmDev6 s;
#pragma unroll
    for (int j = 0; j < 6; ++j) s.x[j] = s.xc[j] = 0.f, s.p[j] = 0.f;
    s.phase = 0, s.status = -1, s.iter = 0, s.nfev = 0;
    s.par = s.delta = s.xnorm = s.fnorm = s.gnorm = s.pnorm = 0.0;
    double cand[28];
#ifndef HOP_LM6_MAX_EVAL
#define HOP_LM6_MAX_EVAL 420  // (maxfev = 400 ends every run; smaller values: timing experiments only)
#endif
    LM_COUNT(0, 1);
    for (int guard = 0; guard < HOP_LM6_MAX_EVAL; ++guard) {
      lm6_eval(M, c, s.xc, cand);
      LM_COUNT(1, 1);
      if (!lm_advance(s, cand)) break;
    }
    if (s.status == 5) LM_COUNT(3, 1);
    lm_warp6(s.x, st.T_inc);
  }
  icp_iteration_bookkeeping(a, st, hl, S[73], cnt);
}
void launch_icp_lm6_solve(const IcpArgs& a, int hb, int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_lm6_solve, dim3((hb + 63) / 64), dim3(64), 0, s, a, hb, nblocks);
}

// One pass over the correspondences of every hypothesis whose minimiser waits for an evaluation.
// FIRST: the pass that follows the correspondence search of an ICP iteration -- applies PCL's surface-normal rejector (strict >
// against the double threshold, left-to-right dot: correspondence_rejection_surface_normal.h), drops rejected correspondences
// from the list for the later passes, and adds the squared distances for the MSE criterion.
template <bool FIRST>
__global__ __launch_bounds__(256) void k_icp_lm_pass(IcpArgs a, int R) {
  __shared__ double red[4][ICP_NACC];
  const int hl = blockIdx.y, h = a.h0 + hl;
  const IcpState& st = a.state[hl];
  if (!st.active) return;
  const LmDev& lm = a.lm[hl];
  if (!FIRST && !lm.waiting) return;
  const float* __restrict__ pose = a.pose + (size_t)h * 16;
  const float* __restrict__ hist = a.hist + (size_t)hl * a.max_iter * 12;
  const float* __restrict__ W = &lm.W[0][0];
  float hj[6], hinv_unused = 0.f;
  (void)hinv_unused;
#pragma unroll
  for (int j = 0; j < 6; ++j) hj[j] = lm.h[j];
  double acc[ICP_NACC];
#pragma unroll
  for (int k = 0; k < ICP_NACC; ++k) acc[k] = 0.0;
  for (int r = 0; r < R; ++r) {
    const int i = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
    if (i >= a.ns) continue;
    const size_t ci = (size_t)hl * a.ns + i;
    const int pos = a.corr_idx[ci];
    if (pos < 0) continue;
    V3 q = v3(a.sx[i], a.sy[i], a.sz[i]);
    const float4 tp = a.cells.pts[pos], tn = a.cells.nrm[pos];
    const V3 nt = m4_dir(pose, v3(tn.x, tn.y, tn.z));
    const V3 tq = m4_point(pose, v3(tp.x, tp.y, tp.z));
    if (FIRST) {
      V3 qn = v3(a.snx[i], a.sny[i], a.snz[i]);
      for (int k = 0; k < a.iter; ++k) {
        q = m4_point(hist + 12 * k, q);
        qn = m4_dir(hist + 12 * k, qn);
      }
      if (!(((qn.x * nt.x + qn.y * nt.y) + qn.z * nt.z) > a.cos_thr)) {
        a.corr_idx[ci] = -1;
        continue;
      }
      acc[28] += (double)sqdist_flann(q, tq);
      acc[29] += 1.0;
    } else {
      for (int k = 0; k < a.iter; ++k) q = m4_point(hist + 12 * k, q);
    }
    const float f0 = lm_residual(W, q, tq, nt);
    double J[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float fj = lm_residual(W + 12 * (1 + j), q, tq, nt);
      J[j] = (double)((fj - f0) / hj[j]);
    }
    const double fd = (double)f0;
    int k = 0;
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
      for (int v = 0; v <= u; ++v) {
        acc[k] = fma(J[u], J[v], acc[k]);  // float-valued factors: the product is exact in double
        ++k;
      }
#pragma unroll
    for (int u = 0; u < 6; ++u) acc[21 + u] = fma(J[u], fd, acc[21 + u]);
    acc[27] = fma(fd, fd, acc[27]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 30; ++k) {
    const double s = lm_wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 30) {
    const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    a.partial[((size_t)hl * gridDim.x + blockIdx.x) * ICP_NACC + threadIdx.x] = s;
  }
}

// One lane per hypothesis: adds the block partials, advances the minimiser; when it stops, the ICP iteration's bookkeeping
// (icp.hpp computeTransformation loop body after estimateRigidTransformation; DefaultConvergenceCriteria::hasConverged).
// n_waiting counts the hypotheses that want another pass.
__global__ __launch_bounds__(64) void k_icp_lm_solve(IcpArgs a, int hb, int nblocks, int first, unsigned* __restrict__ n_waiting) {
  const int hl = blockIdx.x * blockDim.x + threadIdx.x;
  if (hl >= hb) return;
  IcpState& st = a.state[hl];
  if (!st.active) return;
  LmDev& s = a.lm[hl];
  if (!first && !s.waiting) return;
  double acc[30];
  for (int k = 0; k < 30; ++k) acc[k] = 0.0;
  for (int blk = 0; blk < nblocks; ++blk)
    for (int k = 0; k < 30; ++k) acc[k] += a.partial[((size_t)hl * nblocks + blk) * ICP_NACC + k];
  bool estimate = true;
  if (first) {
    s.cnt = (int)acc[29];
    s.mse_sum = acc[28];
    if (s.cnt < 3) {  // icp.hpp: not enough correspondences -> not converged (Utils.cpp:218-225 substitutes identity)
      st.active = 0;
      st.converged = 0;
      s.waiting = 0;
      return;
    }
    estimate = s.cnt >= 4;  // transformation_estimation_lm.hpp:158-164: fewer than 4 -> error message, the matrix keeps its last value
  }
  if (estimate) {
    if (lm_advance(s, acc)) {
      lm_prepare_pass(s);
      s.waiting = 1;
      atomicAdd(n_waiting, 1u);
      return;
    }
    lm_warp6(s.x, st.T_inc);  // warp_point_->setParam(x); transformation_matrix = warp_point_->getTransform()
  }
  s.waiting = 0;
  icp_iteration_bookkeeping(a, st, hl, s.mse_sum, s.cnt);
}

// start of an ICP iteration: every active hypothesis' minimiser at x = 0 (transformation_estimation_lm.hpp:166-168)
__global__ void k_icp_lm_begin(IcpArgs a, int hb) {
  const int hl = blockIdx.x * blockDim.x + threadIdx.x;
  if (hl >= hb) return;
  LmDev& s = a.lm[hl];
  for (int j = 0; j < 6; ++j) s.x[j] = s.xc[j] = 0.f, s.p[j] = 0.f;
  s.phase = 0, s.status = -1, s.iter = 0, s.nfev = 0;
  s.par = s.delta = s.xnorm = s.fnorm = s.gnorm = s.pnorm = 0.0;
  s.waiting = a.state[hl].active ? 1 : 0;
  lm_prepare_pass(s);
}

void launch_icp_lm_begin(const IcpArgs& a, int hb, hipStream_t s) { hipLaunchKernelGGL(k_icp_lm_begin, dim3((hb + 63) / 64), dim3(64), 0, s, a, hb); }
void launch_icp_lm_pass(const IcpArgs& a, int hb, bool first, hipStream_t s) {
  const int nb = icp_blocks_per_hyp(a.ns, true);
  const int R = (a.ns + 256 * nb - 1) / (256 * nb);
  if (first) hipLaunchKernelGGL((k_icp_lm_pass<true>), dim3(nb, hb), dim3(256), 0, s, a, R);
  else hipLaunchKernelGGL((k_icp_lm_pass<false>), dim3(nb, hb), dim3(256), 0, s, a, R);
}
void launch_icp_lm_solve(const IcpArgs& a, int hb, int nblocks, bool first, unsigned* n_waiting, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_lm_solve, dim3((hb + 63) / 64), dim3(64), 0, s, a, hb, nblocks, first ? 1 : 0, n_waiting);
}

}  // namespace hop
