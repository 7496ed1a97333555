// hop_lm_core.h -- Eigen::LevenbergMarquardt<NumericalDiff<Functor>, float>::minimize as the GPU runs it: the 6 x 6 algebra in double,
// lmpar2, the minimizeOneStep state machine and the moment-form function evaluation (nn_mode 6 / 7).  Reference: the vendored
// unsupported/Eigen/src/NonLinearOptimization/{LevenbergMarquardt.h:208-355, lmpar.h:163-293}, NumericalDiff/NumericalDiff.h:64-122
// and PCL 1.9's warp_point_rigid_6d.h / transformation_estimation_lm.hpp (see hop_icp_lm.hip).
//
// Plain scalar code with no HIP dependency, so that the SAME text that the kernels run is also compiled by g++ and checked on the CPU
// against the oracle's independent statement of the algorithm (tests/cpp/lm_core_host.cpp, tests/test_lm_core_cpu.py): a kernel
// that cannot be run where it is written is at least arithmetic-checked there.
// Included INSIDE the includer's namespace (hop_icp_lm.hip: hop::<anonymous>); the includer provides <cmath>, <cfloat>, LM_SQRT_EPS_F
// and, optionally, LM_COUNT(slot, v).
#ifndef HOP_LM_CORE_H_
#define HOP_LM_CORE_H_

#if defined(__HIPCC__)
#define HOP_LM_DEV __device__
#define HOP_LM_INL __device__ __forceinline__
#define HOP_LM_NOINL __device__ __noinline__
#else
#define HOP_LM_DEV inline
#define HOP_LM_INL inline
#define HOP_LM_NOINL inline
#endif
#ifndef LM_COUNT
#define LM_COUNT(slot, v) do { } while (0)
#endif

// (std::max) / (std::min) as Eigen's LevenbergMarquardt.h and lmpar.h call them -- NOT fmax / fmin: when a trial step leaves the unit ball
// of the quaternion its residuals are NaN, and what the minimiser does next (Eigen: the comparison with NaN is false, the step is refused
// and the trust region shrinks) depends on which operand a maximum with NaN returns.  fmax would return the number: round 3's
// `fmax(ff, 0)` turned a NaN sum of squares into 0, "a perfect step", and nn_mode 6 accepted it (found in round 4 by running this text on
// the CPU model against the oracle on the C1 frame, hypothesis 78).
HOP_LM_INL double lm_max(double a, double b) { return (a < b) ? b : a; }
HOP_LM_INL double lm_min(double a, double b) { return (b < a) ? b : a; }

// state of the minimiser for one hypothesis, register-resident (nn_mode 6 / 7: no per-pass warp tables)
struct LmDev6 {
  float x[6], xc[6], p[6];
  double A[21], g[6], ff;
  double diag[6], delta, par, xnorm, fnorm, gnorm, pnorm;
  int iter, nfev, status, phase;
  static constexpr bool fast_lmpar = true;   // lmpar2's common case in registers with hardware reciprocal estimates (lm_par_regs<true>)
  static constexpr bool regs_lmpar = true;
};
// nn_mode 7: the same machine with IEEE operations only (/, sqrt, fma as written): every double it holds is the double the oracle's
// statement of this algorithm holds (oracle/hop_oracle.cpp lm_*_canon)
struct LmDev7 : LmDev6 {
  static constexpr bool fast_lmpar = false;
  static constexpr bool regs_lmpar = true;   // lmpar2's common case in registers, IEEE (lm_par_regs<false>)
};

// ---- 6 x 6 linear algebra in double ---------------------------------------------------------------------------------------------
struct PivChol {
  double R[6][6];
  int perm[6], rank;
};
// Cholesky with diagonal pivoting: R^T R = P^T A P in the pivot order of a column-pivoted QR of J; rank by ColPivHouseholderQR::rank()
HOP_LM_DEV void piv_chol(const double A[6][6], PivChol& c) {
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
    maxpiv = lm_max(maxpiv, d);
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
HOP_LM_DEV void piv_chol_solve(const PivChol& c, const double g[6], double x[6]) {
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
HOP_LM_DEV bool spd_solve6(const double M[6][6], const double b[6], double x[6]) {
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
HOP_LM_INL double norm6(const double v[6]) {
  double s = 0;
  for (int i = 0; i < 6; ++i) s += v[i] * v[i];
  return sqrt(s);
}
HOP_LM_DEV void unpack_sym(const double a21[21], double A[6][6]) {
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
HOP_LM_NOINL void lm_par(const double A[6][6], const double g[6], const double diag[6], double delta, double& par, double x[6]) {
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
  if (paru == 0) paru = dwarf / lm_min(delta, p1);
  par = lm_max(par, parl);
  par = lm_min(par, paru);
  if (par == 0) par = gnorm / dxnorm;
  while (true) {
    ++iter;
    if (par == 0) par = lm_max(dwarf, (double)0.001f * paru);
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
    if (fp > 0) parl = lm_max(parl, par);
    if (fp < 0) paru = lm_min(paru, par);
    par = lm_max(parl, par + parc);
  }
  if (iter == 0) par = 0;
}

// 1 / sqrt(x) to double precision from the hardware estimate and two Newton steps (nn_mode 6 only: its arithmetic is "exact" up to
// 1e-15, no operation order to preserve): the state machine spends most of its instructions in IEEE divisions and square roots
#if defined(__HIPCC__)
HOP_LM_INL double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  y = y * fma(-hx * y, y, 1.5);
  return y;
}
HOP_LM_INL double rcp_nr(double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  return r;
}
#else  // (host build of this header: the estimate-based forms exist on the device only; the IEEE forms are what the host checks)
HOP_LM_INL double rsqrt_nr(double x) { return 1.0 / sqrt(x); }
HOP_LM_INL double rcp_nr(double b) { return 1.0 / b; }
#endif
// The two arithmetic flavours of the register-resident routines: FAST (nn_mode 6) takes reciprocals and reciprocal square roots from
// the hardware estimates; the other (nn_mode 7) uses IEEE /, sqrt and fma only, each a correctly rounded operation that the oracle's
// statement repeats in the same order -- same doubles on the CPU and on the GPU.
template <bool FAST>
struct LmOps {
  static HOP_LM_INL double rcp(double x) { return FAST ? rcp_nr(x) : 1.0 / x; }
  // d = sqrt(v), r = 1 / d for v > 0
  static HOP_LM_INL void root(double v, double& d, double& r) {
    if (FAST) {
      r = rsqrt_nr(v);
      d = v * r;
    } else {
      d = sqrt(v);
      r = 1.0 / d;
    }
  }
  static HOP_LM_INL double sqrt_pos(double q) { return FAST ? q * rsqrt_nr(q) : sqrt(q); }  // q > 1e-290
};

// nn_mode 6 / 7: lmpar2 (lmpar.h:163-293) in registers for a comfortably full-rank Jacobian (unpivoted Cholesky, smallest pivot above 2e-5
// of the largest: ColPivHouseholderQR::rank()'s threshold is 7e-7 of it) -- every index static, no scratch memory, reciprocal
// square roots instead of divisions.  The triangular solves against R become solves against the Cholesky factor of A = J^T J,
// qrsolv's problem [R; sqrt(par) D] the factor of A + par D^2: the same numbers lm_par computes.  false: rank-deficient or
// ill-conditioned -- the caller takes the general, pivoted routine.
template <bool FAST>
HOP_LM_INL bool chol6_regs(const double* __restrict__ a21, const double* __restrict__ diag, double par, double* __restrict__ L,
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
        double d, r;
        LmOps<FAST>::root(lm_max(v, 1e-290), d, r);
        L[i * (i + 1) / 2 + i] = d;
        Linv[i] = r;
        lmin = lm_min(lmin, d), lmax = lm_max(lmax, d);
      } else
        L[i * (i + 1) / 2 + j] = v * Linv[j];
    }
  return ok && lmin > 2e-5 * lmax;
}
HOP_LM_INL void chol6_fwd(const double* __restrict__ L, const double* __restrict__ Linv, const double* __restrict__ b, double* __restrict__ y) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double v = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= L[i * (i + 1) / 2 + k] * y[k];
    y[i] = v * Linv[i];
  }
}
HOP_LM_INL void chol6_bwd(const double* __restrict__ L, const double* __restrict__ Linv, const double* __restrict__ y, double* __restrict__ x) {
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double v = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) v -= L[k * (k + 1) / 2 + i] * x[k];
    x[i] = v * Linv[i];
  }
}
template <bool FAST>
HOP_LM_INL double dnorm6(const double* __restrict__ diag, const double* __restrict__ x, double* __restrict__ wa2) {
  double q = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    wa2[j] = diag[j] * x[j];
    q = fma(wa2[j], wa2[j], q);
  }
  return q > 1e-290 ? LmOps<FAST>::sqrt_pos(q) : 0.0;
}
template <bool FAST>
HOP_LM_INL bool lm_par_regs(const double* __restrict__ a21, const double* __restrict__ g, const double* __restrict__ diag, double delta,
                                            double& par_io, double* __restrict__ x) {
  const double dwarf = (double)FLT_MIN, p1 = (double)0.1f;
  double L[21], Linv[6], y[6], xs[6], wa2[6], w[6];
  if (!chol6_regs<FAST>(a21, diag, 0.0, L, Linv)) return false;
  chol6_fwd(L, Linv, g, y);
  chol6_bwd(L, Linv, y, xs);
  double dxnorm = dnorm6<FAST>(diag, xs, wa2);
  double fp = dxnorm - delta;
  if (fp <= p1 * delta) {
#pragma unroll
    for (int j = 0; j < 6; ++j) x[j] = xs[j];
    par_io = 0;
    LM_COUNT(4, 1);
    return true;
  }
  LM_COUNT(5, 1);
  const double dinv = LmOps<FAST>::rcp(delta);
  // parl = fp / delta / |R^-T D (D x) / |D x||^2 (the Jacobian has full rank here)
  {
    const double ninv = LmOps<FAST>::rcp(dxnorm);
#pragma unroll
    for (int j = 0; j < 6; ++j) w[j] = diag[j] * wa2[j] * ninv;
    chol6_fwd(L, Linv, w, y);
  }
  double t2 = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) t2 = fma(y[j], y[j], t2);
  double parl = fp * dinv * LmOps<FAST>::rcp(t2);
  double gq = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double v = g[j] * LmOps<FAST>::rcp(diag[j]);
    gq = fma(v, v, gq);
  }
  const double gnorm = gq > 1e-290 ? LmOps<FAST>::sqrt_pos(gq) : 0.0;
  double paru = gnorm * dinv;
  if (paru == 0) paru = dwarf / lm_min(delta, p1);
  double par = lm_min(lm_max(par_io, parl), paru);
  if (par == 0) par = gnorm * LmOps<FAST>::rcp(dxnorm);
  for (int iter = 1;; ++iter) {
    if (par == 0) par = lm_max(dwarf, (double)0.001f * paru);
    if (!chol6_regs<FAST>(a21, diag, par, L, Linv)) return false;  // (A + par D^2 is better conditioned than A: does not happen)
    chol6_fwd(L, Linv, g, y);
    chol6_bwd(L, Linv, y, xs);
    dxnorm = dnorm6<FAST>(diag, xs, wa2);
    const double temp = fp;
    fp = dxnorm - delta;
    if (fabs(fp) <= p1 * delta || (parl == 0 && fp <= temp && temp < 0) || iter == 10) break;
    const double ninv = LmOps<FAST>::rcp(dxnorm);
#pragma unroll
    for (int j = 0; j < 6; ++j) w[j] = diag[j] * (wa2[j] * ninv);
    chol6_fwd(L, Linv, w, y);
    double t3 = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) t3 = fma(y[j], y[j], t3);
    const double parc = fp * dinv * LmOps<FAST>::rcp(t3);
    if (fp > 0) parl = lm_max(parl, par);
    if (fp < 0) paru = lm_min(paru, par);
    par = lm_max(parl, par + parc);
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) x[j] = xs[j];
  par_io = par;
  return true;
}

// ---- LevenbergMarquardt::minimize as a state machine around the passes ------------------------------------------------------------
HOP_LM_INL double lm_scaled_norm(const double diag[6], const float v[6]) {
  double q = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) q += (diag[j] * (double)v[j]) * (diag[j] * (double)v[j]);
  return sqrt(q);
}
// do { lmpar; candidate } of minimizeOneStep (LevenbergMarquardt.h:262-275)
template <class LS>
HOP_LM_DEV void lm_inner(LS& s) {
  double xs[6];
  if (!(LS::regs_lmpar && lm_par_regs<LS::fast_lmpar>(s.A, s.g, s.diag, s.delta, s.par, xs))) {
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
  if (s.iter == 1) s.delta = lm_min(s.delta, s.pnorm);
  s.phase = 1;
}
// head of minimizeOneStep (LevenbergMarquardt.h:219-260); false = finished
template <class LS>
HOP_LM_DEV bool lm_outer(LS& s) {
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
        if (wa2[j] != 0) s.gnorm = lm_max(s.gnorm, fabs(s.g[j] * finv * wa2inv[j]));
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (wa2[j] != 0) s.gnorm = lm_max(s.gnorm, fabs(s.g[j] / s.fnorm / wa2[j]));
    }
  }
  if (s.gnorm <= 0) {
    s.status = 4;  // CosinusTooSmall
    return false;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) s.diag[j] = lm_max(s.diag[j], wa2[j]);
  lm_inner(s);
  return true;
}
// consumes the sums of the pass at s.xc (cand: 21 + 6 + 1); true = another pass at the new s.xc
template <class LS>
HOP_LM_DEV bool lm_advance(LS& s, const double* cand) {
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
    // (a NaN sum of squares -- a trial step outside the quaternion's unit ball -- stays NaN: Eigen's `0.1 * fnorm1 < fnorm` is then false and the step is refused)
    fnorm1 = c27 != c27 ? c27 : (c27 > 1e-290 ? c27 * rsqrt_nr(c27) : 0.0);
    const double finv = rcp_nr(s.fnorm), r1 = fnorm1 * finv;
    if (p1 * fnorm1 < s.fnorm) actred = 1.0 - r1 * r1;
    temp1 = lm_max(jp2, 0.0) * finv * finv;
    temp2 = s.par * (s.pnorm * finv) * (s.pnorm * finv);
  } else {
    fnorm1 = sqrt(cand[27]);
    if (p1 * fnorm1 < s.fnorm) actred = 1.0 - (fnorm1 / s.fnorm) * (fnorm1 / s.fnorm);
    const double t1r = sqrt(lm_max(jp2, 0.0)) / s.fnorm, t2r = sqrt(s.par) * s.pnorm / s.fnorm;
    temp1 = t1r * t1r, temp2 = t2r * t2r;
  }
  const double prered = temp1 + temp2 / p5, dirder = -(temp1 + temp2);
  double ratio = 0;
  if (prered != 0) ratio = LS::fast_lmpar ? actred * rcp_nr(prered) : actred / prered;
  if (ratio <= p25) {
    double temp = p5;
    if (actred < 0) temp = p5 * dirder / (dirder + p5 * actred);
    if (p1 * fnorm1 >= s.fnorm || temp < p1) temp = p1;
    s.delta = temp * lm_min(s.delta, s.pnorm / p1);
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

// ---- nn_mode 6: every function evaluation of the minimiser from the 13 x 13 moment matrix of the correspondences --------------------------
// w(x) = (R(x) - I [9], t + (R(x) - I) c [3], 1) in double; R from the quaternion as WarpPointRigid6D::setParam forms it
template <bool FAST>
HOP_LM_DEV void lm6_w(const float x[6], const double c[3], double w[13]) {
  const double qx = (double)x[3], qy = (double)x[4], qz = (double)x[5];
  const double qw2 = 1.0 - (qx * qx + qy * qy + qz * qz);  // (the quaternion's norm is 1 in exact arithmetic: normalize() is the identity)
  const double qw = FAST && qw2 > 1e-290 ? qw2 * rsqrt_nr(qw2) : sqrt(qw2);
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
HOP_LM_INL int sym13(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
// MSTRIDE: distance between consecutive packed entries of M (64: one lane's column of the LDS table; 1: a plain array).
template <bool FAST, int MSTRIDE>
HOP_LM_DEV void lm6_eval(const double* __restrict__ Mlds /* + lane, stride MSTRIDE */, const double c[3], const float xc[6], double* __restrict__ cand) {
  double w0[13], d[3][12], st[3];
  lm6_w<FAST>(xc, c, w0);
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
      lm6_w<FAST>(xx, c, wj);
#pragma unroll
      for (int k = 0; k < 12; ++k) d[j - 3][k] = (wj[k] - w0[k]) * hinv;
    }
  }
  double ff = 0, gr[3] = {0, 0, 0}, Arr[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, gt[3], Atr[3][3], Att[6];
#pragma unroll
  for (int i = 0; i < 13; ++i) {
    double row[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) row[k] = Mlds[sym13(i, k) * MSTRIDE];
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
  cand[27] = lm_max(ff, 0.0);
}


#endif  // HOP_LM_CORE_H_
