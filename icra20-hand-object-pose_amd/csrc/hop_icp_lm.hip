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

#include "hop_lm_core.h"  // the minimiser itself: 6 x 6 algebra, lmpar2, the minimizeOneStep state machine, the moment-form evaluation (host-testable)

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
      lm6_eval<true, 64>(M, c, s.xc, cand);
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

// nn_mode 7: the same kernel on the integer moment sums of k_icp_fusedq_momi, IEEE operations only (LmDev7, lm6_eval<false>): the
// block partials are added in 64-bit integers (any order gives the same sum), scaled back by powers of two (exact), and from there on
// every double is the double the oracle's statement holds (oracle/hop_oracle.cpp "The moment form").
__global__ __launch_bounds__(64) void k_icp_lm7_solve(IcpArgs a, int hb, int nblocks) {
  __shared__ double Msh[91 * 64];
  const int lane = threadIdx.x;
  const int hl = blockIdx.x * 64 + lane;
  if (hl >= hb) return;
  IcpState& st = a.state[hl];
  if (!st.active) return;
  const long long* __restrict__ pp = reinterpret_cast<const long long*>(a.partial) + (size_t)hl * nblocks * ICP_NMOMI_STRIDE;
  long long cnt64 = 0, d2q = 0;
  for (int blk = 0; blk < nblocks; ++blk) cnt64 += pp[(size_t)blk * ICP_NMOMI_STRIDE + ICP_NMOMI], d2q += pp[(size_t)blk * ICP_NMOMI_STRIDE + 91];
  const int cnt = (int)cnt64;
  if (cnt < 3) {
    st.active = 0, st.converged = 0;
    return;
  }
  if (cnt >= 4) {
    double* __restrict__ M = Msh + lane;
#pragma unroll 1
    for (int i = 0; i < 13; ++i) {
      const int ei = i < 9 ? a.mom_k_np : i < 12 ? a.mom_k_n : a.mom_k_r;
#pragma unroll 1
      for (int j = 0; j <= i; ++j) {
        const int ej = j < 9 ? a.mom_k_np : j < 12 ? a.mom_k_n : a.mom_k_r;
        const int idx = i * (i + 1) / 2 + j;
        long long v = 0;
        for (int blk = 0; blk < nblocks; ++blk) v += pp[(size_t)blk * ICP_NMOMI_STRIDE + idx];
        M[idx * 64] = ldexp((double)v, -(ei + ej));  // |v| < 2^53: the conversion and the scaling are exact
      }
    }
    const float* pose = a.pose + (size_t)(a.h0 + hl) * 16;
    const double c[3] = {(double)pose[3], (double)pose[7], (double)pose[11]};
    LmDev7 s;
#pragma unroll
    for (int j = 0; j < 6; ++j) s.x[j] = s.xc[j] = 0.f, s.p[j] = 0.f;
    s.phase = 0, s.status = -1, s.iter = 0, s.nfev = 0;
    s.par = s.delta = s.xnorm = s.fnorm = s.gnorm = s.pnorm = 0.0;
    double cand[28];
    LM_COUNT(0, 1);
    for (int guard = 0; guard < 420; ++guard) {  // (maxfev = 400 ends every run)
      lm6_eval<false, 64>(M, c, s.xc, cand);
      LM_COUNT(1, 1);
      if (!lm_advance(s, cand)) break;
    }
    if (s.status == 5) LM_COUNT(3, 1);
    lm_warp6(s.x, st.T_inc);
  }
  icp_iteration_bookkeeping(a, st, hl, ldexp((double)d2q, -a.mom_k_d), cnt);
}
void launch_icp_lm7_solve(const IcpArgs& a, int hb, int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_lm7_solve, dim3((hb + 63) / 64), dim3(64), 0, s, a, hb, nblocks);
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
