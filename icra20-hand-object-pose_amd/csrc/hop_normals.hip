// hop_normals.hip -- "next" row N3: the two surface-normal estimators the reference driver calls.
//
//   Utils::calNormalIntegralImage(scene_rgb, -1, 0.02, 10, true)   src/perception/src/Utils.cpp:293-329,
//       main_realdata_auto.cpp:61  ->  pcl::IntegralImageNormalEstimation, SIMPLE_3D_GRADIENT, depth-dependent smoothing
//   Utils::calNormalMLS(object1, 0.003)                            Utils.cpp:268-289, main_realdata_auto.cpp:153
//       ->  pcl::MovingLeastSquares, polynomial order 2, normals, no upsampling, SIMPLE projection
//
// PCL 1.9 is not vendored in the reference and not installed here: both algorithms are restated from PCL's published
// sources (features/impl/integral_image_normal.hpp, features/impl/integral_image2D.hpp, surface/impl/mls.hpp,
// common/impl/eigen.hpp: eigen33 / computeRoots, common/impl/centroid.hpp) -- "parity unpinned".  The CPU restatement
// beside them is oracle/normals_oracle.cpp; tests compare the two and check hand-computed planes / spheres.
#include <hip/hip_runtime.h>

#include <chrono>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "hop_ctx_ext.h"
#include "hop_math.h"

using namespace hop;

#define NRCHK(ctx, call)                                                             \
  do {                                                                               \
    hipError_t _e = (call);                                                          \
    if (_e != hipSuccess) {                                                          \
      hop_ctx_set_error((ctx), std::string(#call) + ": " + hipGetErrorString(_e));   \
      return HOP_E_HIP;                                                              \
    }                                                                                \
  } while (0)

namespace {

struct Normals : HopExt {
  DevBuf xyz, nrm, dmap, cmap, out, flags, pos, scan_tmp, scalars;
  ~Normals() override {
    for (DevBuf* b : {&xyz, &nrm, &dmap, &cmap, &out, &flags, &pos, &scan_tmp, &scalars}) b->release();
  }
};
Normals* normals_ext(hop_ctx* c) {
  HopExt*& e = hop_ctx_ext(c, HOP_EXT_NORMALS);
  if (!e) e = new Normals;
  return static_cast<Normals*>(e);
}

// ---------------------------------------------------------------------------------------------------------------------
// Integral-image normals.  The reference's organised cloud marks a dropped pixel with (0,0,0) (Utils.cpp:92,104-108:
// "bad_point = 0"), which is a FINITE point for PCL: it takes part in the depth-change test and in the sums.  A
// non-finite input coordinate is read as 0 here, so both conventions give the reference's result.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fin0(float v) { return isfinite(v) ? v : 0.f; }

// depth-change map (integral_image_normal.hpp, computeFeature): the loop over ri < H-1, ci < W-1 clears index, index+1 /
// index+W when the depth step to the right / downward neighbour exceeds factor * (|depth| + 1) * 2 -- here per pixel as
// a gather of the three ways a pixel can be cleared.  Then the initial distance map: 0 at cleared pixels, W + H elsewhere.
__global__ void k_ii_depth_change(const float* __restrict__ z, int H, int W, float factor, float* __restrict__ dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int r = i / W, c = i - r * W;
  auto edge_right = [&](int rr, int cc) {  // defined for rr < H-1, cc < W-1
    const float d = fin0(z[rr * W + cc]), dr = fin0(z[rr * W + cc + 1]);
    return fabsf(d - dr) > factor * (fabsf(d) + 1.0f) * 2.0f;
  };
  auto edge_down = [&](int rr, int cc) {
    const float d = fin0(z[rr * W + cc]), dd = fin0(z[(rr + 1) * W + cc]);
    return fabsf(d - dd) > factor * (fabsf(d) + 1.0f) * 2.0f;
  };
  bool cleared = false;
  if (r < H - 1 && c < W - 1) cleared = edge_right(r, c) || edge_down(r, c);
  if (!cleared && r < H - 1 && c >= 1) cleared = edge_right(r, c - 1);   // as "index + 1" of the pixel to the left
  if (!cleared && r >= 1 && c < W - 1) cleared = edge_down(r - 1, c);    // as "index + W" of the pixel above
  dist[i] = cleared ? 0.0f : (float)(W + H);
}

// The two raster passes of the distance transform, with PCL's float additions (+1.0f, +1.4f) in PCL's order: the value
// decides static_cast<int>(smoothing), so it has to be the same float.  Row r waits for row r-1 to be two columns ahead:
// one thread per row, skewed by two steps per row, one barrier per step (single workgroup; H <= 1024).
// Indexing is linear exactly as in PCL: previous_row[W] is current_row[0], next_row[-1] is current_row[W-1].
__global__ __launch_bounds__(1024) void k_ii_distance_pass(float* __restrict__ d, int H, int W, int backward) {
  const int t = threadIdx.x;
  if (!backward) {
    const int r = t + 1;  // rows 1 .. H-1
    const int steps = (W - 1) + 2 * (H - 2);
    float left = (r < H) ? d[r * W] : 0.f;  // current_row[0]: never written by this pass
    for (int s = 1; s <= steps; ++s) {
      const int ci = s - 2 * (r - 1);
      if (r < H && ci >= 1 && ci <= W - 1) {
        const float* prev = d + (size_t)(r - 1) * W;
        const float up_left = prev[ci - 1] + 1.4f, up = prev[ci] + 1.0f, up_right = prev[ci + 1] + 1.4f, lf = left + 1.0f;
        const float center = d[(size_t)r * W + ci];
        const float mn = fminf(fminf(up_left, up), fminf(lf, up_right));
        float v = center;
        if (mn < center) v = mn, d[(size_t)r * W + ci] = mn;
        left = v;
      }
      __syncthreads();
    }
  } else {
    const int r = H - 2 - t;  // rows H-2 .. 0
    const int steps = (W - 1) + 2 * (H - 2);
    float right = (r >= 0) ? d[(size_t)r * W + W - 1] : 0.f;  // current_row[W-1]: never written by this pass
    for (int s = 1; s <= steps; ++s) {
      const int ci = (W - 2) - (s - 1 - 2 * t);
      if (r >= 0 && ci >= 0 && ci <= W - 2) {
        const float* next = d + (size_t)(r + 1) * W;
        const float lower_left = next[ci - 1] + 1.4f, lower = next[ci] + 1.0f, lower_right = next[ci + 1] + 1.4f, rt = right + 1.0f;
        const float center = d[(size_t)r * W + ci];
        const float mn = fminf(fminf(lower_left, lower), fminf(rt, lower_right));
        float v = center;
        if (mn < center) v = mn, d[(size_t)r * W + ci] = mn;
        right = v;
      }
      __syncthreads();
    }
  }
}

// computeFeatureFull (BORDER_POLICY_IGNORE) + computePointNormal, SIMPLE_3D_GRADIENT.  The integral image's first-order
// sums over a 1 x h column / w x 1 row are formed directly (double, <= ~12 terms): PCL takes them as differences of a
// double integral image, equal to ~1e-10.
__global__ void k_ii_normals(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const float* __restrict__ dist, int H, int W,
                             float smoothing_size, int depth_dependent, float* __restrict__ nx, float* __restrict__ ny, float* __restrict__ nz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int r = i / W, c = i - r * W;
  const float bad = __builtin_nanf("");
  float ox = bad, oy = bad, oz = bad;
  const int border = (int)smoothing_size;
  if (r >= border && r < H - border && c >= border && c < W - border) {
    const float depth = fin0(z[i]);
    const float smoothing = depth_dependent ? fminf(dist[i], smoothing_size + depth / 10.0f) : fminf(dist[i], smoothing_size);
    if (smoothing > 2.0f) {
      const int rw = (int)smoothing, rh = rw, rw2 = rw / 2, rh2 = rh / 2;
      auto col_sum = [&](int sx, int sy, int h, double o[3]) {
        o[0] = o[1] = o[2] = 0.0;
        for (int k = 0; k < h; ++k) {
          const int j = (sy + k) * W + sx;
          const float px = fin0(x[j]), py = fin0(y[j]), pz = fin0(z[j]);
          o[0] += (double)px, o[1] += (double)py, o[2] += (double)pz;
        }
      };
      auto row_sum = [&](int sx, int sy, int w, double o[3]) {
        o[0] = o[1] = o[2] = 0.0;
        for (int k = 0; k < w; ++k) {
          const int j = sy * W + sx + k;
          const float px = fin0(x[j]), py = fin0(y[j]), pz = fin0(z[j]);
          o[0] += (double)px, o[1] += (double)py, o[2] += (double)pz;
        }
      };
      double a[3], b[3], gx[3], gy[3];
      col_sum(c + rw2, r - rh2, rh, a);
      col_sum(c - rw2, r - rh2, rh, b);
      for (int k = 0; k < 3; ++k) gx[k] = a[k] - b[k];
      row_sum(c - rw2, r + rh2, rw, a);
      row_sum(c - rw2, r - rh2, rw, b);
      for (int k = 0; k < 3; ++k) gy[k] = a[k] - b[k];
      const double n0 = gy[1] * gx[2] - gy[2] * gx[1], n1 = gy[2] * gx[0] - gy[0] * gx[2], n2 = gy[0] * gx[1] - gy[1] * gx[0];  // gradient_y x gradient_x
      const double len = n0 * n0 + n1 * n1 + n2 * n2;
      if (len != 0.0) {
        const double s = sqrt(len);
        float fx = (float)(n0 / s), fy = (float)(n1 / s), fz = (float)(n2 / s);
        // pcl::flipNormalTowardsViewpoint, viewpoint (0,0,0)
        const float vx = 0.f - fin0(x[i]), vy = 0.f - fin0(y[i]), vz = 0.f - fin0(z[i]);
        const float cos_theta = (vx * fx + vy * fy + vz * fz);
        if (cos_theta < 0) fx *= -1, fy *= -1, fz *= -1;
        ox = fx, oy = fy, oz = fz;
      }
    }
  }
  nx[i] = ox, ny[i] = oy, nz[i] = oz;
}

// ---------------------------------------------------------------------------------------------------------------------
// Moving least squares (pcl::MLSResult::computeMLSSurface + projectQueryPoint(SIMPLE), mls.hpp).  One wavefront per
// query point: the neighbours within the radius (FLANN's float L2, strict <, the point itself included) are collected by a
// scan of the whole cloud into an LDS list, then centroid, covariance, smallest eigenvector (pcl::eigen33), the weighted
// second-order fit in the Darboux frame (6 x 6, Cholesky) -- all in double, sums reduced across the wavefront.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int MLS_CAP = 1024;  // neighbours kept per query (a 3 mm ball of a 1 mm voxel cloud holds <= ~120)

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__device__ void roots2(double b, double c, double r[3]) {  // pcl::computeRoots2
  r[0] = 0.0;
  double d = b * b - 4.0 * c;
  if (d < 0.0) d = 0.0;
  const double sd = sqrt(d);
  r[2] = 0.5 * (b + sd);
  r[1] = 0.5 * (b - sd);
}
__device__ void roots3(const double m[3][3], double r[3]) {  // pcl::computeRoots
  const double c0 = m[0][0] * m[1][1] * m[2][2] + 2.0 * m[0][1] * m[0][2] * m[1][2] - m[0][0] * m[1][2] * m[1][2] - m[1][1] * m[0][2] * m[0][2] -
                    m[2][2] * m[0][1] * m[0][1];
  const double c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] + m[1][1] * m[2][2] - m[1][2] * m[1][2];
  const double c2 = m[0][0] + m[1][1] + m[2][2];
  if (fabs(c0) < 2.220446049250313e-16) {
    roots2(c2, c1, r);
    return;
  }
  const double s_inv3 = 1.0 / 3.0, s_sqrt3 = sqrt(3.0);
  const double c2_over_3 = c2 * s_inv3;
  double a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0) a_over_3 = 0.0;
  const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
  double q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0) q = 0.0;
  const double rho = sqrt(-a_over_3);
  const double theta = atan2(sqrt(-q), half_b) * s_inv3;
  const double cos_theta = cos(theta), sin_theta = sin(theta);
  r[0] = c2_over_3 + 2.0 * rho * cos_theta;
  r[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  r[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  auto swp = [](double& a, double& b) {
    const double t = a;
    a = b, b = t;
  };
  if (r[0] >= r[1]) swp(r[0], r[1]);
  if (r[1] >= r[2]) {
    swp(r[1], r[2]);
    if (r[0] >= r[1]) swp(r[0], r[1]);
  }
  if (r[0] <= 0) roots2(c2, c1, r);
}
// pcl::eigen33 (smallest eigenvalue and its eigenvector)
__device__ void eigen33_smallest(const double cov[3][3], double& eval, double evec[3]) {
  double scale = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) scale = fmax(scale, fabs(cov[i][j]));
  if (scale <= 2.2250738585072014e-308) scale = 1.0;
  double m[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m[i][j] = cov[i][j] / scale;
  double r[3];
  roots3(m, r);
  eval = r[0] * scale;
  for (int i = 0; i < 3; ++i) m[i][i] -= r[0];
  auto cross = [](const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1], o[1] = a[2] * b[0] - a[0] * b[2], o[2] = a[0] * b[1] - a[1] * b[0];
  };
  double v1[3], v2[3], v3[3];
  cross(m[0], m[1], v1), cross(m[0], m[2], v2), cross(m[1], m[2], v3);
  const double l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2], l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2],
               l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
  const double* v = v3;
  double l = l3;
  if (l1 >= l2 && l1 >= l3) v = v1, l = l1;
  else if (l2 >= l1 && l2 >= l3) v = v2, l = l2;
  const double s = sqrt(l);
  for (int k = 0; k < 3; ++k) evec[k] = v[k] / s;
}

struct MlsOut {
  float *px, *py, *pz, *nx, *ny, *nz, *curv;
  unsigned* valid;
  int* overflow;
};

__global__ __launch_bounds__(256) void k_mls(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ Z, int n, float r2, int order,
                                             MlsOut o) {
  __shared__ int list[4][MLS_CAP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n) return;
  int* L = list[wave];
  const float qx = X[i], qy = Y[i], qz = Z[i];
  int cnt = 0;
  if (isfinite(qx) && isfinite(qy) && isfinite(qz)) {
    for (int j0 = 0; j0 < n; j0 += 64) {
      const int j = j0 + lane;
      bool in = false;
      if (j < n) {
        const float dx = qx - X[j], dy = qy - Y[j], dz = qz - Z[j];
        in = ((dx * dx + dy * dy) + dz * dz) < r2;  // FLANN L2_Simple, RadiusResultSet: strict
      }
      const unsigned long long m = __ballot(in);
      const int at = cnt + __popcll(m & ((1ull << lane) - 1ull));
      if (in && at < MLS_CAP) L[at] = j;
      cnt += __popcll(m);
    }
  }
  if (cnt > MLS_CAP) {
    if (lane == 0) *o.overflow = 1;
    cnt = MLS_CAP;
  }
  __builtin_amdgcn_wave_barrier();
  if (cnt < 3) {  // mls.hpp performProcessing: such a point produces no output
    if (lane == 0) o.valid[i] = 0u;
    return;
  }
  // pcl::compute3DCentroid, pcl::computeCovarianceMatrix (not normalised)
  double s[3] = {0, 0, 0};
  for (int k = lane; k < cnt; k += 64) s[0] += (double)X[L[k]], s[1] += (double)Y[L[k]], s[2] += (double)Z[L[k]];
  double cen[3];
  for (int a = 0; a < 3; ++a) cen[a] = wsum(s[a]) / (double)cnt;
  double cv[6] = {0, 0, 0, 0, 0, 0};  // xx xy xz yy yz zz
  for (int k = lane; k < cnt; k += 64) {
    const double dx = (double)X[L[k]] - cen[0], dy = (double)Y[L[k]] - cen[1], dz = (double)Z[L[k]] - cen[2];
    cv[0] += dx * dx, cv[1] += dx * dy, cv[2] += dx * dz, cv[3] += dy * dy, cv[4] += dy * dz, cv[5] += dz * dz;
  }
  for (int a = 0; a < 6; ++a) cv[a] = wsum(cv[a]);
  const double cov[3][3] = {{cv[0], cv[1], cv[2]}, {cv[1], cv[3], cv[4]}, {cv[2], cv[4], cv[5]}};
  double eval, nrm[3];
  eigen33_smallest(cov, eval, nrm);
  const double d4 = -(nrm[0] * cen[0] + nrm[1] * cen[1] + nrm[2] * cen[2]);
  const double q[3] = {(double)qx, (double)qy, (double)qz};
  const double distance = q[0] * nrm[0] + q[1] * nrm[1] + q[2] * nrm[2] + d4;
  double mean[3];
  for (int a = 0; a < 3; ++a) mean[a] = q[a] - distance * nrm[a];
  double curvature = cv[0] + cv[3] + cv[5];
  if (curvature != 0) curvature = fabs(eval / curvature);
  // Eigen unitOrthogonal (Geometry/OrthoMethods.h) and the Darboux frame
  double v[3], u[3];
  {
    const double prec = 1e-12;  // NumTraits<double>::dummy_precision()
    auto much_smaller = [&](double a, double b) { return fabs(a) <= fabs(b) * prec; };
    if (!much_smaller(nrm[0], nrm[2]) || !much_smaller(nrm[1], nrm[2])) {
      const double inv = 1.0 / sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1]);
      v[0] = -nrm[1] * inv, v[1] = nrm[0] * inv, v[2] = 0.0;
    } else {
      const double inv = 1.0 / sqrt(nrm[1] * nrm[1] + nrm[2] * nrm[2]);
      v[0] = 0.0, v[1] = -nrm[2] * inv, v[2] = nrm[1] * inv;
    }
    u[0] = nrm[1] * v[2] - nrm[2] * v[1], u[1] = nrm[2] * v[0] - nrm[0] * v[2], u[2] = nrm[0] * v[1] - nrm[1] * v[0];
  }
  double out_n[3] = {nrm[0], nrm[1], nrm[2]}, out_p[3] = {mean[0], mean[1], mean[2]};
  const int nr_coeff = (order + 1) * (order + 2) / 2;
  if (order == 2 && cnt >= nr_coeff) {
    // P W P^T (lower triangle, 21) and P W f (6); term order of mls.hpp: 1, v, v^2, u, u v, u^2
    double A[21], b[6];
    for (double& t : A) t = 0.0;
    for (double& t : b) t = 0.0;
    const double sqr_gauss = (double)r2;
    for (int k = lane; k < cnt; k += 64) {
      const double dx = (double)X[L[k]] - mean[0], dy = (double)Y[L[k]] - mean[1], dz = (double)Z[L[k]] - mean[2];
      const double w = exp(-(dx * dx + dy * dy + dz * dz) / sqr_gauss);
      const double uc = dx * u[0] + dy * u[1] + dz * u[2], vc = dx * v[0] + dy * v[1] + dz * v[2], f = dx * nrm[0] + dy * nrm[1] + dz * nrm[2];
      const double P[6] = {1.0, vc, vc * vc, uc, uc * vc, uc * uc};
      int t = 0;
      for (int a = 0; a < 6; ++a) {
        for (int c = 0; c <= a; ++c) A[t++] += w * P[a] * P[c];
        b[a] += w * P[a] * f;
      }
    }
    for (double& t : A) t = wsum(t);
    for (double& t : b) t = wsum(t);
    // LLT (Eigen::LLT::solveInPlace); a non-positive pivot leaves the plane result (Eigen would carry on with a partial
    // factor: only reachable with degenerate neighbourhoods)
    double Lm[6][6];
    bool ok = true;
    {
      int t = 0;
      for (int a = 0; a < 6; ++a)
        for (int c = 0; c <= a; ++c) Lm[a][c] = A[t++];
    }
    for (int a = 0; a < 6 && ok; ++a) {
      for (int c = 0; c <= a; ++c) {
        double sacc = Lm[a][c];
        for (int k = 0; k < c; ++k) sacc -= Lm[a][k] * Lm[c][k];
        if (a == c) {
          if (!(sacc > 0.0)) {
            ok = false;
            break;
          }
          Lm[a][a] = sqrt(sacc);
        } else
          Lm[a][c] = sacc / Lm[c][c];
      }
    }
    if (ok) {
      double yv[6], cvec[6];
      for (int a = 0; a < 6; ++a) {
        double sacc = b[a];
        for (int k = 0; k < a; ++k) sacc -= Lm[a][k] * yv[k];
        yv[a] = sacc / Lm[a][a];
      }
      for (int a = 5; a >= 0; --a) {
        double sacc = yv[a];
        for (int k = a + 1; k < 6; ++k) sacc -= Lm[k][a] * cvec[k];
        cvec[a] = sacc / Lm[a][a];
      }
      if (isfinite(cvec[0])) {
        // projectPointSimpleToPolynomialSurface(0, 0): z = c[0], dz/du = c[order + 1], dz/dv = c[1]
        const double z0 = cvec[0], zu = cvec[3], zv = cvec[1];
        double nn[3];
        for (int a = 0; a < 3; ++a) nn[a] = nrm[a] - (zu * u[a] + zv * v[a]);
        const double ln = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
        for (int a = 0; a < 3; ++a) out_n[a] = nn[a] / ln, out_p[a] = mean[a] + z0 * nrm[a];
      }
    }
  }
  if (lane == 0) {
    o.px[i] = (float)out_p[0], o.py[i] = (float)out_p[1], o.pz[i] = (float)out_p[2];
    o.nx[i] = (float)out_n[0], o.ny[i] = (float)out_n[1], o.nz[i] = (float)out_n[2];
    o.curv[i] = (float)curvature;
    o.valid[i] = 1u;
  }
}

}  // namespace

// device-level entry (also used by hop_scene_from_depth_normals in hop_physics.hip): x, y, z, nx, ny, nz device planes of H*W
int hop_normals_ii_device(hop_ctx* c, const float* x, const float* y, const float* z, int H, int W, float max_depth_change_factor, float smoothing_size,
                          int depth_dependent, float* nx, float* ny, float* nz) {
  if (H < 2 || W < 2) return HOP_E_INVALID;
  if (H > 1025) return HOP_E_CAPACITY;  // one thread per row in the distance transform
  Normals* nr = normals_ext(c);
  hipStream_t st = hop_ctx_stream(c);
  const int n = H * W;
  NRCHK(c, nr->dmap.ensure(sizeof(float) * (size_t)n));
  float* d = nr->dmap.as<float>();
  k_ii_depth_change<<<(n + 255) / 256, 256, 0, st>>>(z, H, W, max_depth_change_factor, d);
  const int threads = std::max(64, ((H - 1) + 63) / 64 * 64);
  k_ii_distance_pass<<<1, threads, 0, st>>>(d, H, W, 0);
  k_ii_distance_pass<<<1, threads, 0, st>>>(d, H, W, 1);
  k_ii_normals<<<(n + 255) / 256, 256, 0, st>>>(x, y, z, d, H, W, smoothing_size, depth_dependent, nx, ny, nz);
  NRCHK(c, hipGetLastError());
  return HOP_OK;
}

extern "C" {

int hop_normals_integral_image(hop_ctx* c, const float* xyz, int H, int W, float max_depth_change_factor, float normal_smoothing_size,
                               int depth_dependent_smoothing, float* nrm_out) {
  if (!c || !xyz || !nrm_out || H < 2 || W < 2 || !(normal_smoothing_size >= 1.f)) return HOP_E_INVALID;
  NRCHK(c, hipSetDevice(hop_ctx_device(c)));
  Normals* nr = normals_ext(c);
  hipStream_t st = hop_ctx_stream(c);
  const size_t n = (size_t)H * W;
  NRCHK(c, nr->xyz.ensure(sizeof(float) * 3 * n));
  NRCHK(c, nr->nrm.ensure(sizeof(float) * 3 * n));
  static const bool prof = getenv("HOP_PROFILE_NORMALS") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = now();
  NRCHK(c, hop_ctx_h2d(c, nr->xyz.p, xyz, sizeof(float) * 3 * n));
  if (prof) NRCHK(c, hipStreamSynchronize(st));
  const auto t1 = now();
  float* p = nr->xyz.as<float>();
  float* q = nr->nrm.as<float>();
  const int rc = hop_normals_ii_device(c, p, p + n, p + 2 * n, H, W, max_depth_change_factor, normal_smoothing_size, depth_dependent_smoothing, q, q + n, q + 2 * n);
  if (rc) return rc;
  if (prof) NRCHK(c, hipStreamSynchronize(st));
  const auto t2 = now();
  NRCHK(c, hop_ctx_d2h(c, nrm_out, q, sizeof(float) * 3 * n));
  NRCHK(c, hipStreamSynchronize(st));
  if (prof) std::printf("integral-image normals: upload %.2f ms, kernels %.2f ms, download %.2f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, now()));
  return HOP_OK;
}

int hop_normals_mls(hop_ctx* c, const float* xyz, int n, float search_radius, int polynomial_order, float* out_xyz, float* out_nrm, float* out_curvature,
                    int* keep_index, int cap, int* n_out) {
  if (!c || n < 0 || (n > 0 && !xyz) || !n_out || cap < 0 || !(search_radius > 0) || (polynomial_order != 2 && polynomial_order != 1 && polynomial_order != 0))
    return HOP_E_INVALID;
  *n_out = 0;
  if (n == 0) return HOP_OK;
  NRCHK(c, hipSetDevice(hop_ctx_device(c)));
  Normals* nr = normals_ext(c);
  hipStream_t st = hop_ctx_stream(c);
  NRCHK(c, nr->xyz.ensure(sizeof(float) * 3 * (size_t)n));
  NRCHK(c, nr->out.ensure(sizeof(float) * 7 * (size_t)n));
  NRCHK(c, nr->flags.ensure(sizeof(unsigned) * (size_t)n));
  NRCHK(c, nr->scalars.ensure(sizeof(int) * 4));
  NRCHK(c, hop_ctx_h2d(c, nr->xyz.p, xyz, sizeof(float) * 3 * (size_t)n));
  NRCHK(c, hipMemsetAsync(nr->scalars.p, 0, sizeof(int) * 4, st));
  const float* p = nr->xyz.as<float>();
  float* o = nr->out.as<float>();
  MlsOut mo{o, o + n, o + 2 * (size_t)n, o + 3 * (size_t)n, o + 4 * (size_t)n, o + 5 * (size_t)n, o + 6 * (size_t)n, nr->flags.as<unsigned>(), nr->scalars.as<int>()};
  // the radius reaches FLANN as float(radius * radius) of the double search radius (pcl::KdTreeFLANN::radiusSearch)
  const float r2 = (float)((double)search_radius * (double)search_radius);
  k_mls<<<(n + 3) / 4, 256, 0, st>>>(p, p + n, p + 2 * (size_t)n, n, r2, polynomial_order, mo);
  NRCHK(c, hipGetLastError());
  std::vector<float> h(7 * (size_t)n);
  std::vector<unsigned> hv(n);
  int overflow = 0;
  NRCHK(c, hop_ctx_d2h(c, h.data(), o, sizeof(float) * 7 * (size_t)n));
  NRCHK(c, hop_ctx_d2h(c, hv.data(), nr->flags.p, sizeof(unsigned) * (size_t)n));
  NRCHK(c, hop_ctx_d2h(c, &overflow, nr->scalars.p, sizeof(int)));
  NRCHK(c, hipStreamSynchronize(st));
  if (overflow) return HOP_E_CAPACITY;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (!hv[i]) continue;
    if (m < cap) {
      for (int k = 0; k < 3; ++k) {
        if (out_xyz) out_xyz[(size_t)k * cap + m] = h[(size_t)k * n + i];
        if (out_nrm) out_nrm[(size_t)k * cap + m] = h[(size_t)(3 + k) * n + i];
      }
      if (out_curvature) out_curvature[m] = h[(size_t)6 * n + i];
      if (keep_index) keep_index[m] = i;
    }
    ++m;
  }
  *n_out = m;
  return m > cap ? HOP_E_CAPACITY : HOP_OK;
}

}  // extern "C"
