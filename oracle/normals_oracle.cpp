// TEST INFRASTRUCTURE ONLY (imported by tests/, smoke(), bench.py's cpu_baseline leg; never by the product).
//
// CPU restatement of the two PCL normal estimators the reference driver calls ("next" row N3):
//   Utils::calNormalIntegralImage(scene_rgb, -1, 0.02, 10, true)  src/perception/src/Utils.cpp:293-329  (main_realdata_auto.cpp:61)
//   Utils::calNormalMLS(object1, 0.003)                           src/perception/src/Utils.cpp:268-289  (main_realdata_auto.cpp:153)
//
// PARITY UNPINNED: PCL 1.9 (find_package(PCL 1.9), src/perception/CMakeLists.txt:18) is neither vendored in the reference
// nor installed in this image, and the reference has no test for either call.  The algorithms below follow PCL 1.9's
// published sources statement by statement:
//   features/include/pcl/features/impl/integral_image_normal.hpp   computeFeature (depth-change map, two-pass distance map),
//                                                                  computeFeatureFull (BORDER_POLICY_IGNORE), computePointNormal
//                                                                  (SIMPLE_3D_GRADIENT), flipNormalTowardsViewpoint
//   features/include/pcl/features/impl/integral_image2D.hpp        computeIntegralImages (double sums), getFirstOrderSum
//   surface/include/pcl/surface/impl/mls.hpp                       performProcessing, MLSResult::computeMLSSurface,
//                                                                  projectQueryPoint (SIMPLE), projectPointSimpleToPolynomialSurface
//   common/include/pcl/common/impl/eigen.hpp                       eigen33 (smallest eigenpair), computeRoots, computeRoots2
//   common/include/pcl/common/impl/centroid.hpp                    compute3DCentroid, computeCovarianceMatrix (double)
//   Eigen/src/Geometry/OrthoMethods.h                              unitOrthogonal (the reference vendors Eigen 3.3.90)
// and are checked on hand-computed planes and spheres (tests/test_normals_oracle.py).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

extern "C" int orc_voxel_downsample_normals(const float* xyz_planes, const float* nrm_planes, int n, float leaf, float* out_xyz, float* out_nrm, int cap,
                                            int* n_out);

namespace {

inline float fin0(float v) { return std::isfinite(v) ? v : 0.f; }  // the reference's bad_point is (0,0,0) (Utils.cpp:92)

struct D3 {
  double v[3];
};

void roots2(double b, double c, double r[3]) {
  r[0] = 0.0;
  double d = b * b - 4.0 * c;
  if (d < 0.0) d = 0.0;
  const double sd = std::sqrt(d);
  r[2] = 0.5 * (b + sd);
  r[1] = 0.5 * (b - sd);
}
void roots3(const double m[3][3], double r[3]) {
  const double c0 = m[0][0] * m[1][1] * m[2][2] + 2.0 * m[0][1] * m[0][2] * m[1][2] - m[0][0] * m[1][2] * m[1][2] - m[1][1] * m[0][2] * m[0][2] -
                    m[2][2] * m[0][1] * m[0][1];
  const double c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] + m[1][1] * m[2][2] - m[1][2] * m[1][2];
  const double c2 = m[0][0] + m[1][1] + m[2][2];
  if (std::fabs(c0) < std::numeric_limits<double>::epsilon()) {
    roots2(c2, c1, r);
    return;
  }
  const double s_inv3 = 1.0 / 3.0, s_sqrt3 = std::sqrt(3.0);
  const double c2_over_3 = c2 * s_inv3;
  double a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0) a_over_3 = 0.0;
  const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
  double q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0) q = 0.0;
  const double rho = std::sqrt(-a_over_3);
  const double theta = std::atan2(std::sqrt(-q), half_b) * s_inv3;
  const double cos_theta = std::cos(theta), sin_theta = std::sin(theta);
  r[0] = c2_over_3 + 2.0 * rho * cos_theta;
  r[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  r[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  if (r[0] >= r[1]) std::swap(r[0], r[1]);
  if (r[1] >= r[2]) {
    std::swap(r[1], r[2]);
    if (r[0] >= r[1]) std::swap(r[0], r[1]);
  }
  if (r[0] <= 0) roots2(c2, c1, r);
}
void eigen33_smallest(const double cov[3][3], double& eval, double evec[3]) {
  double scale = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) scale = std::max(scale, std::fabs(cov[i][j]));
  if (scale <= std::numeric_limits<double>::min()) scale = 1.0;
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
  const double s = std::sqrt(l);
  for (int k = 0; k < 3; ++k) evec[k] = v[k] / s;
}

}  // namespace

extern "C" {

// pcl::IntegralImageNormalEstimation, SIMPLE_3D_GRADIENT, BORDER_POLICY_IGNORE.  xyz: organised SoA planes (H*W each).
int orc_normals_integral_image(const float* xyz, int H, int W, float max_depth_change_factor, float normal_smoothing_size, int depth_dependent,
                               float* nrm_out) {
  const size_t n = (size_t)H * W;
  const float* X = xyz;
  const float* Y = xyz + n;
  const float* Z = xyz + 2 * n;
  const float bad = std::numeric_limits<float>::quiet_NaN();
  // depth-change map
  std::vector<unsigned char> change(n, 255);
  for (int ri = 0; ri < H - 1; ++ri)
    for (int ci = 0; ci < W - 1; ++ci) {
      const size_t index = (size_t)ri * W + ci;
      const float depth = fin0(Z[index]), depthR = fin0(Z[index + 1]), depthD = fin0(Z[index + W]);
      const float thr = (max_depth_change_factor * (fabsf(depth) + 1.0f) * 2.0f);
      if (std::fabs(depth - depthR) > thr) change[index] = 0, change[index + 1] = 0;
      if (std::fabs(depth - depthD) > thr) change[index] = 0, change[index + W] = 0;
    }
  // distance map, two raster passes (linear indexing across row ends exactly as PCL's row pointers do)
  std::vector<float> dist(n);
  for (size_t i = 0; i < n; ++i) dist[i] = change[i] == 0 ? 0.0f : (float)(W + H);
  {
    float* previous_row = dist.data();
    float* current_row = previous_row + W;
    for (int ri = 1; ri < H; ++ri) {
      for (int ci = 1; ci < W; ++ci) {
        const float upLeft = previous_row[ci - 1] + 1.4f, up = previous_row[ci] + 1.0f, upRight = previous_row[ci + 1] + 1.4f;
        const float left = current_row[ci - 1] + 1.0f, center = current_row[ci];
        const float minValue = std::min(std::min(upLeft, up), std::min(left, upRight));
        if (minValue < center) current_row[ci] = minValue;
      }
      previous_row = current_row;
      current_row += W;
    }
    float* next_row = dist.data() + (size_t)W * (H - 1);
    current_row = next_row - W;
    for (int ri = H - 2; ri >= 0; --ri) {
      for (int ci = W - 2; ci >= 0; --ci) {
        const float lowerLeft = next_row[ci - 1] + 1.4f, lower = next_row[ci] + 1.0f, lowerRight = next_row[ci + 1] + 1.4f;
        const float right = current_row[ci + 1] + 1.0f, center = current_row[ci];
        const float minValue = std::min(std::min(lowerLeft, lower), std::min(right, lowerRight));
        if (minValue < center) current_row[ci] = minValue;
      }
      next_row = current_row;
      current_row -= W;
    }
  }
  // first-order integral image of (x, y, z) in double, (W+1) x (H+1) (IntegralImage2D<float, 3>::computeIntegralImages)
  const int IW = W + 1;
  std::vector<D3> ii((size_t)IW * (H + 1));
  for (int c = 0; c < IW; ++c) ii[c] = D3{{0, 0, 0}};
  for (int r = 0; r < H; ++r) {
    D3* prev = &ii[(size_t)r * IW];
    D3* cur = &ii[(size_t)(r + 1) * IW];
    cur[0] = D3{{0, 0, 0}};
    for (int c = 0; c < W; ++c) {
      for (int k = 0; k < 3; ++k) cur[c + 1].v[k] = prev[c + 1].v[k] + cur[c].v[k] - prev[c].v[k];
      const size_t j = (size_t)r * W + c;
      const float e[3] = {fin0(X[j]), fin0(Y[j]), fin0(Z[j])};
      if (std::isfinite(e[0] + e[1] + e[2]))
        for (int k = 0; k < 3; ++k) cur[c + 1].v[k] += (double)e[k];
    }
  }
  auto first_order_sum = [&](int sx, int sy, int w, int h, double o[3]) {
    const size_t ul = (size_t)sy * IW + sx, ur = ul + w, ll = (size_t)(sy + h) * IW + sx, lr = ll + w;
    for (int k = 0; k < 3; ++k) o[k] = ii[lr].v[k] + ii[ul].v[k] - ii[ur].v[k] - ii[ll].v[k];
  };
  float* NX = nrm_out;
  float* NY = nrm_out + n;
  float* NZ = nrm_out + 2 * n;
  for (size_t i = 0; i < n; ++i) NX[i] = NY[i] = NZ[i] = bad;
  const int border = (int)normal_smoothing_size;
  for (int ri = border; ri < H - border; ++ri)
    for (int ci = border; ci < W - border; ++ci) {
      const size_t index = (size_t)ri * W + ci;
      const float depth = fin0(Z[index]);
      const float smoothing = depth_dependent ? std::min(dist[index], normal_smoothing_size + depth / 10.0f) : std::min(dist[index], normal_smoothing_size);
      if (!(smoothing > 2.0f)) continue;
      const int rw = (int)smoothing, rh = (int)smoothing, rw2 = rw / 2, rh2 = rh / 2;
      double a[3], b[3], gx[3], gy[3];
      first_order_sum(ci + rw2, ri - rh2, 1, rh, a);
      first_order_sum(ci - rw2, ri - rh2, 1, rh, b);
      for (int k = 0; k < 3; ++k) gx[k] = a[k] - b[k];
      first_order_sum(ci - rw2, ri + rh2, rw, 1, a);
      first_order_sum(ci - rw2, ri - rh2, rw, 1, b);
      for (int k = 0; k < 3; ++k) gy[k] = a[k] - b[k];
      const double n0 = gy[1] * gx[2] - gy[2] * gx[1], n1 = gy[2] * gx[0] - gy[0] * gx[2], n2 = gy[0] * gx[1] - gy[1] * gx[0];
      const double len = n0 * n0 + n1 * n1 + n2 * n2;
      if (len == 0.0) continue;
      const double s = std::sqrt(len);
      float nx = (float)(n0 / s), ny = (float)(n1 / s), nz = (float)(n2 / s);
      const float vx = 0.f - fin0(X[index]), vy = 0.f - fin0(Y[index]), vz = 0.f - fin0(Z[index]);
      const float cos_theta = (vx * nx + vy * ny + vz * nz);
      if (cos_theta < 0) nx *= -1, ny *= -1, nz *= -1;
      NX[index] = nx, NY[index] = ny, NZ[index] = nz;
    }
  return 0;
}

// pcl::MovingLeastSquares (order 2, compute normals, SIMPLE projection, no upsampling).  Outputs in input order, planes
// with stride cap; keep_index[k] = input index (mls.getCorrespondingIndices()).
int orc_normals_mls(const float* xyz, int n, float search_radius, int order, float* out_xyz, float* out_nrm, float* out_curv, int* keep_index, int cap,
                    int* n_out) {
  const float* X = xyz;
  const float* Y = xyz + n;
  const float* Z = xyz + 2 * (size_t)n;
  const float r2 = (float)((double)search_radius * (double)search_radius);
  const double sqr_gauss = (double)r2;
  std::vector<float> res((size_t)7 * std::max(n, 1));
  std::vector<unsigned char> valid(std::max(n, 1), 0);
#pragma omp parallel for schedule(dynamic, 64)
  for (int i = 0; i < n; ++i) {
    if (!std::isfinite(X[i]) || !std::isfinite(Y[i]) || !std::isfinite(Z[i])) continue;
    std::vector<int> nn;
    for (int j = 0; j < n; ++j) {
      const float dx = X[i] - X[j], dy = Y[i] - Y[j], dz = Z[i] - Z[j];
      if (((dx * dx + dy * dy) + dz * dz) < r2) nn.push_back(j);
    }
    const int cnt = (int)nn.size();
    if (cnt < 3) continue;
    double cen[3] = {0, 0, 0};
    for (int j : nn) cen[0] += X[j], cen[1] += Y[j], cen[2] += Z[j];
    for (double& c : cen) c /= (double)cnt;
    double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int j : nn) {
      const double dx = X[j] - cen[0], dy = Y[j] - cen[1], dz = Z[j] - cen[2];
      cov[1][1] += dy * dy, cov[1][2] += dy * dz, cov[2][2] += dz * dz;
      cov[0][0] += dx * dx, cov[0][1] += dx * dy, cov[0][2] += dx * dz;
    }
    cov[1][0] = cov[0][1], cov[2][0] = cov[0][2], cov[2][1] = cov[1][2];
    double eval, nrm[3];
    eigen33_smallest(cov, eval, nrm);
    const double d4 = -(nrm[0] * cen[0] + nrm[1] * cen[1] + nrm[2] * cen[2]);
    const double q[3] = {X[i], Y[i], Z[i]};
    const double distance = q[0] * nrm[0] + q[1] * nrm[1] + q[2] * nrm[2] + d4;
    double mean[3];
    for (int a = 0; a < 3; ++a) mean[a] = q[a] - distance * nrm[a];
    double curvature = cov[0][0] + cov[1][1] + cov[2][2];
    if (curvature != 0) curvature = std::fabs(eval / curvature);
    double v[3], u[3];
    {
      const double prec = 1e-12;
      auto much_smaller = [&](double a, double b) { return std::fabs(a) <= std::fabs(b) * prec; };
      if (!much_smaller(nrm[0], nrm[2]) || !much_smaller(nrm[1], nrm[2])) {
        const double inv = 1.0 / std::sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1]);
        v[0] = -nrm[1] * inv, v[1] = nrm[0] * inv, v[2] = 0.0;
      } else {
        const double inv = 1.0 / std::sqrt(nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        v[0] = 0.0, v[1] = -nrm[2] * inv, v[2] = nrm[1] * inv;
      }
      u[0] = nrm[1] * v[2] - nrm[2] * v[1], u[1] = nrm[2] * v[0] - nrm[0] * v[2], u[2] = nrm[0] * v[1] - nrm[1] * v[0];
    }
    double out_n[3] = {nrm[0], nrm[1], nrm[2]}, out_p[3] = {mean[0], mean[1], mean[2]};
    const int nr_coeff = (order + 1) * (order + 2) / 2;
    if (order > 1 && cnt >= nr_coeff) {
      std::vector<double> A((size_t)nr_coeff * nr_coeff, 0.0), b(nr_coeff, 0.0), P(nr_coeff);
      for (int j : nn) {
        const double dx = X[j] - mean[0], dy = Y[j] - mean[1], dz = Z[j] - mean[2];
        const double w = std::exp(-(dx * dx + dy * dy + dz * dz) / sqr_gauss);
        const double uc = dx * u[0] + dy * u[1] + dz * u[2], vc = dx * v[0] + dy * v[1] + dz * v[2], f = dx * nrm[0] + dy * nrm[1] + dz * nrm[2];
        int t = 0;
        double u_pow = 1;
        for (int ui = 0; ui <= order; ++ui) {
          double v_pow = 1;
          for (int vi = 0; vi <= order - ui; ++vi) {
            P[t++] = u_pow * v_pow;
            v_pow *= vc;
          }
          u_pow *= uc;
        }
        for (int a = 0; a < nr_coeff; ++a) {
          for (int c = 0; c < nr_coeff; ++c) A[(size_t)a * nr_coeff + c] += w * P[a] * P[c];
          b[a] += w * P[a] * f;
        }
      }
      // LLT
      std::vector<double> L((size_t)nr_coeff * nr_coeff, 0.0);
      bool ok = true;
      for (int a = 0; a < nr_coeff && ok; ++a)
        for (int c = 0; c <= a; ++c) {
          double s = A[(size_t)a * nr_coeff + c];
          for (int k = 0; k < c; ++k) s -= L[(size_t)a * nr_coeff + k] * L[(size_t)c * nr_coeff + k];
          if (a == c) {
            if (!(s > 0.0)) {
              ok = false;
              break;
            }
            L[(size_t)a * nr_coeff + a] = std::sqrt(s);
          } else
            L[(size_t)a * nr_coeff + c] = s / L[(size_t)c * nr_coeff + c];
        }
      if (ok) {
        std::vector<double> y(nr_coeff), cv(nr_coeff);
        for (int a = 0; a < nr_coeff; ++a) {
          double s = b[a];
          for (int k = 0; k < a; ++k) s -= L[(size_t)a * nr_coeff + k] * y[k];
          y[a] = s / L[(size_t)a * nr_coeff + a];
        }
        for (int a = nr_coeff - 1; a >= 0; --a) {
          double s = y[a];
          for (int k = a + 1; k < nr_coeff; ++k) s -= L[(size_t)k * nr_coeff + a] * cv[k];
          cv[a] = s / L[(size_t)a * nr_coeff + a];
        }
        if (std::isfinite(cv[0])) {
          const double z0 = cv[0], zu = cv[order + 1], zv = cv[1];
          double nnv[3];
          for (int a = 0; a < 3; ++a) nnv[a] = nrm[a] - (zu * u[a] + zv * v[a]);
          const double ln = std::sqrt(nnv[0] * nnv[0] + nnv[1] * nnv[1] + nnv[2] * nnv[2]);
          for (int a = 0; a < 3; ++a) out_n[a] = nnv[a] / ln, out_p[a] = mean[a] + z0 * nrm[a];
        }
      }
    }
    for (int a = 0; a < 3; ++a) res[(size_t)a * n + i] = (float)out_p[a], res[(size_t)(3 + a) * n + i] = (float)out_n[a];
    res[(size_t)6 * n + i] = (float)curvature;
    valid[i] = 1;
  }
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (!valid[i]) continue;
    if (m < cap) {
      for (int k = 0; k < 3; ++k) {
        if (out_xyz) out_xyz[(size_t)k * cap + m] = res[(size_t)k * n + i];
        if (out_nrm) out_nrm[(size_t)k * cap + m] = res[(size_t)(3 + k) * n + i];
      }
      if (out_curv) out_curv[m] = res[(size_t)6 * n + i];
      if (keep_index) keep_index[m] = i;
    }
    ++m;
  }
  *n_out = m;
  return m > cap ? -1 : 0;
}

// main_realdata_auto.cpp:54-96 with normals: organised cloud (bad_point 0), integral-image normals, PassThrough z, VoxelGrid
// over all fields, transform with normals into the hand-base frame, three PassThrough filters, transform back.
int orc_scene_from_depth_normals(const unsigned short* depth_raw, int H, int W, double depth_unit, const float* K9, const float* A /*cam_in_handbase*/,
                                 const float* B /*handbase_in_cam*/, float leaf, const float* crop_min3, const float* crop_max3, float factor, float smoothing,
                                 float* out_xyz, float* out_nrm, int cap, int* n_out) {
  const size_t n = (size_t)H * W;
  std::vector<float> org(3 * n, 0.f), nrm(3 * n);
  for (int u = 0; u < H; ++u)
    for (int v = 0; v < W; ++v) {
      const size_t i = (size_t)u * W + v;
      float depth = (float)((double)(float)depth_raw[i] * depth_unit);
      if (depth > 2.0 || depth < 0.1) depth = 0.0f;
      if (depth > 0.1 && depth < 2.0) org[i] = (float)((v - K9[2]) * depth / K9[0]), org[n + i] = (float)((u - K9[5]) * depth / K9[4]), org[2 * n + i] = depth;
    }
  orc_normals_integral_image(org.data(), H, W, factor, smoothing, 1, nrm.data());
  // PassThrough z in [0.1, 2.0]
  std::vector<float> px, py, pz, qx, qy, qz;
  for (size_t i = 0; i < n; ++i) {
    const float z = org[2 * n + i];
    if (z < 0.1f || z > 2.0f) continue;
    px.push_back(org[i]), py.push_back(org[n + i]), pz.push_back(z);
    qx.push_back(nrm[i]), qy.push_back(nrm[n + i]), qz.push_back(nrm[2 * n + i]);
  }
  const int m = (int)px.size();
  std::vector<float> P(3 * (size_t)m), Q(3 * (size_t)m), VP(3 * (size_t)std::max(m, 1)), VQ(3 * (size_t)std::max(m, 1));
  for (int i = 0; i < m; ++i) {
    P[i] = px[i], P[(size_t)m + i] = py[i], P[2 * (size_t)m + i] = pz[i];
    Q[i] = qx[i], Q[(size_t)m + i] = qy[i], Q[2 * (size_t)m + i] = qz[i];
  }
  int k = 0;
  orc_voxel_downsample_normals(P.data(), Q.data(), m, leaf, VP.data(), VQ.data(), std::max(m, 1), &k);
  const int cp = std::max(m, 1);
  auto point = [](const float* T, float x, float y, float z, float o[3]) {
    for (int r = 0; r < 3; ++r) o[r] = ((T[4 * r] * x + T[4 * r + 1] * y) + T[4 * r + 2] * z) + T[4 * r + 3];
  };
  auto dir = [](const float* T, float x, float y, float z, float o[3]) {
    for (int r = 0; r < 3; ++r) o[r] = (T[4 * r] * x + T[4 * r + 1] * y) + T[4 * r + 2] * z;
  };
  int cnt = 0;
  for (int i = 0; i < k; ++i) {
    float q[3], qn[3], r[3], rn[3];
    point(A, VP[i], VP[(size_t)cp + i], VP[2 * (size_t)cp + i], q);
    dir(A, VQ[i], VQ[(size_t)cp + i], VQ[2 * (size_t)cp + i], qn);
    if (!std::isfinite(q[0]) || !std::isfinite(q[1]) || !std::isfinite(q[2])) continue;
    if (q[2] < crop_min3[2] || q[2] > crop_max3[2]) continue;
    if (q[0] < crop_min3[0] || q[0] > crop_max3[0]) continue;
    if (q[1] < crop_min3[1] || q[1] > crop_max3[1]) continue;
    point(B, q[0], q[1], q[2], r);
    dir(B, qn[0], qn[1], qn[2], rn);
    if (cnt < cap)
      for (int a = 0; a < 3; ++a) out_xyz[(size_t)a * cap + cnt] = r[a], out_nrm[(size_t)a * cap + cnt] = rn[a];
    ++cnt;
  }
  *n_out = cnt;
  return cnt > cap ? -1 : 0;
}

}  // extern "C"
