// sdf_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see hop_oracle.h).  CPU restatement of SURVEY.md 8(f) row N1:
//   * igl::signed_distance with SIGNED_DISTANCE_TYPE_PSEUDONORMAL on float matrices, as SDFchecker calls it
//     (src/perception/src/SDFchecker.cpp:115-134; vendored libigl: include/igl/signed_distance.cpp:18-208,
//     pseudonormal_test.cpp:24-128, point_simplex_squared_distance.cpp:25-132, per_face_normals.cpp:13-38,
//     per_vertex_normals.cpp:40-110 (angle weights: internal_angles.cpp:66-88), per_edge_normals.cpp:23-79,
//     doublearea.cpp:75-200, barycentric_coordinates.cpp:51-100, project_to_line(_segment).cpp);
//   * pcl::VoxelGrid::applyFilter as Utils::downsamplePointCloud uses it (Utils.cpp:334-340) -- PCL is not vendored
//     by the reference and not installed here: restated from the published algorithm (PCL 1.8/1.9
//     filters/include/pcl/filters/impl/voxel_grid.hpp:214-440), PARITY UNPINNED for this function;
//   * PoseEstimator::rejectByCollisionOrNonTouching (src/perception/src/PoseEstimator.cpp:524-735).
//
// Pinning: orc_sdf_signed_distance is checked against oracle/_ref/libref_sdf.so -- the reference's own libigl compiled
// in place (oracle/ref_sdf_driver.cpp) -- by tests/test_sdf_oracle.py and the vectors of tests/golden/sdf_*.npz.
// The closest face is found the way libigl finds it: its AABB tree (AABB.cpp:73-200: one face per leaf, median split of
// the barycentre ranks along the longest box axis) walked depth first (AABB.cpp:360-452: a child that contains the query
// first, otherwise the one with the smaller exterior distance; strict improvements only), because which of several
// faces at exactly the same float distance is reported -- and with it, for queries next to a face plane, the sign --
// depends on that order.  Signed distance AND face index are equal to libigl's on the golden vectors.
//
// Plain IEEE float arithmetic, -ffp-contract=off; 3-element Eigen reductions are c0+(c1+c2) (Redux.h:91-105).
#include "hop_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <utility>
#include <vector>

namespace {

struct F3 {
  float x, y, z;
};
inline F3 f3(float x, float y, float z) { return F3{x, y, z}; }
inline F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline F3 operator*(float s, F3 a) { return f3(s * a.x, s * a.y, s * a.z); }
inline F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
inline float dot3(F3 a, F3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline float sqn3(F3 a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }
inline float norm3(F3 a) { return std::sqrt(sqn3(a)); }
inline F3 cross3(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline bool same3(F3 a, F3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// point_simplex_squared_distance.cpp:43-113 (Ericson, Real-Time Collision Detection ch. 5), Scalar = float
F3 closest_point_triangle(F3 p, F3 a, F3 b, F3 c) {
  const F3 ab = b - a, ac = c - a, ap = p - a;
  const float d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0.0f && d2 <= 0.0f) return a;
  const F3 bp = p - b;
  const float d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0.0f && d4 <= d3) return b;
  const float vc = d1 * d4 - d3 * d2;
  if (!same3(a, b)) {
    if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
      const float v = d1 / (d1 - d3);
      return a + v * ab;
    }
  }
  const F3 cp = p - c;
  const float d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  if (d6 >= 0.0f && d5 <= d6) return c;
  const float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
    const float w = d2 / (d2 - d6);
    return a + w * ac;
  }
  const float va = d3 * d6 - d5 * d4;
  if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
    const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    return b + w * (c - b);
  }
  // `Scalar denom = 1.0 / (va + vb + vc)`: the float sum is promoted, the quotient rounded back to float
  const float denom = (float)(1.0 / (double)((va + vb) + vc));
  const float v = vb * denom;
  const float w = vc * denom;
  return (a + ab * v) + ac * w;
}

struct TreeNode {
  F3 lo, hi;
  int left = -1, right = -1, prim = -1;
};
struct Mesh {
  int nv = 0, nf = 0;
  std::vector<F3> V, FN, VN, EN;
  std::vector<int> F;     // nf x 3
  std::vector<int> EMAP;  // 3*nf: EMAP[c*nf + f] = unique edge of the directed edge opposite corner c of face f
  std::vector<TreeNode> tree;  // igl::AABB, root = 0
};

// igl::AABB<DerivedV,3>::init(V, Ele, SI, I) (AABB.cpp:106-200)
int tree_init(Mesh& m, const std::vector<int>& SI, const std::vector<int>& I) {
  const int idx = (int)m.tree.size();
  m.tree.emplace_back();
  F3 lo = f3(FLT_MAX, FLT_MAX, FLT_MAX), hi = f3(-FLT_MAX, -FLT_MAX, -FLT_MAX);  // AlignedBox::setEmpty
  for (int f : I)
    for (int c = 0; c < 3; ++c) {
      const F3 v = m.V[m.F[3 * f + c]];
      lo = f3(std::min(lo.x, v.x), std::min(lo.y, v.y), std::min(lo.z, v.z));
      hi = f3(std::max(hi.x, v.x), std::max(hi.y, v.y), std::max(hi.z, v.z));
    }
  m.tree[idx].lo = lo, m.tree[idx].hi = hi;
  if (I.size() == 1) {
    m.tree[idx].prim = I[0];
    return idx;
  }
  const float diag[3] = {hi.x - lo.x, hi.y - lo.y, hi.z - lo.z};
  int max_d = 0;  // maxCoeff(&max_d): the first of equal maxima
  for (int d = 1; d < 3; ++d)
    if (diag[d] > diag[max_d]) max_d = d;
  std::vector<int> r(I.size());
  for (size_t i = 0; i < I.size(); ++i) r[i] = SI[3 * (size_t)I[i] + max_d];
  std::vector<int> tmp = r;
  const size_t n = (tmp.size() - 1) / 2;
  std::nth_element(tmp.begin(), tmp.begin() + n, tmp.end());
  const int med = tmp[n];
  std::vector<int> LI, RI;
  for (size_t i = 0; i < I.size(); ++i) (r[i] <= med ? LI : RI).push_back(I[i]);
  if (!LI.empty()) {
    const int l = tree_init(m, SI, LI);
    m.tree[idx].left = l;
  }
  if (!RI.empty()) {
    const int rr = tree_init(m, SI, RI);
    m.tree[idx].right = rr;
  }
  return idx;
}

// AABB.cpp:73-101: barycentres (barycenter.cpp:15-33), per-axis ranks by igl::sort (sort.cpp:287-316: std::sort of an
// index map with IndexLessThan)
void tree_build(Mesh& m) {
  m.tree.clear();
  if (m.nf == 0) return;
  std::vector<int> SI(3 * (size_t)m.nf);
  for (int d = 0; d < 3; ++d) {
    std::vector<float> data(m.nf);
    for (int f = 0; f < m.nf; ++f) {
      const F3 a = m.V[m.F[3 * f]], b = m.V[m.F[3 * f + 1]], c = m.V[m.F[3 * f + 2]];
      const F3 bc = f3(((a.x + b.x) + c.x) / 3.0f, ((a.y + b.y) + c.y) / 3.0f, ((a.z + b.z) + c.z) / 3.0f);
      data[f] = d == 0 ? bc.x : (d == 1 ? bc.y : bc.z);
    }
    std::vector<size_t> index_map(m.nf);
    for (int f = 0; f < m.nf; ++f) index_map[f] = f;
    std::sort(index_map.begin(), index_map.end(), [&data](size_t a, size_t b) { return data[a] < data[b]; });
    for (int i = 0; i < m.nf; ++i) SI[3 * index_map[i] + d] = i;
  }
  std::vector<int> all(m.nf);
  for (int f = 0; f < m.nf; ++f) all[f] = f;
  tree_init(m, SI, all);
}

void mesh_normals(Mesh& m) {
  const int nf = m.nf, nv = m.nv;
  // per_face_normals.cpp:22-37 (Z = 0)
  m.FN.resize(nf);
  for (int i = 0; i < nf; ++i) {
    const F3 v1 = m.V[m.F[3 * i + 1]] - m.V[m.F[3 * i + 0]];
    const F3 v2 = m.V[m.F[3 * i + 2]] - m.V[m.F[3 * i + 0]];
    F3 n = cross3(v1, v2);
    const float r = norm3(n);
    if (r == 0)
      n = f3(0, 0, 0);
    else
      n = f3(n.x / r, n.y / r, n.z / r);
    m.FN[i] = n;
  }
  // per_vertex_normals.cpp:69-107 with PER_VERTEX_NORMALS_WEIGHTING_TYPE_ANGLE: W = internal_angles (float),
  // squared_edge_lengths.cpp:36-40, internal_angles.cpp:75-86
  m.VN.assign(nv, f3(0, 0, 0));
  for (int i = 0; i < nf; ++i) {
    const F3 a = m.V[m.F[3 * i + 0]], b = m.V[m.F[3 * i + 1]], c = m.V[m.F[3 * i + 2]];
    const float L[3] = {sqn3(b - c), sqn3(c - a), sqn3(a - b)};
    for (int j = 0; j < 3; ++j) {
      const float s1 = L[j], s2 = L[(j + 1) % 3], s3 = L[(j + 2) % 3];
      const float w = (float)std::acos((double)((s3 + s2) - s1) / (2. * std::sqrt(s3 * s2)));
      F3& n = m.VN[m.F[3 * i + j]];
      n = n + w * m.FN[i];
    }
  }
  for (int v = 0; v < nv; ++v) {  // N.rowwise().normalize(): Dot.h:121-131, divides only when the squared norm is > 0
    const float z = sqn3(m.VN[v]);
    if (z > 0.f) {
      const float r = std::sqrt(z);
      m.VN[v] = f3(m.VN[v].x / r, m.VN[v].y / r, m.VN[v].z / r);
    }
  }
  // per_edge_normals.cpp:36-78, uniform weights: the (unnormalised) sum of the face normals around each undirected edge
  std::map<std::pair<int, int>, int> ids;
  m.EMAP.assign(3 * (size_t)nf, 0);
  for (int c = 0; c < 3; ++c)
    for (int f = 0; f < nf; ++f) {  // oriented_facets: block c holds the edge opposite corner c
      int u = m.F[3 * f + (c + 1) % 3], v = m.F[3 * f + (c + 2) % 3];
      if (u > v) std::swap(u, v);
      auto it = ids.find({u, v});
      int id;
      if (it == ids.end()) {
        id = (int)ids.size();
        ids[{u, v}] = id;
      } else
        id = it->second;
      m.EMAP[(size_t)c * nf + f] = id;
    }
  m.EN.assign(ids.size(), f3(0, 0, 0));
  for (int f = 0; f < nf; ++f)
    for (int c = 0; c < 3; ++c) {
      F3& n = m.EN[m.EMAP[(size_t)c * nf + f]];
      n = n + m.FN[f];
    }
  tree_build(m);
}

// doublearea.cpp:75-109 (three row vectors, double output) -> :144-199 (Kahan's Heron formula on sorted lengths)
double doublearea3(F3 A, F3 B, F3 C) {
  double l[3] = {(double)norm3(B - C), (double)norm3(C - A), (double)norm3(A - B)};
  std::sort(l, l + 3, [](double x, double y) { return x > y; });
  const double arg = (l[0] + (l[1] + l[2])) * (l[2] - (l[0] - l[1])) * (l[2] + (l[0] - l[1])) * (l[0] + (l[1] - l[2]));
  return 2.0 * 0.25 * std::sqrt(arg);  // NaN stays NaN (nan_replacement = NaN)
}

// pseudonormal_test.cpp:24-128; returns the sign s
float pseudonormal_sign(const Mesh& m, F3 q, int f, F3 c) {
  const F3 A = m.V[m.F[3 * f + 0]], B = m.V[m.F[3 * f + 1]], C = m.V[m.F[3 * f + 2]];
  const double area = doublearea3(A, B, C);
  const double MIN_DOUBLE_AREA = 1e-4, epsilon = 1e-12;
  F3 n = m.FN[f];
  if (area > MIN_DOUBLE_AREA) {
    // barycentric_coordinates.cpp:88-100
    const F3 v0 = B - A, v1 = C - A, v2 = c - A;
    const float d00 = dot3(v0, v0), d01 = dot3(v0, v1), d11 = dot3(v1, v1), d20 = dot3(v2, v0), d21 = dot3(v2, v1);
    const float denom = d00 * d11 - d01 * d01;
    float b[3];
    b[1] = (d11 * d20 - d01 * d21) / denom;
    b[2] = (d00 * d21 - d01 * d20) / denom;
    b[0] = 1.0f - (b[1] + b[2]);
    int type = 0;
    for (int x = 0; x < 3; ++x) type += (b[x] <= (float)epsilon) ? 1 : 0;
    switch (type) {
      case 2:
        for (int x = 0; x < 3; ++x)
          if (b[x] > (float)epsilon) {
            n = m.VN[m.F[3 * f + x]];
            break;
          }
        break;
      case 1:
        for (int x = 0; x < 3; ++x)
          if (b[x] <= (float)epsilon) {
            n = m.EN[m.EMAP[(size_t)m.nf * x + f]];
            break;
          }
        break;
      default:  // 3 (assert in debug builds) falls through to the face normal; NaN barycentrics give type 0
      case 0: n = m.FN[f]; break;
    }
  } else {
    bool found = false;
    for (int v = 0; v < 3 && !found; ++v)
      if ((double)norm3(c - m.V[m.F[3 * f + v]]) < epsilon) {
        found = true;
        n = m.VN[m.F[3 * f + v]];
      }
    for (int e = 0; e < 3 && !found; ++e) {
      const F3 s = m.V[m.F[3 * f + (e + 1) % 3]], d = m.V[m.F[3 * f + (e + 2) % 3]];
      // project_to_line.cpp:36-55 then project_to_line_segment.cpp:27-42; t and sqrD are double, the vectors float
      const F3 DmS = d - s;
      const double v_sqrlen = (double)sqn3(DmS);
      const F3 SmP = s - c;
      const F3 prod = f3(DmS.x * SmP.x, DmS.y * SmP.y, DmS.z * SmP.z);
      double t = (double)(-(prod.x + (prod.y + prod.z))) / v_sqrlen;
      const F3 projP = ((float)(1 - t)) * s + ((float)t) * d;
      double sqrD = (double)sqn3(c - projP);
      if (t < 0)
        sqrD = (double)sqn3(c - s);
      else if (t > 1)
        sqrD = (double)sqn3(c - d);
      if (std::sqrt(sqrD) < epsilon) {
        n = m.EN[m.EMAP[(size_t)m.nf * e + f]];
        found = true;
      }
    }
    if (!found) n = m.FN[f];
  }
  return dot3(q - c, n) >= 0 ? 1.f : -1.f;
}

// Eigen::AlignedBox::contains / squaredExteriorDistance (Geometry/AlignedBox.h)
inline bool box_contains(const TreeNode& n, F3 p) {
  return n.lo.x <= p.x && n.lo.y <= p.y && n.lo.z <= p.z && p.x <= n.hi.x && p.y <= n.hi.y && p.z <= n.hi.z;
}
inline float box_ext_sqdist(const TreeNode& n, F3 p) {
  float dist2 = 0.f;
  const float pv[3] = {p.x, p.y, p.z}, lo[3] = {n.lo.x, n.lo.y, n.lo.z}, hi[3] = {n.hi.x, n.hi.y, n.hi.z};
  for (int k = 0; k < 3; ++k) {
    if (lo[k] > pv[k]) {
      const float aux = lo[k] - pv[k];
      dist2 += aux * aux;
    } else if (pv[k] > hi[k]) {
      const float aux = pv[k] - hi[k];
      dist2 += aux * aux;
    }
  }
  return dist2;
}
// igl::AABB::squared_distance (AABB.cpp:360-452) with leaf_squared_distance / set_min (:768-833)
void tree_squared_distance(const Mesh& m, int node, F3 p, float low_sqr_d, float& sqr_d, int& i, F3& c) {
  const TreeNode& nd = m.tree[node];
  if (nd.prim >= 0) {
    if (low_sqr_d > sqr_d) {
      sqr_d = low_sqr_d;
      return;
    }
    const int f = nd.prim;
    const F3 cc = closest_point_triangle(p, m.V[m.F[3 * f]], m.V[m.F[3 * f + 1]], m.V[m.F[3 * f + 2]]);
    const float d = sqn3(p - cc);
    if (d < sqr_d) sqr_d = d, i = f, c = cc;
    return;
  }
  bool looked_left = false, looked_right = false;
  auto look = [&](int child, bool& flag) {
    // the child starts from the current bound and reports an improvement only (set_min is a strict comparison)
    tree_squared_distance(m, child, p, low_sqr_d, sqr_d, i, c);
    flag = true;
  };
  if (box_contains(m.tree[nd.left], p)) look(nd.left, looked_left);
  if (box_contains(m.tree[nd.right], p)) look(nd.right, looked_right);
  const float l = box_ext_sqdist(m.tree[nd.left], p), r = box_ext_sqdist(m.tree[nd.right], p);
  if (l < r) {
    if (!looked_left && l < sqr_d) look(nd.left, looked_left);
    if (!looked_right && r < sqr_d) look(nd.right, looked_right);
  } else {
    if (!looked_right && r < sqr_d) look(nd.right, looked_right);
    if (!looked_left && l < sqr_d) look(nd.left, looked_left);
  }
}

// signed_distance.cpp:125-206 for one query point
float signed_distance_point(const Mesh& m, F3 q, float low_sqr_d, float up_sqr_d, int* face_out, F3* c_out) {
  float best = up_sqr_d;  // AABB.cpp:379: sqr_d starts at up_sqr_d, strictly smaller candidates replace it
  int bi = -1;
  F3 bc = f3(0, 0, 0);
  if (m.nf > 0 && !(low_sqr_d > up_sqr_d)) tree_squared_distance(m, 0, q, low_sqr_d, best, bi, bc);
  if (face_out) *face_out = bi;
  if (c_out) *c_out = bc;
  if (best >= up_sqr_d || best <= low_sqr_d || bi < 0) {
    if (face_out) *face_out = m.nf + 1;
    return std::numeric_limits<float>::quiet_NaN();
  }
  return pseudonormal_sign(m, q, bi, bc) * std::sqrt(best);
}

void bounds_to_sqr(float lower, float upper, float* low_sqr_d, float* up_sqr_d) {
  // signed_distance.cpp:117-121, Scalar = float
  const float max_abs = std::max(std::abs(lower), std::abs(upper));
  *up_sqr_d = (float)std::pow((double)max_abs, 2.0);
  *low_sqr_d = (float)std::pow((double)std::max(max_abs - (upper - lower), 0.0f), 2.0);
}

Mesh make_mesh(const float* V, int nv, const int* F, int nf, const float* pose) {
  Mesh m;
  m.nv = nv, m.nf = nf;
  m.V.resize(nv);
  for (int i = 0; i < nv; ++i) {
    const float x = V[3 * i], y = V[3 * i + 1], z = V[3 * i + 2];
    if (pose)  // SDFchecker::transformVertices (SDFchecker.cpp:21-33): pose * [V;1], a 4x4 by 4xN float product
      m.V[i] = f3(((pose[0] * x + pose[1] * y) + pose[2] * z) + pose[3], ((pose[4] * x + pose[5] * y) + pose[6] * z) + pose[7],
                  ((pose[8] * x + pose[9] * y) + pose[10] * z) + pose[11]);
    else
      m.V[i] = f3(x, y, z);
  }
  m.F.assign(F, F + 3 * (size_t)nf);
  mesh_normals(m);
  return m;
}

void sdf_minmax(const Mesh& m, const std::vector<F3>& pts, float* min_d, float* max_d, std::vector<float>* dists) {
  // SDFchecker.cpp:115-134 with lower = -FLT_MAX, upper = FLT_MAX as every call site passes them.  S.minCoeff() /
  // maxCoeff() skip NaN here (a point exactly on the surface; Eigen's result for NaN input is unspecified).
  float lo, up;
  bounds_to_sqr(-FLT_MAX, FLT_MAX, &lo, &up);
  float mn = std::numeric_limits<float>::infinity(), mx = -std::numeric_limits<float>::infinity();
  if (dists) dists->resize(pts.size());
  for (size_t i = 0; i < pts.size(); ++i) {
    const float s = signed_distance_point(m, pts[i], lo, up, nullptr, nullptr);
    if (dists) (*dists)[i] = s;
    if (s < mn) mn = s;
    if (s > mx) mx = s;
  }
  *min_d = mn, *max_d = mx;
}

inline void mat4_mul(const float* a, const float* b, float* o) {  // row-major, ((a0 b0 + a1 b1) + a2 b2) + a3 b3
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      o[4 * i + j] = ((a[4 * i] * b[j] + a[4 * i + 1] * b[4 + j]) + a[4 * i + 2] * b[8 + j]) + a[4 * i + 3] * b[12 + j];
}
inline F3 xform(const float* T, F3 p) {  // pcl::transformPointCloud: m00 x + m01 y + m02 z + m03, left to right
  return f3(((T[0] * p.x + T[1] * p.y) + T[2] * p.z) + T[3], ((T[4] * p.x + T[5] * p.y) + T[6] * p.z) + T[7],
            ((T[8] * p.x + T[9] * p.y) + T[10] * p.z) + T[11]);
}
inline float sqdist_flann(F3 a, F3 b) {
  const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return (dx * dx + dy * dy) + dz * dz;
}
int nearest(const std::vector<F3>& cloud, F3 q, float* sq) {
  int bi = -1;
  float bd = std::numeric_limits<float>::infinity();
  for (size_t i = 0; i < cloud.size(); ++i) {
    const float d = sqdist_flann(q, cloud[i]);
    if (d < bd) bd = d, bi = (int)i;
  }
  *sq = bd;
  return bi;
}

// pcl::VoxelGrid<PointT>::applyFilter, xyz only (leaf = (l,l,l), min_points_per_voxel 0, no field filter)
void voxel_grid(const std::vector<F3>& in, float leaf, std::vector<F3>& out) {
  out.clear();
  if (in.empty()) return;
  const float inv = 1.0f / leaf;
  F3 mn = f3(FLT_MAX, FLT_MAX, FLT_MAX), mx = f3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
  bool any = false;
  for (const F3& p : in) {
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    any = true;
    mn = f3(std::min(mn.x, p.x), std::min(mn.y, p.y), std::min(mn.z, p.z));
    mx = f3(std::max(mx.x, p.x), std::max(mx.y, p.y), std::max(mx.z, p.z));
  }
  if (!any) return;
  const int minb[3] = {(int)std::floor(mn.x * inv), (int)std::floor(mn.y * inv), (int)std::floor(mn.z * inv)};
  const int maxb[3] = {(int)std::floor(mx.x * inv), (int)std::floor(mx.y * inv), (int)std::floor(mx.z * inv)};
  const int div[3] = {maxb[0] - minb[0] + 1, maxb[1] - minb[1] + 1, maxb[2] - minb[2] + 1};
  const int mul[3] = {1, div[0], div[0] * div[1]};
  struct Idx {
    unsigned idx;
    unsigned pt;
    bool operator<(const Idx& o) const { return idx < o.idx; }
  };
  std::vector<Idx> v;
  v.reserve(in.size());
  for (size_t i = 0; i < in.size(); ++i) {
    const F3& p = in[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    const int i0 = (int)std::floor(p.x * inv) - minb[0], i1 = (int)std::floor(p.y * inv) - minb[1], i2 = (int)std::floor(p.z * inv) - minb[2];
    v.push_back({(unsigned)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), (unsigned)i});
  }
  std::sort(v.begin(), v.end());
  size_t k = 0;
  while (k < v.size()) {
    size_t e = k + 1;
    while (e < v.size() && v[e].idx == v[k].idx) ++e;
    F3 s = f3(0, 0, 0);  // CentroidPoint: AccumulatorXYZ sums in an Eigen::Vector3f, divides by the count
    for (size_t j = k; j < e; ++j) s = s + in[v[j].pt];
    const float n = (float)(e - k);
    out.push_back(f3(s.x / n, s.y / n, s.z / n));
    k = e;
  }
}

std::vector<F3> planes_to_pts(const float* xyz, int n) {
  std::vector<F3> p(n);
  for (int i = 0; i < n; ++i) p[i] = f3(xyz[i], xyz[(size_t)n + i], xyz[2 * (size_t)n + i]);
  return p;
}

}  // namespace

extern "C" {

int orc_sdf_signed_distance(const float* P, int np, const float* V, int nv, const int* F, int nf, const float* pose16,
                            float lower, float upper, float* S_out, int* I_out) {
  const Mesh m = make_mesh(V, nv, F, nf, pose16);
  float lo, up;
  bounds_to_sqr(lower, upper, &lo, &up);
#pragma omp parallel for schedule(dynamic, 16)
  for (int i = 0; i < np; ++i) {
    int f = -1;
    S_out[i] = signed_distance_point(m, f3(P[3 * i], P[3 * i + 1], P[3 * i + 2]), lo, up, &f, nullptr);
    if (I_out) I_out[i] = f;
  }
  return 0;
}

int orc_voxel_downsample(const float* xyz_planes, int n, float leaf, float* out_planes, int cap, int* n_out) {
  std::vector<F3> out;
  voxel_grid(planes_to_pts(xyz_planes, n), leaf, out);
  *n_out = (int)out.size();
  const int m = std::min((int)out.size(), cap);
  for (int i = 0; i < m; ++i) out_planes[i] = out[i].x, out_planes[(size_t)cap + i] = out[i].y, out_planes[2 * (size_t)cap + i] = out[i].z;
  return (int)out.size() > cap ? -1 : 0;
}

// Scene front end of main_realdata_auto.cpp:54-96 without colours and normals:
//   Utils::readDepthImage (Utils.cpp:36-55), Utils::convert3dOrganizedRGB (Utils.cpp:79-115), PassThrough z in
//   [0.1, 2.0] (main :64-70), VoxelGrid at `leaf` (main :74), transform into the hand-base frame (main :76-77), three
//   PassThrough filters z, x, y (main :79-94; pcl::PassThrough keeps lo <= v <= hi and drops non-finite points),
//   transform back (main :96).  counts4: valid pixels, after the voxel grid, after the crop, (unused).
int orc_scene_from_depth(const unsigned short* depth_raw, int H, int W, double depth_unit, const float* K9, const float* cam_in_handbase16,
                         const float* handbase_in_cam16, float leaf, const float* crop_min3, const float* crop_max3, float* out_planes,
                         int cap, int* n_out, int* counts4) {
  std::vector<F3> cloud;
  cloud.reserve((size_t)H * W);
  for (int u = 0; u < H; ++u)
    for (int v = 0; v < W; ++v) {
      float depth = (float)((double)(float)depth_raw[(size_t)u * W + v] * depth_unit);  // `(float)depthShort * SR300_DEPTH_UNIT`, a double literal
      if (depth > 2.0 || depth < 0.1) depth = 0.0f;  // readDepthImage compares against double literals
      F3 p = f3(0, 0, 0);                             // bad_point
      if (depth > 0.1 && depth < 2.0) p = f3((float)((v - K9[2]) * depth / K9[0]), (float)((u - K9[5]) * depth / K9[4]), depth);
      if (p.z < 0.1f || p.z > 2.0f) continue;         // PassThrough on z
      cloud.push_back(p);
    }
  counts4[0] = (int)cloud.size();
  std::vector<F3> ds;
  voxel_grid(cloud, leaf, ds);
  counts4[1] = (int)ds.size();
  std::vector<F3> out;
  for (const F3& p : ds) {
    const F3 q = xform(cam_in_handbase16, p);
    if (!std::isfinite(q.x) || !std::isfinite(q.y) || !std::isfinite(q.z)) continue;
    if (q.z < crop_min3[2] || q.z > crop_max3[2]) continue;
    if (q.x < crop_min3[0] || q.x > crop_max3[0]) continue;
    if (q.y < crop_min3[1] || q.y > crop_max3[1]) continue;
    out.push_back(xform(handbase_in_cam16, q));
  }
  counts4[2] = (int)out.size(), counts4[3] = 0;
  *n_out = (int)out.size();
  const int m = std::min((int)out.size(), cap);
  for (int i = 0; i < m; ++i) out_planes[i] = out[i].x, out_planes[(size_t)cap + i] = out[i].y, out_planes[2 * (size_t)cap + i] = out[i].z;
  return (int)out.size() > cap ? -1 : 0;
}

// main_realdata_auto.cpp:156-177: the generator's input cloud from the dense hand-free cloud `object1` (with its MLS
// normals and its confidences): VoxelGrid at `leaf` over all fields (CentroidPoint: xyz averaged, normals summed and
// normalised, confidence has no accumulator), removeAllNaNFromPointCloud, flipNormalTowardsViewpoint(0,0,0), confidence
// of the nearest point of object1 (pcl::KdTreeFLANN, 1-NN).
int orc_object_segment(const float* xyz_planes, const float* nrm_planes, const float* conf, int n, float leaf, float* out_xyz, float* out_nrm,
                       float* out_conf, int cap, int* n_out) {
  const std::vector<F3> P = planes_to_pts(xyz_planes, n), N = planes_to_pts(nrm_planes, n);
  *n_out = 0;
  if (n == 0) return 0;
  // the voxel grid again, carrying the normals (same index arithmetic and std::sort as voxel_grid above)
  const float inv = 1.0f / leaf;
  F3 mn = f3(FLT_MAX, FLT_MAX, FLT_MAX), mx = f3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
  bool any = false;
  for (const F3& p : P) {
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    any = true;
    mn = f3(std::min(mn.x, p.x), std::min(mn.y, p.y), std::min(mn.z, p.z));
    mx = f3(std::max(mx.x, p.x), std::max(mx.y, p.y), std::max(mx.z, p.z));
  }
  if (!any) return 0;
  const int minb[3] = {(int)std::floor(mn.x * inv), (int)std::floor(mn.y * inv), (int)std::floor(mn.z * inv)};
  const int maxb[3] = {(int)std::floor(mx.x * inv), (int)std::floor(mx.y * inv), (int)std::floor(mx.z * inv)};
  const int div[3] = {maxb[0] - minb[0] + 1, maxb[1] - minb[1] + 1, maxb[2] - minb[2] + 1};
  const int mul[3] = {1, div[0], div[0] * div[1]};
  struct Idx {
    unsigned idx, pt;
    bool operator<(const Idx& o) const { return idx < o.idx; }
  };
  std::vector<Idx> v;
  for (int i = 0; i < n; ++i) {
    const F3& p = P[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    const int i0 = (int)std::floor(p.x * inv) - minb[0], i1 = (int)std::floor(p.y * inv) - minb[1], i2 = (int)std::floor(p.z * inv) - minb[2];
    v.push_back({(unsigned)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), (unsigned)i});
  }
  std::sort(v.begin(), v.end());
  std::vector<F3> ox, on;
  size_t k = 0;
  while (k < v.size()) {
    size_t e = k + 1;
    while (e < v.size() && v[e].idx == v[k].idx) ++e;
    F3 s = f3(0, 0, 0), sn = f3(0, 0, 0);
    for (size_t j = k; j < e; ++j) s = s + P[v[j].pt], sn = sn + N[v[j].pt];
    const float cnt = (float)(e - k);
    const F3 c = f3(s.x / cnt, s.y / cnt, s.z / cnt);
    const float z = sqn3(sn);  // AccumulatorNormal::get: normal.normalized()
    if (z > 0.f) {
      const float r = std::sqrt(z);
      sn = f3(sn.x / r, sn.y / r, sn.z / r);
    }
    if (std::isfinite(c.x) && std::isfinite(c.y) && std::isfinite(c.z)) ox.push_back(c), on.push_back(sn);  // Utils.cpp:477-497
    k = e;
  }
  *n_out = (int)ox.size();
  const int m = std::min((int)ox.size(), cap);
  for (int i = 0; i < m; ++i) {
    F3 nn = on[i];
    // pcl::flipNormalTowardsViewpoint (features/normal_3d.h): vp - point, flip when the cosine is negative
    const float vx = 0.f - ox[i].x, vy = 0.f - ox[i].y, vz = 0.f - ox[i].z;
    const float cos_theta = (vx * nn.x + vy * nn.y + vz * nn.z);
    if (cos_theta < 0) nn = f3(nn.x * -1, nn.y * -1, nn.z * -1);
    float sq;
    const int j = nearest(P, ox[i], &sq);
    out_xyz[i] = ox[i].x, out_xyz[(size_t)cap + i] = ox[i].y, out_xyz[2 * (size_t)cap + i] = ox[i].z;
    out_nrm[i] = nn.x, out_nrm[(size_t)cap + i] = nn.y, out_nrm[2 * (size_t)cap + i] = nn.z;
    out_conf[i] = j >= 0 ? conf[j] : 0.f;
  }
  return (int)ox.size() > cap ? -1 : 0;
}

// Hand::setCurScene after handbaseICP (Hand.cpp:289-321): the 3 mm hand-region cloud moved into the hand-base frame
// (pcl::transformPointCloudWithNormals), RadiusOutlierRemoval (0.02 m / 30, then 0.04 m / 100: a point stays when the
// radius search -- which finds the point itself too, squared distance strictly below r^2 as FLANN's RadiusResultSet
// tests it -- returns more than min_pts), StatisticalOutlierRemoval (mean distance to the 20 nearest neighbours, kept
// when <= mean + 2 stddev, statistical_outlier_removal.hpp), pass-through x in [-0.25, -0.1].
// keep_noise[n] / keep_swivel[n]: flags of the input points in scene_hand_region_removed_noise / scene_remove_swivel.
static void radius_filter(const std::vector<F3>& in, float radius, int min_pts, std::vector<F3>& out, std::vector<int>& idx_io) {
  std::vector<F3> res;
  std::vector<int> ridx;
  const float r2 = radius * radius;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < (int)in.size(); ++i) {
    int k = 0;
    for (size_t j = 0; j < in.size(); ++j)
      if (sqdist_flann(in[i], in[j]) < r2) ++k;
    if (!(k > min_pts)) idx_io[i] = -1;
  }
  for (size_t i = 0; i < in.size(); ++i)
    if (idx_io[i] >= 0) res.push_back(in[i]), ridx.push_back(idx_io[i]);
  out.swap(res);
  idx_io.swap(ridx);
}
int orc_hand_scene_filters(const float* xyz_planes, const float* nrm_planes, int n, const float* cam_in_handbase16, float* hb_xyz, float* hb_nrm,
                           unsigned char* keep_noise, unsigned char* keep_swivel) {
  std::vector<F3> P = planes_to_pts(xyz_planes, n), N = planes_to_pts(nrm_planes, n);
  const float* T = cam_in_handbase16;
  for (int i = 0; i < n; ++i) {
    P[i] = xform(T, P[i]);
    const F3 m = N[i];  // rotation part only (pcl::transformPointCloudWithNormals)
    N[i] = f3((T[0] * m.x + T[1] * m.y) + T[2] * m.z, (T[4] * m.x + T[5] * m.y) + T[6] * m.z, (T[8] * m.x + T[9] * m.y) + T[10] * m.z);
    hb_xyz[i] = P[i].x, hb_xyz[(size_t)n + i] = P[i].y, hb_xyz[2 * (size_t)n + i] = P[i].z;
    hb_nrm[i] = N[i].x, hb_nrm[(size_t)n + i] = N[i].y, hb_nrm[2 * (size_t)n + i] = N[i].z;
    keep_noise[i] = keep_swivel[i] = 0;
  }
  std::vector<F3> cur = P;
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  radius_filter(cur, 0.02f, 30, cur, idx);
  radius_filter(cur, 0.04f, 100, cur, idx);
  // StatisticalOutlierRemoval, mean_k 20, std_mul 2
  const int mean_k = 20, m = (int)cur.size();
  std::vector<float> distances(m, 0.f);
  if (m > mean_k) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; ++i) {
      std::vector<float> d(m);
      for (int j = 0; j < m; ++j) d[j] = sqdist_flann(cur[i], cur[j]);
      std::partial_sort(d.begin(), d.begin() + mean_k + 1, d.end());
      double dist_sum = 0.0;
      for (int k = 1; k < mean_k + 1; ++k) dist_sum += std::sqrt(d[k]);  // k = 0 is the query point
      distances[i] = (float)(dist_sum / mean_k);
    }
    double sum = 0, sq_sum = 0;
    for (int i = 0; i < m; ++i) sum += distances[i], sq_sum += (double)distances[i] * distances[i];
    const double mean = sum / (double)m;
    const double variance = (sq_sum - sum * sum / (double)m) / ((double)m - 1);
    const double thr = mean + 2.0 * std::sqrt(variance);
    std::vector<F3> res;
    std::vector<int> ridx;
    for (int i = 0; i < m; ++i)
      if (!(distances[i] > thr)) res.push_back(cur[i]), ridx.push_back(idx[i]);
    cur.swap(res), idx.swap(ridx);
  }
  for (size_t i = 0; i < cur.size(); ++i) {
    keep_noise[idx[i]] = 1;
    if (!(cur[i].x < -0.25f || cur[i].x > -0.1f)) keep_swivel[idx[i]] = 1;
  }
  return 0;
}

// pcl::VoxelGrid over xyz + normals (Utils::downsamplePointCloud on a PointXYZRGBNormal cloud): centroids and the
// normalised normal sums, in ascending voxel order
int orc_voxel_downsample_normals(const float* xyz_planes, const float* nrm_planes, int n, float leaf, float* out_xyz, float* out_nrm, int cap, int* n_out) {
  // reuse orc_object_segment's grid by giving every point confidence 0 and undoing nothing: the flip towards the
  // viewpoint is not part of the grid, so the grid is restated here once more without it
  const std::vector<F3> P = planes_to_pts(xyz_planes, n), N = planes_to_pts(nrm_planes, n);
  *n_out = 0;
  std::vector<F3> cen;
  voxel_grid(P, leaf, cen);
  if (cen.empty()) return 0;
  // second walk in the same order for the normals: same index arithmetic as voxel_grid
  const float inv = 1.0f / leaf;
  F3 mn = f3(FLT_MAX, FLT_MAX, FLT_MAX), mx = f3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
  for (const F3& p : P) {
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    mn = f3(std::min(mn.x, p.x), std::min(mn.y, p.y), std::min(mn.z, p.z));
    mx = f3(std::max(mx.x, p.x), std::max(mx.y, p.y), std::max(mx.z, p.z));
  }
  const int minb[3] = {(int)std::floor(mn.x * inv), (int)std::floor(mn.y * inv), (int)std::floor(mn.z * inv)};
  const int maxb[3] = {(int)std::floor(mx.x * inv), (int)std::floor(mx.y * inv), (int)std::floor(mx.z * inv)};
  const int div[3] = {maxb[0] - minb[0] + 1, maxb[1] - minb[1] + 1, maxb[2] - minb[2] + 1};
  const int mul[3] = {1, div[0], div[0] * div[1]};
  struct Idx {
    unsigned idx, pt;
    bool operator<(const Idx& o) const { return idx < o.idx; }
  };
  std::vector<Idx> v;
  for (int i = 0; i < n; ++i) {
    const F3& p = P[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    const int i0 = (int)std::floor(p.x * inv) - minb[0], i1 = (int)std::floor(p.y * inv) - minb[1], i2 = (int)std::floor(p.z * inv) - minb[2];
    v.push_back({(unsigned)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), (unsigned)i});
  }
  std::sort(v.begin(), v.end());
  std::vector<F3> nn;
  size_t k = 0;
  while (k < v.size()) {
    size_t e = k + 1;
    while (e < v.size() && v[e].idx == v[k].idx) ++e;
    F3 sn = f3(0, 0, 0);
    for (size_t j = k; j < e; ++j) sn = sn + N[v[j].pt];
    const float z = sqn3(sn);
    if (z > 0.f) {
      const float r = std::sqrt(z);
      sn = f3(sn.x / r, sn.y / r, sn.z / r);
    }
    nn.push_back(sn);
    k = e;
  }
  *n_out = (int)cen.size();
  const int m = std::min((int)cen.size(), cap);
  for (int i = 0; i < m; ++i) {
    out_xyz[i] = cen[i].x, out_xyz[(size_t)cap + i] = cen[i].y, out_xyz[2 * (size_t)cap + i] = cen[i].z;
    out_nrm[i] = nn[i].x, out_nrm[(size_t)cap + i] = nn[i].y, out_nrm[2 * (size_t)cap + i] = nn[i].z;
  }
  return (int)cen.size() > cap ? -1 : 0;
}

// Hand::handbaseICP, the source cloud (Hand.cpp:685-729): the 5 mm scene moved into the hand-base frame, pass-through
// x in [-0.07, 0.03] and z in [-0.18, 0.01], the finger connections removed (y1,z1 / y2,z2: translation of finger_1_1 /
// finger_2_1 in their parent).  keep[i]: whether input point i survives; hb_xyz / hb_nrm: all points in the hand-base frame.
int orc_handbase_region(const float* xyz_planes, const float* nrm_planes, int n, const float* cam_in_handbase16, float y1, float z1, float y2, float z2,
                        float* hb_xyz, float* hb_nrm, unsigned char* keep) {
  const std::vector<F3> P = planes_to_pts(xyz_planes, n), N = planes_to_pts(nrm_planes, n);
  const float* T = cam_in_handbase16;
  for (int i = 0; i < n; ++i) {
    const F3 pt = xform(T, P[i]);
    const F3 m = N[i];
    const F3 nn = f3((T[0] * m.x + T[1] * m.y) + T[2] * m.z, (T[4] * m.x + T[5] * m.y) + T[6] * m.z, (T[8] * m.x + T[9] * m.y) + T[10] * m.z);
    hb_xyz[i] = pt.x, hb_xyz[(size_t)n + i] = pt.y, hb_xyz[2 * (size_t)n + i] = pt.z;
    hb_nrm[i] = nn.x, hb_nrm[(size_t)n + i] = nn.y, hb_nrm[2 * (size_t)n + i] = nn.z;
    bool k = std::isfinite(pt.x) && std::isfinite(pt.y) && std::isfinite(pt.z);
    k = k && !(pt.x < -0.07f || pt.x > 0.03f) && !(pt.z < -0.18f || pt.z > 0.01f);
    if (k) {
      const float sq_dist1 = (pt.z - z1) * (pt.z - z1) + (pt.y - y1) * (pt.y - y1);
      const float sq_dist2 = (pt.z - z2) * (pt.z - z2) + (pt.y - y2) * (pt.y - y2);
      if (sq_dist1 <= 0.015 * 0.015) k = false;
      else if (sq_dist2 <= 0.015 * 0.015) k = false;
      else if (((pt.y >= y1 && pt.y <= y2) || (pt.y >= y2 && pt.y <= y1)) && std::abs(pt.z - z1) <= 0.01) k = false;  // :718-724 (both operands test z1)
    }
    keep[i] = k;
  }
  return 0;
}

// HandT42::adjustHandHeight, the matching loop (Hand.cpp:1010-1049): for every trial height the hand cloud (hand-base
// frame, with normals) is shifted along z; a hand point counts when its nearest scene point (hand-base frame) is within
// 5 mm and the two normals are within 45 degrees.  counts[t] = cur_match of trial t.
int orc_hand_height_matches(const float* scene_xyz, const float* scene_nrm, int n_scene, const float* hand_xyz, const float* hand_nrm, int n_hand,
                            const float* heights, int n_heights, int* counts) {
  const std::vector<F3> S = planes_to_pts(scene_xyz, n_scene), Sn = planes_to_pts(scene_nrm, n_scene);
  const std::vector<F3> Hd = planes_to_pts(hand_xyz, n_hand), Hn = planes_to_pts(hand_nrm, n_hand);
  const double cos45 = std::cos(45 / 180.0 * M_PI);
  for (int t = 0; t < n_heights; ++t) {
    int cur = 0;
#pragma omp parallel for reduction(+ : cur) schedule(static)
    for (int i = 0; i < n_hand; ++i) {
      // offset = I with (2,3) = height, applied by pcl::transformPointCloudWithNormals: x and y unchanged, z + height
      const F3 pt = f3(Hd[i].x, Hd[i].y, Hd[i].z + heights[t]);
      float sq;
      const int j = nearest(S, pt, &sq);
      if (j < 0) continue;
      if ((double)sq > 0.005 * 0.005) continue;
      if ((double)dot3(Hn[i], Sn[j]) >= cos45) ++cur;
    }
    counts[t] = cur;
  }
  return 0;
}

// PoseEstimator::rejectByCollisionOrNonTouching (PoseEstimator.cpp:524-735).  keep[i] = 1 for the hypotheses the
// reference pushes back into _pose_hypos.  diag (optional, H x 8): stage that decided (0 kept, 1 scene point inside,
// 2 hand point colliding, 3 finger cloud colliding, 4 one side not touching, 5 model inside finger), the two single-
// point distances, the min distances of the four finger clouds (NaN when skipped), the smallest model-to-finger min.
int orc_reject_by_collision(const orc_physics_args* a, const float* poses16, int H, unsigned char* keep, float* diag) {
  static const int ORDER[4] = {0, 1, 2, 3};  // std::map order: finger_1_1, finger_1_2, finger_2_1, finger_2_2
  // finger clouds in the hand-base frame (PoseEstimator.cpp:538-551)
  std::vector<F3> finger_pts[4];
  for (int k = 0; k < 4; ++k) {
    if (!a->finger_status[k]) continue;
    std::vector<F3> p = planes_to_pts(a->finger_xyz[k], a->finger_n[k]);
    for (F3& q : p) q = xform(a->finger2handbase[k], q);
    finger_pts[k] = std::move(p);
  }
  // scene without the hand, hand-base frame, 5 mm voxel grid (:553-557)
  std::vector<F3> cwh = planes_to_pts(a->cloud_without_hand_xyz, a->n_cloud_without_hand), cwh_ds;
  for (F3& q : cwh) q = xform(a->cam2handbase, q);
  voxel_grid(cwh, a->voxel_size, cwh_ds);
  const std::vector<F3> hand_cloud = planes_to_pts(a->hand_cloud_xyz, a->n_hand_cloud);
  const std::vector<F3> model = planes_to_pts(a->model_xyz, a->n_model);
  Mesh finger_mesh[4];
  for (int k = 0; k < 4; ++k) finger_mesh[k] = make_mesh(a->finger_V[k], a->finger_nv[k], a->finger_F[k], a->finger_nf[k], a->finger_mesh_pose[k]);

  const float collision_dist = std::min(-a->smallest_dim * a->collision_thres, -0.007f);
  const float inside_ob_dist = std::min(-a->smallest_dim / 5, -0.01f);
  const float non_touch_dist = a->non_touch_dist;
  const float collision_finger_dist = -a->collision_finger_dist;
  const float NaN = std::numeric_limits<float>::quiet_NaN();

#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < H; ++i) {
    float dg[8] = {0, NaN, NaN, NaN, NaN, NaN, NaN, NaN};
    auto finish = [&](int stage) {
      keep[i] = stage == 0;
      dg[0] = (float)stage;
      if (diag) std::memcpy(diag + 8 * (size_t)i, dg, sizeof(dg));
    };
    float m2h[16];
    mat4_mul(a->cam2handbase, poses16 + 16 * (size_t)i, m2h);
    // sdf.transformMesh("object", model2handbase): the reference transforms the registered vertices there and back
    // for every hypothesis (accumulating rounding); here every hypothesis starts from the registered vertices.
    const Mesh obj = make_mesh(a->object_V, a->object_nv, a->object_F, a->object_nf, m2h);
    const F3 center = f3(((m2h[0] * a->model_center_init[0] + m2h[1] * a->model_center_init[1]) + m2h[2] * a->model_center_init[2]) + m2h[3] * 1.0f,
                         ((m2h[4] * a->model_center_init[0] + m2h[5] * a->model_center_init[1]) + m2h[6] * a->model_center_init[2]) + m2h[7] * 1.0f,
                         ((m2h[8] * a->model_center_init[0] + m2h[9] * a->model_center_init[1]) + m2h[10] * a->model_center_init[2]) + m2h[11] * 1.0f);
    float mn, mx, sq;
    // :598-617 nearest scene point to the object centre inside the object
    int nn = nearest(cwh_ds, center, &sq);
    if (nn >= 0) {
      sdf_minmax(obj, {cwh_ds[nn]}, &mn, &mx, nullptr);
      dg[1] = mn;
      if (mn <= inside_ob_dist) {
        finish(1);
        continue;
      }
    }
    // :620-642 nearest hand point to the object centre
    nn = nearest(hand_cloud, center, &sq);
    if (nn >= 0 && std::sqrt(sq) < a->ob_diameter / 2) {
      sdf_minmax(obj, {hand_cloud[nn]}, &mn, &mx, nullptr);
      dg[2] = mn;
      if (mn < collision_dist) {
        finish(2);
        continue;
      }
    }
    // :646-668 finger clouds against the object
    bool rejected = false, non_touch[4] = {false, false, false, false};
    for (int kk = 0; kk < 4; ++kk) {
      const int k = ORDER[kk];
      if (!a->finger_status[k]) continue;  // not in finger_cloud_eigens
      if (!a->finger_status[0] && (k == 0 || k == 1)) continue;
      if (!a->finger_status[2] && (k == 2 || k == 3)) continue;
      sdf_minmax(obj, finger_pts[k], &mn, &mx, nullptr);
      dg[3 + k] = mn;
      if (mn <= collision_dist) {
        rejected = true;
        break;
      }
      if (mn > non_touch_dist && a->finger_status[k]) non_touch[k] = true;
    }
    if (rejected) {
      finish(3);
      continue;
    }
    if ((non_touch[0] && non_touch[1]) || (non_touch[2] && non_touch[3])) {
      finish(4);
      continue;
    }
    // :684-722 the object's points inside the finger meshes
    std::vector<F3> P(model.size());
    for (size_t j = 0; j < model.size(); ++j) P[j] = xform(m2h, model[j]);
    float smallest = std::numeric_limits<float>::infinity();
    for (int k = 0; k < 4 && !rejected; ++k) {
      std::vector<float> d;
      sdf_minmax(finger_mesh[k], P, &mn, &mx, &d);
      smallest = std::min(smallest, mn);
      if (mn < collision_finger_dist) {
        rejected = true;
        break;
      }
      int num_inside = 0;
      for (float v : d)
        if (v < 0) num_inside++;
      // `num_inside/P.rows() > ratio`: integer division (int / Eigen::Index), :715
      if ((float)((long)num_inside / (long)std::max<size_t>(P.size(), 1)) > a->collision_finger_volume_ratio) {
        rejected = true;
        break;
      }
    }
    dg[7] = smallest;
    finish(rejected ? 5 : 0);
  }
  return 0;
}

}  // extern "C"
