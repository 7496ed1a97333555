// TEST INFRASTRUCTURE ONLY -- see hop_oracle.h for scope, pinning status and conventions.
//
// A literal, serial CPU restatement of the reference's hot path.  Every function cites the reference
// file:line it follows (paths relative to the reference root).  The code is deliberately plain: no GPU,
// no batching tricks; clarity over speed (a kd-tree variant exists only so the CPU baseline in
// bench.py is not penalised by brute force).
#include "hop_oracle.h"

#include <algorithm>
#include <array>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <numeric>
#include <random>
#include <set>
#include <unordered_set>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------
// Small float algebra in Eigen 3.3.90's evaluation order.
//   * 3-element reductions (dot, squaredNorm, sum) are unrolled as c0 + (c1 + c2)
//     (3rdparty/Eigen/Eigen/src/Core/Redux.h:91-105, redux_novec_unroller splits in halves).
//   * normalized(): v / sqrt(squaredNorm) with a true division, identity if the norm is 0
//     (Core/Dot.h:121-131).
//   * cross(): (a1*b2-a2*b1, a2*b0-a0*b2, a0*b1-a1*b0) (Geometry/OrthoMethods.h).
// ------------------------------------------------------------------------------------------------
struct V3 {
  float x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline float sqnorm(V3 a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }
inline float norm(V3 a) { return std::sqrt(sqnorm(a)); }
inline V3 normalized(V3 a) {
  const float z = sqnorm(a);
  if (z > 0.f) return a / std::sqrt(z);
  return a;
}
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct M3 {
  float m[3][3];  // m[row][col]
};
inline V3 row(const M3& a, int r) { return {a.m[r][0], a.m[r][1], a.m[r][2]}; }
inline V3 col(const M3& a, int c) { return {a.m[0][c], a.m[1][c], a.m[2][c]}; }
// lazy coefficient product: coeff(i,j) = (lhs.row(i).cwiseProduct(rhs.col(j))).sum()
// (Core/ProductEvaluators.h:544-547) -> 3-element tree reduction.
inline M3 mul(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = dot(row(a, i), col(b, j));
  return r;
}
inline M3 transpose(const M3& a) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  return r;
}
inline V3 mul(const M3& a, V3 v) { return {dot(row(a, 0), v), dot(row(a, 1), v), dot(row(a, 2), v)}; }

struct M4 {
  float m[4][4];
};
inline M4 identity4() {
  M4 r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) r.m[i][j] = i == j ? 1.f : 0.f;
  return r;
}
inline M4 load4(const float* p) {
  M4 r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) r.m[i][j] = p[4 * i + j];
  return r;
}
inline void store4(const M4& a, float* p) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) p[4 * i + j] = a.m[i][j];
}

// (mat * p.homogeneous()).head<3>() with mat = Ref<const Matrix4f> (cse.hpp:390).  Eigen evaluates
// Matrix * Homogeneous as  block<4,3>(mat) * p  (vectorised over the 4 rows: sequential k = 0,1,2
// accumulation, Core/ProductEvaluators.h etor_product_packet_impl) followed by  += mat.col(3)
// (Geometry/Homogeneous.h homogeneous_left_product_impl::evalTo).  Checked bit for bit against the
// reference build by tests/test_oracle_vs_ref.py::test_probe_transform.
inline V3 transform_point(const M4& T, V3 p) {
  V3 r;
  r.x = ((T.m[0][0] * p.x + T.m[0][1] * p.y) + T.m[0][2] * p.z) + T.m[0][3];
  r.y = ((T.m[1][0] * p.x + T.m[1][1] * p.y) + T.m[1][2] * p.z) + T.m[1][3];
  r.z = ((T.m[2][0] * p.x + T.m[2][1] * p.y) + T.m[2][2] * p.z) + T.m[2][3];
  return r;
}

// pcl::transformPointCloudWithNormals (PCL 1.9 common/impl/transforms.hpp): scalar code,
// x' = m00*x + m01*y + m02*z + m03 evaluated left to right; normals use the 3x3 block only.
inline V3 pcl_transform_point(const M4& T, V3 p) {
  V3 r;
  r.x = ((T.m[0][0] * p.x + T.m[0][1] * p.y) + T.m[0][2] * p.z) + T.m[0][3];
  r.y = ((T.m[1][0] * p.x + T.m[1][1] * p.y) + T.m[1][2] * p.z) + T.m[1][3];
  r.z = ((T.m[2][0] * p.x + T.m[2][1] * p.y) + T.m[2][2] * p.z) + T.m[2][3];
  return r;
}
inline V3 pcl_rotate_normal(const M4& T, V3 n) {
  V3 r;
  r.x = (T.m[0][0] * n.x + T.m[0][1] * n.y) + T.m[0][2] * n.z;
  r.y = (T.m[1][0] * n.x + T.m[1][1] * n.y) + T.m[1][2] * n.z;
  r.z = (T.m[2][0] * n.x + T.m[2][1] * n.y) + T.m[2][2] * n.z;
  return r;
}

// ------------------------------------------------------------------------------------------------
// acosf as glibc 2.35 computes it on x86-64 (sysdeps/ieee754/flt-32/e_acosf.c, the fdlibm float
// routine; no multiarch variant exists for acosf).  Restated so that the HIP side can run the very
// same operation sequence; tests/test_oracle.py::test_acosf_matches_libm checks it against
// std::acos over 2^24 inputs and all breakpoints.  Third-party algorithm: FreeBSD msun / fdlibm
// e_acosf.c as imported in glibc 2.35 (Ubuntu 22.04, the image's libc).
// ------------------------------------------------------------------------------------------------
inline float asfloat(uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
inline uint32_t asuint(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
float acosf_fdlibm(float x) {
  static const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f,
                     pio2_lo = 7.5497894159e-08f, pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f,
                     pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f,
                     pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f,
                     qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
  float z, p, q, r, w, s, c, df;
  int32_t hx, ix;
  hx = (int32_t)asuint(x);
  ix = hx & 0x7fffffff;
  if (ix == 0x3f800000) { /* |x|==1 */
    if (hx > 0) return 0.0f;       /* acos(1) = 0  */
    return pi + (float)2.0 * pio2_lo; /* acos(-1)= pi */
  } else if (ix > 0x3f800000) { /* |x| >= 1 */
    return (x - x) / (x - x);   /* acos(|x|>1) is NaN */
  }
  if (ix < 0x3f000000) { /* |x| < 0.5 */
    if (ix <= 0x32800000) return pio2_hi + pio2_lo; /*if|x|<=2**-26*/
    z = x * x;
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    r = p / q;
    return pio2_hi - (x - (pio2_lo - x * r));
  } else if (hx < 0) { /* x < -0.5 */
    z = (one + x) * (float)0.5;
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    s = std::sqrt(z);
    r = p / q;
    w = r * s - pio2_lo;
    return pi - (float)2.0 * (s + w);
  } else { /* x > 0.5 */
    int32_t idf;
    z = (one - x) * (float)0.5;
    s = std::sqrt(z);
    df = s;
    idf = (int32_t)asuint(df);
    df = asfloat((uint32_t)(idf & 0xfffff000));
    c = (z - df * df) / (s + df);
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    r = p / q;
    w = r * s + c;
    return (float)2.0 * (df + w);
  }
}

// ------------------------------------------------------------------------------------------------
// Point with normal and sampling probability (gr::Point3D, shared.h:61-119).
// ------------------------------------------------------------------------------------------------
struct Pt {
  V3 pos;
  V3 nrm;  // normalised once, as Point3D::set_normal does (shared.h:86-88)
  float prob;
};

inline Pt make_pt(V3 p, V3 raw_normal, float prob) { return {p, normalized(raw_normal), prob}; }

// matchBase.hpp:31-43
inline int ppf_closest_bin(int value, int discretization) {
  int lower_limit = value - (value % discretization);
  int upper_limit = lower_limit + discretization;
  int dist_from_lower = value - lower_limit;
  int dist_from_upper = upper_limit - value;
  return (dist_from_lower < dist_from_upper) ? lower_limit : upper_limit;
}

// static_cast<int>(double) as x86-64 cvttsd2si does it: NaN and out-of-range give INT_MIN.  The
// reference hits this when a dot product of unit vectors rounds above 1 (acos -> NaN).
inline int x86_double_to_int(double v) {
  if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
  return (int)v;
}
inline int x86_float_to_int(float v) {
  if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT_MIN;
  return (int)v;
}

// gr::computePPF, matchBase.hpp:47-68.  Note the normals are normalised twice more here.
void compute_ppf(const Pt& pt1, const Pt& pt2, int key[4]) {
  const int DIST_DISCRET = 5, ANGLE_DISCRET = 10;  // floats converted to int at the call
  V3 n1 = normalized(pt1.nrm);
  V3 n2 = normalized(pt2.nrm);
  n1 = normalized(n1);  // n1.normalize()
  n2 = normalized(n2);
  const V3 p1 = pt1.pos, p2 = pt2.pos;
  const int dist = x86_float_to_int(norm(p1 - p2) * 1000.f);
  const V3 p1p2 = p2 - p1;
  const V3 d = normalized(p1p2);
  const int n1_p1p2 = x86_double_to_int((double)acosf_fdlibm(dot(n1, d)) / M_PI * 180);
  const int n2_p1p2 = x86_double_to_int((double)acosf_fdlibm(dot(n2, d)) / M_PI * 180);
  const int n1_n2 = x86_double_to_int((double)acosf_fdlibm(dot(n1, n2)) / M_PI * 180);
  // INT_MIN % d is fine (no overflow); INT_MIN - rem + d stays in range.
  key[0] = ppf_closest_bin(dist, DIST_DISCRET);
  key[1] = ppf_closest_bin(n1_p1p2, ANGLE_DISCRET);
  key[2] = ppf_closest_bin(n2_p1p2, ANGLE_DISCRET);
  key[3] = ppf_closest_bin(n1_n2, ANGLE_DISCRET);
}

struct KeySet {
  std::set<std::array<int, 4>> keys;
  bool has(const int k[4]) const { return keys.count({k[0], k[1], k[2], k[3]}) != 0; }
};

// gr::pairPPFisGood, PointPairFilter.h:17-38
bool pair_ppf_is_good(const Pt& p, const Pt& q, const Pt& b0, const Pt& b1) {
  const float length1 = norm(p.pos - q.pos);
  const float length2 = norm(b0.pos - b1.pos);
  if ((double)std::fabs(length1 - length2) > 5e-3) return false;
  const V3 pq = normalized(q.pos - p.pos);
  const V3 b0_b1 = normalized(b1.pos - b0.pos);
  const float pq_np = (float)((double)acosf_fdlibm(std::fabs(dot(pq, p.nrm))) / M_PI * 180);
  const float pq_nq = (float)((double)acosf_fdlibm(std::fabs(dot(pq, q.nrm))) / M_PI * 180);
  const float b0_b1_n0 = (float)((double)acosf_fdlibm(std::fabs(dot(b0_b1, b0.nrm))) / M_PI * 180);
  const float b0_b1_n1 = (float)((double)acosf_fdlibm(std::fabs(dot(b0_b1, b1.nrm))) / M_PI * 180);
  const float np_nq = (float)((double)acosf_fdlibm(dot(p.nrm, q.nrm)) / M_PI * 180);
  const float n0_n1 = (float)((double)acosf_fdlibm(dot(b0.nrm, b1.nrm)) / M_PI * 180);
  // NaN compares false: a NaN feature never rejects (same as the reference).
  if (std::fabs(pq_np - b0_b1_n0) > 30 || std::fabs(pq_nq - b0_b1_n1) > 30 || std::fabs(np_nq - n0_n1) > 30)
    return false;
  return true;
}

// MatchBase::ComputeRigidTransformation, matchBase.hpp:229-377 (computeScale=false, max_angle<0).
// Returns false where the reference returns false; the reference's "return FLT_MAX" exits convert to
// `true` with rms = FLT_MAX and an untouched transform (matchBase.hpp:268,282-293), mirrored here.
bool compute_rigid(const V3 ref[3], const V3 cand[3], V3 centroid1, V3 centroid2, M4& transform, float& rms_,
                   size_t ref_size = 4) {
  rms_ = std::numeric_limits<float>::max();
  const float kSmallNumber = 1e-6;
  const V3 p0 = ref[0], p1 = ref[1], p2 = ref[2];
  const V3 q0 = cand[0], q1 = cand[1], q2 = cand[2];

  V3 vector_p1 = p1 - p0;
  if (sqnorm(vector_p1) == 0) return true;
  vector_p1 = normalized(vector_p1);  // normalize(): same arithmetic
  V3 vector_p2 = (p2 - p0) - (dot(p2 - p0, vector_p1)) * vector_p1;
  if (sqnorm(vector_p2) == 0) return true;
  vector_p2 = normalized(vector_p2);
  const V3 vector_p3 = cross(vector_p1, vector_p2);

  V3 vector_q1 = q1 - q0;
  if (sqnorm(vector_q1) == 0) return true;
  vector_q1 = normalized(vector_q1);
  V3 vector_q2 = (q2 - q0) - (dot(q2 - q0, vector_q1)) * vector_q1;
  if (sqnorm(vector_q2) == 0) return true;
  vector_q2 = normalized(vector_q2);
  const V3 vector_q3 = cross(vector_q1, vector_q2);

  M3 rotate_p, rotate_q;
  const V3 rp[3] = {vector_p1, vector_p2, vector_p3}, rq[3] = {vector_q1, vector_q2, vector_q3};
  for (int r = 0; r < 3; ++r) {
    rotate_p.m[r][0] = rp[r].x, rotate_p.m[r][1] = rp[r].y, rotate_p.m[r][2] = rp[r].z;
    rotate_q.m[r][0] = rq[r].x, rotate_q.m[r][1] = rq[r].y, rotate_q.m[r][2] = rq[r].z;
  }
  const M3 rotation = mul(transpose(rotate_p), rotate_q);

  // (rotation.transpose()*rotation).isIdentity(1e-6): Core/CwiseNullaryOp.h isIdentity ->
  // diagonal: isApprox(c,1,prec) = |c-1| <= min(|c|,1)*prec ; off: isMuchSmallerThan(c,1,prec) = |c| <= prec
  const M3 rtr = mul(transpose(rotation), rotation);
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) {
      const float c = rtr.m[i][j];
      if (i == j) {
        if (!(std::fabs(c - 1.f) <= std::min(std::fabs(c), 1.f) * kSmallNumber)) return false;
      } else {
        if (!(std::fabs(c) <= 1.f * kSmallNumber)) return false;
      }
    }

  rms_ = 0.f;
  for (int i = 0; i < 3; ++i) {
    const V3 first = 1.f * cand[i] - centroid2;  // scaleEst*candidate[i].pos() - centroid2
    const V3 transformed = mul(rotation, first);
    rms_ += norm((transformed - ref[i]) + centroid1);
  }
  rms_ /= float(ref_size);

  // etrans = Identity.scale(1).translate(c1).rotate(R).translate(-c2): with an identity start the
  // first three steps are exact, the last one is  t = c1 + R*(-c2)  (Geometry/Transform.h translate()).
  transform = identity4();
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) transform.m[i][j] = rotation.m[i][j];
  const V3 t = centroid1 + mul(rotation, neg(centroid2));
  transform.m[0][3] = t.x, transform.m[1][3] = t.y, transform.m[2][3] = t.z;
  return true;
}

// ------------------------------------------------------------------------------------------------
// Exact nearest-neighbour helpers.  Squared distance as Eigen's (a-b).squaredNorm(): dx^2+(dy^2+dz^2).
// The kd-tree is an accelerator with brute-force semantics (lowest index wins exact ties).
// ------------------------------------------------------------------------------------------------
inline float sqdist_eigen(V3 a, V3 b) { return sqnorm(a - b); }
// FLANN L2_Simple as pcl::KdTreeFLANN uses it: result += diff*diff, x then y then z.
inline float sqdist_flann(V3 a, V3 b) {
  const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return (dx * dx + dy * dy) + dz * dz;
}

template <bool FLANN>
struct KdTree {
  struct Node {
    int lo, hi;       // point range (leaf) in order[]
    int left, right;  // children (-1 leaf)
    int dim;
    float split;
    float bmin[3], bmax[3];
  };
  std::vector<V3> pts;
  std::vector<int> order;
  std::vector<Node> nodes;
  static float dist(V3 a, V3 b) { return FLANN ? sqdist_flann(a, b) : sqdist_eigen(a, b); }
  void build(const std::vector<V3>& p) {
    pts = p;
    order.resize(p.size());
    std::iota(order.begin(), order.end(), 0);
    nodes.clear();
    if (!p.empty()) rec(0, (int)p.size());
  }
  int rec(int lo, int hi) {
    Node n;
    n.lo = lo, n.hi = hi, n.left = n.right = -1, n.dim = 0, n.split = 0;
    for (int k = 0; k < 3; ++k) n.bmin[k] = FLT_MAX, n.bmax[k] = -FLT_MAX;
    for (int i = lo; i < hi; ++i) {
      const float c[3] = {pts[order[i]].x, pts[order[i]].y, pts[order[i]].z};
      for (int k = 0; k < 3; ++k) n.bmin[k] = std::min(n.bmin[k], c[k]), n.bmax[k] = std::max(n.bmax[k], c[k]);
    }
    const int id = (int)nodes.size();
    nodes.push_back(n);
    if (hi - lo > 12) {
      int d = 0;
      float ext = -1;
      for (int k = 0; k < 3; ++k)
        if (n.bmax[k] - n.bmin[k] > ext) ext = n.bmax[k] - n.bmin[k], d = k;
      const int mid = (lo + hi) / 2;
      std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [&](int a, int b) {
        const float ca = d == 0 ? pts[a].x : d == 1 ? pts[a].y : pts[a].z;
        const float cb = d == 0 ? pts[b].x : d == 1 ? pts[b].y : pts[b].z;
        return ca < cb;
      });
      const int l = rec(lo, mid), r = rec(mid, hi);
      nodes[id].left = l, nodes[id].right = r, nodes[id].dim = d;
    }
    return id;
  }
  // conservative lower bound of the distance from q to the node box, with a safety factor so that
  // float rounding of the bound can never prune a point the brute-force scan would accept.
  float box_lb(const Node& n, V3 q) const {
    const float c[3] = {q.x, q.y, q.z};
    double s = 0;
    for (int k = 0; k < 3; ++k) {
      double d = 0;
      if (c[k] < n.bmin[k]) d = (double)n.bmin[k] - c[k];
      else if (c[k] > n.bmax[k]) d = (double)c[k] - n.bmax[k];
      s += d * d;
    }
    return (float)(s * (1.0 - 1e-6));
  }
  // nearest neighbour (brute-force semantics); best_d2/best_i in-out
  void nearest(V3 q, float& best_d2, int& best_i) const {
    if (nodes.empty()) return;
    int stack[64];
    int sp = 0;
    stack[sp++] = 0;
    while (sp) {
      const Node& n = nodes[stack[--sp]];
      if (box_lb(n, q) > best_d2) continue;
      if (n.left < 0) {
        for (int i = n.lo; i < n.hi; ++i) {
          const int id = order[i];
          const float d2 = dist(q, pts[id]);
          if (d2 < best_d2 || (d2 == best_d2 && id < best_i)) best_d2 = d2, best_i = id;
        }
      } else {
        const float c = n.dim == 0 ? q.x : n.dim == 1 ? q.y : q.z;
        const Node& l = nodes[n.left];
        const bool left_first = c <= l.bmax[n.dim];
        stack[sp++] = left_first ? n.right : n.left;
        stack[sp++] = left_first ? n.left : n.right;
      }
    }
  }
  bool any_within(V3 q, float sq) const {
    if (nodes.empty()) return false;
    int stack[64];
    int sp = 0;
    stack[sp++] = 0;
    while (sp) {
      const Node& n = nodes[stack[--sp]];
      if (box_lb(n, q) > sq) continue;
      if (n.left < 0) {
        for (int i = n.lo; i < n.hi; ++i)
          if (dist(q, pts[order[i]]) <= sq) return true;
      } else {
        stack[sp++] = n.right;
        stack[sp++] = n.left;
      }
    }
    return false;
  }
};

template <bool FLANN>
inline void brute_nearest(const std::vector<V3>& pts, V3 q, float& best_d2, int& best_i) {
  for (int i = 0; i < (int)pts.size(); ++i) {
    const float d2 = FLANN ? sqdist_flann(q, pts[i]) : sqdist_eigen(q, pts[i]);
    if (d2 < best_d2) best_d2 = d2, best_i = i;  // first minimal index wins
  }
}

std::vector<V3> soa_to_v3(const float* xyz, int n) {
  std::vector<V3> r(n);
  for (int i = 0; i < n; ++i) r[i] = {xyz[i], xyz[n + i], xyz[2 * n + i]};
  return r;
}

// Verify, cse.hpp:346-435: #{q : exists p, ||T q - p||^2 <= delta^2}.  (kdtree.h:339-404 accepts
// sqdist <= cl_dist, i.e. inclusive.)
int verify_count(const std::vector<V3>& P, const KdTree<false>* tree, const std::vector<V3>& Q, const M4& T,
                 float delta) {
  const float sq_eps = delta * delta;
  int good = 0;
  for (const V3& q : Q) {
    const V3 tq = transform_point(T, q);
    bool hit = false;
    if (tree) hit = tree->any_within(tq, sq_eps);
    else
      for (const V3& p : P)
        if (sqdist_eigen(tq, p) <= sq_eps) {
          hit = true;
          break;
        }
    good += hit ? 1 : 0;
  }
  return good;
}

// ------------------------------------------------------------------------------------------------
// The generator: MatchBase / CongruentSetExplorationBase / Match4pcsBase / FunctorSuper4PCS.
// ------------------------------------------------------------------------------------------------
struct BaseTrace {
  std::array<int, 4> base;
  float inv1, inv2;
  std::vector<std::pair<int, int>> pairs1, pairs2;
  std::vector<std::array<int, 4>> quads;
};

struct Matcher {
  orc_s4pcs_opts opt;
  KeySet ppfs;
  std::mt19937 randomGenerator_;     // matchBase.hpp:73  (options.randomSeed)
  std::mt19937 _point_index_engine;  // matchBase.hpp:76  (seed 0)
  std::vector<Pt> sampled_P_3D_, sampled_Q_3D_;
  std::vector<float> _point_probs;
  V3 centroid_P_{0, 0, 0}, centroid_Q_{0, 0, 0};
  float P_diameter_ = 0, max_base_diameter_ = -1;
  std::array<Pt, 4> base_3D_;
  std::vector<V3> P_pos;  // positions of sampled_P_3D_ for Verify
  KdTree<false> kd_tree_;
  // PairCreationFunctor state (pairCreationFunctor.h:129-161)
  std::vector<V3> unit_points;
  V3 _gcenter{0, 0, 0};
  float _ratio = 1.f;
  // outputs
  std::vector<M4> _pose_hypo;
  std::vector<float> _pose_lcp_scores;
  std::vector<BaseTrace> trace;
  int n_quat_fallback = 0;

  explicit Matcher(const orc_s4pcs_opts& o) : opt(o), randomGenerator_(o.random_seed), _point_index_engine(0) {}

  // ---- sampling.h:127-144 (UniformDistSampler) with its open-addressing voxel hash (:67-124)
  static void uniform_sample(const std::vector<Pt>& in, float delta, std::vector<Pt>& out) {
    const uint64_t MAGIC1 = 100000007, MAGIC2 = 161803409, MAGIC3 = 423606823, NO_DATA = 0xffffffffu;
    const int num_input = (int)in.size();
    const float scale_ = 1.0f / delta;
    std::vector<std::array<int, 3>> voxels_(num_input);
    std::vector<uint64_t> data_(num_input, NO_DATA);
    out.clear();
    for (int i = 0; i < num_input; ++i) {
      const V3 p = in[i].pos;
      std::array<int, 3> c{int(std::floor(p.x * scale_)), int(std::floor(p.y * scale_)), int(std::floor(p.z * scale_))};
      uint64_t key = (MAGIC1 * (uint64_t)(int64_t)c[0] + MAGIC2 * (uint64_t)(int64_t)c[1] + MAGIC3 * (uint64_t)(int64_t)c[2]) % data_.size();
      while (true) {
        if (data_[key] == NO_DATA) {
          voxels_[key] = c;
          break;
        } else if (voxels_[key] == c) {
          break;
        }
        key++;
        if (key == data_.size()) key = 0;
      }
      uint64_t& ind = data_[key];
      if (ind >= (uint64_t)num_input) {
        out.push_back(in[i]);
        ind = out.size();
      }
    }
  }

  // ---- matchBase.hpp:380-462
  void init(const std::vector<Pt>& P, const std::vector<Pt>& Q) {
    centroid_P_ = {0, 0, 0};
    centroid_Q_ = {0, 0, 0};
    sampled_P_3D_ = P;
    sampled_Q_3D_.clear();
    if (Q.size() > (size_t)opt.sample_size) {
      std::vector<Pt> uniform_Q;
      uniform_sample(Q, opt.delta, uniform_Q);
      std::shuffle(uniform_Q.begin(), uniform_Q.end(), randomGenerator_);
      const size_t nb = std::min(uniform_Q.size(), (size_t)opt.sample_size);
      sampled_Q_3D_.assign(uniform_Q.begin(), uniform_Q.begin() + nb);
    } else {
      sampled_Q_3D_ = Q;
    }
    _point_probs.resize(sampled_P_3D_.size());
    for (size_t i = 0; i < sampled_P_3D_.size(); ++i) _point_probs[i] = sampled_P_3D_[i].prob;
    auto center = [](std::vector<Pt>& c, V3& centroid) {
      for (const auto& p : c) centroid = centroid + p.pos;
      centroid = centroid / float(c.size());
      for (auto& p : c) p.pos = p.pos - centroid;
    };
    center(sampled_P_3D_, centroid_P_);
    center(sampled_Q_3D_, centroid_Q_);
    P_pos.resize(sampled_P_3D_.size());
    for (size_t i = 0; i < P_pos.size(); ++i) P_pos[i] = sampled_P_3D_[i].pos;
    kd_tree_.build(P_pos);
    // "diameter of P" measured on Q (matchBase.hpp:439-448)
    P_diameter_ = 0.f;
    for (int i = 0; i < 1000; ++i) {
      const int at = int(randomGenerator_() % sampled_Q_3D_.size());
      const int bt = int(randomGenerator_() % sampled_Q_3D_.size());
      const float l = norm(sampled_Q_3D_[bt].pos - sampled_Q_3D_[at].pos);
      if (l > P_diameter_) P_diameter_ = l;
    }
    // MeanDistance() (matchBase.hpp:83-107) only feeds P_mean_distance_, which nothing reads.
    max_base_diameter_ = P_diameter_;
    synch3DContent();
  }

  // ---- pairCreationFunctor.h:129-161
  void synch3DContent() {
    const size_t n = sampled_Q_3D_.size();
    unit_points.resize(n);
    V3 mn{FLT_MAX, FLT_MAX, FLT_MAX}, mx{-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = 0; i < n; ++i) {
      const V3 q = sampled_Q_3D_[i].pos;
      mn = {std::min(mn.x, q.x), std::min(mn.y, q.y), std::min(mn.z, q.z)};
      mx = {std::max(mx.x, q.x), std::max(mx.y, q.y), std::max(mx.z, q.z)};
    }
    _gcenter = (mn + mx) / 2.f;  // AlignedBox::center(): (m_min+m_max)/2
    const V3 diag = mx - mn;
    _ratio = (float)((double)std::max(diag.x, std::max(diag.y, diag.z)) + 0.001);
    const V3 half{0.5f, 0.5f, 0.5f};
    for (size_t i = 0; i < n; ++i) unit_points[i] = (sampled_Q_3D_[i].pos - _gcenter) / _ratio + half;
  }

  // ---- matchBase.hpp:111-212
  bool SelectRandomTriangle(int& base1, int& base2, int& base3, std::vector<int>& sample_pool) {
    const int number_of_points = (int)sampled_P_3D_.size();
    base1 = base2 = base3 = -1;
    std::discrete_distribution<> sampler(_point_probs.begin(), _point_probs.end());
    const int first_point = sampler(_point_index_engine);
    _point_probs[first_point] *= opt.dispersion;
    sample_pool.clear();
    std::vector<float> point_probs;
    int key[4];
    for (int i = 0; i < number_of_points; ++i) {
      if (i == first_point) continue;
      compute_ppf(sampled_P_3D_[first_point], sampled_P_3D_[i], key);
      if (ppfs.has(key)) {
        sample_pool.push_back(i);
        point_probs.push_back(_point_probs[i]);
      }
    }
    if (sample_pool.size() < 3) return false;
    const float sq_max_base_diameter_ = max_base_diameter_ * max_base_diameter_;
    for (int i = 0; (size_t)i < sample_pool.size() * sample_pool.size() / 4; ++i) {
      std::discrete_distribution<> sampler1(point_probs.begin(), point_probs.end());
      const int second_point = sampler1(_point_index_engine);
      const int third_point = sampler1(_point_index_engine);
      if (second_point == third_point) continue;
      compute_ppf(sampled_P_3D_[sample_pool[second_point]], sampled_P_3D_[sample_pool[third_point]], key);
      if (!ppfs.has(key)) continue;
      point_probs[second_point] *= opt.dispersion;
      point_probs[third_point] *= opt.dispersion;
      const V3 u = sampled_P_3D_[sample_pool[second_point]].pos - sampled_P_3D_[first_point].pos;
      const V3 w = sampled_P_3D_[sample_pool[third_point]].pos - sampled_P_3D_[first_point].pos;
      const float how_wide = dot(normalized(u), normalized(w));
      if ((double)std::fabs(how_wide) <= std::cos(45 * M_PI / 180.0) && sqnorm(u) < sq_max_base_diameter_ &&
          sqnorm(w) < sq_max_base_diameter_) {
        base1 = first_point;
        base2 = sample_pool[second_point];
        base3 = sample_pool[third_point];
        break;
      }
    }
    if (base2 == -1 || base3 == -1) return false;
    // 4th-point pool; NB the reference pushes the LOOP INDEX, not the point id (matchBase.hpp:203)
    std::vector<int> backup = sample_pool;
    sample_pool.clear();
    int k2[4], k3[4];
    for (int i = 0; i < (int)backup.size(); ++i) {
      if (backup[i] == base2 || backup[i] == base3 || backup[i] == base1) continue;
      compute_ppf(sampled_P_3D_[base2], sampled_P_3D_[backup[i]], k2);
      compute_ppf(sampled_P_3D_[base3], sampled_P_3D_[backup[i]], k3);
      if (ppfs.has(k2) && ppfs.has(k3)) sample_pool.push_back(i);
    }
    if (sample_pool.size() < 1) return false;
    return base1 != -1 && base2 != -1 && base3 != -1;
  }

  // ---- match4pcsBase.hpp:283-354
  static float distSegmentToSegment(V3 p1, V3 p2, V3 q1, V3 q2, float& invariant1, float& invariant2) {
    const float kSmallNumber = 0.0001f;
    const V3 u = p2 - p1, v = q2 - q1, w = p1 - q1;
    const float a = dot(u, u), b = dot(u, v), c = dot(v, v), d = dot(u, w), e = dot(v, w);
    const float f = a * c - b * b;
    float s1 = 0.0f, s2 = f, t1 = 0.0f, t2 = f;
    if (f < kSmallNumber) {
      s1 = 0.0f, s2 = 1.0f, t1 = e, t2 = c;
    } else {
      s1 = (b * e - c * d);
      t1 = (a * e - b * d);
      if (s1 < 0.0f) {
        s1 = 0.0f, t1 = e, t2 = c;
      } else if (s1 > s2) {
        s1 = s2, t1 = e + b, t2 = c;
      }
    }
    if (t1 < 0.0f) {
      t1 = 0.0f;
      if (-d < 0.0f) s1 = 0.0f;
      else if (-d > a) s1 = s2;
      else s1 = -d, s2 = a;
    } else if (t1 > t2) {
      t1 = t2;
      if ((-d + b) < 0.0f) s1 = 0;
      else if ((-d + b) > a) s1 = s2;
      else s1 = (-d + b), s2 = a;
    }
    invariant1 = (std::fabs(s1) < kSmallNumber ? 0.0f : s1 / s2);
    invariant2 = (std::fabs(t1) < kSmallNumber ? 0.0f : t1 / t2);
    return norm((w + (invariant1 * u)) - (invariant2 * v));
  }

  // ---- match4pcsBase.hpp:50-101
  bool TryQuadrilateral(float& invariant1, float& invariant2, int& id1, int& id2, int& id3, int& id4) {
    float min_distance = std::numeric_limits<float>::max();
    int best1 = -1, best2 = -1, best3 = -1, best4 = -1;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        if (i == j) continue;
        int k = 0;
        while (k == i || k == j) k++;
        int l = 0;
        while (l == i || l == j || l == k) l++;
        float li1, li2;
        const float sd = distSegmentToSegment(base_3D_[i].pos, base_3D_[j].pos, base_3D_[k].pos, base_3D_[l].pos, li1, li2);
        if (sd < min_distance) {
          min_distance = sd;
          best1 = i, best2 = j, best3 = k, best4 = l;
          invariant1 = li1, invariant2 = li2;
        }
      }
    if (best1 < 0 || best2 < 0 || best3 < 0 || best4 < 0) return false;
    const std::array<Pt, 4> tmp = base_3D_;
    base_3D_[0] = tmp[best1], base_3D_[1] = tmp[best2], base_3D_[2] = tmp[best3], base_3D_[3] = tmp[best4];
    const int tmpId[4] = {id1, id2, id3, id4};
    id1 = tmpId[best1], id2 = tmpId[best2], id3 = tmpId[best3], id4 = tmpId[best4];
    return true;
  }

  // ---- match4pcsBase.hpp:107-189
  bool SelectQuadrilateral(float& invariant1, float& invariant2, int& base1, int& base2, int& base3, int& base4) {
    const float kBaseTooSmall = 0.2f;
    int current_trial = 0;
    while (current_trial < 1000) {  // kNumberOfDiameterTrials
      current_trial++;
      std::vector<int> sample_pool;
      if (!SelectRandomTriangle(base1, base2, base3, sample_pool)) continue;
      const Pt b0 = base_3D_[0] = sampled_P_3D_[base1];
      const Pt b1 = base_3D_[1] = sampled_P_3D_[base2];
      const Pt b2 = base_3D_[2] = sampled_P_3D_[base3];
      const double x1 = b0.pos.x, y1 = b0.pos.y, z1 = b0.pos.z;
      const double x2 = b1.pos.x, y2 = b1.pos.y, z2 = b1.pos.z;
      const double x3 = b2.pos.x, y3 = b2.pos.y, z3 = b2.pos.z;
      const float denom = (float)(-x3 * y2 * z1 + x2 * y3 * z1 + x3 * y1 * z2 - x1 * y3 * z2 - x2 * y1 * z3 + x1 * y2 * z3);
      if (denom != 0) {
        const float A = (float)((-y2 * z1 + y3 * z1 + y1 * z2 - y3 * z2 - y1 * z3 + y2 * z3) / denom);
        const float B = (float)((x2 * z1 - x3 * z1 - x1 * z2 + x3 * z2 + x1 * z3 - x2 * z3) / denom);
        const float C = (float)((-x2 * y1 + x3 * y1 + x1 * y2 - x3 * y2 - x1 * y3 + x2 * y3) / denom);
        base4 = -1;
        float best_distance = std::numeric_limits<float>::max();
        const float too_small = (float)std::pow((double)(max_base_diameter_ * kBaseTooSmall), 2);
        for (unsigned int i = 0; i < sample_pool.size(); ++i) {
          const Pt& p = sampled_P_3D_[sample_pool[i]];
          if (sqnorm(p.pos - b0.pos) >= too_small && sqnorm(p.pos - b1.pos) >= too_small &&
              sqnorm(p.pos - b2.pos) >= too_small) {
            const float distance = (float)std::fabs((double)((A * p.pos.x + B * p.pos.y) + C * p.pos.z) - 1.0);
            if (distance < best_distance) {
              best_distance = distance;
              base4 = int(sample_pool[i]);
            }
          }
        }
        if (base4 != -1) {
          base_3D_[3] = sampled_P_3D_[base4];
          if (TryQuadrilateral(invariant1, invariant2, base1, base2, base3, base4)) return true;
        }
      }
    }
    return false;
  }

  // ---- FunctorSuper4pcs.h:79-116 + pairCreationFunctor.h:189-214.  The octree rasteriser
  // (intersectionFunctor.h:100-234) is an accelerator whose contract is "equals brute force"
  // (tests/pair_extraction.cc:224-225 of the reference); order here: i ascending, j<i ascending,
  // (i,j) before (j,i).
  void ExtractPairs(float pair_distance_f, float eps_f, int base_point1, int base_point2,
                    std::vector<std::pair<int, int>>& pairs) const {
    const double pair_distance = pair_distance_f, pair_distance_epsilon = eps_f;
    pairs.clear();
    const Pt& b0 = base_3D_[base_point1];
    const Pt& b1 = base_3D_[base_point2];
    const int n = (int)sampled_Q_3D_.size();
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < i; ++j) {
        const Pt& p = sampled_Q_3D_[j];
        const Pt& q = sampled_Q_3D_[i];
        const float distance = norm(q.pos - p.pos);
        if (std::fabs((double)distance - pair_distance) > pair_distance_epsilon) continue;
        // AdaptivePointFilter (PointPairFilter.h:88-172) with max_normal_difference, max_color_distance,
        // max_translation_distance, max_angle all negative: only pairPPFisGood, then both orientations.
        if (!pair_ppf_is_good(p, q, b0, b1)) continue;
        pairs.emplace_back(i, j);
        pairs.emplace_back(j, i);
      }
  }

  // ---- normalset.h:99-126 / normalset.hpp:111-253 helpers
  struct NormalSetGeom {
    float _nepsilon, _epsilon;
    int _egSize;
  };
  static NormalSetGeom normalset_geom(float epsilon) {
    NormalSetGeom g;
    g._nepsilon = (float)((double)(1.f / 7.f) + 0.00001);
    const int gridDepth = (int)(-std::log2(epsilon));
    g._egSize = (int)std::pow(2, gridDepth);
    g._epsilon = 1.f / g._egSize;
    return g;
  }
  static void cell_of(const NormalSetGeom& g, V3 p, int c[3]) {
    c[0] = (int)(p.x / g._epsilon), c[1] = (int)(p.y / g._epsilon), c[2] = (int)(p.z / g._epsilon);
  }
  static int index_normal(const NormalSetGeom& g, V3 n) {
    const V3 half{0.5f, 0.5f, 0.5f};
    const V3 c = (n / 2.f + half) / g._nepsilon;
    return (int)c.x + 7 * (int)c.y + 49 * (int)c.z;  // UnrollIndexLoop, utils.h:141-150
  }

  // ---- FunctorSuper4pcs.h:131-293
  bool FindCongruentQuadrilaterals(float invariant1, float invariant2, float distance_threshold2,
                                   const std::vector<std::pair<int, int>>& First_pairs,
                                   const std::vector<std::pair<int, int>>& Second_pairs,
                                   std::vector<std::array<int, 4>>& quads) {
    quads.clear();
    const float alpha = dot(normalized(base_3D_[1].pos - base_3D_[0].pos), normalized(base_3D_[3].pos - base_3D_[2].pos));
    const float eps = distance_threshold2 / _ratio;
    const NormalSetGeom g = normalset_geom(eps);
    // 1. per first pair: cell of its invariant point and its normal bin.  An element lives in the 1-ring
    //    of its cell (normalset.hpp:124-134), so "query cell contains it" == Chebyshev distance <= 1
    //    with both cells inside [0,egSize)^3 (utils.h:141-288 marks out-of-grid neighbours -1).
    struct E {
      int c[3];
      int nid;
    };
    std::vector<E> elems(First_pairs.size());
    for (size_t i = 0; i < First_pairs.size(); ++i) {
      const V3 p1 = unit_points[First_pairs[i].first], p2 = unit_points[First_pairs[i].second];
      const V3 n = normalized(p2 - p1);
      const V3 pos = p1 + invariant1 * (p2 - p1);
      cell_of(g, pos, elems[i].c);
      elems[i].nid = index_normal(g, n);
    }
    // 2. queries
    const float ac = acosf_fdlibm(alpha);                            // std::acos(cosAlpha)
    const float perimeter = (float)((double)2.f * M_PI * (double)std::atan(ac));
    const unsigned int nbSample = (unsigned int)(2 * std::ceil(perimeter * 7.f / 2.f));
    const float angleStep = (float)((double)2.f * M_PI / (double)(float)nbSample);
    const float sinAlpha = std::sin(ac);
    std::vector<V3> ring(nbSample);
    for (unsigned int a = 0; a != nbSample; ++a) {
      const float theta = float(a) * angleStep;
      ring[a] = {sinAlpha * std::cos(theta), sinAlpha * std::sin(theta), alpha};
    }
    std::set<std::pair<unsigned, unsigned>> comb;
    for (unsigned int i = 0; i < Second_pairs.size(); ++i) {
      const V3 p1 = unit_points[Second_pairs[i].first], p2 = unit_points[Second_pairs[i].second];
      const V3 pq1 = sampled_Q_3D_[Second_pairs[i].first].pos, pq2 = sampled_Q_3D_[Second_pairs[i].second].pos;
      const V3 query = p1 + invariant2 * (p2 - p1);
      const V3 queryQ = pq1 + invariant2 * (pq2 - pq1);
      const V3 queryn = normalized(p2 - p1);
      int qc[3];
      cell_of(g, query, qc);
      // Quaternion::setFromTwoVectors((0,0,1), queryn), Geometry/Quaternion.h:578-612
      const V3 v0 = normalized(V3{0.f, 0.f, 1.f});
      const V3 v1 = normalized(queryn);
      float c = dot(v1, v0);
      V3 qv;
      float qw;
      if (c < -1.f + 1e-5f) {
        // Eigen takes the null vector of [v0;v1] from a JacobiSVD here.  Analytic stand-in (same
        // direction up to sign and rounding); counted so that tests can report how often it happens.
        ++n_quat_fallback;
        c = std::max(c, -1.f);
        V3 axis = normalized(V3{-v1.y, v1.x, 0.f});
        if (sqnorm(axis) == 0.f) axis = {1.f, 0.f, 0.f};
        const float w2 = (1.f + c) * 0.5f;
        qw = std::sqrt(w2);
        qv = axis * std::sqrt(1.f - w2);
      } else {
        const V3 axis = cross(v0, v1);
        const float s = std::sqrt((1.f + c) * 2.f);
        const float invs = 1.f / s;
        qv = axis * invs;
        qw = s * 0.5f;
      }
      // colored normal bins
      bool colored[343];
      std::memset(colored, 0, sizeof(colored));
      for (unsigned int a = 0; a != nbSample; ++a) {
        // q * v : Quaternion.h:471-481
        V3 uv = cross(qv, ring[a]);
        uv = uv + uv;
        const V3 rot = (ring[a] + qw * uv) + cross(qv, uv);
        const int id = index_normal(g, normalized(rot));
        if (id >= 0 && id < 343) colored[id] = true;
      }
      for (size_t id = 0; id < elems.size(); ++id) {
        const E& e = elems[id];
        bool near = true;
        for (int k = 0; k < 3; ++k) {
          if (std::abs(e.c[k] - qc[k]) > 1) near = false;
          if (e.c[k] < 0 || e.c[k] >= g._egSize || qc[k] < 0 || qc[k] >= g._egSize) near = false;
        }
        if (!near) continue;
        if (e.nid < 0 || e.nid >= 343 || !colored[e.nid]) continue;
        const V3 pp1 = sampled_Q_3D_[First_pairs[id].first].pos, pp2 = sampled_Q_3D_[First_pairs[id].second].pos;
        const V3 invPoint = pp1 + (pp2 - pp1) * invariant1;
        // squared norm against the UNSQUARED threshold (FunctorSuper4pcs.h:277)
        if (sqnorm(queryQ - invPoint) <= distance_threshold2) comb.emplace((unsigned)id, i);
      }
    }
    for (const auto& it : comb)
      quads.push_back({First_pairs[it.first].first, First_pairs[it.first].second, Second_pairs[it.second].first,
                       Second_pairs[it.second].second});
    return !quads.empty();
  }

  // ---- match4pcsBase.hpp:207-281
  bool generateCongruents(std::array<int, 4>& base, std::vector<std::array<int, 4>>& quads, BaseTrace& t) {
    float invariant1 = 0.f, invariant2 = 0.f;
    if (!SelectQuadrilateral(invariant1, invariant2, base[0], base[1], base[2], base[3])) return false;
    const float distance1 = norm(base_3D_[0].pos - base_3D_[1].pos);
    const float distance2 = norm(base_3D_[2].pos - base_3D_[3].pos);
    const float eps = 1.0f * opt.delta;  // distance_factor * delta
    ExtractPairs(distance1, eps, 0, 1, t.pairs1);
    ExtractPairs(distance2, eps, 2, 3, t.pairs2);
    t.inv1 = invariant1, t.inv2 = invariant2;
    if (t.pairs1.empty() || t.pairs2.empty()) return false;
    if (!FindCongruentQuadrilaterals(invariant1, invariant2, eps, t.pairs1, t.pairs2, quads)) return false;
    return true;
  }

  // ---- cse.hpp:218-340
  void TryCongruentSet(const std::array<int, 4>& base, const std::vector<std::array<int, 4>>& set) {
    V3 ref[3];
    for (int i = 0; i < 3; ++i) ref[i] = sampled_P_3D_[base[i]].pos;
    const V3 centroid1 = ((ref[0] + ref[1]) + ref[2]) / 3.f;
    for (const auto& ids : set) {
      V3 cand[3];
      for (int j = 0; j < 3; ++j) cand[j] = sampled_Q_3D_[ids[j]].pos;
      const V3 centroid2 = ((cand[0] + cand[1]) + cand[2]) / 3.f;
      M4 transform;
      float rms = -1;
      // an untouched `transform` (degenerate exits) is uninitialised in the reference; rms=FLT_MAX then
      // fails the rms<delta test, so its value is never used.
      transform = identity4();
      const bool ok = compute_rigid(ref, cand, centroid1, centroid2, transform, rms);
      if (ok && rms >= 0.f && rms < 1.0f * opt.delta) {
        const int good = verify_count(P_pos, &kd_tree_, q_positions(), transform, opt.delta);
        const float lcp = float((unsigned)good) / float(sampled_Q_3D_.size());
        if (lcp > 0.0f) {
          // getGlobalTransform, cse.hpp:313-321.  The reference rebuilds rot*scale from a JacobiSVD of
          // the rotation block (computeRotationScaling); for an orthonormal block that is the block
          // itself up to rounding (<=1e-6 on the translation), which is what is used here.
          M3 R;
          for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R.m[i][j] = transform.m[i][j];
          const V3 t = (centroid1 + centroid_P_) - mul(R, centroid2 + centroid_Q_);
          M4 g = transform;
          g.m[0][3] = t.x, g.m[1][3] = t.y, g.m[2][3] = t.z, g.m[3][3] = 1.f;
          _pose_hypo.push_back(g);
          _pose_lcp_scores.push_back(lcp);
        }
      }
    }
  }

  mutable std::vector<V3> q_pos_cache;
  const std::vector<V3>& q_positions() const {
    if (q_pos_cache.size() != sampled_Q_3D_.size()) {
      q_pos_cache.resize(sampled_Q_3D_.size());
      for (size_t i = 0; i < q_pos_cache.size(); ++i) q_pos_cache[i] = sampled_Q_3D_[i].pos;
    }
    return q_pos_cache;
  }

  // ---- cse.hpp:66-125 + 133-194 + 201-216
  void ComputeTransformation(const std::vector<Pt>& P, const std::vector<Pt>& Q) {
    if (P.empty() || Q.empty()) return;
    // number_of_trials_: computed from uninitialised state in the reference, clamps to 30
    const int number_of_trials_ = opt.n_trials > 0 ? opt.n_trials : 30;
    init(P, Q);
    q_pos_cache.clear();
    int success_quadrilaterals_times = 0;
    for (int i = 0; i < number_of_trials_; ++i) {
      std::array<int, 4> base{0, 0, 0, 0};
      std::vector<std::array<int, 4>> quads;
      BaseTrace t;
      if (generateCongruents(base, quads, t)) {
        success_quadrilaterals_times++;
        t.base = base;
        t.quads = quads;
        TryCongruentSet(base, quads);
        trace.push_back(std::move(t));
      }
      const float fraction_try = float(i) / float(number_of_trials_);
      if (i > number_of_trials_ || fraction_try >= 0.99f || success_quadrilaterals_times >= opt.success_quadrilaterals)
        break;
    }
  }
};

std::vector<Pt> make_cloud(const float* xyz, const float* nrm, const float* prob, int n) {
  std::vector<Pt> c(n);
  for (int i = 0; i < n; ++i)
    c[i] = make_pt({xyz[i], xyz[n + i], xyz[2 * n + i]}, {nrm[i], nrm[n + i], nrm[2 * n + i]}, prob ? prob[i] : 1.f);
  return c;
}

// ------------------------------------------------------------------------------------------------
// Scoring: computeLCP, ICP, clusterPoses.
// ------------------------------------------------------------------------------------------------
struct Cloud {
  std::vector<V3> pos, nrm;
};
Cloud load_cloud(const float* xyz, const float* nrm, int n) {
  Cloud c;
  c.pos = soa_to_v3(xyz, n);
  c.nrm = soa_to_v3(nrm, n);
  return c;
}

// Utils::computeLCP, Utils.cpp:372-444, flags all true, weights 1.
float compute_lcp(const Cloud& scene, const KdTree<true>* scene_tree, const Cloud& model_t, bool use_tree,
                  float dist_thres, float angle_thres) {
  float cp = 0;
  KdTree<true> model_tree;
  if (use_tree) model_tree.build(model_t.pos);
  const float cos_thres = (float)std::cos((double)(angle_thres / 180.0f) * M_PI);  // std::cos(angle/180.0*M_PI)
  for (size_t i = 0; i < scene.pos.size(); ++i) {
    const V3 pt = scene.pos[i];
    float d2 = FLT_MAX;
    int idx = -1;
    if (use_tree) model_tree.nearest(pt, d2, idx);
    else brute_nearest<true>(model_t.pos, pt, d2, idx);
    if (idx >= 0 && d2 < dist_thres * dist_thres) {
      {
        V3 n1 = normalized(scene.nrm[i]);
        V3 n2 = normalized(model_t.nrm[idx]);
        if (dot(n1, n2) > cos_thres) cp += dot(n1, n2) * (1 - std::sqrt(d2) / dist_thres) * 1.0f;
      }
      {
        const V3 mp = model_t.pos[idx];
        float rd2 = FLT_MAX;
        int ridx = -1;
        if (use_tree && scene_tree) scene_tree->nearest(mp, rd2, ridx);
        else brute_nearest<true>(scene.pos, mp, rd2, ridx);
        if (ridx >= 0) {
          V3 n1 = normalized(model_t.nrm[idx]);
          V3 n2 = normalized(scene.nrm[ridx]);
          if (dot(n1, n2) > cos_thres) cp += dot(n1, n2) * (1 - std::sqrt(rd2) / dist_thres) * 1.0f;
        }
      }
    }
  }
  return cp;
}

// 6x6 symmetric positive (semi)definite solve by Cholesky in double; returns false if not SPD.
bool solve6(double A[6][6], double b[6], double x[6]) {
  double L[6][6] = {};
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[i][j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 1e-300)) return false;
        L[i][i] = std::sqrt(s);
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

inline M4 mul4(const M4& a, const M4& b) {
  M4 r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = 0;
      for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
      r.m[i][j] = s;
    }
  return r;
}
// rigid inverse via double arithmetic of a general 4x4 whose last row is (0,0,0,1)
inline M4 inverse_affine(const M4& a) {
  double m[3][3], inv[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m[i][j] = a.m[i][j];
  const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                     m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
  inv[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) / det;
  inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det;
  inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det;
  inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) / det;
  inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det;
  inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det;
  inv[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) / det;
  inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det;
  inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
  M4 r = identity4();
  for (int i = 0; i < 3; ++i) {
    double t = 0;
    for (int j = 0; j < 3; ++j) {
      r.m[i][j] = (float)inv[i][j];
      t -= inv[i][j] * (double)a.m[j][3];
    }
    r.m[i][3] = (float)t;
  }
  return r;
}

// ------------------------------------------------------------------------------------------------
// The reference's ICP minimiser, restated: PCL 1.9's TransformationEstimationPointToPlane is a
// TransformationEstimationLM (registration/impl/transformation_estimation_lm.hpp:146-197): per ICP iteration
//   Eigen::LevenbergMarquardt<Eigen::NumericalDiff<Functor>, float>::minimize(x), x = (t, quaternion xyz) = 0
// on the float residuals (warp(x) p - q) . n.  The arithmetic is Eigen's unsupported NonLinearOptimization /
// NumericalDiff modules, which ARE in the reference tree (src/OpenGR_4pcs/3rdparty/Eigen/unsupported/Eigen/src/
// NonLinearOptimization/{LevenbergMarquardt.h:208-355, lmpar.h:163-293, qrsolv.h}, NumericalDiff/NumericalDiff.h:64-122);
// oracle/ref_icp_driver.cpp compiles them in place (oracle/_ref/libref_icp.so) and tests/golden/icp_lm_*.npz holds what
// they return.  This restatement (PINNED against those vectors, tests/test_oracle_golden.py) keeps
//   * every residual and every forward-difference Jacobian entry bit-equal to what Eigen evaluates (float, the
//     operation order of PCL's warp_point_rigid_6d.h:77-106 / warp_point_rigid.h:86-93 on Eigen's SSE2 reductions:
//     a 4-element sum is (a0 + a2) + (a1 + a3)), including NumericalDiff's step h = sqrt(eps) |x_j| (or sqrt(eps));
//   * the control flow of minimizeOneStep / lmpar2 with their float constants;
// and replaces the m x 6 Householder QR by the 6 x 6 normal equations in double: R^T R = P^T J^T J P, so every
// quantity the algorithm reads -- the Gauss-Newton direction, |D p|, |J p|, the scaled gradient, the lmpar
// iteration -- is the same function of J^T J and J^T f.  Sums over the correspondences run in index order in double.
// The result differs from Eigen's float run by the rounding of its float reductions (measured on the goldens:
// see tests/test_oracle_golden.py), not bit for bit.  The GPU path (nn_mode 5) computes exactly this restatement.
// ------------------------------------------------------------------------------------------------
inline float sum4_sse2(float a0, float a1, float a2, float a3) { return (a0 + a2) + (a1 + a3); }  // Eigen predux<Packet4f>, SSE2

// WarpPointRigid6D::setParam (warp_point_rigid_6d.h:77-95): Quaternionf(0, x3, x4, x5); w = sqrt(1 - q.dot(q)); normalize; toRotationMatrix.
inline M4 lm_warp6(const float x[6]) {
  M4 T;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) T.m[i][j] = 0.f;
  T.m[0][3] = x[0], T.m[1][3] = x[1], T.m[2][3] = x[2], T.m[3][3] = 1.f;
  float qx = x[3], qy = x[4], qz = x[5], qw = 0.f;
  const float d = sum4_sse2(qx * qx, qy * qy, qz * qz, qw * qw);
  qw = std::sqrt(1 - d);
  const float nn = std::sqrt(sum4_sse2(qx * qx, qy * qy, qz * qz, qw * qw));
  qx = qx / nn, qy = qy / nn, qz = qz / nn, qw = qw / nn;
  const float tx = 2.f * qx, ty = 2.f * qy, tz = 2.f * qz;  // Geometry/Quaternion.h toRotationMatrix
  const float twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  T.m[0][0] = 1.f - (tyy + tzz), T.m[0][1] = txy - twz, T.m[0][2] = txz + twy;
  T.m[1][0] = txy + twz, T.m[1][1] = 1.f - (txx + tzz), T.m[1][2] = tyz - twx;
  T.m[2][0] = txz - twy, T.m[2][1] = tyz + twx, T.m[2][2] = 1.f - (txx + tyy);
  return T;
}
// warpPoint + TransformationEstimationPointToPlane::computeDistance: (Vector4f(p') - Vector4f(q,0)).dot(Vector4f(n,0))
inline float lm_residual(const M4& T, V3 p, V3 q, V3 n) {
  const V3 w = pcl_transform_point(T, p);
  const float dx = w.x - q.x, dy = w.y - q.y, dz = w.z - q.z;
  return sum4_sse2(dx * n.x, dy * n.y, dz * n.z, 0.f);
}

struct LmSums {  // what one pass over the correspondences returns for a parameter vector xc
  double ff;     // sum f_i^2
  double A[6][6];  // J^T J
  double g[6];     // J^T f
};
constexpr float LM_SQRT_EPS = 3.4526698300124393e-04f;  // sqrt(FLT_EPSILON) in float: ftol, xtol and NumericalDiff's eps

// f(xc) and the forward-difference Jacobian NumericalDiff::df returns at xc (NumericalDiff.h:64-122, mode Forward)
void lm_pass(const std::vector<V3>& P, const std::vector<V3>& Q, const std::vector<V3>& Nn, const float xc[6], LmSums& s,
             std::vector<float>* fvec_out = nullptr, std::vector<float>* jac_out = nullptr) {
  const M4 T0 = lm_warp6(xc);
  M4 Tj[6];
  float h[6];
  for (int j = 0; j < 6; ++j) {
    float xx[6];
    std::copy(xc, xc + 6, xx);
    h[j] = LM_SQRT_EPS * std::fabs(xc[j]);
    if (h[j] == 0.f) h[j] = LM_SQRT_EPS;
    xx[j] += h[j];
    Tj[j] = lm_warp6(xx);
  }
  s.ff = 0;
  for (int a = 0; a < 6; ++a) {
    s.g[a] = 0;
    for (int b = 0; b < 6; ++b) s.A[a][b] = 0;
  }
  if (fvec_out) fvec_out->resize(P.size());
  if (jac_out) jac_out->resize(P.size() * 6);
  for (size_t i = 0; i < P.size(); ++i) {
    const float f0 = lm_residual(T0, P[i], Q[i], Nn[i]);
    double J[6];
    for (int j = 0; j < 6; ++j) {
      const float fj = lm_residual(Tj[j], P[i], Q[i], Nn[i]);
      const float jf = (fj - f0) / h[j];
      J[j] = (double)jf;
      if (jac_out) (*jac_out)[6 * i + j] = jf;
    }
    if (fvec_out) (*fvec_out)[i] = f0;
    s.ff += (double)f0 * (double)f0;
    for (int a = 0; a < 6; ++a) {
      s.g[a] += J[a] * (double)f0;
      for (int b = 0; b <= a; ++b) s.A[a][b] += J[a] * J[b];
    }
  }
  for (int a = 0; a < 6; ++a)
    for (int b = a + 1; b < 6; ++b) s.A[a][b] = s.A[b][a];
}

// The same pass in exact arithmetic (double): f(x) = (R(x) p + t - q) . n without the float rounding of each residual, the same
// forward-difference steps.  f is linear in the 12 entries of [R | t], so sum f^2, J^T J and J^T f of ANY parameter vector follow
// from the 13 x 13 moment matrix sum v v^T, v = (n_a p_b, n_a, -q.n), of the correspondences -- which is how the GPU's nn_mode 6
// evaluates the minimiser with ONE pass per ICP iteration; this function is its per-point statement.
void lm_warp6_d(const double x[6], double T[3][4]) {
  double qx = x[3], qy = x[4], qz = x[5];
  double qw = std::sqrt(1 - (qx * qx + qy * qy + qz * qz));
  const double nn = std::sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
  qx /= nn, qy /= nn, qz /= nn, qw /= nn;
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  T[0][0] = 1 - (tyy + tzz), T[0][1] = txy - twz, T[0][2] = txz + twy, T[0][3] = x[0];
  T[1][0] = txy + twz, T[1][1] = 1 - (txx + tzz), T[1][2] = tyz - twx, T[1][3] = x[1];
  T[2][0] = txz - twy, T[2][1] = tyz + twx, T[2][2] = 1 - (txx + tyy), T[2][3] = x[2];
}
void lm_pass_exact(const std::vector<V3>& P, const std::vector<V3>& Q, const std::vector<V3>& Nn, const float xc[6], LmSums& s) {
  double T[7][3][4], h[6], x0[6];
  for (int j = 0; j < 6; ++j) x0[j] = (double)xc[j];
  lm_warp6_d(x0, T[0]);
  for (int j = 0; j < 6; ++j) {
    float hf = LM_SQRT_EPS * std::fabs(xc[j]);
    if (hf == 0.f) hf = LM_SQRT_EPS;
    double xx[6];
    std::copy(x0, x0 + 6, xx);
    xx[j] = (double)(xc[j] + hf);  // the float sum the reference forms, then exact
    h[j] = (double)hf;
    lm_warp6_d(xx, T[1 + j]);
  }
  s.ff = 0;
  for (int a = 0; a < 6; ++a) {
    s.g[a] = 0;
    for (int b = 0; b < 6; ++b) s.A[a][b] = 0;
  }
  for (size_t i = 0; i < P.size(); ++i) {
    double f[7];
    for (int k = 0; k < 7; ++k) {
      double r = 0;
      for (int a = 0; a < 3; ++a) {
        const double w = T[k][a][0] * P[i].x + T[k][a][1] * P[i].y + T[k][a][2] * P[i].z + T[k][a][3];
        const double qa = a == 0 ? Q[i].x : a == 1 ? Q[i].y : Q[i].z, na = a == 0 ? Nn[i].x : a == 1 ? Nn[i].y : Nn[i].z;
        r += (w - qa) * na;
      }
      f[k] = r;
    }
    double J[6];
    for (int j = 0; j < 6; ++j) J[j] = (f[1 + j] - f[0]) / h[j];
    s.ff += f[0] * f[0];
    for (int a = 0; a < 6; ++a) {
      s.g[a] += J[a] * f[0];
      for (int b = 0; b <= a; ++b) s.A[a][b] += J[a] * J[b];
    }
  }
  for (int a = 0; a < 6; ++a)
    for (int b = a + 1; b < 6; ++b) s.A[a][b] = s.A[b][a];
}

// Cholesky with diagonal pivoting of a 6x6 SPD matrix: R upper, R^T R = P^T A P; the pivot order is the one a
// column-pivoted QR of J takes (largest remaining column norm); rank by ColPivHouseholderQR::rank()'s rule
// (|R_ii| > |R|_max * 6 * eps_float).
struct PivChol {
  double R[6][6];
  int perm[6], rank;
};
void piv_chol(const double A[6][6], PivChol& c) {
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
      for (int i = 0; i < 6; ++i) std::swap(S[i][k], S[i][best]);
      for (int j = 0; j < 6; ++j) std::swap(S[k][j], S[best][j]);
      for (int i = 0; i < k; ++i) std::swap(c.R[i][k], c.R[i][best]);
      std::swap(c.perm[k], c.perm[best]);
    }
    const double d = std::sqrt(S[k][k]);
    c.R[k][k] = d;
    maxpiv = std::max(maxpiv, d);
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
// x = least-squares ("basic") solution of J x ~ f from the factor: z1 = R11^-1 R11^-T (P^T g)_1, x = P [z1; 0]  (lmpar.h:196-203)
void piv_chol_solve(const PivChol& c, const double g[6], double x[6]) {
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
// w^T M^-1 w and M^-1 b for SPD M (plain Cholesky, double)
bool spd_solve6(const double M[6][6], const double b[6], double x[6]) {
  double A[6][6], bb[6];
  for (int i = 0; i < 6; ++i) {
    bb[i] = b[i];
    for (int j = 0; j < 6; ++j) A[i][j] = M[i][j];
  }
  return solve6(A, bb, x);
}
inline double norm6(const double v[6]) {
  double s = 0;
  for (int i = 0; i < 6; ++i) s += v[i] * v[i];
  return std::sqrt(s);
}

// internal::lmpar2 (lmpar.h:163-293) on the normal equations: the triangular solves against R become solves
// against A = R^T R, qrsolv's least-squares problem [R; sqrt(par) D] becomes (A + par D^2) x = g.
void lm_par(const double A[6][6], const double g[6], const double diag[6], double delta, double& par, double x[6]) {
  const double dwarf = (double)FLT_MIN;
  PivChol c;
  piv_chol(A, c);
  piv_chol_solve(c, g, x);
  int iter = 0;
  double wa2[6];
  for (int j = 0; j < 6; ++j) wa2[j] = diag[j] * x[j];
  double dxnorm = norm6(wa2);
  double fp = dxnorm - delta;
  if (fp <= (double)0.1f * delta) {
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
      const double temp = std::sqrt(t2);
      parl = fp / delta / temp / temp;
    }
  }
  double wa1[6];
  for (int j = 0; j < 6; ++j) wa1[j] = g[j] / diag[j];
  const double gnorm = norm6(wa1);
  double paru = gnorm / delta;
  if (paru == 0) paru = dwarf / std::min(delta, (double)0.1f);
  par = std::max(par, parl);
  par = std::min(par, paru);
  if (par == 0) par = gnorm / dxnorm;
  while (true) {
    ++iter;
    if (par == 0) par = std::max(dwarf, (double)0.001f * paru);
    double M[6][6];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) M[i][j] = A[i][j] + (i == j ? par * diag[i] * diag[i] : 0.0);
    spd_solve6(M, g, x);
    for (int j = 0; j < 6; ++j) wa2[j] = diag[j] * x[j];
    dxnorm = norm6(wa2);
    double temp = fp;
    fp = dxnorm - delta;
    if (std::fabs(fp) <= (double)0.1f * delta || (parl == 0 && fp <= temp && temp < 0) || iter == 10) break;
    double w[6], u[6];
    for (int j = 0; j < 6; ++j) w[j] = diag[j] * (wa2[j] / dxnorm);
    spd_solve6(M, w, u);
    double t2 = 0;
    for (int j = 0; j < 6; ++j) t2 += w[j] * u[j];
    temp = std::sqrt(t2);
    const double parc = fp / delta / temp / temp;
    if (fp > 0) parl = std::max(parl, par);
    if (fp < 0) paru = std::min(paru, par);
    par = std::max(parl, par + parc);
  }
  if (iter == 0) par = 0;
}

// LevenbergMarquardt::minimize (LevenbergMarquardt.h:157-355) as a state machine around passes over the correspondences:
// begin -> [pass at xc] -> advance -> [pass at xc] -> advance ... until advance returns false.
enum { LM_STATUS_RUNNING = -1, LM_REL_REDUCTION = 1, LM_REL_ERROR = 2, LM_REL_BOTH = 3, LM_COSINUS = 4, LM_MAXFEV = 5, LM_FTOL = 6, LM_XTOL = 7, LM_GTOL = 8 };
struct LmState {
  float x[6], xc[6];
  LmSums cur;
  double diag[6], delta, par, xnorm, fnorm, gnorm, pnorm;
  float p[6];
  int iter, nfev, status, phase;
  bool chol_first = false;  // the moment form (minimiser 7): lmpar2 through the unpivoted Cholesky factor when the Jacobian is comfortably of full rank
};
bool lm_par_chol(const double A[6][6], const double g[6], const double diag[6], double delta, double& par, double x[6]);
void lm_begin(LmState& s) {
  for (int j = 0; j < 6; ++j) s.x[j] = s.xc[j] = 0.f;
  s.phase = 0, s.status = LM_STATUS_RUNNING, s.iter = 0, s.nfev = 0, s.par = 0, s.delta = 0, s.xnorm = 0, s.fnorm = 0, s.gnorm = 0, s.pnorm = 0;
}
inline double lm_scaled_norm(const double diag[6], const float v[6]) {
  double q = 0;
  for (int j = 0; j < 6; ++j) q += (diag[j] * (double)v[j]) * (diag[j] * (double)v[j]);
  return std::sqrt(q);
}
// the do { lmpar; candidate } part of minimizeOneStep (LevenbergMarquardt.h:262-275): leaves the candidate in xc
void lm_inner(LmState& s) {
  double xs[6];
  if (!(s.chol_first && lm_par_chol(s.cur.A, s.cur.g, s.diag, s.delta, s.par, xs))) lm_par(s.cur.A, s.cur.g, s.diag, s.delta, s.par, xs);
  for (int j = 0; j < 6; ++j) {
    s.p[j] = -(float)xs[j];        // wa1 = -wa1
    s.xc[j] = s.x[j] + s.p[j];     // wa2 = x + wa1 (float)
  }
  s.pnorm = lm_scaled_norm(s.diag, s.p);
  if (s.iter == 1) s.delta = std::min(s.delta, s.pnorm);
  s.phase = 1;
}
// the head of minimizeOneStep (LevenbergMarquardt.h:219-260); false = finished (status set)
bool lm_outer(LmState& s) {
  s.nfev += 7;  // NumericalDiff::df returns 1 + 6 evaluations
  double wa2[6];
  for (int j = 0; j < 6; ++j) wa2[j] = std::sqrt(s.cur.A[j][j]);
  if (s.iter == 1) {
    for (int j = 0; j < 6; ++j) s.diag[j] = wa2[j] == 0 ? 1.0 : wa2[j];
    s.xnorm = lm_scaled_norm(s.diag, s.x);
    s.delta = 100.0 * s.xnorm;
    if (s.delta == 0) s.delta = 100.0;
  }
  s.gnorm = 0;
  if (s.fnorm != 0)
    for (int j = 0; j < 6; ++j)
      if (wa2[j] != 0) s.gnorm = std::max(s.gnorm, std::fabs(s.cur.g[j] / s.fnorm / wa2[j]));
  if (s.gnorm <= 0) {
    s.status = LM_COSINUS;
    return false;
  }
  for (int j = 0; j < 6; ++j) s.diag[j] = std::max(s.diag[j], wa2[j]);
  lm_inner(s);
  return true;
}
// consumes the sums of the pass at s.xc; true = another pass at (the new) s.xc is wanted
bool lm_advance(LmState& s, const LmSums& cand) {
  if (s.phase == 0) {  // minimizeInit
    s.cur = cand;
    s.nfev = 1;
    s.fnorm = std::sqrt(cand.ff);
    s.par = 0;
    s.iter = 1;
    return lm_outer(s);
  }
  const double ftol = (double)LM_SQRT_EPS, xtol = (double)LM_SQRT_EPS, eps = (double)FLT_EPSILON;
  const double p1 = (double)0.1f, p25 = 0.25, p5 = 0.5, p75 = 0.75, p0001 = (double)1e-4f;
  ++s.nfev;
  const double fnorm1 = std::sqrt(cand.ff);
  double actred = -1;
  if (p1 * fnorm1 < s.fnorm) actred = 1.0 - (fnorm1 / s.fnorm) * (fnorm1 / s.fnorm);
  double jp2 = 0;  // |J p|^2 = p^T A p  (wa3 = R P^T wa1)
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) jp2 += (double)s.p[a] * s.cur.A[a][b] * (double)s.p[b];
  const double t1r = std::sqrt(std::max(jp2, 0.0)) / s.fnorm, t2r = std::sqrt(s.par) * s.pnorm / s.fnorm;
  const double temp1 = t1r * t1r, temp2 = t2r * t2r;
  const double prered = temp1 + temp2 / p5, dirder = -(temp1 + temp2);
  double ratio = 0;
  if (prered != 0) ratio = actred / prered;
  if (ratio <= p25) {
    double temp = p5;
    if (actred < 0) temp = p5 * dirder / (dirder + p5 * actred);
    if (p1 * fnorm1 >= s.fnorm || temp < p1) temp = p1;
    s.delta = temp * std::min(s.delta, s.pnorm / p1);
    s.par /= temp;
  } else if (!(s.par != 0 && ratio < p75)) {
    s.delta = s.pnorm / p5;
    s.par = p5 * s.par;
  }
  if (ratio >= p0001) {
    for (int j = 0; j < 6; ++j) s.x[j] = s.xc[j];
    s.cur = cand;
    s.xnorm = lm_scaled_norm(s.diag, s.x);
    s.fnorm = fnorm1;
    ++s.iter;
  }
  const bool small_red = std::fabs(actred) <= ftol && prered <= ftol && p5 * ratio <= 1.0;
  const bool small_err = s.delta <= xtol * s.xnorm;
  if (small_red && small_err) s.status = LM_REL_BOTH;
  else if (small_red) s.status = LM_REL_REDUCTION;
  else if (small_err) s.status = LM_REL_ERROR;
  else if (s.nfev >= 400) s.status = LM_MAXFEV;
  else if (std::fabs(actred) <= eps && prered <= eps && p5 * ratio <= 1.0) s.status = LM_FTOL;
  else if (s.delta <= eps * s.xnorm) s.status = LM_XTOL;
  else if (s.gnorm <= eps) s.status = LM_GTOL;
  if (s.status != LM_STATUS_RUNNING) return false;
  if (ratio < p0001) {  // do { } while (ratio < 1e-4): same Jacobian, new lmpar
    lm_inner(s);
    return true;
  }
  return lm_outer(s);
}

// TransformationEstimationLM::estimateRigidTransformation on the correspondences (P moved source, Q target, N target
// normals); fewer than 4: PCL prints an error and leaves the matrix as it was (returns false here).
bool lm_point_to_plane(const std::vector<V3>& P, const std::vector<V3>& Q, const std::vector<V3>& Nn, M4& T, float x_out[6] = nullptr,
                       int* stats = nullptr, bool exact = false) {
  if (P.size() < 4) return false;
  LmState s;
  lm_begin(s);
  LmSums sums;
  do {
    if (exact) lm_pass_exact(P, Q, Nn, s.xc, sums);
    else lm_pass(P, Q, Nn, s.xc, sums);
  } while (lm_advance(s, sums));
  T = lm_warp6(s.x);
  if (x_out) std::copy(s.x, s.x + 6, x_out);
  if (stats) stats[0] = s.status, stats[1] = s.nfev, stats[2] = s.iter;
  return true;
}

// ------------------------------------------------------------------------------------------------
// The moment form of the same minimiser (minimiser 7; what the GPU's nn_mode 7 computes, bit for bit).
//
// PCL's residual f_i(x) = n_i . (R(x) p_i + t - q_i) (transformation_estimation_point_to_plane.h:77-84) is linear in the twelve entries
// of [R | t].  With c a fixed point near the object (the translation of the hypothesis: conditioning only), p' = p - c,
//   u_i = (n_a p'_b [9 entries, a major], n_a [3], r0_i),  r0_i = n_i . (p_i - q_i),   w(x) = (R - I [9], t + (R - I) c [3], 1):
//   f_i(x) = w(x) . u_i,
// so sum f^2, J^T J and J^T f of EVERY parameter vector Eigen's minimiser evaluates -- NumericalDiff's six forward-difference columns
// included -- are quadratic forms in the 13 x 13 moment matrix M = sum_i u_i u_i^T of the ICP iteration's correspondences: one pass over
// the points per ICP iteration instead of one per function evaluation, and the algorithm is evaluated in exact arithmetic where PCL
// rounds every residual to float (lm_pass_exact above is the per-point statement of that; profiles/r0*_icp_lm_deltas.json measures both
// against Eigen's own run).
//
// What makes the form reproducible on ANY summation order (a GPU adds the terms of M in the order its lanes, wavefronts and workgroups
// happen to be laid out): u is put on a fixed binary grid first -- each component class scaled by a power of two chosen from the model's
// radius and the correspondence gate, the EXACT scaled value rounded to the nearest integer (ties to even), |integer| <= 2^bits -- and M is the EXACT integer
// matrix sum_i U_i U_i^T (products < 2^(2 bits), sums in 64 bits).  Integer addition is associative: every order gives the same M.  The
// grid is far below what the float run carries as rounding noise in its Jacobian (12 bits: 30 um on p', 2e-4 on n, 4 um on r0; the
// reference's forward differences carry 1e-3 relative), and the squared correspondence distances of the MSE stop rule are summed the
// same way.  From M on, every operation is an IEEE double operation (+ - * / sqrt fma) in a fixed order, stated once here and once in
// csrc/hop_lm_core.h -- two texts, same doubles.
// Differences to minimiser 6 (lm_pass_exact), all at rounding level: (a) the grid; (b) the moved source point of ICP iteration k is
// final_k p0 with final_k the float product of the increments so far, applied with fused multiply-adds (PCL moves a stored cloud once
// per iteration: T_k(...T_1(p0))); (c) lmpar2 runs on the unpivoted Cholesky factor of J^T J when it is comfortably of full rank.
// ------------------------------------------------------------------------------------------------
struct MomSpec {
  int k_np, k_n, k_r, k_d;  // power-of-two scales of n_a p'_b, n_a, r0 and of the squared distance
  float lim, lim_d;         // clamps (2^bits; 2^24 for the distances)
};
// largest power of two 2^k with bound * 2^k <= lim
inline int mom_scale_exp(double lim, double bound) { return std::ilogb(lim / bound); }
MomSpec mom_spec(double model_radius, float max_corr_dist, int bits) {
  MomSpec q;
  q.lim = std::ldexp(1.0f, bits), q.lim_d = 16777216.0f;
  const double gate = (double)max_corr_dist;
  q.k_n = bits;                                                     // |n_a| <= 1
  q.k_np = mom_scale_exp((double)q.lim, (model_radius + gate) * 1.01);  // |n_a p'_b| <= |p'| <= model radius + gate (p' is within the gate of a posed model point)
  q.k_r = mom_scale_exp((double)q.lim, gate);                       // |r0| <= |p - q| <= gate
  q.k_d = mom_scale_exp((double)q.lim_d, gate * gate);
  return q;
}
inline int32_t mom_quant(float v, int k, float lim) {
  float s = std::ldexp(v, k);  // exact (power of two)
  s = std::nearbyintf(s);      // ties to even
  s = std::min(std::max(s, -lim), lim);
  return (int32_t)s;
}
struct MomSums {
  int64_t M[13][13];  // lower triangle used
  int64_t d2q;
  int cnt;
  void clear() {
    std::memset(M, 0, sizeof M);
    d2q = 0, cnt = 0;
  }
};
// one accepted correspondence: p moved source point, q its target (posed model point), n the target's posed normal, d2 their squared distance
inline void mom_add(MomSums& m, const MomSpec& sp, V3 p, V3 q, V3 n, V3 ctr, float d2) {
  const float pc[3] = {p.x - ctr.x, p.y - ctr.y, p.z - ctr.z}, nv[3] = {n.x, n.y, n.z};
  const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
  const float r0 = dx * n.x + (dy * n.y + dz * n.z);
  // U = the EXACT product n_a (p'_b 2^k) -- two floats: 48 bits, exact in a double -- rounded to the nearest integer, ties to even (the GPU
  // obtains it with one fused multiply-add onto 1.5 2^23; see csrc/hop_kernels.hip momi_qp), clamped
  auto q_exact = [&](float x, float y_scaled) -> int32_t {
    const double r = std::nearbyint((double)x * (double)y_scaled);
    return (int32_t)std::min(std::max(r, -(double)sp.lim), (double)sp.lim);
  };
  int32_t U[13];
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) U[3 * a + b] = q_exact(nv[a], std::ldexp(pc[b], sp.k_np));
    U[9 + a] = q_exact(nv[a], std::ldexp(1.0f, sp.k_n));
  }
  U[12] = q_exact(r0, std::ldexp(1.0f, sp.k_r));
  for (int i = 0; i < 13; ++i)
    for (int j = 0; j <= i; ++j) m.M[i][j] += (int64_t)U[i] * (int64_t)U[j];
  m.d2q += (int64_t)mom_quant(d2, sp.k_d, sp.lim_d);
  m.cnt += 1;
}
inline int mom_exp_of(const MomSpec& sp, int i) { return i < 9 ? sp.k_np : i < 12 ? sp.k_n : sp.k_r; }
// the moment matrix in metres etc.: exact (an integer below 2^53 times a power of two)
void mom_to_double(const MomSums& m, const MomSpec& sp, double Md[13][13]) {
  for (int i = 0; i < 13; ++i)
    for (int j = 0; j <= i; ++j) Md[i][j] = Md[j][i] = std::ldexp((double)m.M[i][j], -(mom_exp_of(sp, i) + mom_exp_of(sp, j)));
}

// w(x) = (R(x) - I, t + (R(x) - I) c, 1); R from the quaternion as WarpPointRigid6D::setParam forms it (|q| = 1 in exact arithmetic)
void mom_w(const float x[6], const double c[3], double w[13]) {
  const double qx = (double)x[3], qy = (double)x[4], qz = (double)x[5];
  const double qw2 = 1.0 - (qx * qx + qy * qy + qz * qz);
  const double qw = std::sqrt(qw2);
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  w[0] = -(tyy + tzz), w[1] = txy - twz, w[2] = txz + twy;
  w[3] = txy + twz, w[4] = -(txx + tzz), w[5] = tyz - twx;
  w[6] = txz - twy, w[7] = tyz + twx, w[8] = -(txx + tyy);
  for (int a = 0; a < 3; ++a) w[9 + a] = (double)x[a] + (w[3 * a] * c[0] + w[3 * a + 1] * c[1] + w[3 * a + 2] * c[2]);
  w[12] = 1.0;
}
// sum f^2, J^T J, J^T f at xc from M, with NumericalDiff's forward differences (NumericalDiff.h:64-122): column j of the Jacobian as a
// functional on u is d_j = (w(xc + h_j e_j) - w(xc)) / h_j, h_j = sqrt(eps) |xc_j| (or sqrt(eps)), the step taken in float as the
// reference takes it.  For the three translations d_j = s_j e_{9+j} with s_j = (float step actually taken) / h_j.
void lm_eval_moments(const double M[13][13], const double c[3], const float xc[6], LmSums& s) {
  double w0[13], d[3][12], st[3];
  mom_w(xc, c, w0);
  for (int j = 0; j < 6; ++j) {
    float xx[6];
    std::copy(xc, xc + 6, xx);
    float h = LM_SQRT_EPS * std::fabs(xc[j]);
    if (h == 0.f) h = LM_SQRT_EPS;
    xx[j] += h;
    const double hinv = 1.0 / (double)h;
    if (j < 3) st[j] = ((double)xx[j] - (double)xc[j]) * hinv;
    else {
      double wj[13];
      mom_w(xx, c, wj);
      for (int k = 0; k < 12; ++k) d[j - 3][k] = (wj[k] - w0[k]) * hinv;
    }
  }
  double ff = 0, gr[3] = {0, 0, 0}, Arr[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, gt[3], Atr[3][3], Att[3][3];
  for (int i = 0; i < 13; ++i) {
    const double* row = M[i];
    double y0 = 0, z[3] = {0, 0, 0};
    for (int k = 0; k < 13; ++k) y0 += row[k] * w0[k];
    for (int v = 0; v < 3; ++v)
      for (int k = 0; k < 12; ++k) z[v] += row[k] * d[v][k];
    ff += w0[i] * y0;
    if (i < 12)
      for (int u = 0; u < 3; ++u) {
        gr[u] += d[u][i] * y0;
        for (int v = 0; v <= u; ++v) Arr[u][v] += d[u][i] * z[v];
      }
    if (i >= 9 && i < 12) {
      const int t = i - 9;
      gt[t] = st[t] * y0;
      for (int v = 0; v < 3; ++v) Atr[t][v] = st[t] * z[v];
      for (int t2 = 0; t2 <= t; ++t2) Att[t][t2] = st[t] * st[t2] * row[9 + t2];
    }
  }
  for (int u = 0; u < 6; ++u)
    for (int v = 0; v <= u; ++v) {
      double a;
      if (u < 3) a = Att[u][v];
      else if (v < 3) a = Atr[v][u - 3];
      else a = Arr[u - 3][v - 3];
      s.A[u][v] = s.A[v][u] = a;
    }
  for (int u = 0; u < 6; ++u) s.g[u] = u < 3 ? gt[u] : gr[u - 3];
  s.ff = std::max(ff, 0.0);
}

// lmpar2 (lmpar.h:163-293) when J^T J is comfortably positive definite (smallest Cholesky pivot above 2e-5 of the largest; Eigen's rank
// decision would call it full rank down to 7e-7): the triangular solves against R are solves against L, L L^T = A, and qrsolv's problem
// [R; sqrt(par) D] is the factor of A + par D^2.  Reciprocals of the pivots are formed once and multiplied with (one IEEE division per
// pivot).  false: not that case -- the caller runs the general, pivoted lm_par.
static bool canon_chol(const double A[6][6], const double diag[6], double par, double L[6][6], double Linv[6]) {
  double lmin = 1e300, lmax = 0;
  bool ok = true;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double v = A[i][j];
      if (i == j) v = std::fma(par * diag[i], diag[i], v);
      for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
      if (i == j) {
        ok = ok && v > 1e-290;
        const double d = std::sqrt(std::max(v, 1e-290));
        L[i][i] = d;
        Linv[i] = 1.0 / d;
        lmin = std::min(lmin, d), lmax = std::max(lmax, d);
      } else
        L[i][j] = v * Linv[j];
    }
  return ok && lmin > 2e-5 * lmax;
}
static void canon_fwd(const double L[6][6], const double Linv[6], const double b[6], double y[6]) {
  for (int i = 0; i < 6; ++i) {
    double v = b[i];
    for (int k = 0; k < i; ++k) v -= L[i][k] * y[k];
    y[i] = v * Linv[i];
  }
}
static void canon_bwd(const double L[6][6], const double Linv[6], const double y[6], double x[6]) {
  for (int i = 5; i >= 0; --i) {
    double v = y[i];
    for (int k = i + 1; k < 6; ++k) v -= L[k][i] * x[k];
    x[i] = v * Linv[i];
  }
}
static double canon_dnorm(const double diag[6], const double x[6], double wa2[6]) {
  double q = 0;
  for (int j = 0; j < 6; ++j) {
    wa2[j] = diag[j] * x[j];
    q = std::fma(wa2[j], wa2[j], q);
  }
  return q > 1e-290 ? std::sqrt(q) : 0.0;
}
bool lm_par_chol(const double A[6][6], const double g[6], const double diag[6], double delta, double& par_io, double x[6]) {
  const double dwarf = (double)FLT_MIN, p1 = (double)0.1f;
  double L[6][6] = {}, Linv[6], y[6], xs[6], wa2[6], w[6];
  if (!canon_chol(A, diag, 0.0, L, Linv)) return false;
  canon_fwd(L, Linv, g, y);
  canon_bwd(L, Linv, y, xs);
  double dxnorm = canon_dnorm(diag, xs, wa2);
  double fp = dxnorm - delta;
  if (fp <= p1 * delta) {
    std::copy(xs, xs + 6, x);
    par_io = 0;
    return true;
  }
  const double dinv = 1.0 / delta;
  {
    const double ninv = 1.0 / dxnorm;
    for (int j = 0; j < 6; ++j) w[j] = diag[j] * wa2[j] * ninv;
    canon_fwd(L, Linv, w, y);
  }
  double t2 = 0;
  for (int j = 0; j < 6; ++j) t2 = std::fma(y[j], y[j], t2);
  double parl = fp * dinv * (1.0 / t2);
  double gq = 0;
  for (int j = 0; j < 6; ++j) {
    const double v = g[j] * (1.0 / diag[j]);
    gq = std::fma(v, v, gq);
  }
  const double gnorm = gq > 1e-290 ? std::sqrt(gq) : 0.0;
  double paru = gnorm * dinv;
  if (paru == 0) paru = dwarf / std::min(delta, p1);
  double par = std::min(std::max(par_io, parl), paru);
  if (par == 0) par = gnorm * (1.0 / dxnorm);
  for (int iter = 1;; ++iter) {
    if (par == 0) par = std::max(dwarf, (double)0.001f * paru);
    if (!canon_chol(A, diag, par, L, Linv)) return false;
    canon_fwd(L, Linv, g, y);
    canon_bwd(L, Linv, y, xs);
    dxnorm = canon_dnorm(diag, xs, wa2);
    const double temp = fp;
    fp = dxnorm - delta;
    if (std::fabs(fp) <= p1 * delta || (parl == 0 && fp <= temp && temp < 0) || iter == 10) break;
    const double ninv = 1.0 / dxnorm;
    for (int j = 0; j < 6; ++j) w[j] = diag[j] * (wa2[j] * ninv);
    canon_fwd(L, Linv, w, y);
    double t3 = 0;
    for (int j = 0; j < 6; ++j) t3 = std::fma(y[j], y[j], t3);
    const double parc = fp * dinv * (1.0 / t3);
    if (fp > 0) parl = std::max(parl, par);
    if (fp < 0) paru = std::min(paru, par);
    par = std::max(parl, par + parc);
  }
  std::copy(xs, xs + 6, x);
  par_io = par;
  return true;
}

// estimateRigidTransformation from the moment matrix (cnt >= 4 correspondences went into it)
void lm_point_to_plane_moments(const double M[13][13], const double c[3], M4& T, float x_out[6] = nullptr, int* stats = nullptr) {
  LmState s;
  lm_begin(s);
  s.chol_first = true;
  LmSums sums;
  do lm_eval_moments(M, c, s.xc, sums);
  while (lm_advance(s, sums));
  T = lm_warp6(s.x);
  if (x_out) std::copy(s.x, s.x + 6, x_out);
  if (stats) stats[0] = s.status, stats[1] = s.nfev, stats[2] = s.iter;
}
// the composed form of the moved source: final p0 with fused multiply-adds (what the GPU's fused kernels evaluate)
inline V3 fma_transform_point(const M4& T, V3 p) {
  V3 r;
  r.x = std::fma(T.m[0][0], p.x, std::fma(T.m[0][1], p.y, std::fma(T.m[0][2], p.z, T.m[0][3])));
  r.y = std::fma(T.m[1][0], p.x, std::fma(T.m[1][1], p.y, std::fma(T.m[1][2], p.z, T.m[1][3])));
  r.z = std::fma(T.m[2][0], p.x, std::fma(T.m[2][1], p.y, std::fma(T.m[2][2], p.z, T.m[2][3])));
  return r;
}
inline V3 fma_rotate_normal(const M4& T, V3 n) {
  V3 r;
  r.x = std::fma(T.m[0][0], n.x, std::fma(T.m[0][1], n.y, T.m[0][2] * n.z));
  r.y = std::fma(T.m[1][0], n.x, std::fma(T.m[1][1], n.y, T.m[1][2] * n.z));
  r.z = std::fma(T.m[2][0], n.x, std::fma(T.m[2][1], n.y, T.m[2][2] * n.z));
  return r;
}

// hook for the minimiser compiled from the reference's vendored Eigen (oracle/_ref/libref_icp.so, ref_lm_point_to_plane);
// set from oracle/orc.py in the build container (and wherever the prebuilt file travelled)
typedef int (*lm_estimator_fn)(int m, const float* src_xyz, const float* tgt_xyz, const float* tgt_nrm, float* T16_out, float* x6_out,
                               int* stats, float* fnorm_out);
lm_estimator_fn g_ref_lm_estimator = nullptr;

// pcl::IterativeClosestPoint as configured by Utils::runICP (Utils.cpp:188-229) -- PARITY UNPINNED.
// Restated behaviour (PCL 1.9 registration/impl/icp.hpp, correspondence_estimation.hpp,
// correspondence_rejection_surface_normal.h, default_convergence_criteria.hpp):
//   per iteration: (1) NN of every (already moved) source point in the target, kept if d^2 <= max^2;
//   (2) surface-normal rejector keeps n_src . n_tgt >= cos(threshold) (source normals move with the cloud)
//   (3) < 3 correspondences -> not converged;  (4) point-to-plane minimisation: PCL runs Levenberg-
//   Marquardt on the 6-vector (t, quaternion xyz); here one Gauss-Newton step of the linearised
//   point-to-plane problem (the same minimiser for the small per-iteration motions ICP produces);
//   (5) final = T * final; (6) stop: iterations >= max_iter (counts as converged), or
//   |mse - mse_prev| < 1e-6 absolute (Utils.cpp:208), or |mse-mse_prev|/mse_prev < 1e-10 (Utils.cpp:207),
//   with mse the mean squared correspondence distance, for which DefaultConvergenceCriteria requires
//   max_iterations_similar_transforms_ (=0) consecutive hits -> the first hit stops.
struct IcpResult {
  M4 final_tf;
  bool converged;
  int iterations;
};
// Variants for the sensitivity test (tests/test_icp_sensitivity.py): where this restatement departs from what PCL is
// documented to do, how much does the final pose care?
//   minimiser 0: one Gauss-Newton step applied about the centroid of the matched points (the oracle and the GPU)
//             1: the same step applied about the origin
//             2: the non-linear point-to-plane problem on (t, unit-quaternion xyz) minimised to convergence by damped
//                Gauss-Newton (what pcl::registration::TransformationEstimationPointToPlane's Levenberg-Marquardt returns)
//   strict_normal 1: the surface-normal rejector keeps n_src . n_tgt > cos (PCL's operator); 0: >= (here)
//   relative_stop 0: only the absolute MSE criterion (PCL overwrites the relative one, SURVEY.md 8(a) notes); 1: both (here)
//   minimiser 3: Gauss-Newton steps at the fixed correspondences until the step is below 1e-3 rad (<= 3): a second,
//             independent way to reach the non-linear minimiser (to ~1e-6 rad)
// The test's finding: > vs >= and the relative stop change nothing on any frame; the minimisers agree in
// translation (< 1 mm) and are equally close to the ground truth, but the pose the chain finally SELECTS can differ by
// degrees between any two of them -- also between 2 and 3, which compute the same minimiser -- because the early MSE stop
// and the argmax over near-equivalent cluster heads amplify 1e-6 differences on this near-symmetric object.
//   minimiser 4 / 5 / 6: the reference's Levenberg-Marquardt -- restated in float / Eigen's own code (hook) / in exact arithmetic per point
//   minimiser 7: the moment form with integer-exact sums (see "The moment form" above): needs the hypothesis' translation (ctr), the
//             model's radius about its origin and the grid's bits
struct IcpVariant {
  int minimiser = 0, strict_normal = 0, relative_stop = 1;
  float ctr[3] = {0, 0, 0};
  double model_radius = 0;
  int mom_bits = 12;
};
constexpr int ICP_INNER_MAX = 3;
constexpr double ICP_INNER_W = 1.0e-3;

IcpResult run_icp(const Cloud& src, const Cloud& tgt, bool use_tree, int max_iter, float rejection_angle_deg,
                  float max_corr_dist, const IcpVariant& var = IcpVariant()) {
  IcpResult res;
  res.final_tf = identity4();
  res.converged = false;
  res.iterations = 0;
  KdTree<true> tree;
  if (use_tree) tree.build(tgt.pos);
  const float cos_thr = (float)std::cos((double)(rejection_angle_deg / 180.0f) * M_PI);
  const float max_d2 = max_corr_dist * max_corr_dist;
  // minimiser >= 4 (the reference's Levenberg-Marquardt): the gates as PCL 1.9 writes them --
  //   CorrespondenceEstimation::determineCorrespondences: double max_dist_sqr = max_distance * max_distance; skip if distance[0] > max_dist_sqr
  //   CorrespondenceRejectorSurfaceNormal: double((s0*t0 + s1*t1) + s2*t2) > threshold_, threshold_ = std::cos(angle/180.0*M_PI) (Utils.cpp:205)
  const bool pcl_gates = var.minimiser >= 4;
  const double max_d2_d = (double)max_corr_dist * (double)max_corr_dist;
  const double cos_thr_d = std::cos((double)rejection_angle_deg / 180.0 * M_PI);
  std::vector<V3> sp = src.pos, sn = src.nrm;  // moved source
  double mse_prev = std::numeric_limits<double>::max();
  M4 T_lm = identity4();  // transformation_: survives an iteration whose estimator returns early (fewer than 4 correspondences)
  const MomSpec mspec = mom_spec(var.model_radius, max_corr_dist, var.mom_bits);
  MomSums mom;
  while (true) {
    // correspondences
    double A[6][6] = {}, b[6] = {};
    double mse = 0;
    double cs[3] = {0, 0, 0};  // sum of the matched source points
    int cnt = 0;
    std::vector<V3> corr_p, corr_q, corr_n;  // minimiser 2 keeps the correspondences
    mom.clear();
    for (size_t i = 0; i < sp.size(); ++i) {
      float d2 = FLT_MAX;
      int idx = -1;
      if (use_tree) tree.nearest(sp[i], d2, idx);
      else brute_nearest<true>(tgt.pos, sp[i], d2, idx);
      if (idx < 0) continue;
      if (pcl_gates ? (double)d2 > max_d2_d : !(d2 <= max_d2)) continue;
      const V3 nt = tgt.nrm[idx];
      if (pcl_gates) {
        if (!((double)((sn[i].x * nt.x + sn[i].y * nt.y) + sn[i].z * nt.z) > cos_thr_d)) continue;
      } else if (var.strict_normal ? !(dot(sn[i], nt) > cos_thr) : !(dot(sn[i], nt) >= cos_thr)) continue;
      if (var.minimiser == 7) mom_add(mom, mspec, sp[i], tgt.pos[idx], nt, V3{var.ctr[0], var.ctr[1], var.ctr[2]}, d2);
      else if (var.minimiser >= 2) corr_p.push_back(sp[i]), corr_q.push_back(tgt.pos[idx]), corr_n.push_back(nt);
      ++cnt;
      mse += (double)d2;
      cs[0] += (double)sp[i].x, cs[1] += (double)sp[i].y, cs[2] += (double)sp[i].z;
      // residual r = (p - q).n ; jacobian wrt (rx,ry,rz,tx,ty,tz) = [p x n, n]
      const V3 p = sp[i], q = tgt.pos[idx];
      const V3 c = cross(p, nt);
      const double J[6] = {c.x, c.y, c.z, nt.x, nt.y, nt.z};
      const double r = (double)dot(p - q, nt);
      for (int a = 0; a < 6; ++a) {
        b[a] -= J[a] * r;
        for (int bb = 0; bb <= a; ++bb) A[a][bb] += J[a] * J[bb];
      }
    }
    if (cnt < 3) {
      res.converged = false;
      return res;
    }
    // one Gauss-Newton step of the linearised point-to-plane problem from the normal equations (lower triangle in A):
    // (w, t) solve  p' = p + w x p + t.  The increment is applied as the exact rotation exp(w) about the centroid c of
    // the matched points, with the translation the linear model assigns to c:  p' = R (p - c) + c + (t + w x c)
    // (about the origin of a camera frame 0.4 m away the O(|w|^2 |p|) error of the first-order model is centimetres).
    auto gn_step = [&](double A[6][6], double b[6], const double cs[3], int cnt, M4& T, double& wnorm) -> bool {
      for (int a = 0; a < 6; ++a)
        for (int bb = a + 1; bb < 6; ++bb) A[a][bb] = A[bb][a];
      double x[6];
      // tiny Tikhonov term keeps the solve defined on degenerate (e.g. planar/spherical) geometry
      double tr = 0;
      for (int a = 0; a < 6; ++a) tr += A[a][a];
      for (int a = 0; a < 6; ++a) A[a][a] += 1e-9 * tr + 1e-30;
      if (!solve6(A, b, x)) return false;
      // rotation from the rotation vector (exact exponential map), double -> float
      const double th = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
      double Rm[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
      if (th > 1e-12) {
        const double kx = x[0] / th, ky = x[1] / th, kz = x[2] / th, s = std::sin(th), c = std::cos(th), v = 1 - c;
        Rm[0][0] = c + kx * kx * v, Rm[0][1] = kx * ky * v - kz * s, Rm[0][2] = kx * kz * v + ky * s;
        Rm[1][0] = ky * kx * v + kz * s, Rm[1][1] = c + ky * ky * v, Rm[1][2] = ky * kz * v - kx * s;
        Rm[2][0] = kz * kx * v - ky * s, Rm[2][1] = kz * ky * v + kx * s, Rm[2][2] = c + kz * kz * v;
      }
      double c0 = cs[0] / cnt, c1 = cs[1] / cnt, c2 = cs[2] / cnt;
      if (var.minimiser == 1) c0 = c1 = c2 = 0.0;  // about the origin
      const double tc[3] = {x[3] + (x[1] * c2 - x[2] * c1), x[4] + (x[2] * c0 - x[0] * c2), x[5] + (x[0] * c1 - x[1] * c0)};
      const double cc[3] = {c0, c1, c2};
      T = identity4();
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T.m[i][j] = (float)Rm[i][j];
        T.m[i][3] = (float)(cc[i] - (Rm[i][0] * c0 + Rm[i][1] * c1 + Rm[i][2] * c2) + tc[i]);
      }
      wnorm = th;
      return true;
    };
    M4 T;
    double wnorm = 0;
    if (var.minimiser >= 4) {
      // 4: the restated Levenberg-Marquardt above; 5: the one compiled from the reference's vendored Eigen (hook)
      if (var.minimiser == 5) {
        if (!g_ref_lm_estimator) {
          res.converged = false;
          res.iterations = -1;
          return res;
        }
        std::vector<float> a(3 * cnt), bq(3 * cnt), cn(3 * cnt);
        for (int i = 0; i < cnt; ++i) {
          a[3 * i] = corr_p[i].x, a[3 * i + 1] = corr_p[i].y, a[3 * i + 2] = corr_p[i].z;
          bq[3 * i] = corr_q[i].x, bq[3 * i + 1] = corr_q[i].y, bq[3 * i + 2] = corr_q[i].z;
          cn[3 * i] = corr_n[i].x, cn[3 * i + 1] = corr_n[i].y, cn[3 * i + 2] = corr_n[i].z;
        }
        float T16[16];
        if (g_ref_lm_estimator(cnt, a.data(), bq.data(), cn.data(), T16, nullptr, nullptr, nullptr) == 0) T_lm = load4(T16);
      } else if (var.minimiser == 7) {
        if (cnt >= 4) {
          double Md[13][13];
          mom_to_double(mom, mspec, Md);
          const double cd[3] = {(double)var.ctr[0], (double)var.ctr[1], (double)var.ctr[2]};
          lm_point_to_plane_moments(Md, cd, T_lm);
        }
        mse = std::ldexp((double)mom.d2q, -mspec.k_d);  // the exact sum of the gridded squared distances
      } else
        lm_point_to_plane(corr_p, corr_q, corr_n, T_lm, nullptr, nullptr, var.minimiser == 6);  // 6: the same minimiser in exact arithmetic
      T = T_lm;
    } else if (!gn_step(A, b, cs, cnt, T, wnorm)) {
      res.converged = false;
      return res;
    }
    if (var.minimiser == 3) {
      // PCL's estimator minimises the NON-linear problem at the correspondences of this iteration (Levenberg-Marquardt to
      // convergence).  Gauss-Newton converges quadratically on it: further steps at the FIXED correspondences, evaluated
      // at the points moved by the increment so far, while the last step still turned by more than 1e-3 rad (its neglected
      // second-order term is then below 1e-6 rad), three steps at most.
      for (int inner = 1; inner < ICP_INNER_MAX && wnorm > ICP_INNER_W; ++inner) {
        double A2[6][6] = {}, b2[6] = {}, cs2[3] = {0, 0, 0};
        for (size_t i = 0; i < corr_p.size(); ++i) {
          const V3 p = pcl_transform_point(T, corr_p[i]), nt = corr_n[i];
          cs2[0] += (double)p.x, cs2[1] += (double)p.y, cs2[2] += (double)p.z;
          const V3 c = cross(p, nt);
          const double J[6] = {c.x, c.y, c.z, nt.x, nt.y, nt.z};
          const double r = (double)dot(p - corr_q[i], nt);
          for (int a = 0; a < 6; ++a) {
            b2[a] -= J[a] * r;
            for (int bb = 0; bb <= a; ++bb) A2[a][bb] += J[a] * J[bb];
          }
        }
        M4 T2;
        if (!gn_step(A2, b2, cs2, cnt, T2, wnorm)) break;
        T = mul4(T2, T);
      }
    }
    if (var.minimiser == 2) {
      // minimise sum_i ((R(q) p_i + t - q_i) . n_i)^2 over x = (t, q_xyz), q_w = sqrt(1 - |q_xyz|^2), from x = 0:
      // damped Gauss-Newton with a forward-difference Jacobian (as Eigen::NumericalDiff gives PCL's functor) to convergence
      double xv[6] = {0, 0, 0, 0, 0, 0};
      auto rot = [](const double* q3, double R[3][3]) {
        const double qx = q3[0], qy = q3[1], qz = q3[2], qw = std::sqrt(std::max(0.0, 1.0 - qx * qx - qy * qy - qz * qz));
        R[0][0] = 1 - 2 * (qy * qy + qz * qz), R[0][1] = 2 * (qx * qy - qz * qw), R[0][2] = 2 * (qx * qz + qy * qw);
        R[1][0] = 2 * (qx * qy + qz * qw), R[1][1] = 1 - 2 * (qx * qx + qz * qz), R[1][2] = 2 * (qy * qz - qx * qw);
        R[2][0] = 2 * (qx * qz - qy * qw), R[2][1] = 2 * (qy * qz + qx * qw), R[2][2] = 1 - 2 * (qx * qx + qy * qy);
      };
      auto residuals = [&](const double* xx, std::vector<double>& r) {
        double R[3][3];
        rot(xx + 3, R);
        r.resize(corr_p.size());
        for (size_t i = 0; i < corr_p.size(); ++i) {
          const double p[3] = {corr_p[i].x, corr_p[i].y, corr_p[i].z};
          double m[3];
          for (int a = 0; a < 3; ++a) m[a] = R[a][0] * p[0] + R[a][1] * p[1] + R[a][2] * p[2] + xx[a];
          r[i] = (m[0] - corr_q[i].x) * corr_n[i].x + (m[1] - corr_q[i].y) * corr_n[i].y + (m[2] - corr_q[i].z) * corr_n[i].z;
        }
      };
      std::vector<double> r0, r1;
      residuals(xv, r0);
      double cost = 0;
      for (double v : r0) cost += v * v;
      double lambda = 1e-6;
      for (int it = 0; it < 60; ++it) {
        double JtJ[6][6] = {}, Jtr[6] = {};
        std::vector<std::vector<double>> Jc(6);
        for (int a = 0; a < 6; ++a) {
          double xp[6];
          std::copy(xv, xv + 6, xp);
          const double hstep = 1e-7;
          xp[a] += hstep;
          residuals(xp, r1);
          Jc[a].resize(r0.size());
          for (size_t i = 0; i < r0.size(); ++i) Jc[a][i] = (r1[i] - r0[i]) / hstep;
        }
        for (int a = 0; a < 6; ++a) {
          for (int bb = 0; bb < 6; ++bb) {
            double sacc = 0;
            for (size_t i = 0; i < r0.size(); ++i) sacc += Jc[a][i] * Jc[bb][i];
            JtJ[a][bb] = sacc;
          }
          double sacc = 0;
          for (size_t i = 0; i < r0.size(); ++i) sacc += Jc[a][i] * r0[i];
          Jtr[a] = -sacc;
        }
        bool improved = false;
        for (int tries = 0; tries < 12 && !improved; ++tries) {
          double Ad[6][6], bd[6], dx[6];
          for (int a = 0; a < 6; ++a) {
            bd[a] = Jtr[a];
            for (int bb = 0; bb < 6; ++bb) Ad[a][bb] = JtJ[a][bb];
            Ad[a][a] += lambda * (JtJ[a][a] + 1e-30);
          }
          if (!solve6(Ad, bd, dx)) {
            lambda *= 10;
            continue;
          }
          double xn[6];
          for (int a = 0; a < 6; ++a) xn[a] = xv[a] + dx[a];
          residuals(xn, r1);
          double c1n = 0;
          for (double v : r1) c1n += v * v;
          if (c1n <= cost) {
            const double step = std::sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2] + dx[3] * dx[3] + dx[4] * dx[4] + dx[5] * dx[5]);
            std::copy(xn, xn + 6, xv);
            r0 = r1;
            const double dc = cost - c1n;
            cost = c1n;
            lambda = std::max(lambda * 0.3, 1e-12);
            improved = true;
            if (step < 1e-12 || dc <= 1e-18 * (cost + 1e-30)) it = 1000;
          } else
            lambda *= 10;
        }
        if (!improved) break;
      }
      double R[3][3];
      rot(xv + 3, R);
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T.m[i][j] = (float)R[i][j];
        T.m[i][3] = (float)xv[i];
      }
    }
    if (var.minimiser == 7) {  // the composed form: final p0, not T_k(...T_1(p0))
      const M4 F = mul4(T, res.final_tf);
      for (size_t i = 0; i < sp.size(); ++i) {
        sp[i] = fma_transform_point(F, src.pos[i]);
        sn[i] = fma_rotate_normal(F, src.nrm[i]);
      }
    } else
      for (size_t i = 0; i < sp.size(); ++i) {
        sp[i] = pcl_transform_point(T, sp[i]);
        sn[i] = pcl_rotate_normal(T, sn[i]);
      }
    res.final_tf = mul4(T, res.final_tf);
    res.iterations++;
    mse /= cnt;
    // convergence
    if (res.iterations >= max_iter) {
      res.converged = true;
      return res;
    }
    if (pcl_gates) {
      // DefaultConvergenceCriteria::hasConverged, criterion 2 with the thresholds ICP installs (rotation 1.0 - 0, translation 0):
      // fires only for an increment that is exactly the identity
      const double cos_angle = 0.5 * (double)(T.m[0][0] + T.m[1][1] + T.m[2][2] - 1);
      const double translation_sqr = (double)(T.m[0][3] * T.m[0][3] + T.m[1][3] * T.m[1][3] + T.m[2][3] * T.m[2][3]);
      if (cos_angle >= 1.0 && translation_sqr <= 0.0) {
        res.converged = true;
        return res;
      }
    }
    if (std::fabs(mse - mse_prev) < 1e-6) {
      res.converged = true;
      return res;
    }
    if (var.relative_stop && std::fabs(mse - mse_prev) / mse_prev < 1e-10) {
      res.converged = true;
      return res;
    }
    mse_prev = mse;
  }
}

// Eigen 3.3 Matrix3f::eulerAngles(2,1,0) (Geometry/EulerAngles.h:36-108): first angle in [0,pi].
void euler_zyx(const float R[3][3], float res[3]) {
  const int i = 2, j = 1, k = 0;  // odd = 1
  res[0] = std::atan2(R[j][k], R[k][k]);
  const float c2 = std::sqrt(R[i][i] * R[i][i] + R[i][j] * R[i][j]);
  if (res[0] < 0.f) {
    res[0] += (float)M_PI;
    res[1] = std::atan2(-R[i][k], -c2);
  } else
    res[1] = std::atan2(-R[i][k], c2);
  const float s1 = std::sin(res[0]), c1 = std::cos(res[0]);
  res[2] = std::atan2(s1 * R[k][i] - c1 * R[j][i], c1 * R[j][j] - s1 * R[k][j]);
}

// ------------------------------------------------------------------------------------------------
// Hand-state search: objFuncPSO + pso_int.  PARITY UNPINNED (PCL, Armadillo absent).
// ------------------------------------------------------------------------------------------------
struct FingerScene {
  std::vector<V3> model_pos, model_nrm, scene_pos, lookup_nrm, swivel_pos;
  KdTree<true> tree;
};
FingerScene load_finger_scene(const orc_finger_args* a) {
  FingerScene f;
  f.model_pos = soa_to_v3(a->model_xyz, a->n_model);
  f.model_nrm = soa_to_v3(a->model_nrm, a->n_model);
  f.scene_pos = soa_to_v3(a->scene_xyz, a->n_scene);
  f.lookup_nrm = soa_to_v3(a->scene_nrm_lookup, a->n_lookup);
  f.swivel_pos = soa_to_v3(a->swivel_xyz, a->n_swivel);
  f.tree.build(f.scene_pos);
  return f;
}

inline int bin_along_z(const orc_finger_args* a, float z) {  // Hand.cpp:244-250
  int bin = (int)(std::max(z - a->fp_min[2], 0.0f) / a->fp_stride_z);
  bin = std::max(bin, 0);
  bin = std::min(bin, a->fp_num_division - 1);
  return bin;
}

inline void mat4_vec4(const M4& T, const float v[4], float out[4]) {
  for (int i = 0; i < 4; ++i) out[i] = ((T.m[i][0] * v[0] + T.m[i][1] * v[1]) + T.m[i][2] * v[2]) + T.m[i][3] * v[3];
}

// objFuncPSO, Hand.cpp:10-178
double objective(const orc_finger_args* a, const FingerScene& fs, double X0) {
  const float dist_thres = a->dist_thres;
  float score = 0;
  M4 tf_self = identity4();
  {  // AngleAxisf(X, UnitX) as a matrix: float cos/sin of the float angle
    const float ang = (float)X0;
    const float c = std::cos(ang), s = std::sin(ang);
    tf_self.m[1][1] = c, tf_self.m[1][2] = -s, tf_self.m[2][1] = s, tf_self.m[2][2] = c;
  }
  const M4 model2handbase = load4(a->model2handbase);
  const M4 cur_model2handbase = mul4(model2handbase, tf_self);
  float tip1[4], tip2[4], tmp[4];
  if (a->is_palm_side) {
    const float t1[4] = {a->fo_min[0], a->fo_max[1], a->fo_min[2], 1};
    const M4 cur_finger_out2handbase = mul4(cur_model2handbase, load4(a->finger_out2parent));
    mat4_vec4(cur_finger_out2handbase, t1, tip1);
    const float t2[4] = {a->fp_min[0], a->fp_max[1], a->fp_min[2], 1};
    mat4_vec4(cur_model2handbase, t2, tip2);
  } else {
    const float t1[4] = {a->fp_min[0], a->fp_max[1], a->fp_min[2], 1};
    mat4_vec4(cur_model2handbase, t1, tip1);
    const float t2[4] = {a->fp_min[0], a->fp_max[1], a->fp_max[2], 1};
    mat4_vec4(cur_model2handbase, t2, tip2);
  }
  (void)tmp;
  float gripper_dist1, gripper_dist2;
  if (a->is_right_side) {
    gripper_dist1 = tip1[1] - a->pair_tip1[1];
    gripper_dist2 = tip2[1] - a->pair_tip2[1];
  } else {
    gripper_dist1 = -tip1[1] + a->pair_tip1[1];
    gripper_dist2 = -tip2[1] + a->pair_tip2[1];
  }
  const float GRIPPER_MIN_DIST = a->gripper_min_dist;
  if (gripper_dist1 < GRIPPER_MIN_DIST || gripper_dist2 < GRIPPER_MIN_DIST) {
    const float penalty = (float)(1e3 + 1e3 * (double)std::fabs(GRIPPER_MIN_DIST - gripper_dist1));
    score -= penalty;
    return -score;
  }
  float num_match = 0;
  for (size_t ii = 0; ii < fs.model_pos.size(); ++ii) {
    const V3 pt = pcl_transform_point(cur_model2handbase, fs.model_pos[ii]);
    const V3 pn = pcl_rotate_normal(cur_model2handbase, fs.model_nrm[ii]);
    float d2 = FLT_MAX;
    int idx = -1;
    fs.tree.nearest(pt, d2, idx);
    if (idx < 0) continue;
    if (d2 <= dist_thres * dist_thres) {
      if (!a->check_normal) {
        num_match = (float)((double)num_match + (1 + X0));
        continue;
      }
      // Hand.cpp:91: the neighbour is fetched from scene_hand_region (unfiltered cloud) with an index
      // into the filtered one; only its normal is used.
      const V3 nn = (idx < (int)fs.lookup_nrm.size()) ? fs.lookup_nrm[idx] : V3{0, 0, 0};
      if (nn.x == 0 && nn.y == 0 && nn.z == 0) {
        num_match = (float)((double)num_match + (1 + X0));
        continue;
      }
      if (std::isfinite(nn.x) && std::isfinite(nn.y) && std::isfinite(nn.z)) {
        if (dot(pn, nn) >= a->cos_normal_thres) num_match = (float)((double)num_match + (1 + X0));
        continue;
      }
    }
  }
  score += num_match;
  if (num_match == 0) {
    score = (float)(-100 + X0);
    return -score;
  }
  float outer_dist_sum = 0;
  int num_outer = 0;
  const M4 inv = inverse_affine(cur_model2handbase);
  for (const V3& s : fs.swivel_pos) {
    const V3 pt = pcl_transform_point(inv, s);
    const int cur_bin = bin_along_z(a, pt.z);
    if (pt.y >= a->fp_hist_min_y[cur_bin]) continue;
    outer_dist_sum += std::fabs(pt.y - a->fp_hist_min_y[cur_bin]);
    num_outer++;
  }
  float penalty_outer = 0;
  const float avg_outer_dist = outer_dist_sum / num_outer;  // 0/0 = NaN when nothing is outside
  if (num_outer >= a->max_outter_pts || avg_outer_dist >= 0.005) {
    penalty_outer = (float)(1e3 + (double)(a->outter_pt_dist_weight * std::max(avg_outer_dist - a->outter_pt_dist, 0.0f)));
    score -= penalty_outer;
  } else if (num_outer >= 0 && avg_outer_dist - a->outter_pt_dist > 0) {
    penalty_outer = a->outter_pt_dist_weight * std::exp(avg_outer_dist * 1000);
    score -= penalty_outer;
  }
  return -score;
}

}  // namespace

// ==================================================================================================
extern "C" {

float orc_acosf(float x) { return acosf_fdlibm(x); }

void orc_compute_ppf(const float* p1, const float* n1, const float* p2, const float* n2, int* key4) {
  const Pt a = make_pt({p1[0], p1[1], p1[2]}, {n1[0], n1[1], n1[2]}, 1.f);
  const Pt b = make_pt({p2[0], p2[1], p2[2]}, {n2[0], n2[1], n2[2]}, 1.f);
  compute_ppf(a, b, key4);
}

int orc_pair_ppf_is_good(const float* p, const float* q, const float* b0, const float* b1) {
  auto mk = [](const float* v) { return make_pt({v[0], v[1], v[2]}, {v[3], v[4], v[5]}, 1.f); };
  return pair_ppf_is_good(mk(p), mk(q), mk(b0), mk(b1)) ? 1 : 0;
}

int orc_rigid(const float* ref9, const float* cand9, float* T16, float* rms) {
  V3 r[3], c[3];
  for (int i = 0; i < 3; ++i) {
    r[i] = {ref9[3 * i], ref9[3 * i + 1], ref9[3 * i + 2]};
    c[i] = {cand9[3 * i], cand9[3 * i + 1], cand9[3 * i + 2]};
  }
  const V3 c1 = ((r[0] + r[1]) + r[2]) / 3.f;
  const V3 c2 = ((c[0] + c[1]) + c[2]) / 3.f;
  M4 T = identity4();
  float e = -1;
  const bool ok = compute_rigid(r, c, c1, c2, T, e);
  store4(T, T16);
  *rms = e;
  return ok ? 1 : 0;
}

void orc_probe_transform(const float* T16, const float* p3, float* out3) {
  const V3 o = transform_point(load4(T16), {p3[0], p3[1], p3[2]});
  out3[0] = o.x, out3[1] = o.y, out3[2] = o.z;
}

void orc_probe_vec(const float* a3, const float* b3, float* out9) {
  const V3 a{a3[0], a3[1], a3[2]}, b{b3[0], b3[1], b3[2]};
  out9[0] = sqnorm(a - b);
  out9[1] = dot(a, b);
  out9[2] = norm(a - b);
  const V3 n = normalized(a), c = cross(a, b);
  out9[3] = n.x, out9[4] = n.y, out9[5] = n.z, out9[6] = c.x, out9[7] = c.y, out9[8] = c.z;
}

void orc_probe_quat(const float* n3, const float* v3, float* out7) {
  const V3 v0 = normalized(V3{0.f, 0.f, 1.f});
  const V3 v1 = normalized(V3{n3[0], n3[1], n3[2]});
  const float c = dot(v1, v0);
  const V3 axis = cross(v0, v1);
  const float s = std::sqrt((1.f + c) * 2.f);
  const float invs = 1.f / s;
  const V3 qv = axis * invs;
  const float qw = s * 0.5f;
  const V3 v{v3[0], v3[1], v3[2]};
  V3 uv = cross(qv, v);
  uv = uv + uv;
  const V3 r = (v + qw * uv) + cross(qv, uv);
  out7[0] = qw, out7[1] = qv.x, out7[2] = qv.y, out7[3] = qv.z, out7[4] = r.x, out7[5] = r.y, out7[6] = r.z;
}

void* orc_s4pcs_create(const orc_s4pcs_opts* o) { return new Matcher(*o); }
void orc_s4pcs_destroy(void* h) { delete static_cast<Matcher*>(h); }
void orc_s4pcs_set_keys(void* h, const int* keys4, int n) {
  auto* m = static_cast<Matcher*>(h);
  m->ppfs.keys.clear();
  for (int i = 0; i < n; ++i) m->ppfs.keys.insert({keys4[4 * i], keys4[4 * i + 1], keys4[4 * i + 2], keys4[4 * i + 3]});
}
int orc_s4pcs_run(void* h, const float* Pxyz, const float* Pnrm, const float* Pprob, int nP, const float* Qxyz,
                  const float* Qnrm, int nQ, int n_calls) {
  auto* m = static_cast<Matcher*>(h);
  const std::vector<Pt> P = make_cloud(Pxyz, Pnrm, Pprob, nP);
  const std::vector<Pt> Q = make_cloud(Qxyz, Qnrm, nullptr, nQ);
  for (int k = 0; k < n_calls; ++k) m->ComputeTransformation(P, Q);
  return (int)m->_pose_hypo.size();
}
int orc_s4pcs_num_hypos(void* h) { return (int)static_cast<Matcher*>(h)->_pose_hypo.size(); }
void orc_s4pcs_get_hypos(void* h, float* pose16, float* lcp) {
  auto* m = static_cast<Matcher*>(h);
  for (size_t i = 0; i < m->_pose_hypo.size(); ++i) {
    store4(m->_pose_hypo[i], pose16 + 16 * i);
    lcp[i] = m->_pose_lcp_scores[i];
  }
}
int orc_s4pcs_num_bases(void* h) { return (int)static_cast<Matcher*>(h)->trace.size(); }
void orc_s4pcs_get_base(void* h, int i, int* base4, float* inv2, int* counts3) {
  const auto& t = static_cast<Matcher*>(h)->trace[i];
  for (int k = 0; k < 4; ++k) base4[k] = t.base[k];
  inv2[0] = t.inv1, inv2[1] = t.inv2;
  counts3[0] = (int)t.pairs1.size(), counts3[1] = (int)t.pairs2.size(), counts3[2] = (int)t.quads.size();
}
void orc_s4pcs_get_base_lists(void* h, int i, int* pairs1, int* pairs2, int* quads) {
  const auto& t = static_cast<Matcher*>(h)->trace[i];
  for (size_t k = 0; k < t.pairs1.size(); ++k) pairs1[2 * k] = t.pairs1[k].first, pairs1[2 * k + 1] = t.pairs1[k].second;
  for (size_t k = 0; k < t.pairs2.size(); ++k) pairs2[2 * k] = t.pairs2[k].first, pairs2[2 * k + 1] = t.pairs2[k].second;
  for (size_t k = 0; k < t.quads.size(); ++k)
    for (int j = 0; j < 4; ++j) quads[4 * k + j] = t.quads[k][j];
}
int orc_s4pcs_num_sampled_q(void* h) { return (int)static_cast<Matcher*>(h)->sampled_Q_3D_.size(); }
void orc_s4pcs_get_state(void* h, float* Qs_xyz, float* Qs_nrm, float* cP3, float* cQ3, float* diameter,
                         int* n_quat_fallback) {
  auto* m = static_cast<Matcher*>(h);
  const int n = (int)m->sampled_Q_3D_.size();
  for (int i = 0; i < n; ++i) {
    const Pt& q = m->sampled_Q_3D_[i];
    Qs_xyz[i] = q.pos.x, Qs_xyz[n + i] = q.pos.y, Qs_xyz[2 * n + i] = q.pos.z;
    Qs_nrm[i] = q.nrm.x, Qs_nrm[n + i] = q.nrm.y, Qs_nrm[2 * n + i] = q.nrm.z;
  }
  cP3[0] = m->centroid_P_.x, cP3[1] = m->centroid_P_.y, cP3[2] = m->centroid_P_.z;
  cQ3[0] = m->centroid_Q_.x, cQ3[1] = m->centroid_Q_.y, cQ3[2] = m->centroid_Q_.z;
  *diameter = m->P_diameter_;
  *n_quat_fallback = m->n_quat_fallback;
}
float orc_s4pcs_verify(void* h, const float* T16) {
  auto* m = static_cast<Matcher*>(h);
  const int good = verify_count(m->P_pos, &m->kd_tree_, m->q_positions(), load4(T16), m->opt.delta);
  return float((unsigned)good) / float(m->sampled_Q_3D_.size());
}

void orc_verify_batch(const float* Pxyz, int nP, const float* Qxyz, int nQ, const float* T16, int H, float delta,
                      int use_tree, int* count_out) {
  const std::vector<V3> P = soa_to_v3(Pxyz, nP), Q = soa_to_v3(Qxyz, nQ);
  KdTree<false> tree;
  if (use_tree) tree.build(P);
#pragma omp parallel for schedule(dynamic)
  for (int h = 0; h < H; ++h) count_out[h] = verify_count(P, use_tree ? &tree : nullptr, Q, load4(T16 + 16 * h), delta);
}

void orc_compute_lcp_batch(const float* Sxyz, const float* Snrm, int nS, const float* Mxyz, const float* Mnrm, int nM,
                           const float* pose16, int H, float dist_thres, float angle_deg, int use_tree,
                           float* score_out) {
  const Cloud scene = load_cloud(Sxyz, Snrm, nS), model = load_cloud(Mxyz, Mnrm, nM);
  KdTree<true> scene_tree;
  if (use_tree) scene_tree.build(scene.pos);
#pragma omp parallel for schedule(dynamic)
  for (int h = 0; h < H; ++h) {
    const M4 T = load4(pose16 + 16 * h);
    Cloud mt;
    mt.pos.resize(nM), mt.nrm.resize(nM);
    for (int i = 0; i < nM; ++i) {  // pcl::transformPointCloudWithNormals, PoseEstimator.cpp:487
      mt.pos[i] = pcl_transform_point(T, model.pos[i]);
      mt.nrm[i] = pcl_rotate_normal(T, model.nrm[i]);
    }
    score_out[h] = compute_lcp(scene, use_tree ? &scene_tree : nullptr, mt, use_tree != 0, dist_thres, angle_deg);
  }
}

void orc_icp_refine_batch(const float* Sxyz, const float* Snrm, int nS, const float* Mxyz, const float* Mnrm, int nM,
                          float* pose16, int H, int max_iter, float angle_deg, float max_corr_dist, int use_tree,
                          int* iters_out, int* converged_out) {
  const Cloud scene = load_cloud(Sxyz, Snrm, nS), model = load_cloud(Mxyz, Mnrm, nM);
#pragma omp parallel for schedule(dynamic)
  for (int h = 0; h < H; ++h) {
    const M4 pose = load4(pose16 + 16 * h);
    Cloud mt;
    mt.pos.resize(nM), mt.nrm.resize(nM);
    for (int i = 0; i < nM; ++i) {  // PoseEstimator.cpp:263
      mt.pos[i] = pcl_transform_point(pose, model.pos[i]);
      mt.nrm[i] = pcl_rotate_normal(pose, model.nrm[i]);
    }
    IcpResult r = run_icp(scene, mt, use_tree != 0, max_iter, angle_deg, max_corr_dist);
    M4 T = r.converged ? r.final_tf : identity4();  // Utils.cpp:218-225
    const M4 out = mul4(inverse_affine(T), pose);   // PoseEstimator.cpp:267
    store4(out, pose16 + 16 * h);
    if (iters_out) iters_out[h] = r.iterations;
    if (converged_out) converged_out[h] = r.converged ? 1 : 0;
  }
}

// radius of a cloud about its origin: sqrt of the largest squared norm, both in double from the float coordinates
double cloud_radius(const std::vector<V3>& p) {
  double m = 0;
  for (const V3& v : p) m = std::max(m, (double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z);
  return std::sqrt(m);
}
int g_mom_bits = 12;  // grid of the moment form (minimiser 7); orc_set_mom_bits: precision experiments only
void orc_set_mom_bits(int bits) { g_mom_bits = bits; }
void orc_icp_refine_batch_variant(const float* Sxyz, const float* Snrm, int nS, const float* Mxyz, const float* Mnrm, int nM, float* pose16, int H,
                                  int max_iter, float angle_deg, float max_corr_dist, int minimiser, int strict_normal, int relative_stop,
                                  int* iters_out, int* converged_out) {
  const Cloud scene = load_cloud(Sxyz, Snrm, nS), model = load_cloud(Mxyz, Mnrm, nM);
  IcpVariant var0;
  var0.minimiser = minimiser, var0.strict_normal = strict_normal, var0.relative_stop = relative_stop;
  var0.model_radius = cloud_radius(model.pos), var0.mom_bits = g_mom_bits;
#pragma omp parallel for schedule(dynamic)
  for (int h = 0; h < H; ++h) {
    const M4 pose = load4(pose16 + 16 * h);
    IcpVariant var = var0;
    var.ctr[0] = pose.m[0][3], var.ctr[1] = pose.m[1][3], var.ctr[2] = pose.m[2][3];
    Cloud mt;
    mt.pos.resize(nM), mt.nrm.resize(nM);
    for (int i = 0; i < nM; ++i) {
      mt.pos[i] = pcl_transform_point(pose, model.pos[i]);
      mt.nrm[i] = pcl_rotate_normal(pose, model.nrm[i]);
    }
    IcpResult r = run_icp(scene, mt, true, max_iter, angle_deg, max_corr_dist, var);
    M4 T = r.converged ? r.final_tf : identity4();
    const M4 out = mul4(inverse_affine(T), pose);
    store4(out, pose16 + 16 * h);
    if (iters_out) iters_out[h] = r.iterations;
    if (converged_out) converged_out[h] = r.converged ? 1 : 0;
  }
}

void orc_set_lm_estimator(void* fn) { g_ref_lm_estimator = (lm_estimator_fn)fn; }

/* the restated minimiser alone (one estimateRigidTransformation call); AoS m x 3 inputs like ref_lm_point_to_plane */
int orc_lm_point_to_plane(int m, const float* src_xyz, const float* tgt_xyz, const float* tgt_nrm, float* T16_out, float* x6_out, int* stats) {
  std::vector<V3> P(m), Q(m), Nn(m);
  for (int i = 0; i < m; ++i) {
    P[i] = {src_xyz[3 * i], src_xyz[3 * i + 1], src_xyz[3 * i + 2]};
    Q[i] = {tgt_xyz[3 * i], tgt_xyz[3 * i + 1], tgt_xyz[3 * i + 2]};
    Nn[i] = {tgt_nrm[3 * i], tgt_nrm[3 * i + 1], tgt_nrm[3 * i + 2]};
  }
  M4 T;
  if (!lm_point_to_plane(P, Q, Nn, T, x6_out, stats)) return -1;
  store4(T, T16_out);
  return 0;
}
/* the moment form alone (minimiser 7): Eigen's minimiser from a 13 x 13 moment matrix (row-major, symmetric) and the centre c */
void orc_lm_point_to_plane_moments(const double* M169, const double* c3, float* T16_out, float* x6_out, int* stats) {
  double M[13][13];
  for (int i = 0; i < 13; ++i)
    for (int j = 0; j < 13; ++j) M[i][j] = M169[13 * i + j];
  M4 T = identity4();
  lm_point_to_plane_moments(M, c3, T, x6_out, stats);
  if (T16_out) store4(T, T16_out);
}
/* the integer moment sums of m correspondences (AoS m x 3 inputs; d2[m]); M169_out: the exact integer matrix, Md169_out: in metres */
void orc_mom_accumulate(int m, const float* src_xyz, const float* tgt_xyz, const float* tgt_nrm, const float* d2, const float* ctr3, double model_radius,
                        float max_corr_dist, int bits, long long* M169_out, long long* d2q_out, double* Md169_out, int* scales4_out) {
  const MomSpec sp = mom_spec(model_radius, max_corr_dist, bits);
  MomSums ms;
  ms.clear();
  for (int i = 0; i < m; ++i)
    mom_add(ms, sp, V3{src_xyz[3 * i], src_xyz[3 * i + 1], src_xyz[3 * i + 2]}, V3{tgt_xyz[3 * i], tgt_xyz[3 * i + 1], tgt_xyz[3 * i + 2]},
            V3{tgt_nrm[3 * i], tgt_nrm[3 * i + 1], tgt_nrm[3 * i + 2]}, V3{ctr3[0], ctr3[1], ctr3[2]}, d2[i]);
  double Md[13][13];
  mom_to_double(ms, sp, Md);
  for (int i = 0; i < 13; ++i)
    for (int j = 0; j < 13; ++j) {
      if (M169_out) M169_out[13 * i + j] = ms.M[std::max(i, j)][std::min(i, j)];
      if (Md169_out) Md169_out[13 * i + j] = Md[i][j];
    }
  if (d2q_out) *d2q_out = ms.d2q;
  if (scales4_out) scales4_out[0] = sp.k_np, scales4_out[1] = sp.k_n, scales4_out[2] = sp.k_r, scales4_out[3] = sp.k_d;
}
void orc_lm_warp6(const float* x6, float* T16) { store4(lm_warp6(x6), T16); }
/* f(x) and NumericalDiff's Jacobian at x (row-major m x 6) */
void orc_lm_residuals_jacobian(int m, const float* src_xyz, const float* tgt_xyz, const float* tgt_nrm, const float* x6, float* fvec_out, float* jac_out) {
  std::vector<V3> P(m), Q(m), Nn(m);
  for (int i = 0; i < m; ++i) {
    P[i] = {src_xyz[3 * i], src_xyz[3 * i + 1], src_xyz[3 * i + 2]};
    Q[i] = {tgt_xyz[3 * i], tgt_xyz[3 * i + 1], tgt_xyz[3 * i + 2]};
    Nn[i] = {tgt_nrm[3 * i], tgt_nrm[3 * i + 1], tgt_nrm[3 * i + 2]};
  }
  LmSums s;
  std::vector<float> f, J;
  lm_pass(P, Q, Nn, x6, s, &f, &J);
  std::copy(f.begin(), f.end(), fvec_out);
  std::copy(J.begin(), J.end(), jac_out);
}
// Utils::rotationGeodesicDistance, Utils.cpp:29-32: std::acos(((R1 * R2).trace()-1) / 2.0) returned as float
float rotation_geodesic_distance(const float R0[3][3], const float R1[3][3]) {
  M3 a, b;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a.m[i][j] = R0[i][j], b.m[i][j] = R1[i][j];
  const M3 p = mul(a, b);
  const float tr = p.m[0][0] + (p.m[1][1] + p.m[2][2]);
  return (float)std::acos((double)(tr - 1.0f) / 2.0);  // trace() - 1 in float, / 2.0 in double
}
float orc_geodesic(const float* R1_9, const float* R2_9) {
  float a[3][3], b[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a[i][j] = R1_9[3 * i + j], b[i][j] = R2_9[3 * i + j];
  return rotation_geodesic_distance(a, b);
}
float orc_tdiff_norm(const float* t0, const float* t1) { return norm(V3{t0[0], t0[1], t0[2]} - V3{t1[0], t1[1], t1[2]}); }
void orc_inverse_times(const float* Ticp16, const float* pose16, float* out16) { store4(mul4(inverse_affine(load4(Ticp16)), load4(pose16)), out16); }
void orc_mul4(const float* A16, const float* B16, float* out16) { store4(mul4(load4(A16), load4(B16)), out16); }
void orc_euler_zyx(const float* R9, float* out3) {
  float R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = R9[3 * i + j];
  euler_zyx(R, out3);
}

int orc_cluster_poses(const float* pose16, const float* lcp, const int* ids, int H, float angle_diff, float dist_diff,
                      const float* sym_deg3, int* keep_out) {
  if (H <= 0) return 0;
  std::vector<int> order(H);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) {  // HypoCompare, PoseEstimator.cpp:110-123
    if (lcp[a] > lcp[b]) return true;
    if (lcp[a] < lcp[b]) return false;
    if (ids[a] < ids[b]) return true;
    return false;
  });
  const float radian_thres = (float)((double)(angle_diff / 180.0f) * M_PI);
  const float sym[3] = {(float)((double)sym_deg3[0] / 180 * M_PI), (float)((double)sym_deg3[1] / 180 * M_PI),
                        (float)((double)sym_deg3[2] / 180 * M_PI)};
  std::vector<int> clusters;
  clusters.push_back(order[0]);
  for (int oi = 1; oi < H; ++oi) {
    const float* cur = pose16 + 16 * order[oi];
    bool isnew = true;
    for (int ci : clusters) {
      const float* cl = pose16 + 16 * ci;
      const V3 t0{cl[3], cl[7], cl[11]}, t1{cur[3], cur[7], cur[11]};
      if (norm(t0 - t1) >= dist_diff) continue;
      float R0[3][3], R1[3][3];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R0[i][j] = cl[4 * i + j], R1[i][j] = cur[4 * i + j];
      float e0[3], e1[3];
      euler_zyx(R0, e0);
      euler_zyx(R1, e1);
      float roll_diff = std::fabs(e0[2] - e1[2]), pitch_diff = std::fabs(e0[1] - e1[1]), yaw_diff = std::fabs(e0[0] - e1[0]);
      if (sym[0] == 0) roll_diff = 0;
      else if (sym[0] > 0) roll_diff = std::min(roll_diff, sym[0] - roll_diff);
      if (sym[1] == 0) pitch_diff = 0;
      else if (sym[1] > 0) pitch_diff = std::min(pitch_diff, sym[1] - pitch_diff);
      if (sym[2] == 0) yaw_diff = 0;
      else if (sym[2] > 0) yaw_diff = std::min(yaw_diff, sym[2] - yaw_diff);
      if (pitch_diff <= radian_thres && roll_diff <= radian_thres && yaw_diff <= radian_thres) {
        isnew = false;
        break;
      }
      // Utils::rotationGeodesicDistance, Utils.cpp:29-32: acos((trace(R1*R2)-1)/2.0), R1*R2 (not R1^T R2)
      const float rot_diff = rotation_geodesic_distance(R0, R1);
      if (rot_diff <= radian_thres) {
        isnew = false;
        break;
      }
    }
    if (isnew) clusters.push_back(order[oi]);
  }
  for (size_t i = 0; i < clusters.size(); ++i) keep_out[i] = clusters[i];
  return (int)clusters.size();
}

double orc_pso_objective(const orc_finger_args* a, double angle) {
  const FingerScene fs = load_finger_scene(a);
  return objective(a, fs, angle);
}

void orc_pso_objective_batch(const orc_finger_args* a, const double* angles, int n, double* cost_out) {
  const FingerScene fs = load_finger_scene(a);
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < n; ++i) cost_out[i] = objective(a, fs, angles[i]);
}

// pso_int, pso.hpp:146-351 (vals_bound=true, center particle, inertia method 1, velocity method 1)
int orc_pso_search(const orc_finger_args* a, const orc_pso_settings* s, double* best_angle, double* objval) {
  const FingerScene fs = load_finger_scene(a);
  std::mt19937_64 eng(s->seed);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  const size_t n_pop = (size_t)s->n_pop + 1;
  const size_t n_gen = (size_t)s->n_gen;
  const size_t check_freq = s->check_freq > 0 ? (size_t)s->check_freq : n_gen;
  double par_w = s->initial_w;
  auto inv_tf = [&](double v) { return v * (s->upper_rad - s->lower_rad) + s->lower_rad; };  // transform_vals.hpp:135-137
  auto eval = [&](double p) {
    double v = objective(a, fs, inv_tf(p));
    if (!std::isfinite(v)) v = std::numeric_limits<double>::max();
    return v;
  };
  std::vector<double> P(n_pop), V(n_pop), objfn(n_pop);
  for (size_t i = 0; i < n_pop; ++i) P[i] = U(eng);
  auto center = [&]() {
    double sum = 0;
    for (size_t i = 0; i + 1 < n_pop; ++i) sum += P[i];
    P[n_pop - 1] = sum / double(n_pop - 1);
  };
  center();
  for (size_t i = 0; i < n_pop; ++i) objfn[i] = eval(P[i]);
  std::vector<double> best_vals = objfn, best_vecs = P;
  size_t gi = std::min_element(objfn.begin(), objfn.end()) - objfn.begin();
  double cur_global_best_val = objfn[gi], global_best_val_check = cur_global_best_val, global_best_vec = P[gi];
  size_t iter = 0;
  double err = 2.0 * s->err_tol;
  for (size_t i = 0; i < n_pop; ++i) V[i] = U(eng);
  while (err > s->err_tol && iter < n_gen) {
    iter++;
    std::vector<double> r1(n_pop), r2(n_pop);
    for (size_t i = 0; i < n_pop; ++i) r1[i] = U(eng);
    for (size_t i = 0; i < n_pop; ++i) r2[i] = U(eng);
    for (size_t i = 0; i < n_pop; ++i) {
      V[i] = par_w * V[i] + s->c_cog * r1[i] * (best_vecs[i] - P[i]) + s->c_soc * r2[i] * (global_best_vec - P[i]);
      P[i] += V[i];
    }
    center();
    for (size_t i = 0; i < n_pop; ++i) P[i] = std::min(std::max(P[i], 0.0), 1.0);
    for (size_t i = 0; i < n_pop; ++i) {
      objfn[i] = eval(P[i]);
      if (objfn[i] < best_vals[i]) best_vals[i] = objfn[i], best_vecs[i] = P[i];
    }
    const size_t mi = std::min_element(best_vals.begin(), best_vals.end()) - best_vals.begin();
    if (best_vals[mi] < cur_global_best_val) cur_global_best_val = best_vals[mi], global_best_vec = best_vecs[mi];
    if (iter % check_freq == 0)
      err = std::fabs(cur_global_best_val - global_best_val_check) / (1e-20 + std::fabs(global_best_val_check));
    if (cur_global_best_val < global_best_val_check) global_best_val_check = cur_global_best_val;
    par_w = s->w_min + (s->w_max - s->w_min) * double(iter + 1) / double(n_gen);
  }
  *best_angle = inv_tf(global_best_vec);
  *objval = (double)(float)global_best_val_check;  // ArgPasser::objval is a float (pso.hpp:43,347)
  return 1;
}

void orc_finger_property(const float* xyz, int n, int num_division, float* min3, float* max3, float* stride_z,
                         float* hist) {
  // FingerProperty::FingerProperty, Hand.cpp:182-236
  const std::vector<V3> p = soa_to_v3(xyz, n);
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (const V3& q : p) {
    mn[0] = std::min(mn[0], q.x), mn[1] = std::min(mn[1], q.y), mn[2] = std::min(mn[2], q.z);
    mx[0] = std::max(mx[0], q.x), mx[1] = std::max(mx[1], q.y), mx[2] = std::max(mx[2], q.z);
  }
  for (int k = 0; k < 3; ++k) min3[k] = mn[k], max3[k] = mx[k];
  const float stride = (mx[2] - mn[2]) / num_division;
  *stride_z = stride;
  auto H = [&](int r, int c) -> float& { return hist[r * num_division + c]; };
  for (int c = 0; c < num_division; ++c)
    for (int r = 0; r < 6; ++r) H(r, c) = r < 3 ? FLT_MAX : -FLT_MAX;
  std::vector<bool> changed(num_division, false);
  for (const V3& q : p) {
    int bin = (int)(std::max(q.z - mn[2], 0.0f) / stride);
    bin = std::min(std::max(bin, 0), num_division - 1);
    H(0, bin) = std::min(H(0, bin), q.x), H(1, bin) = std::min(H(1, bin), q.y), H(2, bin) = std::min(H(2, bin), q.z);
    H(3, bin) = std::max(H(3, bin), q.x), H(4, bin) = std::max(H(4, bin), q.y), H(5, bin) = std::max(H(5, bin), q.z);
    changed[bin] = true;
  }
  for (int i = 0; i < num_division; ++i) {
    if (changed[i]) continue;
    for (int j = i + 1; j < num_division; ++j)
      if (changed[j]) {
        for (int r = 0; r < 6; ++r) H(r, i) = H(r, j);
        changed[i] = true;
        break;
      }
  }
  if (!changed[num_division - 1])
    for (int i = num_division - 2; i >= 0; --i)
      if (changed[i]) {
        for (int r = 0; r < 6; ++r) H(r, num_division - 1) = H(r, i);
        changed[num_division - 1] = true;
        break;
      }
}


// "next" row N3a -- HandT42::removeSurroundingPointsAndAssignProbability (Hand.cpp:779-888), restated.
// scene: camera frame, SoA xyz / normal planes.  links: the clouds of Hand::makeHandCloud (Hand.cpp:537-556, hand-base
// frame) in std::map (name) order, each with its local_dist_thres (Hand.cpp:812-821, squared metres).  A point survives
// when no link is "near" (NN within the threshold, or planar distance within it and |dz| <= 5 mm, Hand.cpp:826-840) and
// it is not on the outer side of either distal finger (Hand.cpp:869-884).  Survivors come back in input order (the
// reference's order depends on OpenMP scheduling), in the camera frame after the reference's round trip through the
// hand-base frame, with confidence 1 - exp(-lambda * min_dist) (Hand.cpp:844).  keep_index: input index of each.
int orc_hand_remove_surrounding(const float* scene_xyz, const float* scene_nrm, int n, const float* handbase_in_cam16, const float* const* link_xyz,
                                const int* link_n, const float* link_sq_thres, int n_links, const float* finger12_in_handbase16,
                                const float* finger22_in_handbase16, float min_z, float* out_xyz, float* out_nrm, float* out_conf,
                                int* keep_index) {
  const M4 hb = load4(handbase_in_cam16), cam2hb = inverse_affine(hb);
  const M4 f1i = inverse_affine(load4(finger12_in_handbase16)), f2i = inverse_affine(load4(finger22_in_handbase16));
  const std::vector<V3> sp = soa_to_v3(scene_xyz, n), sn = soa_to_v3(scene_nrm, n);
  std::vector<std::vector<V3>> links(n_links);
  for (int l = 0; l < n_links; ++l) links[l] = soa_to_v3(link_xyz[l], link_n[l]);
  const float lambda = 231.04906018664843f;
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    const V3 p = pcl_transform_point(cam2hb, sp[i]);
    const V3 nn = pcl_rotate_normal(cam2hb, sn[i]);
    bool is_near = false;
    float min_dist = 1.0f;
    for (int l = 0; l < n_links && !is_near; ++l) {
      if (links[l].empty()) continue;  // nearestKSearch returns 0
      float best = FLT_MAX;
      int bj = -1;
      for (int j = 0; j < (int)links[l].size(); ++j) {
        const V3& q = links[l][j];
        const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
        const float d2 = (dx * dx + dy * dy) + dz * dz;  // FLANN L2_Simple order
        if (d2 < best) best = d2, bj = j;
      }
      min_dist = std::min(min_dist, std::sqrt(best));
      if (best <= link_sq_thres[l]) {
        is_near = true;
        break;
      }
      const V3& nei = links[l][bj];
      const float sq_planar = (p.x - nei.x) * (p.x - nei.x) + (p.y - nei.y) * (p.y - nei.y);
      if (sq_planar <= link_sq_thres[l] && std::abs(p.z - nei.z) <= 0.005) is_near = true;
    }
    if (is_near) continue;
    const V3 p1 = pcl_transform_point(f1i, p), p2 = pcl_transform_point(f2i, p);
    if (p1.y < 0 && p1.z >= min_z) continue;
    if (p2.y < 0 && p2.z >= min_z) continue;
    const V3 pc = pcl_transform_point(hb, p), nc = pcl_rotate_normal(hb, nn);
    out_xyz[kept] = pc.x, out_xyz[n + kept] = pc.y, out_xyz[2 * (size_t)n + kept] = pc.z;
    out_nrm[kept] = nc.x, out_nrm[n + kept] = nc.y, out_nrm[2 * (size_t)n + kept] = nc.z;
    out_conf[kept] = 1 - std::exp(-lambda * min_dist);
    keep_index[kept] = i;
    ++kept;
  }
  return kept;
}


// "next" row N4 -- the pair loop of the offline computePPF tool (computePPF.cpp:17-38,88-100), restated: key of
// (points[i], points[j]) for every i < j; normals normalised once (n.normalize()); float arithmetic in Eigen's order;
// std::acos on a float argument is acosf.  Returns the number of distinct keys, the first `cap` of them sorted.
int orc_model_ppf_keys(const float* xyz, const float* nrm, int n, int* keys4, int cap) {
  const std::vector<V3> p = soa_to_v3(xyz, n), nn = soa_to_v3(nrm, n);
  std::vector<V3> un(n);
  for (int i = 0; i < n; ++i) un[i] = normalized(nn[i]);
  std::set<std::array<int, 4>> keys;
  auto closest = [](int value, int disc) {
    const int lower = value - (value % disc), upper = lower + disc;
    return (value - lower) < (upper - value) ? lower : upper;
  };
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      const float dist_f = norm(p[i] - p[j]) * 1000;
      if (!(dist_f < 2147483648.0f)) continue;
      const V3 d = normalized(p[j] - p[i]);
      const float c1 = dot(un[i], d), c2 = dot(un[j], d), c3 = dot(un[i], un[j]);
      const float a1 = acosf_fdlibm(c1), a2 = acosf_fdlibm(c2), a3 = acosf_fdlibm(c3);
      if (!(a1 == a1) || !(a2 == a2) || !(a3 == a3)) continue;  // the tool stores an INT_MIN key nobody can look up
      std::array<int, 4> k{closest((int)dist_f, 5), closest((int)((double)a1 / M_PI * 180), 10), closest((int)((double)a2 / M_PI * 180), 10),
                           closest((int)((double)a3 / M_PI * 180), 10)};
      keys.insert(k);
    }
  int c = 0;
  for (const auto& k : keys) {
    if (c < cap) std::memcpy(keys4 + 4 * (size_t)c, k.data(), sizeof(int) * 4);
    ++c;
  }
  return c;
}

}  // extern "C"
