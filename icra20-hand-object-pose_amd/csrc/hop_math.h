// hop_math.h -- small float algebra shared by host and device code of libhop.
//
// Everything here is plain IEEE float arithmetic compiled with -ffp-contract=off, in the operation
// order the reference's Eigen 3.3.90 expressions evaluate to (3-element reductions are c0+(c1+c2),
// 3rdparty/Eigen/Eigen/src/Core/Redux.h:91-105; normalized() divides by sqrt(squaredNorm),
// Core/Dot.h:121-131), so that integer decisions taken from these values (PPF bins, grid cells,
// inlier counts) are the same on the GPU, on the host and in the reference.
#ifndef HOP_MATH_H_
#define HOP_MATH_H_

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HOP_HD __host__ __device__ __forceinline__
#else
#define HOP_HD inline
#endif

namespace hop {

struct V3 {
  float x, y, z;
};
HOP_HD V3 v3(float x, float y, float z) {
  V3 r;
  r.x = x, r.y = y, r.z = z;
  return r;
}
HOP_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
HOP_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
HOP_HD V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
HOP_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
HOP_HD V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
HOP_HD V3 vneg(V3 a) { return v3(-a.x, -a.y, -a.z); }
HOP_HD float vdot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
HOP_HD float vsqn(V3 a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }
HOP_HD float vnorm(V3 a) { return sqrtf(vsqn(a)); }
HOP_HD V3 vnormalized(V3 a) {
  const float z = vsqn(a);
  if (z > 0.f) return a / sqrtf(z);
  return a;
}
HOP_HD V3 vcross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// squared distance as Eigen's (a-b).squaredNorm() -- used by the generator's Verify (kdtree.h:368)
HOP_HD float sqdist_eigen(V3 a, V3 b) { return vsqn(a - b); }
// squared distance as FLANN's L2_Simple accumulates it -- PCL kd-tree searches (computeLCP, ICP, PSO)
HOP_HD float sqdist_flann(V3 a, V3 b) {
  const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return (dx * dx + dy * dy) + dz * dz;
}

// HypoCompare order of a score as an unsigned key that ASCENDS with the score: used by every sort / merge of (score, id) rows on the
// device and on the host, so the two cannot disagree.  Canonical form: -0 counts as +0 and a NaN score as lower than every number
// (float comparison would make NaN "equal" to everything, which is no order at all).
HOP_HD uint32_t score_order_key(float s) {
  if (s != s) return 0u;
  if (s == 0.f) s = 0.f;  // (-0 -> +0)
  uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
  u = __float_as_uint(s);
#else
  memcpy(&u, &s, 4);
#endif
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct M4 {
  float m[16];  // row-major
};
HOP_HD M4 m4_identity() {
  M4 r;
  for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.f : 0.f;
  return r;
}
// p' = ((m0*x + m1*y) + m2*z) + m3: both Eigen's Matrix4f*homogeneous (cse.hpp:390) and PCL's
// transformPointCloudWithNormals evaluate in this order (checked against the reference build by the
// oracle's probes).
HOP_HD V3 m4_point(const float* T, V3 p) {
  return v3(((T[0] * p.x + T[1] * p.y) + T[2] * p.z) + T[3], ((T[4] * p.x + T[5] * p.y) + T[6] * p.z) + T[7],
            ((T[8] * p.x + T[9] * p.y) + T[10] * p.z) + T[11]);
}
HOP_HD V3 m4_dir(const float* T, V3 n) {
  return v3((T[0] * n.x + T[1] * n.y) + T[2] * n.z, (T[4] * n.x + T[5] * n.y) + T[6] * n.z,
            (T[8] * n.x + T[9] * n.y) + T[10] * n.z);
}
HOP_HD M4 m4_mul(const M4& a, const M4& b) {
  M4 r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += a.m[4 * i + k] * b.m[4 * k + j];
      r.m[4 * i + j] = s;
    }
  return r;
}
// inverse of an affine 4x4 (last row 0 0 0 1), adjugate in double
HOP_HD M4 m4_inverse_affine(const M4& a) {
  double m[3][3], inv[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m[i][j] = a.m[4 * i + j];
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
  M4 r = m4_identity();
  for (int i = 0; i < 3; ++i) {
    double t = 0;
    for (int j = 0; j < 3; ++j) {
      r.m[4 * i + j] = (float)inv[i][j];
      t -= inv[i][j] * (double)a.m[4 * j + 3];
    }
    r.m[4 * i + 3] = (float)t;
  }
  return r;
}

HOP_HD uint32_t f2u(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(f);
#else
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
#endif
}
HOP_HD float u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float f;
  memcpy(&f, &u, 4);
  return f;
#endif
}

// acosf with the operation sequence of glibc 2.35's sysdeps/ieee754/flt-32/e_acosf.c (fdlibm), so the
// device returns bit-for-bit what std::acos(float) returns on the host (and returned inside the
// reference when it evaluated gr::computePPF / pairPPFisGood).  Only +,-,*,/ and sqrtf: all correctly
// rounded on gfx950 with -ffp-contract=off and hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt.
HOP_HD float acosf_fdlibm(float x) {
  const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f,
              pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
              pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f,
              qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
  const int32_t hx = (int32_t)f2u(x);
  const int32_t ix = hx & 0x7fffffff;
  if (ix == 0x3f800000) {
    if (hx > 0) return 0.0f;
    return pi + 2.0f * pio2_lo;
  } else if (ix > 0x3f800000) {
    return (x - x) / (x - x);
  }
  if (ix < 0x3f000000) {
    if (ix <= 0x32800000) return pio2_hi + pio2_lo;
    const float z = x * x;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    return pio2_hi - (x - (pio2_lo - x * r));
  } else if (hx < 0) {
    const float z = (one + x) * 0.5f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float s = sqrtf(z);
    const float r = p / q;
    const float w = r * s - pio2_lo;
    return pi - 2.0f * (s + w);
  } else {
    const float z = (one - x) * 0.5f;
    const float s = sqrtf(z);
    const float df = u2f(f2u(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    const float w = r * s + c;
    return 2.0f * (df + w);
  }
}

#define HOP_PI_D 3.14159265358979323846

// ---- PPF key (gr::computePPF, matchBase.hpp:47-68; ppfClosestBin :31-43) -----------------------
// The reference casts float/double to int with x86 cvtt semantics: NaN / out of range -> INT_MIN, which
// can never be a table key.  Here an invalid component makes the whole key invalid (returns false).
HOP_HD int ppf_closest_bin(int value, int disc) {
  const int lower = value - (value % disc);
  const int upper = lower + disc;
  return ((value - lower) < (upper - value)) ? lower : upper;
}
HOP_HD bool ppf_angle_bin(float c, int* bin) {
  const float a = acosf_fdlibm(c);
  if (!(a == a)) return false;  // NaN: |c| > 1
  const double deg = (double)a / HOP_PI_D * 180;
  *bin = ppf_closest_bin((int)deg, 10);
  return true;
}
// The angle component of the key is a step function of the cosine c: ppf_closest_bin((int)deg, 10) = 10 * #{k in 0..17 :
// deg(c) >= 5 + 10 k}, and deg(c) is non-increasing in c.  With thr[k] = the largest float whose bin is >= 10 (k+1)
// (found and verified on the host against ppf_angle_bin itself, hop_select.h: build_angle_thresholds), the bin is a
// count of thresholds >= c: five comparisons of a binary search instead of three divisions, a square root and a
// double division per angle.  thr has 32 entries, [18..31] = -2 (never reached).
HOP_HD bool ppf_angle_bin_thr(float c, const float* thr, int* bin) {
  if (!(fabsf(c) <= 1.0f)) return false;  // NaN or outside [-1,1]: acosf gives NaN, the key is invalid
  int pos = 0;
  if (c <= thr[pos + 15]) pos += 16;
  if (c <= thr[pos + 7]) pos += 8;
  if (c <= thr[pos + 3]) pos += 4;
  if (c <= thr[pos + 1]) pos += 2;
  if (c <= thr[pos]) pos += 1;
  *bin = 10 * pos;
  return true;
}
HOP_HD bool ppf_key_thr(V3 p1, V3 n1p, V3 p2, V3 n2p, const float* thr, int key[4]) {
  const float nrm = vnorm(p1 - p2) * 1000.f;
  if (!(nrm < 2147483648.0f)) return false;
  key[0] = ppf_closest_bin((int)nrm, 5);
  const V3 d = vnormalized(p2 - p1);
  if (!ppf_angle_bin_thr(vdot(n1p, d), thr, &key[1])) return false;
  if (!ppf_angle_bin_thr(vdot(n2p, d), thr, &key[2])) return false;
  if (!ppf_angle_bin_thr(vdot(n1p, n2p), thr, &key[3])) return false;
  return true;
}
// n1p/n2p: the point normals after the two extra normalisations computePPF applies (matchBase.hpp:53-56)
HOP_HD bool ppf_key(V3 p1, V3 n1p, V3 p2, V3 n2p, int key[4]) {
  const float nrm = vnorm(p1 - p2) * 1000.f;
  if (!(nrm < 2147483648.0f)) return false;
  key[0] = ppf_closest_bin((int)nrm, 5);
  const V3 d = vnormalized(p2 - p1);
  if (!ppf_angle_bin(vdot(n1p, d), &key[1])) return false;
  if (!ppf_angle_bin(vdot(n2p, d), &key[2])) return false;
  if (!ppf_angle_bin(vdot(n1p, n2p), &key[3])) return false;
  return true;
}

// ---- pair filter (gr::pairPPFisGood, PointPairFilter.h:17-38) ----------------------------------
// p,q from Q with normals np,nq; len2 / b_n0 / b_n1 / n0_n1 are the base-edge features, computed once
// per base edge with base_edge_features().
struct EdgeFeat {
  float len, f_n0, f_n1, n0_n1;  // degrees as float, exactly as the reference stores them
};
HOP_HD float ppf_deg(float c) { return (float)((double)acosf_fdlibm(c) / HOP_PI_D * 180); }
HOP_HD EdgeFeat base_edge_features(V3 b0, V3 n0, V3 b1, V3 n1) {
  EdgeFeat e;
  e.len = vnorm(b0 - b1);
  const V3 d = vnormalized(b1 - b0);
  e.f_n0 = ppf_deg(fabsf(vdot(d, n0)));
  e.f_n1 = ppf_deg(fabsf(vdot(d, n1)));
  e.n0_n1 = ppf_deg(vdot(n0, n1));
  return e;
}
HOP_HD bool pair_ppf_is_good(V3 p, V3 np, V3 q, V3 nq, const EdgeFeat& e) {
  const float length1 = vnorm(p - q);
  if ((double)fabsf(length1 - e.len) > 5e-3) return false;
  const V3 pq = vnormalized(q - p);
  const float pq_np = ppf_deg(fabsf(vdot(pq, np)));
  const float pq_nq = ppf_deg(fabsf(vdot(pq, nq)));
  const float np_nq = ppf_deg(vdot(np, nq));
  // NaN never rejects (comparisons are false), as in the reference
  if (fabsf(pq_np - e.f_n0) > 30.f || fabsf(pq_nq - e.f_n1) > 30.f || fabsf(np_nq - e.n0_n1) > 30.f) return false;
  return true;
}

// ---- 3-point rigid fit (MatchBase::ComputeRigidTransformation, matchBase.hpp:229-377) ------------
// computeScale=false, max_angle<0.  Returns false where the reference returns false; degenerate inputs
// give true with rms = FLT_MAX (the reference's "return FLT_MAX" in a bool function).
HOP_HD bool rigid_3pt(const V3 ref[3], const V3 cand[3], V3 c1, V3 c2, float T[16], float* rms_out) {
  const float FMAX = 3.402823466e+38f;
  *rms_out = FMAX;
  V3 vp1 = ref[1] - ref[0];
  if (vsqn(vp1) == 0.f) return true;
  vp1 = vnormalized(vp1);
  V3 vp2 = (ref[2] - ref[0]) - vdot(ref[2] - ref[0], vp1) * vp1;
  if (vsqn(vp2) == 0.f) return true;
  vp2 = vnormalized(vp2);
  const V3 vp3 = vcross(vp1, vp2);
  V3 vq1 = cand[1] - cand[0];
  if (vsqn(vq1) == 0.f) return true;
  vq1 = vnormalized(vq1);
  V3 vq2 = (cand[2] - cand[0]) - vdot(cand[2] - cand[0], vq1) * vq1;
  if (vsqn(vq2) == 0.f) return true;
  vq2 = vnormalized(vq2);
  const V3 vq3 = vcross(vq1, vq2);
  // rotation = rotate_p^T * rotate_q with rows (vp1,vp2,vp3) / (vq1,vq2,vq3):
  // R(i,j) = vp1[i]*vq1[j] + (vp2[i]*vq2[j] + vp3[i]*vq3[j])
  const float P[3][3] = {{vp1.x, vp1.y, vp1.z}, {vp2.x, vp2.y, vp2.z}, {vp3.x, vp3.y, vp3.z}};
  const float Q[3][3] = {{vq1.x, vq1.y, vq1.z}, {vq2.x, vq2.y, vq2.z}, {vq3.x, vq3.y, vq3.z}};
  float R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = P[0][i] * Q[0][j] + (P[1][i] * Q[1][j] + P[2][i] * Q[2][j]);
  // (R^T R).isIdentity(1e-6)
  const float k = 1e-6f;
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) {
      const float c = R[0][i] * R[0][j] + (R[1][i] * R[1][j] + R[2][i] * R[2][j]);
      if (i == j) {
        const float ac = fabsf(c);
        if (!(fabsf(c - 1.f) <= (ac < 1.f ? ac : 1.f) * k)) return false;
      } else {
        if (!(fabsf(c) <= k)) return false;
      }
    }
  float rms = 0.f;
  for (int i = 0; i < 3; ++i) {
    const V3 first = 1.f * cand[i] - c2;
    const V3 tr = v3(R[0][0] * first.x + (R[0][1] * first.y + R[0][2] * first.z),
                     R[1][0] * first.x + (R[1][1] * first.y + R[1][2] * first.z),
                     R[2][0] * first.x + (R[2][1] * first.y + R[2][2] * first.z));
    rms += vnorm((tr - ref[i]) + c1);
  }
  rms /= 4.f;  // divided by ref.size() == 4 after summing 3 terms (matchBase.hpp:350-357)
  *rms_out = rms;
  const V3 m = vneg(c2);
  const V3 t = c1 + v3(R[0][0] * m.x + (R[0][1] * m.y + R[0][2] * m.z), R[1][0] * m.x + (R[1][1] * m.y + R[1][2] * m.z),
                      R[2][0] * m.x + (R[2][1] * m.y + R[2][2] * m.z));
  T[0] = R[0][0], T[1] = R[0][1], T[2] = R[0][2], T[3] = t.x;
  T[4] = R[1][0], T[5] = R[1][1], T[6] = R[1][2], T[7] = t.y;
  T[8] = R[2][0], T[9] = R[2][1], T[10] = R[2][2], T[11] = t.z;
  T[12] = 0.f, T[13] = 0.f, T[14] = 0.f, T[15] = 1.f;
  return true;
}

// ---- IndexedNormalSet geometry (normalset.h:99-126, utils.h:141-150) ----------------------------
struct NsetGeom {
  float nepsilon;  // 1/7 + 1e-5 (as float)
  float epsilon;   // 1/egSize
  int eg_size;
};
HOP_HD int nset_index_normal(const NsetGeom& g, V3 n) {
  const V3 half = v3(0.5f, 0.5f, 0.5f);
  const V3 c = (n / 2.f + half) / g.nepsilon;
  return (int)c.x + 7 * (int)c.y + 49 * (int)c.z;
}

}  // namespace hop
#endif
