// hop_select.h -- host-only part of the generator (no HIP types): cloud containers, MatchBase::init, and the
// sequential, RNG-driven base selection (SelectRandomTriangle / SelectQuadrilateral / TryQuadrilateral).
// Kept free of device headers so that tools/select_bench.cpp can exercise and time it on a CPU.
#ifndef HOP_SELECT_H_
#define HOP_SELECT_H_
#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <chrono>
#include <climits>
#include <limits>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>
#include "../../include/hop.h"
#include "hop_math.h"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace hop {

// 4th base point (match4pcsBase.hpp:154-175): among candidate point indices cand[0..nc), the first one with the
// smallest |A x + B y + C z - 1| that is at least sqrt(too_small) away from b0, b1, b2.  Every float operation is
// the reference's ((A*x + B*y) + C*z, then "- 1.0" in double and back to float == one float subtraction, exactly);
// the AVX2 form evaluates 8 candidates per instruction with the same IEEE operations (no FMA), so the choice is
// bit-identical to the scalar loop.  Returns the chosen index or -1.
inline int fourth_point_scalar(const float* X, const float* Y, const float* Z, const int* cand, int nc, const float b[3][3],
                               float too_small, float A, float B, float C) {
  int best = -1;
  float best_distance = FLT_MAX;
  for (int k = 0; k < nc; ++k) {
    const int r = cand[k];
    const float px = X[r], py = Y[r], pz = Z[r];
    bool ok = true;
    for (int q = 0; q < 3 && ok; ++q) {
      const float dx = px - b[q][0], dy = py - b[q][1], dz = pz - b[q][2];
      ok = (dx * dx + (dy * dy + dz * dz)) >= too_small;
    }
    if (!ok) continue;
    const float distance = std::fabs(((A * px + B * py) + C * pz) - 1.0f);
    if (distance < best_distance) {
      best_distance = distance;
      best = r;
    }
  }
  return best;
}

// The same choice when the candidates are given as a bit mask over the index range [0, nr): bit r set <=> point r is
// a candidate.  Contiguous loads instead of gathers.
inline int fourth_point_mask_scalar(const float* X, const float* Y, const float* Z, const unsigned long long* mask, int nr,
                                    const float b[3][3], float too_small, float A, float B, float C) {
  int best = -1;
  float best_distance = FLT_MAX;
  for (int w = 0; w * 64 < nr; ++w) {
    unsigned long long bits = mask[w];
    while (bits) {
      const int r = w * 64 + __builtin_ctzll(bits);
      bits &= bits - 1;
      if (r >= nr) break;
      const float px = X[r], py = Y[r], pz = Z[r];
      bool ok = true;
      for (int q = 0; q < 3 && ok; ++q) {
        const float dx = px - b[q][0], dy = py - b[q][1], dz = pz - b[q][2];
        ok = (dx * dx + (dy * dy + dz * dz)) >= too_small;
      }
      if (!ok) continue;
      const float distance = std::fabs(((A * px + B * py) + C * pz) - 1.0f);
      if (distance < best_distance) best_distance = distance, best = r;
    }
  }
  return best;
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) inline int fourth_point_mask_avx2(const float* X, const float* Y, const float* Z,
                                                                 const unsigned long long* mask, int nr, const float b[3][3],
                                                                 float too_small, float A, float B, float C, float* dist_tmp) {
  const __m256 vts = _mm256_set1_ps(too_small), vA = _mm256_set1_ps(A), vB = _mm256_set1_ps(B), vC = _mm256_set1_ps(C);
  const __m256 one = _mm256_set1_ps(1.0f), vmax = _mm256_set1_ps(FLT_MAX);
  const __m256 absmask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7fffffff));
  const __m256i bitsel = _mm256_setr_epi32(1, 2, 4, 8, 16, 32, 64, 128);
  __m256 b0x = _mm256_set1_ps(b[0][0]), b0y = _mm256_set1_ps(b[0][1]), b0z = _mm256_set1_ps(b[0][2]);
  __m256 b1x = _mm256_set1_ps(b[1][0]), b1y = _mm256_set1_ps(b[1][1]), b1z = _mm256_set1_ps(b[1][2]);
  __m256 b2x = _mm256_set1_ps(b[2][0]), b2y = _mm256_set1_ps(b[2][1]), b2z = _mm256_set1_ps(b[2][2]);
  __m256 vmin = vmax;
  const int nfull = nr & ~7;
  const unsigned char* mbytes = reinterpret_cast<const unsigned char*>(mask);
  for (int k = 0; k < nfull; k += 8) {
    const unsigned m8 = mbytes[k >> 3];
    if (!m8) {
      _mm256_storeu_ps(dist_tmp + k, vmax);
      continue;
    }
    const __m256 px = _mm256_loadu_ps(X + k), py = _mm256_loadu_ps(Y + k), pz = _mm256_loadu_ps(Z + k);
    __m256 ok = _mm256_castsi256_ps(_mm256_cmpeq_epi32(_mm256_and_si256(_mm256_set1_epi32((int)m8), bitsel), bitsel));
#define HOP_FAR_ENOUGH(bx, by, bz)                                                                                        \
  {                                                                                                                       \
    const __m256 dx = _mm256_sub_ps(px, bx), dy = _mm256_sub_ps(py, by), dz = _mm256_sub_ps(pz, bz);                         \
    const __m256 d2 = _mm256_add_ps(_mm256_mul_ps(dx, dx), _mm256_add_ps(_mm256_mul_ps(dy, dy), _mm256_mul_ps(dz, dz)));    \
    ok = _mm256_and_ps(ok, _mm256_cmp_ps(d2, vts, _CMP_GE_OQ));                                                            \
  }
    HOP_FAR_ENOUGH(b0x, b0y, b0z)
    HOP_FAR_ENOUGH(b1x, b1y, b1z)
    HOP_FAR_ENOUGH(b2x, b2y, b2z)
#undef HOP_FAR_ENOUGH
    const __m256 s = _mm256_add_ps(_mm256_add_ps(_mm256_mul_ps(vA, px), _mm256_mul_ps(vB, py)), _mm256_mul_ps(vC, pz));
    __m256 dist = _mm256_and_ps(_mm256_sub_ps(s, one), absmask);
    const __m256 good = _mm256_and_ps(ok, _mm256_cmp_ps(dist, dist, _CMP_ORD_Q));  // NaN never wins, as in the scalar loop
    dist = _mm256_blendv_ps(vmax, dist, good);
    _mm256_storeu_ps(dist_tmp + k, dist);
    vmin = _mm256_min_ps(vmin, dist);
  }
  float lanes[8];
  _mm256_storeu_ps(lanes, vmin);
  float best_distance = FLT_MAX;
  for (int q = 0; q < 8; ++q) best_distance = std::min(best_distance, lanes[q]);
  int best = -1;
  if (best_distance < FLT_MAX)
    for (int t = 0; t < nfull; ++t)
      if (dist_tmp[t] == best_distance) {
        best = t;
        break;
      }
  for (int r = nfull; r < nr; ++r) {  // tail (< 8 entries), scalar; strict '<' keeps the earlier index on ties
    if (!((mask[r >> 6] >> (r & 63)) & 1ull)) continue;
    const float px = X[r], py = Y[r], pz = Z[r];
    bool ok = true;
    for (int q = 0; q < 3 && ok; ++q) {
      const float dx = px - b[q][0], dy = py - b[q][1], dz = pz - b[q][2];
      ok = (dx * dx + (dy * dy + dz * dz)) >= too_small;
    }
    if (!ok) continue;
    const float distance = std::fabs(((A * px + B * py) + C * pz) - 1.0f);
    if (distance < best_distance) best_distance = distance, best = r;
  }
  return best;
}
#endif

#if defined(__x86_64__)
__attribute__((target("avx512f"))) inline int fourth_point_mask_avx512(const float* X, const float* Y, const float* Z,
                                                                      const unsigned long long* mask, int nr, const float b[3][3],
                                                                      float too_small, float A, float B, float C, float* /*dist_tmp*/) {
  const __m512 vts = _mm512_set1_ps(too_small), vA = _mm512_set1_ps(A), vB = _mm512_set1_ps(B), vC = _mm512_set1_ps(C);
  const __m512 one = _mm512_set1_ps(1.0f);
  __m512 bx[3], by[3], bz[3];
  for (int q = 0; q < 3; ++q) bx[q] = _mm512_set1_ps(b[q][0]), by[q] = _mm512_set1_ps(b[q][1]), bz[q] = _mm512_set1_ps(b[q][2]);
  // The scalar loop keeps ONE running best and takes a candidate only if it is strictly below it.  The same here, sixteen
  // at a time: lanes not below the running best are out before anything else is computed for them, and the three
  // "far enough from the base points" tests -- most of the arithmetic -- run only for blocks that still hold such a lane
  // (a few dozen of the ~900 blocks: the best falls quickly).  Within a block the lowest index among the lanes with the
  // smallest distance wins, which is what the scalar loop does when it walks the block in order.
  float best_distance = FLT_MAX;
  int best = -1;
  __m512 vbest = _mm512_set1_ps(FLT_MAX);
  const int nfull = nr & ~15;
  const unsigned short* m16 = reinterpret_cast<const unsigned short*>(mask);
  for (int k = 0; k < nfull; k += 16) {
    __mmask16 ok = m16[k >> 4];
    if (!ok) continue;
    const __m512 px = _mm512_loadu_ps(X + k), py = _mm512_loadu_ps(Y + k), pz = _mm512_loadu_ps(Z + k);
    const __m512 s = _mm512_add_ps(_mm512_add_ps(_mm512_mul_ps(vA, px), _mm512_mul_ps(vB, py)), _mm512_mul_ps(vC, pz));
    const __m512 dist = _mm512_abs_ps(_mm512_sub_ps(s, one));
    ok = _mm512_mask_cmp_ps_mask(ok, dist, vbest, _CMP_LT_OQ);  // false for NaN, like the scalar loop
    if (!ok) continue;
    for (int q = 0; q < 3; ++q) {
      const __m512 dx = _mm512_sub_ps(px, bx[q]), dy = _mm512_sub_ps(py, by[q]), dz = _mm512_sub_ps(pz, bz[q]);
      const __m512 d2 = _mm512_add_ps(_mm512_mul_ps(dx, dx), _mm512_add_ps(_mm512_mul_ps(dy, dy), _mm512_mul_ps(dz, dz)));
      ok = _mm512_mask_cmp_ps_mask(ok, d2, vts, _CMP_GE_OQ);
    }
    if (!ok) continue;
    best_distance = _mm512_mask_reduce_min_ps(ok, dist);
    vbest = _mm512_set1_ps(best_distance);
    best = k + __builtin_ctz((unsigned)_mm512_mask_cmp_ps_mask(ok, dist, vbest, _CMP_EQ_OQ));
  }
  for (int r = nfull; r < nr; ++r) {  // tail (< 16 entries), scalar; strict '<' keeps the earlier index on ties
    if (!((mask[r >> 6] >> (r & 63)) & 1ull)) continue;
    const float px = X[r], py = Y[r], pz = Z[r];
    bool ok = true;
    for (int q = 0; q < 3 && ok; ++q) {
      const float dx = px - b[q][0], dy = py - b[q][1], dz = pz - b[q][2];
      ok = (dx * dx + (dy * dy + dz * dz)) >= too_small;
    }
    if (!ok) continue;
    const float distance = std::fabs(((A * px + B * py) + C * pz) - 1.0f);
    if (distance < best_distance) best_distance = distance, best = r;
  }
  return best;
}

// pool of one matrix row: ids and weights of the set bits in order, first rank and weight sum per 64-bit word.
// `weights` must be readable up to 64*W entries.
__attribute__((target("avx512f"))) inline int pool_build_avx512(const unsigned long long* row, int W, const float* weights, int* ids,
                                                               float* pr, int* wstart, double* bsum) {
  const __m512i iota = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
  int k = 0;
  for (int w = 0; w < W; ++w) {
    const unsigned long long bits = row[w];
    wstart[w] = k;
    if (!bits) {
      bsum[w] = 0.0;
      continue;
    }
    __m512d acc = _mm512_setzero_pd();
    for (int c = 0; c < 4; ++c) {
      const __mmask16 m = (__mmask16)(bits >> (16 * c));
      if (!m) continue;
      const int base = w * 64 + c * 16;
      const __m512 p = _mm512_maskz_loadu_ps(m, weights + base);
      // compress in registers and store whole vectors (the buffers are padded): the memory-destination form of
      // vpcompressd is microcoded on Zen 4/5
      _mm512_storeu_si512(ids + k, _mm512_maskz_compress_epi32(m, _mm512_add_epi32(iota, _mm512_set1_epi32(base))));
      _mm512_storeu_ps(pr + k, _mm512_maskz_compress_ps(m, p));
      acc = _mm512_add_pd(acc, _mm512_add_pd(_mm512_cvtps_pd(_mm512_castps512_ps256(p)), _mm512_cvtps_pd(_mm256_castpd_ps(_mm512_extractf64x4_pd(_mm512_castps_pd(p), 1)))));
      k += __builtin_popcount((unsigned)m);
    }
    bsum[w] = _mm512_reduce_add_pd(acc);
  }
  wstart[W] = k;
  return k;
}
// number of leading entries of the non-decreasing array c[0..n) (padded to a multiple of 8 with +inf) that are < target
__attribute__((target("avx512f"))) inline int count_below_avx512(const double* c, int n, double target) {
  const __m512d t = _mm512_set1_pd(target);
  int cnt = 0;
  for (int k = 0; k < n; k += 8) cnt += __builtin_popcount((unsigned)_mm512_cmp_pd_mask(_mm512_loadu_pd(c + k), t, _CMP_LT_OQ));
  return cnt;
}
// c[from..n) += d   (n padded to a multiple of 8; the padding is +inf and stays +inf)
__attribute__((target("avx512f"))) inline void suffix_add_avx512(double* c, int from, int n, double d) {
  const __m512d dv = _mm512_set1_pd(d);
  int k = from & ~7;
  const __mmask8 first = (__mmask8)(0xffu << (from & 7));
  _mm512_storeu_pd(c + k, _mm512_mask_add_pd(_mm512_loadu_pd(c + k), first, _mm512_loadu_pd(c + k), dv));
  for (k += 8; k < n; k += 8) _mm512_storeu_pd(c + k, _mm512_add_pd(_mm512_loadu_pd(c + k), dv));
}
// first j in [0, cnt) with lo + w[0] + ... + w[j] >= target (cnt-1 if none); *lo_out = the cumulative weight before
// entry j.  Eight entries per step, prefix sums formed in registers, early exit.
__attribute__((target("avx512f"))) inline int word_scan_avx512(const float* w, int cnt, double lo, double target, double* lo_out) {
  const __m512i sh1 = _mm512_setr_epi64(0, 0, 1, 2, 3, 4, 5, 6), sh2 = _mm512_setr_epi64(0, 0, 0, 1, 2, 3, 4, 5), sh4 = _mm512_setr_epi64(0, 0, 0, 0, 0, 1, 2, 3);
  alignas(64) double buf[8][8];
  __m512d v[8];
  const int G = (cnt + 7) >> 3;
  // local prefix sums of all groups first (independent chains), then the short carry chain over the group totals
  for (int g = 0; g < G; ++g) {
    const int left = cnt - 8 * g;
    const __mmask16 m = left >= 8 ? (__mmask16)0xff : (__mmask16)((1u << left) - 1u);
    __m512d x = _mm512_cvtps_pd(_mm512_castps512_ps256(_mm512_maskz_loadu_ps(m, w + 8 * g)));
    x = _mm512_mask_add_pd(x, 0xfe, x, _mm512_permutexvar_pd(sh1, x));
    x = _mm512_mask_add_pd(x, 0xfc, x, _mm512_permutexvar_pd(sh2, x));
    x = _mm512_mask_add_pd(x, 0xf0, x, _mm512_permutexvar_pd(sh4, x));
    v[g] = x;
    _mm512_store_pd(buf[g], x);
  }
  const __m512d t = _mm512_set1_pd(target);
  double carry[8];
  double c = lo;
  int below = 0;
  for (int g = 0; g < G; ++g) {
    carry[g] = c;
    const int left = cnt - 8 * g;
    const __mmask8 m = left >= 8 ? (__mmask8)0xff : (__mmask8)((1u << left) - 1u);
    below += __builtin_popcount((unsigned)_mm512_mask_cmp_pd_mask(m, _mm512_add_pd(v[g], _mm512_set1_pd(c)), t, _CMP_LT_OQ));
    c += buf[g][7];
  }
  const int j = below > cnt - 1 ? cnt - 1 : below;  // the last member takes whatever is left
  *lo_out = j > 0 ? buf[(j - 1) >> 3][(j - 1) & 7] + carry[(j - 1) >> 3] : lo;
  return j;
}
// c[from..n) += d for a short array (n a multiple of 8, +inf padding stays +inf)
__attribute__((target("avx512f"))) inline int count_below_off_avx512(const double* c, int n, double base, double target) {
  const __m512d t = _mm512_set1_pd(target), b = _mm512_set1_pd(base);
  int cnt = 0;
  for (int k = 0; k < n; k += 8) cnt += __builtin_popcount((unsigned)_mm512_cmp_pd_mask(_mm512_add_pd(_mm512_loadu_pd(c + k), b), t, _CMP_LT_OQ));
  return cnt;
}
// ---- pool without a materialised copy (AVX-512 path of the pair draws) ---------------------------------------------
// The pool of a first point is the set bits of its matrix row; its weights are read through the row's masks straight
// from the per-call weight array (point order), summed per 16-lane chunk, per word and in two running-sum levels above.
// Returns the pool size; cs[4 w + c] = weight of chunk c of word w, bsum[w] = weight of word w, wstart[w] = pool rank of the
// word's first member; *first / *last = lowest / highest member id (-1 if empty).
__attribute__((target("avx512f"))) inline int pool_sums_avx512(const unsigned long long* row, int W, const float* weights, double* cs, double* bsum,
                                                              int* wstart, int* first, int* last) {
  int k = 0, lo_id = -1, hi_id = -1;
  for (int w = 0; w < W; ++w) {
    const unsigned long long bits = row[w];
    wstart[w] = k;
    if (!bits) {
      cs[4 * w] = cs[4 * w + 1] = cs[4 * w + 2] = cs[4 * w + 3] = 0.0;
      bsum[w] = 0.0;
      continue;
    }
    if (lo_id < 0) lo_id = w * 64 + __builtin_ctzll(bits);
    hi_id = w * 64 + 63 - __builtin_clzll(bits);
    double tot = 0.0;
    for (int c = 0; c < 4; ++c) {
      const __mmask16 m = (__mmask16)(bits >> (16 * c));
      double v = 0.0;
      if (m) {
        const __m512 p = _mm512_maskz_loadu_ps(m, weights + w * 64 + c * 16);
        v = _mm512_reduce_add_pd(
            _mm512_add_pd(_mm512_cvtps_pd(_mm512_castps512_ps256(p)), _mm512_cvtps_pd(_mm256_castpd_ps(_mm512_extractf64x4_pd(_mm512_castps_pd(p), 1)))));
      }
      cs[4 * w + c] = v;
      tot += v;
    }
    bsum[w] = tot;
    k += __builtin_popcountll(bits);
  }
  wstart[W] = k;
  *first = lo_id, *last = hi_id;
  return k;
}
// lowest set lane j of the chunk mask m with lo + (weights of the set lanes <= j) >= target, the highest set lane if none;
// *lo_out = the cumulative weight before lane j.  w16: the 16 weights of the chunk (unset lanes are not read).
__attribute__((target("avx512f"))) inline int chunk_scan_avx512(const float* w16, unsigned m, double lo, double target, double* lo_out) {
  const __m512i sh1 = _mm512_setr_epi64(0, 0, 1, 2, 3, 4, 5, 6), sh2 = _mm512_setr_epi64(0, 0, 0, 1, 2, 3, 4, 5), sh4 = _mm512_setr_epi64(0, 0, 0, 0, 0, 1, 2, 3);
  const __m512 p = _mm512_maskz_loadu_ps((__mmask16)m, w16);
  __m512d a = _mm512_cvtps_pd(_mm512_castps512_ps256(p)), b = _mm512_cvtps_pd(_mm256_castpd_ps(_mm512_extractf64x4_pd(_mm512_castps_pd(p), 1)));
  a = _mm512_mask_add_pd(a, 0xfe, a, _mm512_permutexvar_pd(sh1, a)), b = _mm512_mask_add_pd(b, 0xfe, b, _mm512_permutexvar_pd(sh1, b));
  a = _mm512_mask_add_pd(a, 0xfc, a, _mm512_permutexvar_pd(sh2, a)), b = _mm512_mask_add_pd(b, 0xfc, b, _mm512_permutexvar_pd(sh2, b));
  a = _mm512_mask_add_pd(a, 0xf0, a, _mm512_permutexvar_pd(sh4, a)), b = _mm512_mask_add_pd(b, 0xf0, b, _mm512_permutexvar_pd(sh4, b));
  alignas(64) double incl[16];
  _mm512_store_pd(incl, a);
  b = _mm512_add_pd(b, _mm512_set1_pd(incl[7]));
  _mm512_store_pd(incl + 8, b);
  const __m512d t = _mm512_set1_pd(target), l = _mm512_set1_pd(lo);
  const unsigned ge = ((unsigned)_mm512_cmp_pd_mask(_mm512_add_pd(a, l), t, _CMP_GE_OQ) | ((unsigned)_mm512_cmp_pd_mask(_mm512_add_pd(b, l), t, _CMP_GE_OQ) << 8)) & m;
  const int j = ge ? __builtin_ctz(ge) : 31 - __builtin_clz(m);
  *lo_out = j > 0 ? lo + incl[j - 1] : lo;
  return j;
}
#endif

inline int fourth_point_mask(const float* X, const float* Y, const float* Z, const unsigned long long* mask, int nr, const float b[3][3],
                             float too_small, float A, float B, float C, std::vector<float>& tmp) {
#if defined(__x86_64__)
  if (nr >= 64 && __builtin_cpu_supports("avx512f")) {
    tmp.resize((size_t)nr + 16);
    return fourth_point_mask_avx512(X, Y, Z, mask, nr, b, too_small, A, B, C, tmp.data());
  }
  if (nr >= 64 && __builtin_cpu_supports("avx2")) {
    tmp.resize((size_t)nr + 8);
    return fourth_point_mask_avx2(X, Y, Z, mask, nr, b, too_small, A, B, C, tmp.data());
  }
#endif
  return fourth_point_mask_scalar(X, Y, Z, mask, nr, b, too_small, A, B, C);
}

}  // namespace hop

namespace hop {


struct CloudHost {
  std::vector<float> x, y, z, nx, ny, nz;
  int n = 0;
  void resize(int m) {
    n = m;
    x.resize(m), y.resize(m), z.resize(m), nx.resize(m), ny.resize(m), nz.resize(m);
  }
};

// Point3D::set_normal (shared.h:86-88): stored normals are normalised once
inline void load_cloud_host(CloudHost& h, const float* xyz, const float* nrm, int n, bool normalise) {
  h.resize(n);
  for (int i = 0; i < n; ++i) {
    h.x[i] = xyz[i], h.y[i] = xyz[n + i], h.z[i] = xyz[2 * (size_t)n + i];
    V3 nn = v3(nrm[i], nrm[n + i], nrm[2 * (size_t)n + i]);
    if (normalise) nn = vnormalized(nn);
    h.nx[i] = nn.x, h.ny[i] = nn.y, h.nz[i] = nn.z;
  }
}

// the PPF key set as a direct-address bitmap: bit ((d/5*19 + a1/10)*19 + a2/10)*19 + a3/10
inline void build_key_bitmap(const int32_t* keys4, int nkeys, std::vector<unsigned>& bitmap, int& dist_bins) {
  int max_d = 0;
  for (int i = 0; i < nkeys; ++i) max_d = std::max(max_d, keys4[4 * i]);
  dist_bins = max_d / 5 + 1;
  const size_t nbits = (size_t)dist_bins * 19 * 19 * 19;
  bitmap.assign((nbits + 31) / 32 + 1, 0u);
  for (int i = 0; i < nkeys; ++i) {
    const int* k = keys4 + 4 * i;
    // keys are multiples of 5 / 10 by construction (ppfClosestBin); anything else can never be produced
    if (k[0] < 0 || k[0] % 5 || k[1] < 0 || k[1] > 180 || k[1] % 10 || k[2] < 0 || k[2] > 180 || k[2] % 10 || k[3] < 0 || k[3] > 180 || k[3] % 10)
      continue;
    const size_t bit = (((size_t)(k[0] / 5) * 19 + k[1] / 10) * 19 + k[2] / 10) * 19 + k[3] / 10;
    bitmap[bit >> 5] |= 1u << (bit & 31);
  }
}

// key membership of the ordered pair (p1 -> p2) on the host (same arithmetic as k_ppf_matrix)
// Thresholds of ppf_angle_bin_thr (hop_math.h), derived from ppf_angle_bin itself and verified against it:
//  * thr[k] by bisection over the ordered floats of [-1, 1] (bin >= 10 (k+1) holds on a prefix of them),
//  * every float within 2048 ulps of each threshold, and 400 000 floats spread over [-1, 1] (plus the endpoints and
//    the branch points of acosf), must classify identically through the thresholds and through acosf.
// Returns false if any check fails (the caller then keeps the literal kernel).
inline bool build_angle_thresholds(float thr[32]) {
  auto ord = [](float f) { int32_t i; std::memcpy(&i, &f, 4); return i < 0 ? (int32_t)0x80000000 - i : i; };  // monotone float -> int
  auto unord = [](int32_t o) { int32_t i = o < 0 ? (int32_t)0x80000000 - o : o; float f; std::memcpy(&f, &i, 4); return f; };
  auto exact = [](float c) { int b = -1; return ppf_angle_bin(c, &b) ? b : -1; };
  const int32_t lo0 = ord(-1.0f), hi0 = ord(1.0f);
  for (int k = 0; k < 32; ++k) thr[k] = -2.0f;
  for (int k = 0; k < 18; ++k) {
    // largest c with exact(c) >= 10 (k+1); exact(-1) = 180 always qualifies
    int32_t lo = lo0, hi = hi0;  // invariant: exact(unord(lo)) >= target; search the last such
    const int target = 10 * (k + 1);
    if (exact(unord(lo)) < target) return false;
    if (exact(unord(hi)) >= target) return false;
    while (hi - lo > 1) {
      const int32_t mid = lo + (hi - lo) / 2;
      if (exact(unord(mid)) >= target) lo = mid;
      else hi = mid;
    }
    thr[k] = unord(lo);
  }
  auto check = [&](float c) {
    int b = -1;
    const bool ok = ppf_angle_bin_thr(c, thr, &b);
    return ok && b == exact(c);
  };
  for (int k = 0; k < 18; ++k) {
    const int32_t o = ord(thr[k]);
    for (int32_t d = -2048; d <= 2048; ++d) {
      const int32_t q = o + d;
      if (q < lo0 || q > hi0) continue;
      if (!check(unord(q))) return false;
    }
  }
  const long long span = (long long)hi0 - (long long)lo0;
  for (int s = 0; s <= 400000; ++s)
    if (!check(unord((int32_t)(lo0 + span * s / 400000)))) return false;
  for (float c : {-1.0f, 1.0f, 0.0f, -0.0f, 0.5f, -0.5f, 1e-9f, -1e-9f})
    for (int32_t d = -64; d <= 64; ++d) {
      const int32_t q = ord(c) + d;
      if (q >= lo0 && q <= hi0 && !check(unord(q))) return false;
    }
  int b;
  if (ppf_angle_bin_thr(1.0000001f, thr, &b) || ppf_angle_bin_thr(-1.0000001f, thr, &b) || ppf_angle_bin_thr(NAN, thr, &b)) return false;
  return true;
}

inline bool ppf_member_host(V3 p1, V3 n1p, V3 p2, V3 n2p, const std::vector<unsigned>& bitmap, int dist_bins) {
  int key[4];
  if (!ppf_key(p1, n1p, p2, n2p, key)) return false;
  const int d = key[0] / 5, a1 = key[1] / 10, a2 = key[2] / 10, a3 = key[3] / 10;
  if (key[0] < 0 || d >= dist_bins || (unsigned)a1 >= 19u || (unsigned)a2 >= 19u || (unsigned)a3 >= 19u) return false;
  const unsigned bit = ((unsigned)(d * 19 + a1) * 19u + (unsigned)a2) * 19u + (unsigned)a3;
  return (bitmap[bit >> 5] >> (bit & 31)) & 1u;
}

// bits of `cand` (a subset of `rb`) re-indexed by rank within `rb` (the k-th set bit of rb -> bit k)
inline unsigned long long compress_bits_generic(unsigned long long cand, unsigned long long rb) {
  unsigned long long out = 0;
  int k = 0;
  while (rb) {
    const unsigned long long low = rb & (0ull - rb);
    if (cand & low) out |= 1ull << k;
    rb ^= low;
    ++k;
  }
  return out;
}
#if defined(__x86_64__)
__attribute__((target("bmi2"))) inline unsigned long long compress_bits_bmi2(unsigned long long cand, unsigned long long rb) {
  return _pext_u64(cand, rb);
}
#endif

// ---- weighted index draws -------------------------------------------------------------------------
// std::discrete_distribution<int>(w.begin(), w.end())(engine) in libstdc++ (bits/random.tcc, _M_initialize and
// operator()):  sum = accumulate(w as double, 0.0) ; p_i = w_i / sum ; cp = partial_sum(p) ; cp.back() = 1.0 ;
// u = generate_canonical<double,53>(engine) ; return lower_bound(cp, u) - cp.begin().
// exact_discrete_index() repeats exactly that for a given u.  The selection below normally answers the same
// question from a Fenwick tree of the raw weights in O(log n) and falls back to the exact routine whenever u is
// within 1e-10 (relative) of a bin boundary, where the different summation order could matter (the orders differ
// by < 2e-12), so the drawn indices are provably the reference's.
inline int exact_discrete_index(const float* w, int n, double u) {
  double sum = 0.0;
  for (int i = 0; i < n; ++i) sum += (double)w[i];
  double cp = 0.0;
  for (int i = 0; i < n - 1; ++i) {
    cp += (double)w[i] / sum;
    if (!(cp < u)) return i;  // lower_bound: first cp >= u
  }
  return n - 1;  // cp.back() is forced to 1.0 and u < 1
}

struct Fenwick {
  std::vector<double> t;  // 1-based
  int n = 0, top = 1;
  void build(const float* w, int n_) {
    n = n_;
    t.assign((size_t)n + 1, 0.0);
    for (int i = 1; i <= n; ++i) {
      t[i] += (double)w[i - 1];
      const int j = i + (i & -i);
      if (j <= n) t[j] += t[i];
    }
    top = 1;
    while ((top << 1) <= n) top <<= 1;
  }
  void build_d(const double* w, int n_) {
    n = n_;
    t.assign((size_t)n + 1, 0.0);
    for (int i = 1; i <= n; ++i) {
      t[i] += w[i - 1];
      const int j = i + (i & -i);
      if (j <= n) t[j] += t[i];
    }
    top = 1;
    while ((top << 1) <= n) top <<= 1;
  }
  void add(int i, double d) {
    for (++i; i <= n; i += i & -i) t[i] += d;
  }
  double prefix(int count) const {  // sum of the first `count` weights
    double s = 0.0;
    for (int i = count; i > 0; i -= i & -i) s += t[i];
    return s;
  }
  double total() const { return prefix(n); }
  // smallest index r (0-based) with prefix(r+1) >= target; n-1 if none.  *before = prefix(r).
  int find(double target, double* before = nullptr) const {
    int pos = 0;
    double acc = 0.0;
    for (int step = top; step > 0; step >>= 1) {
      const int nxt = pos + step;
      if (nxt <= n && acc + t[nxt] < target) {
        pos = nxt;
        acc += t[nxt];
      }
    }
    if (pos >= n) {
      pos = n - 1;
      acc = prefix(pos);
    }
    if (before) *before = acc;
    return pos;
  }
};

struct GenState {
  CloudHost scene_h;              // _scene_high_confidence, Point3D-style normals (normalised once)
  std::vector<float> scene_conf;
  CloudHost model_h[2];
  CloudHost gp_h;                 // centred P
  std::vector<float> gp_prob;
  CloudHost gq_h;                 // sampled, centred Q
  std::vector<float> gq_unit[3];  // the same points in the unit cube (pairCreationFunctor.h:129-161)
  float centroid_p[3] = {0, 0, 0}, centroid_q[3] = {0, 0, 0};
  float diameter = 0, ratio = 1;
};

// per-base constants the device side needs (see BaseDev in hop_device.h); plain floats so this header stays host-only
struct BaseHostOut {
  float bpos[4][3];
  float dist1, dist2, inv1, inv2;
  EdgeFeat e1, e2;
  int nb_sample;
  float ring[64][3];
};

// ------------------------------------------------------------------------------------------------
// generator, host part
// ------------------------------------------------------------------------------------------------
struct GenHost {
  GenState* c;
  hop_s4pcs_opts opt;
  std::mt19937 randomGenerator_;     // matchBase.hpp:73
  std::mt19937 point_index_engine_;  // matchBase.hpp:76, seed 0
  std::vector<float> point_probs_;
  int n = 0;  // |P|
  const unsigned long long* M = nullptr;
  int W = 0;
  float max_base_diameter_ = -1;
  std::array<V3, 4> bpos, bnrm;
  bool use_fast = true;
  double guard_tol = 1e-10;  // relative half-width of the zone around a bin boundary that is resolved exactly
  Fenwick fw_all_, fw_blk_;
  std::vector<int> pool_ids_;
  bool fw_all_valid_ = false;
  std::vector<float> probs_;
  std::vector<float> dist4_;
  std::vector<unsigned long long> mask4_;  // 4th-point candidates as a bit mask over pool ranks
  std::vector<double> bsum_;               // per matrix word: sum of the pool weights of that word
  std::vector<double> c0_, c1_;            // AVX-512 path: running sums of bsum_ over / inside super-blocks of 16 words
  int nsb_ = 0;
  std::vector<int> wstart_;                // per matrix word: rank of its first pool member
  int npool_ = 0, n4_ = 0;
  double pool_total_ = 0.0;
  bool have_bmi2_ = false, have_avx512_ = false, simd_draw_ = false;
  int lookahead_ = 2;
  // profile (seconds) of the selection phases, for tools/select_bench.cpp
  double t_first = 0, t_pool = 0, t_pairdraw = 0, t_pool4 = 0, t_fourth = 0;
  long long n_tri_calls = 0, n_pair_iters = 0, n_fallbacks = 0, sum_pool = 0, n_same = 0, n_nobit = 0, n_geom = 0;
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  void print_profile() const {
    std::printf("  avg pool %.0f, same %lld, nobit %lld, geom-tested %lld\n  triangle calls %lld, pair-draw iterations %lld, exact fallbacks %lld\n"
                "  first %.1f ms, pool %.1f ms, pair draws %.1f ms, pool4 %.1f ms, fourth %.1f ms\n",
                (double)sum_pool / (double)std::max(1ll, n_tri_calls), n_same, n_nobit, n_geom, n_tri_calls, n_pair_iters, n_fallbacks,
                1e3 * t_first, 1e3 * t_pool, 1e3 * t_pairdraw, 1e3 * t_pool4, 1e3 * t_fourth);
  }

  GenHost(GenState* ctx, const hop_s4pcs_opts& o) : c(ctx), opt(o), randomGenerator_(o.random_seed), point_index_engine_(0) {}

  bool bit(int i, int j) const { return (M[(size_t)i * W + (j >> 6)] >> (j & 63)) & 1ull; }
  V3 ppos(int i) const { return v3(c->gp_h.x[i], c->gp_h.y[i], c->gp_h.z[i]); }
  V3 pnrm(int i) const { return v3(c->gp_h.nx[i], c->gp_h.ny[i], c->gp_h.nz[i]); }

  // UniformDistSampler (sampling.h:67-144): first point of every delta-voxel, open-addressing hash
  static void uniform_sample(const CloudHost& in, float delta, std::vector<int>& keep) {
    const uint64_t MAGIC1 = 100000007, MAGIC2 = 161803409, MAGIC3 = 423606823, NO_DATA = 0xffffffffu;
    const int num_input = in.n;
    const float scale_ = 1.0f / delta;
    std::vector<std::array<int, 3>> voxels_(num_input);
    std::vector<uint64_t> data_(num_input, NO_DATA);
    keep.clear();
    for (int i = 0; i < num_input; ++i) {
      const std::array<int, 3> cell{int(std::floor(in.x[i] * scale_)), int(std::floor(in.y[i] * scale_)), int(std::floor(in.z[i] * scale_))};
      uint64_t key = (MAGIC1 * (uint64_t)(int64_t)cell[0] + MAGIC2 * (uint64_t)(int64_t)cell[1] + MAGIC3 * (uint64_t)(int64_t)cell[2]) % data_.size();
      while (true) {
        if (data_[key] == NO_DATA) {
          voxels_[key] = cell;
          break;
        } else if (voxels_[key] == cell)
          break;
        if (++key == data_.size()) key = 0;
      }
      if (data_[key] >= (uint64_t)num_input) {
        keep.push_back(i);
        data_[key] = keep.size();
      }
    }
  }

  // MatchBase::init (matchBase.hpp:380-462) minus the kd-tree; fills ctx->gp_h / gq_h / centroids / diameter
  void init_clouds() {
    const CloudHost& P = c->scene_h;
    const CloudHost& Q = c->model_h[HOP_MODEL_5MM];
    std::vector<int> qsel;
    if (Q.n > opt.sample_size) {
      uniform_sample(Q, opt.delta, qsel);
      std::shuffle(qsel.begin(), qsel.end(), randomGenerator_);
      if ((int)qsel.size() > opt.sample_size) qsel.resize(opt.sample_size);
    } else {
      qsel.resize(Q.n);
      std::iota(qsel.begin(), qsel.end(), 0);
    }
    CloudHost& gp = c->gp_h;
    CloudHost& gq = c->gq_h;
    gp = P;
    c->gp_prob = c->scene_conf;
    gq.resize((int)qsel.size());
    for (int k = 0; k < gq.n; ++k) {
      const int i = qsel[k];
      gq.x[k] = Q.x[i], gq.y[k] = Q.y[i], gq.z[k] = Q.z[i], gq.nx[k] = Q.nx[i], gq.ny[k] = Q.ny[i], gq.nz[k] = Q.nz[i];
    }
    auto centre = [](CloudHost& cl, float cen[3]) {
      V3 s = v3(0, 0, 0);
      for (int i = 0; i < cl.n; ++i) s = s + v3(cl.x[i], cl.y[i], cl.z[i]);
      s = s / float(cl.n);
      for (int i = 0; i < cl.n; ++i) {
        const V3 p = v3(cl.x[i], cl.y[i], cl.z[i]) - s;
        cl.x[i] = p.x, cl.y[i] = p.y, cl.z[i] = p.z;
      }
      cen[0] = s.x, cen[1] = s.y, cen[2] = s.z;
    };
    centre(gp, c->centroid_p);
    centre(gq, c->centroid_q);
    // "diameter of P", measured on sampled Q (matchBase.hpp:439-448)
    float diam = 0.f;
    for (int i = 0; i < 1000; ++i) {
      const int at = int(randomGenerator_() % (unsigned long)gq.n);
      const int bt = int(randomGenerator_() % (unsigned long)gq.n);
      const float l = vnorm(v3(gq.x[bt], gq.y[bt], gq.z[bt]) - v3(gq.x[at], gq.y[at], gq.z[at]));
      if (l > diam) diam = l;
    }
    c->diameter = diam;
    max_base_diameter_ = diam;
    // PairCreationFunctor::synch3DContent (pairCreationFunctor.h:129-161)
    V3 mn = v3(FLT_MAX, FLT_MAX, FLT_MAX), mx = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int i = 0; i < gq.n; ++i) {
      mn = v3(std::min(mn.x, gq.x[i]), std::min(mn.y, gq.y[i]), std::min(mn.z, gq.z[i]));
      mx = v3(std::max(mx.x, gq.x[i]), std::max(mx.y, gq.y[i]), std::max(mx.z, gq.z[i]));
    }
    const V3 gcenter = (mn + mx) / 2.f;
    const V3 diag = mx - mn;
    c->ratio = (float)((double)std::max(diag.x, std::max(diag.y, diag.z)) + 0.001);
    for (int k = 0; k < 3; ++k) c->gq_unit[k].resize(gq.n);
    const V3 half = v3(0.5f, 0.5f, 0.5f);
    for (int i = 0; i < gq.n; ++i) {
      const V3 u = (v3(gq.x[i], gq.y[i], gq.z[i]) - gcenter) / c->ratio + half;
      c->gq_unit[0][i] = u.x, c->gq_unit[1][i] = u.y, c->gq_unit[2][i] = u.z;
    }
    n = gp.n;
    point_probs_ = c->gp_prob;
  }


  // index drawn by std::discrete_distribution over w[0..n) for this engine state (see exact_discrete_index)
  int draw_index(const Fenwick& fw, const float* w, int n) {
    const double u = std::generate_canonical<double, std::numeric_limits<double>::digits>(point_index_engine_);
    const double T = fw.total();
    const double target = u * T;
    const int r = fw.find(target);
    const double hi = fw.prefix(r + 1), lo = hi - (double)w[r];
    const double tol = guard_tol * T;
    if ((r == 0 || target - lo > tol) && (r == n - 1 || hi - target > tol) && T > 0.0) return r;
    ++n_fallbacks;
    return exact_discrete_index(w, n, u);
  }

  // index drawn by std::discrete_distribution over the pool weights probs_[0..npool_) (pool order), answered from the
  // per-word block sums; same exactness guard as draw_index.
  struct PoolDraw {
    int r;          // pool rank
    double lo, hi;  // cumulative weight before / through rank r when the draw was resolved
    double u;       // the canonical variate
  };
  PoolDraw draw_pool_index(double u) {
    const double T = pool_total_;
    const double target = u * T;
    double lo = 0.0, hi;
    int r;
#if defined(__x86_64__)
    if (simd_draw_) {
      // first word whose cumulative block sum reaches the target: count of cumulative sums below it (no branches to
      // mispredict); then the same inside the word on prefix sums formed in registers
      const int sb = std::min(count_below_avx512(c0_.data(), (int)c0_.size(), target), nsb_ - 1);
      const double base0 = sb > 0 ? c0_[sb - 1] : 0.0;
      const int wi = count_below_off_avx512(c1_.data() + 16 * sb, 16, base0, target);
      const int w = std::min(16 * sb + wi, W - 1);
      const int rs = wstart_[w], cnt = wstart_[w + 1] - rs;
      if (cnt == 0) {
        ++n_fallbacks;
        return {exact_discrete_index(probs_.data(), npool_, u), 0.0, -1.0, u};
      }
      lo = base0 + ((w & 15) > 0 ? c1_[w - 1] : 0.0);
      r = rs + word_scan_avx512(probs_.data() + rs, cnt, lo, target, &lo);
      hi = lo + (double)probs_[r];
    } else
#endif
    {
      const int w = fw_blk_.find(target, &lo);  // first word whose cumulative block sum reaches the target
      if (wstart_[w + 1] == wstart_[w]) {  // an empty word can only be hit through rounding: resolve exactly
        ++n_fallbacks;
        return {exact_discrete_index(probs_.data(), npool_, u), 0.0, -1.0, u};
      }
      r = wstart_[w];
      const int rend = wstart_[w + 1];
      for (; r < rend - 1; ++r) {
        if (lo + (double)probs_[r] >= target) break;
        lo += (double)probs_[r];
      }
      hi = lo + (double)probs_[r];
    }
    const double tol = guard_tol * T;
    // accept only when the target is clearly inside (lo, hi]; anything near a boundary (or a block-level miss) is
    // resolved by the exact routine
    if (T > 0.0 && target - lo > tol && hi - target > tol) return {r, lo, hi, u};
    if (T > 0.0 && r == 0 && target <= hi - tol) return {r, lo, hi, u};
    if (T > 0.0 && r == npool_ - 1 && target - lo > tol) return {r, lo, hi, u};
    ++n_fallbacks;
    return {exact_discrete_index(probs_.data(), npool_, u), 0.0, -1.0, u};
  }
  // A draw resolved before the weights of ranks a and b dropped by da, db (<= 0): still the same rank afterwards?
  // (the cumulative bounds move by the deltas of the lower ranks, the total by both; same guard zone as above)
  bool redraw_unchanged(PoolDraw& d, int a, double da, int b, double db) const {
    if (d.hi < d.lo) return false;  // came from the exact routine: no bounds kept
    const double lo = d.lo + (a < d.r ? da : 0.0) + (b < d.r ? db : 0.0);
    const double hi = lo + (double)probs_[d.r];
    const double T = pool_total_, target = d.u * T, tol = guard_tol * T;
    if (!(T > 0.0)) return false;
    const bool ok = (target - lo > tol && hi - target > tol) || (d.r == 0 && target <= hi - tol) || (d.r == npool_ - 1 && target - lo > tol);
    if (ok) d.lo = lo, d.hi = hi;
    return ok;
  }
  // std::generate_canonical<double, 53>(std::mt19937&): two 32-bit draws a, b -> round_to_double(a + b 2^32) / 2^64
  // (random.tcc: the sum is accumulated in double, one rounding, then divided by the exact range 2^64; results
  // >= 1 are replaced by the largest double below 1).  The u64 -> double conversion rounds identically.
  double canonical53() {
    const uint64_t a = point_index_engine_(), b = point_index_engine_();
    const double r = (double)(a | (b << 32)) * 5.421010862427522e-20;  // 2^-64
    return r >= 1.0 ? 0.99999999999999988898 : r;
  }

  // SelectRandomTriangle with the same draws, decisions and side effects as the literal version below, but
  //  * index draws from partial sums (Fenwick tree over all points, per-word block sums over the pool) with an
  //    exactness guard instead of rebuilding std::discrete_distribution for every draw,
  //  * the 4th-point candidates as a bit mask over pool ranks, from word-wise ANDs of three matrix rows.
  bool SelectRandomTriangleFast(int& base1, int& base2, int& base3) {
    base1 = base2 = base3 = -1;
    ++n_tri_calls;
    double tp = now();
    if (!fw_all_valid_) {
      fw_all_.build(point_probs_.data(), n);
      fw_all_valid_ = true;
      point_probs_.resize((size_t)W * 64 + 16, 0.f);  // vector loads of whole words stay in bounds (n itself is unchanged)
#if defined(__x86_64__)
      have_bmi2_ = __builtin_cpu_supports("bmi2");
      have_avx512_ = __builtin_cpu_supports("avx512f");
      simd_draw_ = have_avx512_;
      if (const char* e = getenv("HOP_SELECT_SIMD_DRAW")) simd_draw_ = have_avx512_ && atoi(e) != 0;
      if (const char* e = getenv("HOP_SELECT_LOOKAHEAD")) lookahead_ = std::max(1, std::min(8, atoi(e)));
#endif
      bsum_.assign(W, 0.0);
      wstart_.assign(W + 1, 0);
    }
#if defined(__x86_64__)
    if (simd_draw_) return SelectRandomTriangleNoPool(base1, base2, base3, tp);
#endif
    const int first_point = draw_index(fw_all_, point_probs_.data(), n);
    {
      const float old = point_probs_[first_point];
      point_probs_[first_point] *= opt.dispersion;
      fw_all_.add(first_point, (double)point_probs_[first_point] - (double)old);
    }
    t_first += now() - tp, tp = now();
    const unsigned long long* row = M + (size_t)first_point * W;
    int npool = 0;
    for (int w = 0; w < W; ++w) npool += __builtin_popcountll(row[w]);
    npool_ = npool;
    if (pool_ids_.size() < (size_t)n + 64) pool_ids_.resize((size_t)n + 64), probs_.resize((size_t)n + 64);
    bool built = false;
#if defined(__x86_64__)
    if (have_avx512_) {
      pool_build_avx512(row, W, point_probs_.data(), pool_ids_.data(), probs_.data(), wstart_.data(), bsum_.data());
      built = true;
    }
#endif
    if (!built) {
      int k = 0;
      int* ids = pool_ids_.data();
      float* pr = probs_.data();
      const float* pp = point_probs_.data();
      for (int w = 0; w < W; ++w) {
        unsigned long long bits = row[w];
        const int base = w * 64;
        wstart_[w] = k;
        double s = 0.0;
        while (bits) {
          const int i = base + __builtin_ctzll(bits);
          bits &= bits - 1;
          ids[k] = i;
          pr[k] = pp[i];
          s += (double)pp[i];
          ++k;
        }
        bsum_[w] = s;
      }
      wstart_[W] = k;
    }
    if (simd_draw_) {
      // two levels of inclusive running sums over the words: c1_ inside super-blocks of 16 words, c0_ over super-blocks
      nsb_ = (W + 15) / 16;
      c1_.assign((size_t)nsb_ * 16, HUGE_VAL);
      c0_.assign((size_t)((nsb_ + 7) & ~7), HUGE_VAL);
      double tot = 0.0;
      for (int sb = 0; sb < nsb_; ++sb) {
        double acc = 0.0;
        for (int w = 16 * sb; w < std::min(W, 16 * sb + 16); ++w) acc += bsum_[w], c1_[w] = acc;
        tot += acc;
        c0_[sb] = tot;
      }
      pool_total_ = tot;
    } else {
      fw_blk_.build_d(bsum_.data(), W);
      pool_total_ = fw_blk_.total();
    }
    t_pool += now() - tp, tp = now();
    sum_pool += npool;
    if (npool < 3) return false;
    const float sq_max = max_base_diameter_ * max_base_diameter_;
    const V3 p0 = ppos(first_point);
    const size_t max_it = (size_t)npool * (size_t)npool / 4;
    // Every iteration consumes exactly two canonical variates (four engine calls), whatever its outcome, so the
    // variates of the next iteration can be drawn one iteration early and its matrix word prefetched: the key test
    // bit(p2,p3) is a random access into the N^2/8-byte matrix (a DRAM miss at C2 sizes).  The indices computed
    // ahead are only reused if the current iteration did not change the weights; the engine ends exactly where the
    // reference's would (an unused look-ahead pair is rolled back).
    const std::mt19937 engine_before = point_index_engine_;
    // Every iteration consumes exactly two canonical variates (four engine calls), whatever its outcome, so the pairs
    // of the next LOOKAHEAD iterations are drawn early and their matrix words prefetched: the key test bit(p2,p3) is a
    // random access into the N^2/8-byte matrix (a DRAM + TLB miss at C2 sizes, longer than one iteration).  When an
    // iteration changes the weights, each queued draw is re-validated in O(1) (redraw_unchanged) or redone; the engine
    // ends exactly where the reference's would (unused look-ahead pairs are rolled back).
    constexpr int MAXLA = 8;
    const int LOOKAHEAD = lookahead_;
    PoolDraw q0[MAXLA + 1], q1[MAXLA + 1];
    int qhead = 0, qlen = 0;
    size_t drawn_pairs = 0, used_pairs = 0;
    auto push_pair = [&]() {
      const double u0 = canonical53(), u1 = canonical53();
      ++drawn_pairs;
      const int at = (qhead + qlen) % (LOOKAHEAD + 1);
      q0[at] = draw_pool_index(u0), q1[at] = draw_pool_index(u1);
      if (q0[at].r != q1[at].r) __builtin_prefetch(&M[(size_t)pool_ids_[q0[at].r] * W + (pool_ids_[q1[at].r] >> 6)], 0, 1);
      ++qlen;
    };
    for (size_t it = 0; it < max_it && it < (size_t)INT_MAX; ++it) {  // the reference's counter is an int
      ++n_pair_iters;
      while (qlen < LOOKAHEAD + 1 && it + (size_t)qlen < max_it) push_pair();
      const int second = q0[qhead].r, third = q1[qhead].r;
      qhead = (qhead + 1) % (LOOKAHEAD + 1), --qlen;
      used_pairs = it + 1;
      if (second == third) { ++n_same; continue; }
      if (!bit(pool_ids_[second], pool_ids_[third])) { ++n_nobit; continue; }
      ++n_geom;
      // the weights of both points drop (matchBase.hpp:163-164); the block sums follow by the exact differences
      double dlt[2];
      int q = 0;
      for (int r : {second, third}) {
        const float old = probs_[r];
        probs_[r] = old * opt.dispersion;
        const double d = (double)probs_[r] - (double)old;
        const int w = pool_ids_[r] >> 6;
        bsum_[w] += d;
#if defined(__x86_64__)
        if (simd_draw_) {
          suffix_add_avx512(c1_.data() + 16 * (w >> 4), w & 15, 16, d);
          suffix_add_avx512(c0_.data(), w >> 4, (int)c0_.size(), d);
        }
        else
#endif
          fw_blk_.add(w, d);
        pool_total_ += d;
        dlt[q++] = d;
      }
      const V3 u = ppos(pool_ids_[second]) - p0;
      const V3 w = ppos(pool_ids_[third]) - p0;
      const float how_wide = vdot(vnormalized(u), vnormalized(w));
      if ((double)std::fabs(how_wide) <= std::cos(45 * M_PI / 180.0) && vsqn(u) < sq_max && vsqn(w) < sq_max) {
        base1 = first_point;
        base2 = pool_ids_[second];
        base3 = pool_ids_[third];
        break;
      }
      // keep the queued draws that the change did not move, redo the others
      for (int k = 0; k < qlen; ++k) {
        const int at = (qhead + k) % (LOOKAHEAD + 1);
        bool moved = false;
        if (!redraw_unchanged(q0[at], second, dlt[0], third, dlt[1])) q0[at] = draw_pool_index(q0[at].u), moved = true;
        if (!redraw_unchanged(q1[at], second, dlt[0], third, dlt[1])) q1[at] = draw_pool_index(q1[at].u), moved = true;
        if (moved && q0[at].r != q1[at].r) __builtin_prefetch(&M[(size_t)pool_ids_[q0[at].r] * W + (pool_ids_[q1[at].r] >> 6)], 0, 1);
      }
    }
    if (drawn_pairs != used_pairs) {  // roll the engine back to "used_pairs pairs consumed"
      point_index_engine_ = engine_before;
      point_index_engine_.discard(4ull * used_pairs);
    }
    t_pairdraw += now() - tp, tp = now();
    if (base2 == -1 || base3 == -1) return false;
    const bool ok4 = build_mask4(row, base1, base2, base3, npool);
    t_pool4 += now() - tp;
    return ok4;
  }

#if defined(__x86_64__)
  // ---- AVX-512 form of the fast path: no materialised pool -------------------------------------------------------
  // The pool members are the set bits of the first point's matrix row; draws return POINT IDS (ranks and ids order the
  // pool identically, so every comparison the rank form makes carries over), weights are read through the row masks from
  // wloc_ = point_probs_ with this call's drops applied (undone when the call ends), and a draw descends four levels of
  // sums: super-blocks of 16 words, words, 16-lane chunks, lanes -- at most 16 weights are scanned per draw instead of 64.
  std::vector<float> wloc_;
  std::vector<double> cs_;  // per 16-lane chunk of a matrix word: weight of its pool members
  std::vector<std::pair<int, float>> undo_;
  int pool_first_ = -1, pool_last_ = -1;
  const unsigned long long* row_ = nullptr;
  struct IdDraw {
    int id;         // point id
    double lo, hi;  // cumulative weight before / through it when the draw was resolved (hi < lo: from the exact routine)
    double u;
  };
  // the reference's own arithmetic on the pool as a vector (rare: a draw within the guard zone of a bin boundary)
  IdDraw exact_pool_id(double u) {
    ++n_fallbacks;
    int k = 0;
    for (int w = 0; w < W; ++w) {
      unsigned long long bits = row_[w];
      while (bits) {
        const int i = w * 64 + __builtin_ctzll(bits);
        bits &= bits - 1;
        pool_ids_[k] = i, probs_[k] = wloc_[i], ++k;
      }
    }
    return {pool_ids_[exact_discrete_index(probs_.data(), npool_, u)], 0.0, -1.0, u};
  }
  bool id_draw_ok(int id, double lo, double hi, double target, double T) const {
    const double tol = guard_tol * T;
    if (!(T > 0.0)) return false;
    return (target - lo > tol && hi - target > tol) || (id == pool_first_ && target <= hi - tol) || (id == pool_last_ && target - lo > tol);
  }
  IdDraw draw_pool_id(double u) {
    const double T = pool_total_, target = u * T;
    const int sb = std::min(count_below_avx512(c0_.data(), (int)c0_.size(), target), nsb_ - 1);
    const double base0 = sb > 0 ? c0_[sb - 1] : 0.0;
    const int wi = count_below_off_avx512(c1_.data() + 16 * sb, 16, base0, target);
    const int w = std::min(16 * sb + wi, W - 1);
    const unsigned long long bits = row_[w];
    if (!bits) return exact_pool_id(u);  // an empty word can only be hit through rounding
    // chunk: the first whose running sum reaches the target, the last non-empty one if none does (no branches to mispredict)
    const double* cw = cs_.data() + 4 * w;
    double run[5];
    run[0] = base0 + ((w & 15) > 0 ? c1_[w - 1] : 0.0);
    run[1] = run[0] + cw[0], run[2] = run[1] + cw[1], run[3] = run[2] + cw[2], run[4] = run[3] + cw[3];
    const int below = (run[1] < target) + (run[2] < target) + (run[3] < target) + (run[4] < target);
    const int sel = std::min(below, 3 - (__builtin_clzll(bits) >> 4));
    if (!((bits >> (16 * sel)) & 0xffffull)) return exact_pool_id(u);  // (an empty chunk: only through rounding)
    const double before = run[sel];
    double lo;
    const int lane = chunk_scan_avx512(wloc_.data() + w * 64 + sel * 16, (unsigned)((bits >> (16 * sel)) & 0xffffull), before, target, &lo);
    const int id = w * 64 + sel * 16 + lane;
    const double hi = lo + (double)wloc_[id];
    if (id_draw_ok(id, lo, hi, target, T)) return {id, lo, hi, u};
    return exact_pool_id(u);
  }
  // a draw resolved before the weights of ids a and b dropped by da, db (<= 0): still the same member afterwards?
  bool redraw_unchanged_id(IdDraw& d, int a, double da, int b, double db) const {
    if (d.hi < d.lo) return false;
    const double lo = d.lo + (a < d.id ? da : 0.0) + (b < d.id ? db : 0.0);
    const double hi = lo + (double)wloc_[d.id];
    const double T = pool_total_;
    const bool ok = id_draw_ok(d.id, lo, hi, d.u * T, T);
    if (ok) d.lo = lo, d.hi = hi;
    return ok;
  }
  bool SelectRandomTriangleNoPool(int& base1, int& base2, int& base3, double tp) {
    if (wloc_.size() != point_probs_.size()) {
      wloc_ = point_probs_;
      cs_.assign((size_t)4 * W, 0.0);
      if (pool_ids_.size() < (size_t)n + 64) pool_ids_.resize((size_t)n + 64), probs_.resize((size_t)n + 64);
    }
    const int first_point = draw_index(fw_all_, point_probs_.data(), n);
    {
      const float old = point_probs_[first_point];
      point_probs_[first_point] *= opt.dispersion;
      wloc_[first_point] = point_probs_[first_point];
      fw_all_.add(first_point, (double)point_probs_[first_point] - (double)old);
    }
    t_first += now() - tp, tp = now();
    const unsigned long long* row = M + (size_t)first_point * W;
    row_ = row;
    const int npool = pool_sums_avx512(row, W, wloc_.data(), cs_.data(), bsum_.data(), wstart_.data(), &pool_first_, &pool_last_);
    npool_ = npool;
    nsb_ = (W + 15) / 16;
    c1_.assign((size_t)nsb_ * 16, HUGE_VAL);
    c0_.assign((size_t)((nsb_ + 7) & ~7), HUGE_VAL);
    {
      double tot = 0.0;
      for (int sb = 0; sb < nsb_; ++sb) {
        double acc = 0.0;
        for (int w = 16 * sb; w < std::min(W, 16 * sb + 16); ++w) acc += bsum_[w], c1_[w] = acc;
        tot += acc;
        c0_[sb] = tot;
      }
      pool_total_ = tot;
    }
    t_pool += now() - tp, tp = now();
    sum_pool += npool;
    if (npool < 3) return false;
    const float sq_max = max_base_diameter_ * max_base_diameter_;
    const V3 p0 = ppos(first_point);
    const size_t max_it = (size_t)npool * (size_t)npool / 4;
    // (look-ahead queue, engine roll-back: as in the rank form above)
    const std::mt19937 engine_before = point_index_engine_;
    constexpr int MAXLA = 8;
    const int LOOKAHEAD = lookahead_;
    IdDraw q0[MAXLA + 1], q1[MAXLA + 1];
    int qhead = 0, qlen = 0;
    size_t drawn_pairs = 0, used_pairs = 0;
    undo_.clear();
    auto push_pair = [&]() {
      const double u0 = canonical53(), u1 = canonical53();
      ++drawn_pairs;
      const int at = (qhead + qlen) % (LOOKAHEAD + 1);
      q0[at] = draw_pool_id(u0), q1[at] = draw_pool_id(u1);
      if (q0[at].id != q1[at].id) __builtin_prefetch(&M[(size_t)q0[at].id * W + (q1[at].id >> 6)], 0, 1);
      ++qlen;
    };
    for (size_t it = 0; it < max_it && it < (size_t)INT_MAX; ++it) {
      ++n_pair_iters;
      while (qlen < LOOKAHEAD + 1 && it + (size_t)qlen < max_it) push_pair();
      const int second = q0[qhead].id, third = q1[qhead].id;
      qhead = (qhead + 1) % (LOOKAHEAD + 1), --qlen;
      used_pairs = it + 1;
      if (second == third) { ++n_same; continue; }
      if (!bit(second, third)) { ++n_nobit; continue; }
      ++n_geom;
      double dlt[2];
      int q = 0;
      for (int id : {second, third}) {
        const float old = wloc_[id];
        undo_.emplace_back(id, old);
        wloc_[id] = old * opt.dispersion;
        const double d = (double)wloc_[id] - (double)old;
        const int w = id >> 6;
        cs_[4 * w + ((id >> 4) & 3)] += d;
        bsum_[w] += d;
        suffix_add_avx512(c1_.data() + 16 * (w >> 4), w & 15, 16, d);
        suffix_add_avx512(c0_.data(), w >> 4, (int)c0_.size(), d);
        pool_total_ += d;
        dlt[q++] = d;
      }
      const V3 u = ppos(second) - p0;
      const V3 w = ppos(third) - p0;
      const float how_wide = vdot(vnormalized(u), vnormalized(w));
      if ((double)std::fabs(how_wide) <= std::cos(45 * M_PI / 180.0) && vsqn(u) < sq_max && vsqn(w) < sq_max) {
        base1 = first_point, base2 = second, base3 = third;
        break;
      }
      for (int k = 0; k < qlen; ++k) {
        const int at = (qhead + k) % (LOOKAHEAD + 1);
        bool moved = false;
        if (!redraw_unchanged_id(q0[at], second, dlt[0], third, dlt[1])) q0[at] = draw_pool_id(q0[at].u), moved = true;
        if (!redraw_unchanged_id(q1[at], second, dlt[0], third, dlt[1])) q1[at] = draw_pool_id(q1[at].u), moved = true;
        if (moved && q0[at].id != q1[at].id) __builtin_prefetch(&M[(size_t)q0[at].id * W + (q1[at].id >> 6)], 0, 1);
      }
    }
    if (drawn_pairs != used_pairs) {
      point_index_engine_ = engine_before;
      point_index_engine_.discard(4ull * used_pairs);
    }
    for (size_t k = undo_.size(); k-- > 0;) wloc_[undo_[k].first] = undo_[k].second;  // the drops were this call's only
    t_pairdraw += now() - tp, tp = now();
    if (base2 == -1 || base3 == -1) return false;
    const bool ok = build_mask4(row, base1, base2, base3, npool);
    t_pool4 += now() - tp;
    return ok;
  }
#endif
  // 4th-point candidates: pool RANKS (not point ids -- the reference stores the loop index, matchBase.hpp:203) of the
  // pool members compatible with base2 and base3, as a bit mask over [0, npool)
  bool build_mask4(const unsigned long long* row, int base1, int base2, int base3, int npool) {
    const unsigned long long* r2 = M + (size_t)base2 * W;
    const unsigned long long* r3 = M + (size_t)base3 * W;
    mask4_.assign(((size_t)npool + 63) / 64 + 1, 0ull);
    int n4 = 0;
    for (int w = 0; w < W; ++w) {
      const unsigned long long rb = row[w];
      if (!rb) continue;
      unsigned long long cand = rb & r2[w] & r3[w];
      if ((base2 >> 6) == w) cand &= ~(1ull << (base2 & 63));
      if ((base3 >> 6) == w) cand &= ~(1ull << (base3 & 63));
      if ((base1 >> 6) == w) cand &= ~(1ull << (base1 & 63));
      if (!cand) continue;
      n4 += __builtin_popcountll(cand);
#if defined(__x86_64__)
      const unsigned long long comp = have_bmi2_ ? compress_bits_bmi2(cand, rb) : compress_bits_generic(cand, rb);
#else
      const unsigned long long comp = compress_bits_generic(cand, rb);
#endif
      const int pos = wstart_[w], word = pos >> 6, off = pos & 63;
      mask4_[word] |= comp << off;
      if (off) mask4_[word + 1] |= comp >> (64 - off);
    }
    n4_ = n4;
    return n4 > 0;
  }

  // MatchBase::SelectRandomTriangle, matchBase.hpp:111-212, with key membership read from the bit matrix (literal form)
  bool SelectRandomTriangle(int& base1, int& base2, int& base3, std::vector<int>& sample_pool) {
    base1 = base2 = base3 = -1;
    ++n_tri_calls;
    double tp = now();
    std::discrete_distribution<> sampler(point_probs_.begin(), point_probs_.end());
    const int first_point = sampler(point_index_engine_);
    point_probs_[first_point] *= opt.dispersion;
    t_first += now() - tp, tp = now();
    sample_pool.clear();
    std::vector<float> probs;
    const unsigned long long* row = M + (size_t)first_point * W;
    for (int w = 0; w < W; ++w) {
      unsigned long long bits = row[w];
      while (bits) {
        const int i = w * 64 + __builtin_ctzll(bits);
        bits &= bits - 1;
        if (i == first_point || i >= n) continue;
        sample_pool.push_back(i);
        probs.push_back(point_probs_[i]);
      }
    }
    t_pool += now() - tp, tp = now();
    sum_pool += (long long)sample_pool.size();
    if (sample_pool.size() < 3) return false;
    const float sq_max = max_base_diameter_ * max_base_diameter_;
    const V3 p0 = ppos(first_point);
    for (int i = 0; (size_t)i < sample_pool.size() * sample_pool.size() / 4; ++i) {
      ++n_pair_iters;
      std::discrete_distribution<> sampler1(probs.begin(), probs.end());
      const int second = sampler1(point_index_engine_);
      const int third = sampler1(point_index_engine_);
      if (second == third) { ++n_same; continue; }
      if (!bit(sample_pool[second], sample_pool[third])) { ++n_nobit; continue; }
      ++n_geom;
      probs[second] *= opt.dispersion;
      probs[third] *= opt.dispersion;
      const V3 u = ppos(sample_pool[second]) - p0;
      const V3 w = ppos(sample_pool[third]) - p0;
      const float how_wide = vdot(vnormalized(u), vnormalized(w));
      if ((double)std::fabs(how_wide) <= std::cos(45 * M_PI / 180.0) && vsqn(u) < sq_max && vsqn(w) < sq_max) {
        base1 = first_point;
        base2 = sample_pool[second];
        base3 = sample_pool[third];
        break;
      }
    }
    t_pairdraw += now() - tp, tp = now();
    if (base2 == -1 || base3 == -1) return false;
    // pool for the 4th point; the reference stores the LOOP INDEX here (matchBase.hpp:203) and later
    // uses it as a point index (match4pcsBase.hpp:159) -- mirrored.
    std::vector<int> backup;
    backup.swap(sample_pool);
    for (int i = 0; i < (int)backup.size(); ++i) {
      const int id = backup[i];
      if (id == base2 || id == base3 || id == base1) continue;
      if (bit(base2, id) && bit(base3, id)) sample_pool.push_back(i);
    }
    t_pool4 += now() - tp;
    if (sample_pool.empty()) return false;
    return base1 != -1 && base2 != -1 && base3 != -1;
  }

  // Match4pcsBase::distSegmentToSegment, match4pcsBase.hpp:283-354
  static float distSegmentToSegment(V3 p1, V3 p2, V3 q1, V3 q2, float& invariant1, float& invariant2) {
    const float kSmall = 0.0001f;
    const V3 u = p2 - p1, v = q2 - q1, w = p1 - q1;
    const float a = vdot(u, u), b = vdot(u, v), cc = vdot(v, v), d = vdot(u, w), e = vdot(v, w);
    const float f = a * cc - b * b;
    float s1 = 0.0f, s2 = f, t1 = 0.0f, t2 = f;
    if (f < kSmall) {
      s1 = 0.0f, s2 = 1.0f, t1 = e, t2 = cc;
    } else {
      s1 = (b * e - cc * d);
      t1 = (a * e - b * d);
      if (s1 < 0.0f) s1 = 0.0f, t1 = e, t2 = cc;
      else if (s1 > s2) s1 = s2, t1 = e + b, t2 = cc;
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
    invariant1 = (std::fabs(s1) < kSmall ? 0.0f : s1 / s2);
    invariant2 = (std::fabs(t1) < kSmall ? 0.0f : t1 / t2);
    return vnorm((w + (invariant1 * u)) - (invariant2 * v));
  }

  // Match4pcsBase::TryQuadrilateral, match4pcsBase.hpp:50-101
  bool TryQuadrilateral(float& invariant1, float& invariant2, int ids[4]) {
    float min_distance = FLT_MAX;
    int best[4] = {-1, -1, -1, -1};
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        if (i == j) continue;
        int k = 0;
        while (k == i || k == j) k++;
        int l = 0;
        while (l == i || l == j || l == k) l++;
        float li1, li2;
        const float sd = distSegmentToSegment(bpos[i], bpos[j], bpos[k], bpos[l], li1, li2);
        if (sd < min_distance) {
          min_distance = sd;
          best[0] = i, best[1] = j, best[2] = k, best[3] = l;
          invariant1 = li1, invariant2 = li2;
        }
      }
    if (best[0] < 0) return false;
    const std::array<V3, 4> tp = bpos, tn = bnrm;
    const int tid[4] = {ids[0], ids[1], ids[2], ids[3]};
    for (int k = 0; k < 4; ++k) bpos[k] = tp[best[k]], bnrm[k] = tn[best[k]], ids[k] = tid[best[k]];
    return true;
  }

  // Match4pcsBase::SelectQuadrilateral, match4pcsBase.hpp:107-189
  bool SelectQuadrilateral(float& invariant1, float& invariant2, int ids[4]) {
    const float kBaseTooSmall = 0.2f;
    int current_trial = 0;
    std::vector<int> sample_pool;
    while (current_trial < 1000) {
      current_trial++;
      int base1, base2, base3, base4;
      if (!(use_fast ? SelectRandomTriangleFast(base1, base2, base3) : SelectRandomTriangle(base1, base2, base3, sample_pool))) continue;
      const V3 b0 = ppos(base1), b1 = ppos(base2), b2 = ppos(base3);
      const double x1 = b0.x, y1 = b0.y, z1 = b0.z, x2 = b1.x, y2 = b1.y, z2 = b1.z, x3 = b2.x, y3 = b2.y, z3 = b2.z;
      const float denom = (float)(-x3 * y2 * z1 + x2 * y3 * z1 + x3 * y1 * z2 - x1 * y3 * z2 - x2 * y1 * z3 + x1 * y2 * z3);
      if (denom != 0) {
        const float A = (float)((-y2 * z1 + y3 * z1 + y1 * z2 - y3 * z2 - y1 * z3 + y2 * z3) / denom);
        const float B = (float)((x2 * z1 - x3 * z1 - x1 * z2 + x3 * z2 + x1 * z3 - x2 * z3) / denom);
        const float C = (float)((-x2 * y1 + x3 * y1 + x1 * y2 - x3 * y2 - x1 * y3 + x2 * y3) / denom);
        base4 = -1;
        const double tf = now();
        const float too_small = (float)std::pow((double)(max_base_diameter_ * kBaseTooSmall), 2);
        if (use_fast) {
          const float bb[3][3] = {{b0.x, b0.y, b0.z}, {b1.x, b1.y, b1.z}, {b2.x, b2.y, b2.z}};
          base4 = fourth_point_mask(c->gp_h.x.data(), c->gp_h.y.data(), c->gp_h.z.data(), mask4_.data(), npool_, bb, too_small, A, B, C,
                                    dist4_);
        } else {
          float best_distance = FLT_MAX;
          for (size_t i = 0; i < sample_pool.size(); ++i) {
            const V3 p = ppos(sample_pool[i]);
            if (vsqn(p - b0) >= too_small && vsqn(p - b1) >= too_small && vsqn(p - b2) >= too_small) {
              const float distance = (float)std::fabs((double)((A * p.x + B * p.y) + C * p.z) - 1.0);
              if (distance < best_distance) {
                best_distance = distance;
                base4 = sample_pool[i];
              }
            }
          }
        }
        t_fourth += now() - tf;
        if (base4 != -1) {
          ids[0] = base1, ids[1] = base2, ids[2] = base3, ids[3] = base4;
          for (int k = 0; k < 4; ++k) bpos[k] = ppos(ids[k]), bnrm[k] = pnrm(ids[k]);
          if (TryQuadrilateral(invariant1, invariant2, ids)) return true;
        }
      }
    }
    return false;
  }

  // per-base constants of the device side of generateCongruents (match4pcsBase.hpp:244-261,
  // FunctorSuper4pcs.h:163-170, normalset.hpp:205-212)
  template <class BaseT>
  void fill_base(BaseT& B, float inv1, float inv2) const {
    for (int k = 0; k < 4; ++k) B.bpos[k][0] = bpos[k].x, B.bpos[k][1] = bpos[k].y, B.bpos[k][2] = bpos[k].z;
    B.dist1 = vnorm(bpos[0] - bpos[1]);
    B.dist2 = vnorm(bpos[2] - bpos[3]);
    B.inv1 = inv1, B.inv2 = inv2;
    B.e1 = base_edge_features(bpos[0], bnrm[0], bpos[1], bnrm[1]);
    B.e2 = base_edge_features(bpos[2], bnrm[2], bpos[3], bnrm[3]);
    const float alpha = vdot(vnormalized(bpos[1] - bpos[0]), vnormalized(bpos[3] - bpos[2]));
    const float ac = acosf_fdlibm(alpha);
    const float perimeter = (float)((double)2.f * M_PI * (double)std::atan(ac));
    unsigned nb = (unsigned)(2 * std::ceil(perimeter * 7.f / 2.f));
    if (!(nb <= 64u)) nb = alpha == alpha ? 64u : 0u;  // NaN alpha -> no samples
    const float angleStep = (float)((double)2.f * M_PI / (double)(float)nb);
    const float sinAlpha = std::sin(ac);
    B.nb_sample = (int)nb;
    for (unsigned a = 0; a < nb; ++a) {
      const float theta = float(a) * angleStep;
      B.ring[a][0] = sinAlpha * std::cos(theta), B.ring[a][1] = sinAlpha * std::sin(theta), B.ring[a][2] = alpha;
    }
  }
};


}  // namespace hop
#endif
