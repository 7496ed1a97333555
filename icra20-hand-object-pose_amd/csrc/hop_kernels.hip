// hop_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4, wave64).
//
// Layout rules used throughout
//   * clouds live in HBM as SoA planes (x[], y[], z[], nx[], ny[], nz[]): a wave reads 64 consecutive
//     floats per plane = one 256-B transaction.
//   * nearest-neighbour scans keep the QUERY points in registers (R per lane) and stream the TARGET
//     points through an LDS tile as float4{x,y,z,_}; every lane reads the same LDS address per step, which
//     the LDS serves as a broadcast (no bank conflict), and one ds_read_b128 feeds R*9 VALU ops.
//   * squared distances use the exact operation order of the CPU code they replace (Eigen's
//     dx^2+(dy^2+dz^2) for the generator's Verify, FLANN's (dx^2+dy^2)+dz^2 for PCL searches), compiled
//     with -ffp-contract=off, so inlier counts and nearest indices are bit-reproducible.
//   * per-hypothesis integer results are reduced with wave ballots + one atomic per (wave, hypothesis).
#include "../../include/hop.h"
#include "hop_device.h"

// Packed f32 arithmetic, kernel by kernel.  This unit is compiled with the target feature packed-fp32-ops OFF (csrc/Makefile: -Xclang -target-feature
// -Xclang -packed-fp32-ops) and every kernel below switches it back ON for itself (HOP_PK_F32) EXCEPT the two lookup kernels of the shipped
// configuration, k_icp_fusedq_momm and k_lcp_cells_fast.  Why: the compiler pairs independent float operations into v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 -- its cost model has them at twice the scalar rate -- and pays v_mov_b32s to line the operands up in register pairs; on this part a
// packed f32 operation issues in 5.2 cycles per wavefront against 2.5-2.7 for each of the two scalar ones (profiles/r02_valu_issue_rates.txt, measured
// on the MI355X), so a pair buys nothing and every move costs 2.3 cycles.  Why this way round: a kernel that switches the feature OFF for itself can no
// longer inline the HIP header's __syncthreads / __ballot / __shfl_xor (a callee with more target features than its caller is not inlined: tried); a
// kernel that switches it ON can.  With the attribute a kernel compiles to exactly the assembly it had -- the kernels that have run on hardware stay
// what ran (checked body by body) -- and the arithmetic is the same IEEE operations either way.
#if defined(__HIP_DEVICE_COMPILE__)
#define HOP_PK_F32 __attribute__((target("packed-fp32-ops")))
#else
#define HOP_PK_F32
#endif

namespace hop {

// ------------------------------------------------------------------------------------------------
// brute-force nearest neighbour core
// ------------------------------------------------------------------------------------------------
#define HOP_FAR 1.0e15f  // padding coordinate: d^2 ~ 3e30, finite, never the minimum of a real cloud

template <bool FLANN>
__device__ __forceinline__ float sqdist(const V3& q, const float4& t) {
  const float dx = q.x - t.x, dy = q.y - t.y, dz = q.z - t.z;
  if (FLANN) return (dx * dx + dy * dy) + dz * dz;
  return dx * dx + (dy * dy + dz * dz);
}

// Scans `tn` targets (multiple of NN_CH) of an LDS tile for R register queries.  Keeps, per query, the
// smallest squared distance and the index of the FIRST target that attains it (the tie rule of a linear
// scan).  The per-pair work is sub/mul/add + one v_min; the index is resolved once per tile from the
// winning NN_CH-chunk, so the hot loop carries no compare/select.
template <int R, bool FLANN, bool WANT_INDEX>
__device__ __forceinline__ void nn_scan_tile(const float4* __restrict__ tile, int tn, int tile_base, const V3 (&q)[R],
                                             float (&best)[R], int (&bidx)[R]) {
  int chunk_of[R];
#pragma unroll
  for (int r = 0; r < R; ++r) chunk_of[r] = -1;
  for (int c = 0; c < tn; c += NN_CH) {
    float cm[R];
#pragma unroll
    for (int r = 0; r < R; ++r) cm[r] = 3.0e38f;
#pragma unroll
    for (int k = 0; k < NN_CH; ++k) {
      const float4 t = tile[c + k];
#pragma unroll
      for (int r = 0; r < R; ++r) cm[r] = fminf(cm[r], sqdist<FLANN>(q[r], t));
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (cm[r] < best[r]) {
        best[r] = cm[r];
        chunk_of[r] = c;
      }
  }
  if (WANT_INDEX) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (chunk_of[r] >= 0) {
        int found = chunk_of[r];
        for (int k = NN_CH - 1; k >= 0; --k)  // last write wins -> lowest k with equality
          if (sqdist<FLANN>(q[r], tile[chunk_of[r] + k]) == best[r]) found = chunk_of[r] + k;
        bidx[r] = tile_base + found;
      }
    }
  }
}

// cooperative staging of a tile of raw SoA points into LDS (coalesced plane reads -> float4 writes)
__device__ __forceinline__ void stage_tile_raw(float4* tile, const float* __restrict__ X, const float* __restrict__ Y,
                                               const float* __restrict__ Z, int start, int n_total, int tn) {
  for (int t = threadIdx.x; t < tn; t += blockDim.x) {
    const int j = start + t;
    float4 v;
    if (j < n_total) v = make_float4(X[j], Y[j], Z[j], 0.f);
    else v = make_float4(HOP_FAR, HOP_FAR, HOP_FAR, 0.f);
    tile[t] = v;
  }
}
// same, with a rigid transform applied on the way in (batched SE(3) transform fused into the scan)
__device__ __forceinline__ void stage_tile_tf(float4* tile, const float* __restrict__ X, const float* __restrict__ Y,
                                              const float* __restrict__ Z, int start, int n_total, int tn,
                                              const float* T) {
  for (int t = threadIdx.x; t < tn; t += blockDim.x) {
    const int j = start + t;
    float4 v;
    if (j < n_total) {
      const V3 p = m4_point(T, v3(X[j], Y[j], Z[j]));
      v = make_float4(p.x, p.y, p.z, 0.f);
    } else
      v = make_float4(HOP_FAR, HOP_FAR, HOP_FAR, 0.f);
    tile[t] = v;
  }
}

__device__ __forceinline__ int round_up(int a, int b) { return (a + b - 1) / b * b; }

// one atomic per (wave, key) for lanes that vote `flag` on integer `key`
__device__ __forceinline__ void wave_count_by_key(bool flag, int key, int* __restrict__ counts) {
  unsigned long long pending = __ballot(flag);
  while (pending) {
    const int leader = __ffsll((long long)pending) - 1;
    const int k = __shfl(key, leader);
    const unsigned long long same = __ballot(flag && key == k);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&counts[k], __popcll(same));
    pending &= ~same;
  }
}

// ------------------------------------------------------------------------------------------------
// K2: PPF key membership matrix.  bit (i,j) = key(P[i] -> P[j]) is in the table
// (gr::computePPF(sampled_P_3D_[i], sampled_P_3D_[j]) + _ppfs.find, matchBase.hpp:131-134,158-159,196-201).
// One wave owns 64 consecutive columns j (coalesced plane reads, kept in registers) and walks a tile of
// rows i whose data sits in LDS (broadcast reads); __ballot packs the 64 answers of a row into one word.
// ------------------------------------------------------------------------------------------------
template <bool THR>
__global__ HOP_PK_F32 __launch_bounds__(256) void k_ppf_matrix(PpfMatrixArgs a) {
  __shared__ float rows[PPF_ROWS][8];
  __shared__ float sthr[32];
  if (THR && threadIdx.x < 32) sthr[threadIdx.x] = a.angle_thr[threadIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jb = blockIdx.x * 4 + wave;  // 64-column block
  const int j = jb * 64 + lane;
  const int i0 = blockIdx.y * PPF_ROWS;
  for (int t = threadIdx.x; t < PPF_ROWS; t += blockDim.x) {
    const int i = i0 + t;
    if (i < a.n) {
      rows[t][0] = a.x[i], rows[t][1] = a.y[i], rows[t][2] = a.z[i];
      rows[t][3] = a.nx[i], rows[t][4] = a.ny[i], rows[t][5] = a.nz[i];
    }
  }
  __syncthreads();
  if (jb >= a.words) return;
  V3 pj = v3(0, 0, 0), nj = v3(0, 0, 1);
  const bool jvalid = j < a.n;
  if (jvalid) {
    pj = v3(a.x[j], a.y[j], a.z[j]);
    nj = v3(a.nx[j], a.ny[j], a.nz[j]);
  }
  const int rmax = min(PPF_ROWS, a.n - i0);
  for (int r = 0; r < rmax; ++r) {
    const V3 pi = v3(rows[r][0], rows[r][1], rows[r][2]);
    const V3 ni = v3(rows[r][3], rows[r][4], rows[r][5]);
    bool member = false;
    int key[4];
    if (jvalid && (i0 + r) != j && (THR ? ppf_key_thr(pi, ni, pj, nj, sthr, key) : ppf_key(pi, ni, pj, nj, key))) {
      // direct-address bitmap: dist bin k0/5 in [0,dmax], angle bins k/10 in [0,18]
      const int d = key[0] / 5, a1 = key[1] / 10, a2 = key[2] / 10, a3 = key[3] / 10;
      if (key[0] >= 0 && d < a.dist_bins && (unsigned)a1 < 19u && (unsigned)a2 < 19u && (unsigned)a3 < 19u) {
        const unsigned bit = ((unsigned)(d * 19 + a1) * 19u + (unsigned)a2) * 19u + (unsigned)a3;
        member = (a.bitmap[bit >> 5] >> (bit & 31)) & 1u;
      }
    }
    const unsigned long long word = __ballot(member);
    if (lane == 0) a.out[(size_t)(i0 + r) * a.words + jb] = word;
  }
}

// The same matrix from half the pair evaluations (threshold-bin path only).  key(j -> i) shares with key(i -> j) the distance
// bin, the unit direction up to its sign and the three dot products: (pi - pj) = -(pj - pi), a quotient and a sum of products
// change sign exactly with their operands, and ni . nj commutes -- so the reverse key is (k0, bin(-(nj . d)), bin(-(ni . d)), k3)
// from values already in registers, bit for bit what the direct evaluation gives.  A wave owns the 64 columns jb and walks the 64
// rows of word ib <= jb: the forward bits of a row go out as one ballot word M[i][jb], the reverse bits accumulate per lane
// into M[j][ib]; diagonal blocks (ib == jb) hold both orders already.
__global__ HOP_PK_F32 __launch_bounds__(256) void k_ppf_matrix_sym(PpfMatrixArgs a) {
  __shared__ float rows[PPF_ROWS][8];
  __shared__ float sthr[32];
  static_assert(PPF_ROWS == 64, "one matrix word of rows per block");
  const int ib = blockIdx.y;
  if ((int)blockIdx.x * 4 + 3 < ib) return;  // the whole block lies below the diagonal
  if (threadIdx.x < 32) sthr[threadIdx.x] = a.angle_thr[threadIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jb = blockIdx.x * 4 + wave;
  const int j = jb * 64 + lane;
  const int i0 = ib * PPF_ROWS;
  for (int t = threadIdx.x; t < PPF_ROWS; t += blockDim.x) {
    const int i = i0 + t;
    if (i < a.n) {
      rows[t][0] = a.x[i], rows[t][1] = a.y[i], rows[t][2] = a.z[i];
      rows[t][3] = a.nx[i], rows[t][4] = a.ny[i], rows[t][5] = a.nz[i];
    }
  }
  __syncthreads();
  if (jb >= a.words || jb < ib) return;
  const bool both = jb != ib;
  V3 pj = v3(0, 0, 0), nj = v3(0, 0, 1);
  const bool jvalid = j < a.n;
  if (jvalid) {
    pj = v3(a.x[j], a.y[j], a.z[j]);
    nj = v3(a.nx[j], a.ny[j], a.nz[j]);
  }
  auto member = [&](int d, int a1, int a2, int a3) -> bool {
    // direct-address bitmap: dist bin in [0, dist_bins), angle bins in [0, 18]
    const unsigned bit = ((unsigned)(d * 19 + a1) * 19u + (unsigned)a2) * 19u + (unsigned)a3;
    return (a.bitmap[bit >> 5] >> (bit & 31)) & 1u;
  };
  unsigned long long rev = 0ull;
  const int rmax = min(PPF_ROWS, a.n - i0);
  for (int r = 0; r < rmax; ++r) {
    const V3 pi = v3(rows[r][0], rows[r][1], rows[r][2]);
    const V3 ni = v3(rows[r][3], rows[r][4], rows[r][5]);
    bool fwd = false, bwd = false;
    if (jvalid && (i0 + r) != j) {
      const float nrm = vnorm(pi - pj) * 1000.f;  // (ppf_key_thr: p1 = the row's point, p2 = the column's)
      if (nrm < 2147483648.0f) {
        const int k0 = ppf_closest_bin((int)nrm, 5);
        const V3 dir = vnormalized(pj - pi);
        const float s1 = vdot(ni, dir), s2 = vdot(nj, dir), s3 = vdot(ni, nj);
        int b1, b2, b3;
        if (ppf_angle_bin_thr(s1, sthr, &b1) && ppf_angle_bin_thr(s2, sthr, &b2) && ppf_angle_bin_thr(s3, sthr, &b3)) {
          const int d = k0 / 5, a3 = b3 / 10;
          if (k0 >= 0 && d < a.dist_bins) {
            fwd = member(d, b1 / 10, b2 / 10, a3);
            if (both) {
              int c1, c2;
              (void)ppf_angle_bin_thr(-s2, sthr, &c1), (void)ppf_angle_bin_thr(-s1, sthr, &c2);  // (|s| <= 1 holds for both already)
              bwd = member(d, c1 / 10, c2 / 10, a3);
            }
          }
        }
      }
    }
    const unsigned long long word = __ballot(fwd);
    if (lane == 0) a.out[(size_t)(i0 + r) * a.words + jb] = word;
    rev |= (unsigned long long)bwd << r;
  }
  if (both && jvalid) a.out[(size_t)j * a.words + ib] = rev;
}

// ------------------------------------------------------------------------------------------------
// K3a: pair extraction for a batch of bases (FunctorSuper4PCS::ExtractPairs + PairCreationFunctor::process
// + AdaptivePointFilter, FunctorSuper4pcs.h:79-116, pairCreationFunctor.h:189-214, PointPairFilter.h:88-172)
// as the brute force the reference's octree accelerator is specified to equal.  One pass serves both
// base edges: a pair (i>j) is tested against edge (0,1) and edge (2,3).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_append_pair(bool flag, unsigned a, unsigned b, unsigned* list, int* counter, int cap,
                                                 int* overflow) {
  const unsigned long long m = __ballot(flag);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == __ffsll((long long)m) - 1) base = atomicAdd(counter, 2 * __popcll(m));
  base = __shfl(base, __ffsll((long long)m) - 1);
  if (flag) {
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    const int pos = base + 2 * rank;
    if (pos + 1 < cap) {
      list[pos] = (a << 16) | b;      // (i,j)
      list[pos + 1] = (b << 16) | a;  // (j,i)
    } else
      *overflow = 1;
  }
}

__global__ HOP_PK_F32 __launch_bounds__(256) void k_pairs(PairArgs a) {
  const int b = blockIdx.y;
  const BaseDev& B = a.bases[b];
  const int n = a.nq;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = (int)(t / n), j = (int)(t % n);
  bool ok1 = false, ok2 = false;
  if (i < n && j < i) {
    const V3 p = v3(a.qx[j], a.qy[j], a.qz[j]), np = v3(a.qnx[j], a.qny[j], a.qnz[j]);
    const V3 q = v3(a.qx[i], a.qy[i], a.qz[i]), nq = v3(a.qnx[i], a.qny[i], a.qnz[i]);
    const float distance = vnorm(q - p);
    const double dd = (double)distance;
    if (!(fabs(dd - (double)B.dist1) > (double)a.eps)) ok1 = pair_ppf_is_good(p, np, q, nq, B.e1);
    if (!(fabs(dd - (double)B.dist2) > (double)a.eps)) ok2 = pair_ppf_is_good(p, np, q, nq, B.e2);
  }
  unsigned* l1 = a.pairs1 + (size_t)b * a.cap;
  unsigned* l2 = a.pairs2 + (size_t)b * a.cap;
  wave_append_pair(ok1, (unsigned)i, (unsigned)j, l1, &a.cnt1[b], a.cap, a.overflow);
  wave_append_pair(ok2, (unsigned)i, (unsigned)j, l2, &a.cnt2[b], a.cap, a.overflow);
}

// ------------------------------------------------------------------------------------------------
// K3b: congruent quadrilaterals (FunctorSuper4PCS::FindCongruentQuadrilaterals, FunctorSuper4pcs.h:131-293)
// with the IndexedNormalSet lookups (normalset.hpp:111-253) expressed as a predicate on
// (first pair, second pair): the query cell lies in the element's 1-ring, the element's normal bin is one
// of the bins painted by the query cone, and the world-space invariant points are close.
//   prep: per first pair  -> cell, normal bin, world invariant point
//         per second pair -> cell, 343-bit painted-bin mask, world query point
//   main: the n1 x n2 predicate matrix, fused with the 3-point rigid fit (ComputeRigidTransformation) and
//         the rms<delta gate of TryCongruentSet (cse.hpp:273-291); survivors become Verify candidates.
// ------------------------------------------------------------------------------------------------
__global__ HOP_PK_F32 __launch_bounds__(256) void k_quad_prep(QuadPrepArgs a) {
  const int b = blockIdx.y;
  const BaseDev& B = a.bases[b];
  const int n1 = min(a.cnt1[b], a.cap), n2 = min(a.cnt2[b], a.cap);
  const NsetGeom g = a.geom;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n1; t += gridDim.x * blockDim.x) {
    {
      const unsigned pr = a.pairs1[(size_t)b * a.cap + t];
      const int i0 = pr >> 16, i1 = pr & 0xffff;
      const V3 u1 = v3(a.ux[i0], a.uy[i0], a.uz[i0]), u2 = v3(a.ux[i1], a.uy[i1], a.uz[i1]);
      const V3 nrm = vnormalized(u2 - u1);
      const V3 pos = u1 + B.inv1 * (u2 - u1);
      QuadElem e;
      e.cx = (short)(int)(pos.x / g.epsilon), e.cy = (short)(int)(pos.y / g.epsilon), e.cz = (short)(int)(pos.z / g.epsilon);
      e.nid = (short)nset_index_normal(g, nrm);
      const V3 w1 = v3(a.qx[i0], a.qy[i0], a.qz[i0]), w2 = v3(a.qx[i1], a.qy[i1], a.qz[i1]);
      const V3 ip = w1 + (w2 - w1) * B.inv1;
      e.px = ip.x, e.py = ip.y, e.pz = ip.z;
      a.elems[(size_t)b * a.cap + t] = e;
    }
  }
  // second pairs: one wave per query, one lane per ring sample (nb_sample <= 64): the 64 quaternion rotations and
  // normal-bin look-ups of a query run side by side instead of as a 64-step loop in one lane
  __shared__ unsigned smask[4][12];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int s = blockIdx.x * 4 + wave; s < n2; s += gridDim.x * 4) {
    const unsigned pr = a.pairs2[(size_t)b * a.cap + s];
    const int i0 = pr >> 16, i1 = pr & 0xffff;
    const V3 u1 = v3(a.ux[i0], a.uy[i0], a.uz[i0]), u2 = v3(a.ux[i1], a.uy[i1], a.uz[i1]);
    const V3 query = u1 + B.inv2 * (u2 - u1);
    const V3 queryn = vnormalized(u2 - u1);
    // Quaternion::setFromTwoVectors((0,0,1), queryn) then q * ringvec (Geometry/Quaternion.h:578-612,471-481)
    const V3 v1 = vnormalized(queryn);
    float c = v1.x * 0.f + (v1.y * 0.f + v1.z * 1.f);  // v1.dot((0,0,1))
    V3 qv;
    float qw;
    if (c < -1.f + 1e-5f) {
      // Eigen falls back to a JacobiSVD null vector here; analytic stand-in (DESIGN.md "known deviations")
      c = fmaxf(c, -1.f);
      V3 axis = vnormalized(v3(-v1.y, v1.x, 0.f));
      if (vsqn(axis) == 0.f) axis = v3(1.f, 0.f, 0.f);
      const float w2q = (1.f + c) * 0.5f;
      qw = sqrtf(w2q);
      qv = axis * sqrtf(1.f - w2q);
    } else {
      const V3 axis = vcross(v3(0.f, 0.f, 1.f), v1);
      const float s2 = sqrtf((1.f + c) * 2.f);
      const float invs = 1.f / s2;
      qv = axis * invs;
      qw = s2 * 0.5f;
    }
    if (lane < 11) smask[wave][lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    if (lane < B.nb_sample) {
      const V3 rv = v3(B.ring[lane][0], B.ring[lane][1], B.ring[lane][2]);
      V3 uv = vcross(qv, rv);
      uv = uv + uv;
      const V3 rot = (rv + qw * uv) + vcross(qv, uv);
      const int id = nset_index_normal(g, vnormalized(rot));
      if (id >= 0 && id < 343) atomicOr(&smask[wave][id >> 5], 1u << (id & 31));
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      QuadQuery q;
      q.cx = (short)(int)(query.x / g.epsilon), q.cy = (short)(int)(query.y / g.epsilon), q.cz = (short)(int)(query.z / g.epsilon);
      q.pad = 0;
      const V3 w1 = v3(a.qx[i0], a.qy[i0], a.qz[i0]), w2 = v3(a.qx[i1], a.qy[i1], a.qz[i1]);
      const V3 qq = w1 + B.inv2 * (w2 - w1);
      q.px = qq.x, q.py = qq.y, q.pz = qq.z;
      for (int w = 0; w < 11; ++w) q.mask[w] = smask[wave][w];
      a.queries[(size_t)b * a.cap + s] = q;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// TryCongruentSet (cse.hpp:229-291) for one accepted (first pair, second pair): 3-point rigid fit, RMS gate, candidate
// record.  Called with all lanes of the wave converged (`have` marks the lanes that carry a pair): the slot in the
// candidate array comes from one wave-aggregated atomic.
__device__ __forceinline__ void quad_fit_emit(const QuadArgs& a, const BaseDev& B, int b, bool have, int id1, int id2) {
  bool emit = false;
  Candidate cd;
  if (have) {
    const unsigned p1 = a.pairs1[(size_t)b * a.cap + id1], p2 = a.pairs2[(size_t)b * a.cap + id2];
    const int qa = p1 >> 16, qb = p1 & 0xffff, qc = p2 >> 16;
    // only the first three points enter the fit
    V3 ref[3], cand[3];
    for (int k = 0; k < 3; ++k) ref[k] = v3(B.bpos[k][0], B.bpos[k][1], B.bpos[k][2]);
    cand[0] = v3(a.qx[qa], a.qy[qa], a.qz[qa]);
    cand[1] = v3(a.qx[qb], a.qy[qb], a.qz[qb]);
    cand[2] = v3(a.qx[qc], a.qy[qc], a.qz[qc]);
    const V3 c1 = ((ref[0] + ref[1]) + ref[2]) / 3.f;
    const V3 c2 = ((cand[0] + cand[1]) + cand[2]) / 3.f;
    float T[16], rms;
    const bool fit = rigid_3pt(ref, cand, c1, c2, T, &rms);
    if (fit && rms >= 0.f && rms < a.delta) {
      emit = true;
      for (int k = 0; k < 12; ++k) cd.T[k] = T[k];
      cd.c1[0] = c1.x, cd.c1[1] = c1.y, cd.c1[2] = c1.z;
      cd.c2[0] = c2.x, cd.c2[1] = c2.y, cd.c2[2] = c2.z;
      // canonical order key: base trial, then first pair in (i,j,flip) order, then second pair
      const unsigned long long k1 = ((unsigned long long)max(qa, qb) << 12 | (unsigned long long)min(qa, qb)) << 1 | (qa < qb ? 1u : 0u);
      const int qd = p2 & 0xffff;
      const unsigned long long k2 = ((unsigned long long)max(qc, qd) << 12 | (unsigned long long)min(qc, qd)) << 1 | (qc < qd ? 1u : 0u);
      cd.key = ((unsigned long long)(a.base_index0 + b) << 50) | (k1 << 25) | k2;
    }
  }
  const unsigned long long m = __ballot(emit);
  if (!m) return;
  const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(a.cand_count, __popcll(m));
  base = __shfl(base, leader);
  if (!emit) return;
  const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
  if (slot >= a.cand_cap) {
    *a.overflow = 1;
    return;
  }
  a.cands[slot] = cd;
  a.cand_counts[slot] = 0;
}

// Stage 1 of K3b: the cheap tests of FindCongruentQuadrilaterals on (first pair) x (second pair).  One second pair
// (query) per wave at a time -- wave-uniform, read once -- against all first pairs (elements), 64 per step, read
// coalesced.  The ~0.2 % that pass go into a device-wide queue (one wave-aggregated atomic per step that has any), so that
// the long rigid-fit code of stage 2 always runs with full lanes whatever the per-wave yield.
__global__ HOP_PK_F32 __launch_bounds__(256) void k_quads(QuadArgs a, int n1_above) {
  __shared__ int queue_s[4][2][128];  // per wave: accepted (id1, id2) on their way to the device-wide queue
  const int b = blockIdx.y;
  const int n1 = min(a.cnt1[b], a.cap), n2 = min(a.cnt2[b], a.cap);
  if (n1 <= n1_above) return;  // (bases k_quads_hash takes)
  const int eg = a.geom.eg_size;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int* queue1 = queue_s[wave][0];
  int* queue2 = queue_s[wave][1];
  int nq_wave = 0, qn = 0;
  // FIT_QUEUES sub-queues (by block column) keep the counters from becoming one hot address; a wave pays for one
  // returning atomic per 64 accepted pairs (plus one at its end), not per step
  const int qi = blockIdx.x & (FIT_QUEUES - 1);
  auto flush = [&](int from, int count) {
    int base = 0;
    if (lane == 0) base = atomicAdd(a.fit_count + qi, count);
    base = __shfl(base, 0);
    if (lane < count) {
      const int at = base + lane;
      if (at < a.fit_cap) a.fit_queue[(size_t)qi * a.fit_cap + at] = make_int4(b, queue1[from + lane], queue2[from + lane], 0);
      else *a.overflow = 1;
    }
  };
  for (int id2 = blockIdx.x * 4 + wave; id2 < n2; id2 += gridDim.x * 4) {
    const QuadQuery& q = a.queries[(size_t)b * a.cap + id2];
    const bool qok = q.cx >= 0 && q.cx < eg && q.cy >= 0 && q.cy < eg && q.cz >= 0 && q.cz < eg;
    if (!qok) continue;
    for (int id1 = lane; id1 < ((n1 + 63) & ~63); id1 += 64) {
      bool ok = id1 < n1;
      if (ok) {
        const QuadElem e = a.elems[(size_t)b * a.cap + id1];
        const int dx = e.cx - q.cx, dy = e.cy - q.cy, dz = e.cz - q.cz;
        ok = dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
        ok = ok && e.cx >= 0 && e.cx < eg && e.cy >= 0 && e.cy < eg && e.cz >= 0 && e.cz < eg;
        ok = ok && e.nid >= 0 && e.nid < 343 && ((q.mask[e.nid >> 5] >> (e.nid & 31)) & 1u);
        if (ok) {
          const V3 d = v3(q.px, q.py, q.pz) - v3(e.px, e.py, e.pz);
          ok = vsqn(d) <= a.dist_thr2;  // squared norm vs the UNSQUARED threshold (FunctorSuper4pcs.h:277)
        }
      }
      const unsigned long long m = __ballot(ok);
      if (!m) continue;
      if (ok) {
        const int at = qn + __popcll(m & ((1ull << lane) - 1ull));
        queue1[at] = id1, queue2[at] = id2;
      }
      qn += __popcll(m);
      nq_wave += __popcll(m);
      __builtin_amdgcn_wave_barrier();
      if (qn >= 64) {
        flush(qn - 64, 64);
        qn -= 64;
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (qn > 0) flush(0, qn);
  if (lane == 0 && nq_wave) atomicAdd(&a.nquads[b], nq_wave);
}

// The same stage the way the reference does it (IndexedNormalSet: first pairs binned by the cell of their invariant point, a second pair
// looks into the 27 cells around its own, FunctorSuper4pcs.h:131-293) instead of every second pair against every first pair: a block
// hashes the base's first pairs by cell into LDS (chained, open hashing on the three cell coordinates), then one second pair per lane
// walks the chains of its 27 neighbour cells.  ~30 x fewer instructions than k_quads at C2's ~600 x 600 pairs per base; bases with more
// than QH_MAX first pairs are left to k_quads.  The accepted pairs reach the fit queue in another order; quad_fit_emit's canonical key
// orders the candidates afterwards, as it already had to (the queue is filled through atomics).
constexpr int QH_MAX = 4096, QH_SLOTS = 4096;
__device__ __forceinline__ int qh_hash(int cx, int cy, int cz) {
  return (int)(((unsigned)cx * 73856093u) ^ ((unsigned)cy * 19349663u) ^ ((unsigned)cz * 83492791u)) & (QH_SLOTS - 1);
}
__global__ HOP_PK_F32 __launch_bounds__(256) void k_quads_hash(QuadArgs a) {
  __shared__ int head[QH_SLOTS];
  __shared__ unsigned short nxt[QH_MAX];
  __shared__ uint2 ekey[QH_MAX];
  __shared__ int queue_s[4][2][128];
  const int b = blockIdx.y;
  const int n1 = min(a.cnt1[b], a.cap), n2 = min(a.cnt2[b], a.cap);
  if (n1 > QH_MAX || n1 == 0 || n2 == 0) return;
  const int eg = a.geom.eg_size;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* __restrict__ eb = reinterpret_cast<const char*>(a.elems + (size_t)b * a.cap);
  static_assert(sizeof(QuadElem) == 20, "an element's first two words and its position are read by offset");
  for (int i = threadIdx.x; i < QH_SLOTS; i += 256) head[i] = -1;
  __syncthreads();
  for (int id1 = threadIdx.x; id1 < n1; id1 += 256) {
    const unsigned* w = reinterpret_cast<const unsigned*>(eb + (size_t)id1 * 20);
    const unsigned w0 = w[0], w1 = w[1];
    ekey[id1] = make_uint2(w0, w1);
    const int ecx = (short)(w0 & 0xffffu), ecy = (short)(w0 >> 16), ecz = (short)(w1 & 0xffffu), nid = (short)(w1 >> 16);
    int old = -1;
    const bool valid = ecx >= 0 && ecx < eg && ecy >= 0 && ecy < eg && ecz >= 0 && ecz < eg && nid >= 0 && nid < 343;
    if (valid) old = atomicExch(&head[qh_hash(ecx, ecy, ecz)], id1);
    nxt[id1] = (unsigned short)(old < 0 ? 0xFFFF : old);
  }
  __syncthreads();
  int* queue1 = queue_s[wave][0];
  int* queue2 = queue_s[wave][1];
  int nq_wave = 0, qn = 0;
  const int qi = (blockIdx.y * gridDim.x + blockIdx.x) & (FIT_QUEUES - 1);
  auto flush = [&](int from, int count) {
    int base = 0;
    if (lane == 0) base = atomicAdd(a.fit_count + qi, count);
    base = __shfl(base, 0);
    if (lane < count) {
      const int at = base + lane;
      if (at < a.fit_cap) a.fit_queue[(size_t)qi * a.fit_cap + at] = make_int4(b, queue1[from + lane], queue2[from + lane], 0);
      else *a.overflow = 1;
    }
  };
  const int tiles = (n2 + 63) >> 6;
  for (int tile = blockIdx.x * 4 + wave; tile < tiles; tile += gridDim.x * 4) {
    const int id2 = tile * 64 + lane;
    const QuadQuery* __restrict__ qq = a.queries + (size_t)b * a.cap + min(id2, n2 - 1);
    const int qcx = qq->cx, qcy = qq->cy, qcz = qq->cz;
    const bool qok = id2 < n2 && qcx >= 0 && qcx < eg && qcy >= 0 && qcy < eg && qcz >= 0 && qcz < eg;
    const V3 qp = v3(qq->px, qq->py, qq->pz);
    for (int nb = 0; nb < 27; ++nb) {
      const int ncx = qcx + nb % 3 - 1, ncy = qcy + (nb / 3) % 3 - 1, ncz = qcz + nb / 9 - 1;
      int cur = -1;
      if (qok && ncx >= 0 && ncx < eg && ncy >= 0 && ncy < eg && ncz >= 0 && ncz < eg) cur = head[qh_hash(ncx, ncy, ncz)];
      while (__ballot(cur >= 0)) {  // (wave-uniform: the lanes whose chain has ended idle)
        bool ok = false;
        const int id1 = cur;
        if (cur >= 0) {
          const uint2 k = ekey[cur];
          const int ecx = (short)(k.x & 0xffffu), ecy = (short)(k.x >> 16), ecz = (short)(k.y & 0xffffu), nid = (short)(k.y >> 16);
          const unsigned short nx = nxt[cur];
          cur = nx == 0xFFFF ? -1 : (int)nx;
          ok = ecx == ncx && ecy == ncy && ecz == ncz;  // (a chain holds every cell that hashes to its slot)
          if (ok) ok = (qq->mask[nid >> 5] >> (nid & 31)) & 1u;
          if (ok) {
            const float* pp = reinterpret_cast<const float*>(eb + (size_t)id1 * 20 + 8);
            const V3 d = qp - v3(pp[0], pp[1], pp[2]);
            ok = vsqn(d) <= a.dist_thr2;  // squared norm vs the UNSQUARED threshold (FunctorSuper4pcs.h:277)
          }
        }
        const unsigned long long m = __ballot(ok);
        if (!m) continue;
        if (ok) {
          const int at = qn + __popcll(m & ((1ull << lane) - 1ull));
          queue1[at] = id1, queue2[at] = id2;
        }
        qn += __popcll(m);
        nq_wave += __popcll(m);
        __builtin_amdgcn_wave_barrier();
        if (qn >= 64) {
          flush(qn - 64, 64);
          qn -= 64;
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  }
  if (qn > 0) flush(0, qn);
  if (lane == 0 && nq_wave) atomicAdd(&a.nquads[b], nq_wave);
}

// Stage 2: one queued quadrilateral per lane
__global__ HOP_PK_F32 __launch_bounds__(256) void k_quad_fit(QuadArgs a) {
  const int qi = blockIdx.y;
  const int n = min(a.fit_count[qi], a.fit_cap);
  const int stride = gridDim.x * blockDim.x;
  for (int i0 = blockIdx.x * blockDim.x; i0 < n; i0 += stride) {
    const int i = i0 + threadIdx.x;
    const bool have = i < n;
    const int4 e = have ? a.fit_queue[(size_t)qi * a.fit_cap + i] : make_int4(0, 0, 0, 0);
    quad_fit_emit(a, a.bases[e.x], e.x, have, e.y, e.z);
  }
}

// ------------------------------------------------------------------------------------------------
// K4: Verify (CongruentSetExplorationBase::Verify, cse.hpp:346-435), brute force.
// query g = (candidate c, sample s): T_c * Qs[s]; targets: all of P (centred), streamed through LDS.
// ------------------------------------------------------------------------------------------------
template <int R>
__global__ HOP_PK_F32 __launch_bounds__(256) void k_verify_brute(VerifyArgs a) {
  __shared__ float4 tile[NN_TILE];
  const int n_cand = a.n_cand_ptr ? min(*a.n_cand_ptr, a.cand_cap) : a.n_cand;
  const long long total = (long long)n_cand * a.nq;
  const int QB = 256 * R;
  const long long nchunks = (total + QB - 1) / QB;
  for (long long chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    V3 q[R];
    float best[R];
    int cidx[R], dummy[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long g = chunk * QB + (long long)r * 256 + threadIdx.x;
      best[r] = 3.0e38f;
      if (g < total) {
        const int c = (int)(g / a.nq), s = (int)(g % a.nq);
        cidx[r] = c;
        const float* T = a.T + (size_t)c * a.t_stride;
        q[r] = m4_point(T, v3(a.qx[s], a.qy[s], a.qz[s]));
      } else {
        cidx[r] = -1;
        q[r] = v3(-HOP_FAR, -HOP_FAR, -HOP_FAR);
      }
    }
    for (int start = 0; start < a.np; start += NN_TILE) {
      const int tn = min(NN_TILE, round_up(a.np - start, NN_CH));
      __syncthreads();
      stage_tile_raw(tile, a.px, a.py, a.pz, start, a.np, tn);
      __syncthreads();
      nn_scan_tile<R, false, false>(tile, tn, start, q, best, dummy);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) wave_count_by_key(cidx[r] >= 0 && best[r] <= a.sq_eps, cidx[r], a.counts);
  }
}
template __global__ void k_verify_brute<4>(VerifyArgs);
template __global__ void k_verify_brute<8>(VerifyArgs);

// Verify on a voxel grid over P (cell >= delta): a transformed sample can only have an inlier partner in
// the 27 cells around its own.  Same arithmetic for the distance test, hence identical counts.
__global__ HOP_PK_F32 __launch_bounds__(256) void k_verify_grid(VerifyArgs a, GridDev gd) {
  const int n_cand = a.n_cand_ptr ? min(*a.n_cand_ptr, a.cand_cap) : a.n_cand;
  const long long total = (long long)n_cand * a.nq;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ((total + 63) / 64) * 64;
       g += (long long)gridDim.x * blockDim.x) {
    bool hit = false;
    int c = -1;
    if (g < total) {
      c = (int)(g / a.nq);
      const int s = (int)(g % a.nq);
      const float* T = a.T + (size_t)c * a.t_stride;
      const V3 q = m4_point(T, v3(a.qx[s], a.qy[s], a.qz[s]));
      const float fx = (q.x - gd.ox) * gd.inv_cell, fy = (q.y - gd.oy) * gd.inv_cell, fz = (q.z - gd.oz) * gd.inv_cell;
      // cells outside [-1, dim] cannot have partners
      if (fx >= -1.f && fy >= -1.f && fz >= -1.f && fx < (float)(gd.dx + 1) && fy < (float)(gd.dy + 1) && fz < (float)(gd.dz + 1)) {
        const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
        for (int z = max(cz - 1, 0); z <= min(cz + 1, gd.dz - 1) && !hit; ++z)
          for (int y = max(cy - 1, 0); y <= min(cy + 1, gd.dy - 1) && !hit; ++y) {
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, gd.dx - 1);
            if (x0 > x1) continue;
            const int row = (z * gd.dy + y) * gd.dx;
            const int beg = gd.cell_start[row + x0], end = gd.cell_start[row + x1 + 1];  // x-adjacent cells are contiguous
            for (int k = beg; k < end; ++k) {
              const float4 t = gd.pts[k];
              const float dx = q.x - t.x, dy = q.y - t.y, dz = q.z - t.z;
              if (dx * dx + (dy * dy + dz * dz) <= a.sq_eps) {
                hit = true;
                break;
              }
            }
          }
      }
    }
    wave_count_by_key(hit, c, a.counts);
  }
}

// Verify on NN cell lists over P built in EXIST mode (verify_mode 2): one cell lookup, then a handful of candidates
// (one, when every query of the cell is known to hit).  Same distance expression and threshold as the other modes.
__global__ HOP_PK_F32 __launch_bounds__(256) void k_verify_cells(VerifyArgs a, CellListDev cl) {
  const int n_cand = a.n_cand_ptr ? min(*a.n_cand_ptr, a.cand_cap) : a.n_cand;
  const long long total = (long long)n_cand * a.nq;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ((total + 63) / 64) * 64;
       g += (long long)gridDim.x * blockDim.x) {
    bool hit = false;
    int c = -1;
    if (g < total) {
      c = (int)(g / a.nq);
      const int s = (int)(g % a.nq);
      const float* T = a.T + (size_t)c * a.t_stride;
      const V3 q = m4_point(T, v3(a.qx[s], a.qy[s], a.qz[s]));
      const float fx = (q.x - cl.ox) * cl.inv_cell, fy = (q.y - cl.oy) * cl.inv_cell, fz = (q.z - cl.oz) * cl.inv_cell;
      if (fx >= 0.f && fy >= 0.f && fz >= 0.f && fx < (float)cl.dx && fy < (float)cl.dy && fz < (float)cl.dz) {
        const int cidx = ((int)fz * cl.dy + (int)fy) * cl.dx + (int)fx;
        const int2 rg = cl.range[cidx];
        const int beg = rg.x, end = rg.y;
        for (int k = beg; k < end; ++k) {
          const float4 t = cl.pts[k];
          const float dx = q.x - t.x, dy = q.y - t.y, dz = q.z - t.z;
          if (dx * dx + (dy * dy + dz * dz) <= a.sq_eps) {
            hit = true;
            break;
          }
        }
      }
    }
    wave_count_by_key(hit, c, a.counts);
  }
}

// K3c: candidates with at least one inlier become hypotheses (cse.hpp:313-333): the translation is
// re-expressed for the un-centred clouds, t = c1 + centroid_P - R (c2 + centroid_Q).
__global__ HOP_PK_F32 __launch_bounds__(256) void k_emit(EmitArgs a) {
  const int n_cand = min(*a.cand_count, a.cand_cap);
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.cand_total, n_cand);
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n_cand; c += gridDim.x * blockDim.x) {
    const int cnt = a.cand_counts[c];
    if (cnt <= 0) continue;
    const int slot = atomicAdd(a.hyp_count, 1);
    if (slot >= a.hyp_cap) {
      *a.overflow = 1;
      continue;
    }
    const Candidate& cd = a.cands[c];
    const V3 u = v3(cd.c2[0], cd.c2[1], cd.c2[2]) + v3(a.cq[0], a.cq[1], a.cq[2]);
    const float* R = cd.T;
    const V3 ru = v3(R[0] * u.x + (R[1] * u.y + R[2] * u.z), R[4] * u.x + (R[5] * u.y + R[6] * u.z), R[8] * u.x + (R[9] * u.y + R[10] * u.z));
    const V3 t = (v3(cd.c1[0], cd.c1[1], cd.c1[2]) + v3(a.cp[0], a.cp[1], a.cp[2])) - ru;
    float* P = a.pose + (size_t)slot * 16;
    P[0] = R[0], P[1] = R[1], P[2] = R[2], P[3] = t.x;
    P[4] = R[4], P[5] = R[5], P[6] = R[6], P[7] = t.y;
    P[8] = R[8], P[9] = R[9], P[10] = R[10], P[11] = t.z;
    P[12] = 0.f, P[13] = 0.f, P[14] = 0.f, P[15] = 1.f;
    a.score[slot] = (float)((unsigned)cnt) / (float)a.nq;
    a.key[slot] = cd.key;
    a.inv_count[slot] = (unsigned)(a.nq - cnt);
  }
}

// gather of the sorted hypothesis set
__global__ HOP_PK_F32 void k_gather_hypos(const unsigned* __restrict__ perm, int n, const float* __restrict__ pose_in,
                               const float* __restrict__ score_in, float* __restrict__ pose_out, float* __restrict__ score_out,
                               int* __restrict__ id_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 16) return;
  const int h = t >> 4, k = t & 15;
  const unsigned src = perm[h];
  pose_out[t] = pose_in[(size_t)src * 16 + k];
  if (k == 0) {
    score_out[h] = score_in[src];
    id_out[h] = h;
  }
}
// row r of the top-k table (hop.h: score, id, pose[16]) from the r-th entry of the sorted order; rows beyond the set: score -FLT_MAX, id -1
__global__ HOP_PK_F32 void k_topk_pack(const unsigned* __restrict__ order, int H, int k, int id_offset, const float* __restrict__ pose, const float* __restrict__ score,
                            const int* __restrict__ ids, float* __restrict__ rows) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = t / HOP_TOPK_ROW_FLOATS, q = t % HOP_TOPK_ROW_FLOATS;
  if (r >= k) return;
  float v;
  if (r < H) {
    const unsigned h = order[r];
    v = q == 0 ? score[h] : q == 1 ? __int_as_float(ids[h] + id_offset) : pose[(size_t)h * 16 + (q - 2)];
  } else
    v = q == 0 ? -3.402823466e+38f : q == 1 ? __int_as_float(-1) : 0.f;
  rows[t] = v;
}
void launch_topk_pack(const unsigned* order, int H, int k, int id_offset, const float* pose, const float* score, const int* ids, float* rows, hipStream_t s) {
  const int n = k * HOP_TOPK_ROW_FLOATS;
  hipLaunchKernelGGL(k_topk_pack, dim3((n + 255) / 256), dim3(256), 0, s, order, H, k, id_offset, pose, score, ids, rows);
}
__global__ HOP_PK_F32 void k_iota(unsigned* p, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) p[t] = (unsigned)t;
}
__global__ HOP_PK_F32 void k_score_keys(const float* __restrict__ score, const int* __restrict__ ids, int n, unsigned long long* __restrict__ key) {
  // descending score, then ascending id, as one ascending 64-bit key (score_order_key: -0 = +0, NaN below every number -- the
  // same key the host's hop_topk_merge and the device merge of hop_comm.hip order by)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  key[t] = ((unsigned long long)(~score_order_key(score[t])) << 32) | (unsigned)ids[t];
}

// ------------------------------------------------------------------------------------------------
// K4': computeLCP (Utils::computeLCP, Utils.cpp:372-444; flags true, weights 1) for hypotheses
// [h0, h0+hb).  Pass 1 (k_lcp_reverse): nearest scene point of every transformed model point.
// Pass 2 (k_lcp_forward): nearest transformed model point of every scene point, both score terms.
// Pass 3 (k_lcp_sum): the reference adds the terms into one float in scene order; one lane per
// hypothesis repeats exactly that, so scores are bit-equal to the CPU.
// ------------------------------------------------------------------------------------------------
template <int R>
__global__ HOP_PK_F32 __launch_bounds__(256) void k_lcp_reverse(LcpArgs a) {
  __shared__ float4 tile[NN_TILE];
  const int h = a.h0 + blockIdx.y;
  const float* T = a.pose + (size_t)h * 16;
  V3 q[R];
  float best[R];
  int bidx[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int j = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
    best[r] = 3.0e38f;
    bidx[r] = -1;
    q[r] = j < a.nm ? m4_point(T, v3(a.mx[j], a.my[j], a.mz[j])) : v3(-HOP_FAR, -HOP_FAR, -HOP_FAR);
  }
  for (int start = 0; start < a.ns; start += NN_TILE) {
    const int tn = min(NN_TILE, round_up(a.ns - start, NN_CH));
    __syncthreads();
    stage_tile_raw(tile, a.sx, a.sy, a.sz, start, a.ns, tn);
    __syncthreads();
    nn_scan_tile<R, true, true>(tile, tn, start, q, best, bidx);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int j = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
    if (j < a.nm) {
      a.rev_idx[(size_t)blockIdx.y * a.nm + j] = bidx[r];
      a.rev_d2[(size_t)blockIdx.y * a.nm + j] = best[r];
    }
  }
}
template __global__ void k_lcp_reverse<4>(LcpArgs);

__device__ __forceinline__ float lcp_term(V3 n1, V3 n2, float d2, float dist_thres, float cos_thres) {
  n1 = vnormalized(n1);
  n2 = vnormalized(n2);
  const float d = vdot(n1, n2);
  if (d > cos_thres) return d * (1 - sqrtf(d2) / dist_thres) * 1.0f;
  return -1.f;  // "no contribution" marker (real terms are >= 0: d>cos>0 and d2<dist^2)
}

// the same term for normals that are already normalised (the reference normalises both at every use; a scene normal's
// normalised value does not depend on the hypothesis, so the cell-list path keeps it precomputed, and the posed model
// normal is normalised once for the forward and the reciprocal term)
__device__ __forceinline__ float lcp_term_unit(V3 u1, V3 u2, float d2, float dist_thres, float cos_thres) {
  const float d = vdot(u1, u2);
  if (d > cos_thres) return d * (1 - sqrtf(d2) / dist_thres) * 1.0f;
  return -1.f;
}
__global__ HOP_PK_F32 __launch_bounds__(256) void k_unit_normals(const float* __restrict__ nx, const float* __restrict__ ny, const float* __restrict__ nz, int n,
                                                     float* __restrict__ ux, float* __restrict__ uy, float* __restrict__ uz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 u = vnormalized(v3(nx[i], ny[i], nz[i]));
  ux[i] = u.x, uy[i] = u.y, uz[i] = u.z;
}
void launch_unit_normals(const float* nx, const float* ny, const float* nz, int n, float* ux, float* uy, float* uz, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_unit_normals, dim3((n + 255) / 256), dim3(256), 0, s, nx, ny, nz, n, ux, uy, uz);
}

template <int R>
__global__ HOP_PK_F32 __launch_bounds__(256) void k_lcp_forward(LcpArgs a) {
  __shared__ float4 tile[NN_TILE];
  const int h = a.h0 + blockIdx.y;
  const float* T = a.pose + (size_t)h * 16;
  V3 q[R];
  float best[R];
  int bidx[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
    best[r] = 3.0e38f;
    bidx[r] = -1;
    q[r] = i < a.ns ? v3(a.sx[i], a.sy[i], a.sz[i]) : v3(-HOP_FAR, -HOP_FAR, -HOP_FAR);
  }
  for (int start = 0; start < a.nm; start += NN_TILE) {
    const int tn = min(NN_TILE, round_up(a.nm - start, NN_CH));
    __syncthreads();
    stage_tile_tf(tile, a.mx, a.my, a.mz, start, a.nm, tn, T);
    __syncthreads();
    nn_scan_tile<R, true, true>(tile, tn, start, q, best, bidx);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
    if (i >= a.ns) continue;
    float f = -1.f, g = -1.f;
    if (bidx[r] >= 0 && best[r] < a.dist * a.dist) {
      const int j = bidx[r];
      const V3 ns = v3(a.snx[i], a.sny[i], a.snz[i]);
      const V3 nmod = m4_dir(T, v3(a.mnx[j], a.mny[j], a.mnz[j]));
      f = lcp_term(ns, nmod, best[r], a.dist, a.cos_thres);
      const int ri = a.rev_idx[(size_t)blockIdx.y * a.nm + j];
      if (ri >= 0) {
        const float rd2 = a.rev_d2[(size_t)blockIdx.y * a.nm + j];
        g = lcp_term(nmod, v3(a.snx[ri], a.sny[ri], a.snz[ri]), rd2, a.dist, a.cos_thres);
      }
    }
    a.terms[((size_t)blockIdx.y * a.ns + i) * 2 + 0] = f;
    a.terms[((size_t)blockIdx.y * a.ns + i) * 2 + 1] = g;
  }
}
template __global__ void k_lcp_forward<4>(LcpArgs);

__global__ HOP_PK_F32 __launch_bounds__(64) void k_lcp_sum(LcpArgs a, int hb) {
  const int hl = blockIdx.x * blockDim.x + threadIdx.x;
  if (hl >= hb) return;
  // (forward term, reciprocal term) of a scene point: 8-byte pairs -- a hypothesis' row starts at a multiple of 8 bytes whatever ns is (16-byte
  // loads would be misaligned for odd hypotheses when ns is odd: found by the UBSan build of the CPU model)
  const float2* t = reinterpret_cast<const float2*>(a.terms + (size_t)hl * a.ns * 2);
  float cp = 0.f;
  for (int k = 0; k < a.ns; ++k) {
    const float2 v = t[k];
    if (v.x >= 0.f) cp += v.x;
    if (v.y >= 0.f) cp += v.y;
  }
  a.score[a.h0 + hl] = cp;
}

// ------------------------------------------------------------------------------------------------
// Exact nearest neighbour on a voxel grid of the TARGET cloud in its own rest frame (nn_mode 1).
// Candidates come from grid cells; every candidate's squared distance is evaluated with the same float
// expression as the brute-force scan, and the result is the lexicographic minimum of (d^2, original index),
// i.e. exactly what a linear scan over the whole cloud returns -- provided the cells visited cover every
// point that could win.  Coverage argument: the cube of half-width k cells around the query's cell contains
// the ball of radius k*cell around the query; a query given in another frame (scene point mapped by the
// inverse pose) moves by less than GRID_MARGIN, so after ring k every unvisited point is farther than
// k*cell - GRID_MARGIN in the frame where distances are compared.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void nn_update(float d2, int j, float& best, int& bj) {
  if (d2 < best || (d2 == best && j < bj)) {
    best = d2;
    bj = j;
  }
}

template <bool TRANSFORM>
__device__ __forceinline__ void grid_scan_run(const GridDev& g, int row, int x0, int x1, const float* T, V3 q, float& best, int& bj) {
  const int beg = g.cell_start[row + x0], end = g.cell_start[row + x1 + 1];
  for (int k = beg; k < end; ++k) {
    const float4 t = g.pts[k];
    V3 p = v3(t.x, t.y, t.z);
    if (TRANSFORM) p = m4_point(T, p);
    nn_update(sqdist_flann(q, p), __float_as_int(t.w), best, bj);
  }
}

// all grid points within one ring (27 cells) of the cell of qg (= the query expressed in the grid's frame)
template <bool TRANSFORM>
__device__ __forceinline__ void grid_nn_ring1(const GridDev& g, V3 qg, const float* T, V3 q, float& best, int& bj) {
  const float fx = (qg.x - g.ox) * g.inv_cell, fy = (qg.y - g.oy) * g.inv_cell, fz = (qg.z - g.oz) * g.inv_cell;
  if (!(fx >= -1.f && fy >= -1.f && fz >= -1.f && fx < (float)(g.dx + 1) && fy < (float)(g.dy + 1) && fz < (float)(g.dz + 1))) return;
  const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
  const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dx - 1);
  if (x0 > x1) return;
  for (int z = max(cz - 1, 0); z <= min(cz + 1, g.dz - 1); ++z)
    for (int y = max(cy - 1, 0); y <= min(cy + 1, g.dy - 1); ++y) grid_scan_run<TRANSFORM>(g, (z * g.dy + y) * g.dx, x0, x1, T, q, best, bj);
}

// ring-expanding search: exact nearest neighbour if it lies within sqrt(max_d2), otherwise "whatever was
// seen" (which the caller rejects by distance, exactly as it rejects the true NN).
template <bool TRANSFORM>
__device__ __forceinline__ void grid_nn_rings(const GridDev& g, V3 qg, const float* T, V3 q, int max_ring, float max_d2, float& best,
                                              int& bj) {
  const float fx = (qg.x - g.ox) * g.inv_cell, fy = (qg.y - g.oy) * g.inv_cell, fz = (qg.z - g.oz) * g.inv_cell;
  const float lim = (float)max_ring + 1.f;
  if (!(fx >= -lim && fy >= -lim && fz >= -lim && fx < (float)g.dx + lim && fy < (float)g.dy + lim && fz < (float)g.dz + lim)) return;
  const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
  for (int k = 0; k <= max_ring; ++k) {
    for (int ddz = -k; ddz <= k; ++ddz) {
      const int z = cz + ddz;
      if (z < 0 || z >= g.dz) continue;
      for (int ddy = -k; ddy <= k; ++ddy) {
        const int y = cy + ddy;
        if (y < 0 || y >= g.dy) continue;
        const int row = (z * g.dy + y) * g.dx;
        if (ddz == -k || ddz == k || ddy == -k || ddy == k) {
          const int x0 = max(cx - k, 0), x1 = min(cx + k, g.dx - 1);
          if (x0 <= x1) grid_scan_run<TRANSFORM>(g, row, x0, x1, T, q, best, bj);
        } else {
          const int xa = cx - k, xb = cx + k;
          if (xa >= 0 && xa < g.dx) grid_scan_run<TRANSFORM>(g, row, xa, xa, T, q, best, bj);
          if (xb >= 0 && xb < g.dx) grid_scan_run<TRANSFORM>(g, row, xb, xb, T, q, best, bj);
        }
      }
    }
    if (k >= 1) {
      const float cov = (float)k * g.cell - GRID_MARGIN;
      const float cov2 = cov * cov;
      if (best <= cov2 || cov2 >= max_d2) break;
    }
  }
}

__device__ __forceinline__ void block_pose_and_inverse(const float* pose, float* sT, float* sTi) {
  if (threadIdx.x == 0) {
    M4 P;
    for (int k = 0; k < 16; ++k) P.m[k] = pose[k];
    const M4 inv = m4_inverse_affine(P);
    for (int k = 0; k < 12; ++k) sT[k] = P.m[k], sTi[k] = inv.m[k];
  }
  __syncthreads();
}

// computeLCP on grids: forward NN (scene point -> transformed model) through the model's rest-frame grid,
// reciprocal NN (transformed model point -> scene) through the scene grid; same terms[] as the brute-force pair.
__global__ HOP_PK_F32 __launch_bounds__(256) void k_lcp_grid(LcpArgs a) {
  __shared__ float sT[12], sTi[12];
  const int h = a.h0 + blockIdx.y;
  block_pose_and_inverse(a.pose + (size_t)h * 16, sT, sTi);
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.ns) return;
  const int i = a.perm[k];  // caller's index of this scene point
  const V3 s = v3(a.qx[k], a.qy[k], a.qz[k]);
  float best = 3.0e38f;
  int bj = -1;
  grid_nn_ring1<true>(a.model_grid, m4_point(sTi, s), sT, s, best, bj);
  float f = -1.f, g = -1.f;
  if (bj >= 0 && best < a.dist * a.dist) {
    const V3 nmod = m4_dir(sT, v3(a.mnx[bj], a.mny[bj], a.mnz[bj]));
    f = lcp_term(v3(a.qnx[k], a.qny[k], a.qnz[k]), nmod, best, a.dist, a.cos_thres);
    const V3 pm = m4_point(sT, v3(a.mx[bj], a.my[bj], a.mz[bj]));
    float rbest = 3.0e38f;
    int rk = -1;
    grid_nn_ring1<false>(a.scene_grid, pm, nullptr, pm, rbest, rk);
    if (rk >= 0) g = lcp_term(nmod, v3(a.snx[rk], a.sny[rk], a.snz[rk]), rbest, a.dist, a.cos_thres);
  }
  a.terms[((size_t)blockIdx.y * a.ns + i) * 2 + 0] = f;
  a.terms[((size_t)blockIdx.y * a.ns + i) * 2 + 1] = g;
}

// ------------------------------------------------------------------------------------------------
// NN cell lists (nn_mode 2).  For a voxel C and the cloud M:
//   U(C)    = min_m  maxdist(m, C)          -- every query q in C has its nearest neighbour within U(C)
//   list(C) = { m : mindist(m, C) <= min(U(C), max_dist) + margin }
// If the true nearest neighbour m* of q (q in C) is within max_dist it satisfies
// mindist(m*,C) <= |q-m*| <= U(C), hence m* is in list(C); scanning list(C) with the exact distance expression
// and the (d^2, index) order therefore reproduces the linear scan.  `margin` absorbs the float error of
// mapping the query into the cloud's rest frame (GRID_MARGIN) and of the box arithmetic below.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cell_box(const CellListBuildArgs& a, int cidx, float lo[3], float hi[3]) {
  const int cx = cidx % a.dx, cy = (cidx / a.dx) % a.dy, cz = cidx / (a.dx * a.dy);
  lo[0] = a.ox + (float)cx * a.cell, lo[1] = a.oy + (float)cy * a.cell, lo[2] = a.oz + (float)cz * a.cell;
  for (int k = 0; k < 3; ++k) hi[k] = lo[k] + a.cell;
}
__device__ __forceinline__ float box_maxdist2(const float lo[3], const float hi[3], float x, float y, float z) {
  const float dx = fmaxf(fabsf(x - lo[0]), fabsf(x - hi[0])), dy = fmaxf(fabsf(y - lo[1]), fabsf(y - hi[1])),
              dz = fmaxf(fabsf(z - lo[2]), fabsf(z - hi[2]));
  return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ float box_mindist2(const float lo[3], const float hi[3], float x, float y, float z) {
  const float dx = fmaxf(fmaxf(lo[0] - x, x - hi[0]), 0.f), dy = fmaxf(fmaxf(lo[1] - y, y - hi[1]), 0.f),
              dz = fmaxf(fmaxf(lo[2] - z, z - hi[2]), 0.f);
  return dx * dx + dy * dy + dz * dz;
}

__global__ HOP_PK_F32 __launch_bounds__(256) void k_cell_list_bounds(CellListBuildArgs a) {
  __shared__ float red[4];
  const int cidx = blockIdx.x;
  float lo[3], hi[3];
  cell_box(a, cidx, lo, hi);
  float best = 3.0e38f;
  for (int i = threadIdx.x; i < a.n; i += blockDim.x) best = fminf(best, box_maxdist2(lo, hi, a.x[i], a.y[i], a.z[i]));
  for (int off = 32; off > 0; off >>= 1) best = fminf(best, __shfl_down(best, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) a.u2[cidx] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
}

__device__ __forceinline__ float cell_list_thr2(const CellListBuildArgs& a, int cidx) {
  const float u = sqrtf(a.u2[cidx]) * 1.00001f + a.margin;  // rounded up
  const float t = fminf(u, a.max_dist + a.margin);
  return t * t * 1.00001f;
}

// Pruning (second stage of the list construction).  m in list(C) is DOMINATED when some other m' of the list is
// closer than m to every point of the (margin-expanded) box by more than `dom_eps`:
//     max_{q in box} ( |q-m'|^2 - |q-m|^2 ) < -dom_eps ,   the left side being linear in q  (maximum at a corner).
// A dominated point can neither be the nearest neighbour of a query in C nor tie with it (dom_eps is far above the
// float error of the distance expression at these magnitudes), and domination is a strict partial order, so the
// surviving (maximal) points still contain the linear scan's answer.  What is left is essentially the set of points
// whose Voronoi cell meets the voxel: ~5 instead of ~26 at cell = spacing.
// o dominates m over the box [blo, bhi]:  max_q ( |q-o|^2 - |q-m|^2 ) < -eps ;  f(q) = -2 q.e + |o|^2 - |m|^2 with e = o - m
__device__ __forceinline__ bool dominates(const float4& o, const float4& m, const double blo[3], const double bhi[3], float eps) {
  const double ex = (double)o.x - m.x, ey = (double)o.y - m.y, ez = (double)o.z - m.z;
  const double oo = (double)o.x * o.x + (double)o.y * o.y + (double)o.z * o.z;
  const double mm = (double)m.x * m.x + (double)m.y * m.y + (double)m.z * m.z;
  const double qe = fmin(ex * blo[0], ex * bhi[0]) + fmin(ey * blo[1], ey * bhi[1]) + fmin(ez * blo[2], ez * bhi[2]);
  return (oo - mm) - 2.0 * qe < -(double)eps;
}
#define CELL_LCAP 1024
template <bool WRITE>
__global__ HOP_PK_F32 __launch_bounds__(64) void k_cell_list_build(CellListBuildArgs a) {
  __shared__ float4 lp[CELL_LCAP];
  __shared__ unsigned char keep[CELL_LCAP];
  const int cidx = blockIdx.x;
  float lo[3], hi[3];
  cell_box(a, cidx, lo, hi);
  const float thr2 = cell_list_thr2(a, cidx);
  const int lane = threadIdx.x;
  int L = 0;       // candidates staged in LDS (ascending point order)
  int extra = 0;   // candidates beyond the LDS capacity: kept without the pruning test
  int pos = WRITE ? a.start[cidx] : 0;
  // pass 1: U-bound candidates
  for (int i0 = 0; i0 < a.n; i0 += 64) {
    const int i = i0 + lane;
    float x = 0, y = 0, z = 0;
    bool in = false;
    if (i < a.n) {
      x = a.x[i], y = a.y[i], z = a.z[i];
      in = box_mindist2(lo, hi, x, y, z) <= thr2;
    }
    const unsigned long long m = __ballot(in);
    const int at = L + __popcll(m & ((1ull << lane) - 1ull));
    if (in && at < CELL_LCAP) lp[at] = make_float4(x, y, z, __int_as_float(i));
    L += __popcll(m);
  }
  if (L > CELL_LCAP) extra = L - CELL_LCAP, L = CELL_LCAP;
  __syncthreads();
  // pass 2: domination test in double
  const double blo[3] = {(double)lo[0] - a.margin, (double)lo[1] - a.margin, (double)lo[2] - a.margin};
  const double bhi[3] = {(double)hi[0] + a.margin, (double)hi[1] + a.margin, (double)hi[2] + a.margin};
  for (int u = lane; u < L; u += 64) {
    const float4 m = lp[u];
    const double mm = (double)m.x * m.x + (double)m.y * m.y + (double)m.z * m.z;
    bool dominated = false;
    for (int v = 0; v < L && !dominated; ++v) {
      if (v == u) continue;
      const float4 o = lp[v];
      const double ex = (double)o.x - m.x, ey = (double)o.y - m.y, ez = (double)o.z - m.z;
      const double oo = (double)o.x * o.x + (double)o.y * o.y + (double)o.z * o.z;
      // f(q) = -2 q.e + |o|^2 - |m|^2 ; max over the box takes the corner minimising q.e
      const double qe = fmin(ex * blo[0], ex * bhi[0]) + fmin(ey * blo[1], ey * bhi[1]) + fmin(ez * blo[2], ez * bhi[2]);
      dominated = (oo - mm) - 2.0 * qe < -(double)a.dom_eps;
    }
    keep[u] = dominated ? 0 : 1;
  }
  __syncthreads();
  // pass 3: count / write the survivors in ascending point order
  int kept = 0;
  for (int u0 = 0; u0 < L; u0 += 64) {
    const int u = u0 + lane;
    const bool k = u < L && keep[u];
    const unsigned long long m = __ballot(k);
    if (WRITE && k) {
      const int at = pos + kept + __popcll(m & ((1ull << lane) - 1ull));
      const float4 t = lp[u];
      a.pts[at] = t;
      if (a.nrm) {
        const int i = __float_as_int(t.w);
        a.nrm[at] = make_float4(a.nx[i], a.ny[i], a.nz[i], 0.f);
      }
    }
    kept += __popcll(m);
  }
  if (extra > 0) {  // overflow (not reached at the sizes this build sees): second scan, unpruned tail
    int seen = 0;
    for (int i0 = 0; i0 < a.n; i0 += 64) {
      const int i = i0 + lane;
      float x = 0, y = 0, z = 0;
      bool in = false;
      if (i < a.n) {
        x = a.x[i], y = a.y[i], z = a.z[i];
        in = box_mindist2(lo, hi, x, y, z) <= thr2;
      }
      const unsigned long long m = __ballot(in);
      const int ord = seen + __popcll(m & ((1ull << lane) - 1ull));
      if (WRITE && in && ord >= CELL_LCAP) {
        const int at = pos + kept + (ord - CELL_LCAP);
        a.pts[at] = make_float4(x, y, z, __int_as_float(i));
        if (a.nrm) a.nrm[at] = make_float4(a.nx[i], a.ny[i], a.nz[i], 0.f);
      }
      seen += __popcll(m);
    }
    kept += extra;
  }
  if (!WRITE && lane == 0) a.count[cidx] = kept;
}
template __global__ void k_cell_list_build<false>(CellListBuildArgs);
template __global__ void k_cell_list_build<true>(CellListBuildArgs);

// The same lists built from a ring grid of the cloud (ring cell >= max_dist + margin), one thread per list cell: every
// point with mindist(m, C) <= max_dist + margin lies in the few ring cells that the box of C grown by that radius
// touches, so those are the only candidates.  The list grid is the ring grid padded by one ring cell on every side
// (queries up to max_dist outside the cloud's box) and subdivided `sub` times per axis.  Used for per-frame clouds
// (the scene), where scanning the whole cloud per cell would be far too slow.
// EXIST mode (Verify only asks "is any point within max_dist"): a cell all of whose queries are within max_dist of one
// and the same point (U(C) <= max_dist - margin) keeps just that point -- the first candidate then always hits.
// stage 1 (one thread per list cell): does any grid row the cell can draw candidates from hold a point?
__device__ __forceinline__ void local_ranges(const CellListBuildArgs& a, const GridDev& g, const float lo[3], const float hi[3], int& x0, int& x1,
                                             int& y0, int& y1, int& z0, int& z1) {
  const float R = a.max_dist + a.margin + 1.0e-6f;
  x0 = max((int)floorf((lo[0] - R - g.ox) * g.inv_cell), 0), x1 = min((int)floorf((hi[0] + R - g.ox) * g.inv_cell), g.dx - 1);
  y0 = max((int)floorf((lo[1] - R - g.oy) * g.inv_cell), 0), y1 = min((int)floorf((hi[1] + R - g.oy) * g.inv_cell), g.dy - 1);
  z0 = max((int)floorf((lo[2] - R - g.oz) * g.inv_cell), 0), z1 = min((int)floorf((hi[2] + R - g.oz) * g.inv_cell), g.dz - 1);
}
__global__ HOP_PK_F32 __launch_bounds__(256) void k_cell_list_local_flag(CellListBuildArgs a, GridDev g, int* __restrict__ flag) {
  const int cidx = blockIdx.x * blockDim.x + threadIdx.x;
  const int ncell = a.dx * a.dy * a.dz;
  if (cidx > ncell) return;
  if (cidx == ncell) {  // sentinel slot of the scans
    flag[cidx] = 0;
    a.count[cidx] = 0;
    return;
  }
  float lo[3], hi[3];
  cell_box(a, cidx, lo, hi);
  int x0, x1, y0, y1, z0, z1;
  local_ranges(a, g, lo, hi, x0, x1, y0, y1, z0, z1);
  int any = 0;
  if (x0 <= x1)
    for (int z = z0; z <= z1 && !any; ++z)
      for (int y = y0; y <= y1 && !any; ++y) {
        const int row = (z * g.dy + y) * g.dx;
        any = g.cell_start[row + x1 + 1] > g.cell_start[row + x0];
      }
  flag[cidx] = any;
  a.count[cidx] = 0;
}
// work[k] = k-th flagged cell (flag_scan = exclusive scan of flag)
__global__ HOP_PK_F32 __launch_bounds__(256) void k_cell_list_local_work(const int* __restrict__ flag, const int* __restrict__ flag_scan, int ncell,
                                                              int* __restrict__ work) {
  const int cidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (cidx < ncell && flag[cidx]) work[flag_scan[cidx]] = cidx;
}

// stage 2 (one wave per flagged cell): U(C) over the candidate rows, survivors of the threshold and of the point that
// realises U(C) into an LDS list, pairwise domination among them, ballot-compacted output.  Lanes own candidate rows.
#define LOCAL_WCAP 512
#define LOCAL_KEEP 192  /* results of the counting pass up to this length are replayed by the writing pass */
// SUB lanes work on one voxel (256 / SUB voxels per workgroup): the candidate rows of a voxel are few (3 x 3 ring cells: 9 rows), so
// a whole wavefront per voxel left most lanes idle in a latency-bound kernel.
template <bool WRITE, int SUB>
__global__ HOP_PK_F32 __launch_bounds__(256) void k_cell_list_local(CellListBuildArgs a, GridDev g, int exist_mode, const int* __restrict__ work, int nwork,
                                                         int* __restrict__ keep_buf) {
  __shared__ int list_s[256 / SUB][LOCAL_WCAP];
  const int grp = threadIdx.x / SUB, lane = threadIdx.x % SUB;
  const int gsh = ((threadIdx.x & 63) / SUB) * SUB;  // this group's bits in a wavefront ballot
  const unsigned long long gmask = SUB == 64 ? ~0ull : ((1ull << (SUB & 63)) - 1ull);
#define GBALLOT(x) ((__ballot(x) >> gsh) & gmask)
  const int wi = blockIdx.x * (256 / SUB) + grp;
  if (wi >= nwork) return;  // whole groups leave; no block-level barrier below
  int* list = list_s[grp];
  const int cidx = work[wi];
  int* keepw = keep_buf + (size_t)wi * LOCAL_KEEP;
  if (WRITE) {
    const int pos = a.start[cidx], cnt = a.start[cidx + 1] - pos;
    if (cnt == 0) return;
    if (cnt <= LOCAL_KEEP) {  // replay
      for (int i = lane; i < cnt; i += SUB) {
        const float4 m = g.pts[keepw[i]];
        a.pts[pos + i] = m;
        if (a.nrm) {
          const int id = __float_as_int(m.w);
          a.nrm[pos + i] = make_float4(a.nx[id], a.ny[id], a.nz[id], 0.f);
        }
      }
      return;
    }
  }
  float lo[3], hi[3];
  cell_box(a, cidx, lo, hi);
  int x0, x1, y0, y1, z0, z1;
  local_ranges(a, g, lo, hi, x0, x1, y0, y1, z0, z1);
  const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
  // step 1: U(C) and the point that realises it (ties: lowest grid position)
  float u2 = 3.0e38f;
  int kb = 0x7fffffff;
  int maxlen = 0;
  for (int r = lane; r < nrows; r += SUB) {
    const int row = ((z0 + r / ny) * g.dy + (y0 + r % ny)) * g.dx;
    const int b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
    maxlen = max(maxlen, e - b);
    for (int k = b; k < e; ++k) {
      const float4 t = g.pts[k];
      const float m2 = box_maxdist2(lo, hi, t.x, t.y, t.z);
      if (m2 < u2 || (m2 == u2 && k < kb)) u2 = m2, kb = k;
    }
  }
#pragma unroll
  for (int off = SUB / 2; off > 0; off >>= 1) {
    const float ou = __shfl_xor(u2, off, SUB);
    const int ok = __shfl_xor(kb, off, SUB);
    if (ou < u2 || (ou == u2 && ok < kb)) u2 = ou, kb = ok;
    maxlen = max(maxlen, __shfl_xor(maxlen, off, SUB));
  }
  const int pos = WRITE ? a.start[cidx] : 0;
  const float4 best = g.pts[kb];
  const float all_r = a.max_dist - a.margin;
  if (exist_mode && all_r > 0.f && sqrtf(u2) * 1.00001f <= all_r) {
    if (lane == 0) {
      if (WRITE) a.pts[pos] = best;
      else a.count[cidx] = 1, keepw[0] = kb;
    }
    return;
  }
  const float u = sqrtf(u2) * 1.00001f + a.margin;
  const float th = fminf(u, a.max_dist + a.margin);
  const float thr2 = th * th * 1.00001f;
  const double blo[3] = {(double)lo[0] - a.margin, (double)lo[1] - a.margin, (double)lo[2] - a.margin};
  const double bhi[3] = {(double)hi[0] + a.margin, (double)hi[1] + a.margin, (double)hi[2] + a.margin};
  // step 2: survivors of the threshold and of `best`, in (position-in-row, row) order
  int L = 0;
  for (int r0 = 0; r0 < nrows; r0 += SUB) {
    const int r = r0 + lane;
    int b = 0, e = 0;
    if (r < nrows) {
      const int row = ((z0 + r / ny) * g.dy + (y0 + r % ny)) * g.dx;
      b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
    }
    for (int j = 0; j < maxlen; ++j) {
      const int k = b + j;
      bool in = false;
      if (k < e) {
        const float4 t = g.pts[k];
        in = box_mindist2(lo, hi, t.x, t.y, t.z) <= thr2 && (k == kb || !dominates(best, t, blo, bhi, a.dom_eps));
      }
      const unsigned long long m = GBALLOT(in);
      if (in) {
        const int at = L + __popcll(m & ((1ull << lane) - 1ull));
        if (at < LOCAL_WCAP) list[at] = k;
      }
      L += __popcll(m);
    }
  }
  __builtin_amdgcn_wave_barrier();
  int kept = 0;
  if (L <= LOCAL_WCAP) {
    // step 3a: a few rounds of pruning against the next-closest survivor (by farthest-corner distance): these remove
    // almost everything that pairwise testing would, at O(L) per round.  Removed entries are marked ~k.
    float pm2 = u2;
    int pk = kb;
    for (int round = 0; round < 48 && L > 8; ++round) {
      float c2 = 3.0e38f;
      int ck = 0x7fffffff;
      for (int i = lane; i < L; i += SUB) {
        const int k = list[i];
        if (k < 0) continue;
        const float4 t = g.pts[k];
        const float m2 = box_maxdist2(lo, hi, t.x, t.y, t.z);
        const bool after = m2 > pm2 || (m2 == pm2 && k > pk);  // pivots advance in (m2, k) order
        if (after && (m2 < c2 || (m2 == c2 && k < ck))) c2 = m2, ck = k;
      }
#pragma unroll
      for (int off = SUB / 2; off > 0; off >>= 1) {
        const float oc = __shfl_xor(c2, off, SUB);
        const int ok = __shfl_xor(ck, off, SUB);
        if (oc < c2 || (oc == c2 && ok < ck)) c2 = oc, ck = ok;
      }
      if (ck == 0x7fffffff) break;
      pm2 = c2, pk = ck;
      const float4 pv = g.pts[ck];
      for (int i = lane; i < L; i += SUB) {
        const int k = list[i];
        if (k < 0 || k == ck) continue;
        if (dominates(pv, g.pts[k], blo, bhi, a.dom_eps)) list[i] = ~k;
      }
      __builtin_amdgcn_wave_barrier();
    }
    // compact the survivors (order preserved)
    int L2 = 0;
    for (int i0 = 0; i0 < L; i0 += SUB) {
      const int i = i0 + lane;
      const int k = i < L ? list[i] : -1;
      const unsigned long long m = GBALLOT(k >= 0);
      __builtin_amdgcn_wave_barrier();
      if (k >= 0) list[L2 + __popcll(m & ((1ull << lane) - 1ull))] = k;
      L2 += __popcll(m);
      __builtin_amdgcn_wave_barrier();
    }
    L = L2;
    // step 3b: pairwise domination among what is left
    for (int i0 = 0; i0 < L; i0 += SUB) {
      const int i = i0 + lane;
      bool keep = false;
      float4 m = make_float4(0, 0, 0, 0);
      int km = -1;
      if (i < L) {
        km = list[i];
        m = g.pts[km];
        keep = true;
        // (a few cells far from a dense cloud keep hundreds of survivors: the O(L^2) pass is skipped there -- longer
        // lists, same answers -- rather than letting a handful of waves set the kernel's duration)
        for (int j = 0; j < L && keep && L <= a.pair_max; ++j)
          if (j != i) keep = !dominates(g.pts[list[j]], m, blo, bhi, a.dom_eps);
      }
      const unsigned long long kmask = GBALLOT(keep);
      if (keep) {
        const int n = kept + __popcll(kmask & ((1ull << lane) - 1ull));
        if (WRITE) {
          a.pts[pos + n] = m;
          if (a.nrm) {
            const int id = __float_as_int(m.w);
            a.nrm[pos + n] = make_float4(a.nx[id], a.ny[id], a.nz[id], 0.f);
          }
        } else if (n < LOCAL_KEEP) keepw[n] = km;
      }
      kept += __popcll(kmask);
    }
  } else {
    // more survivors than the list holds (not seen at the sizes built here): keep them all, same order as step 2
    for (int r0 = 0; r0 < nrows; r0 += SUB) {
      const int r = r0 + lane;
      int b = 0, e = 0;
      if (r < nrows) {
        const int row = ((z0 + r / ny) * g.dy + (y0 + r % ny)) * g.dx;
        b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
      }
      for (int j = 0; j < maxlen; ++j) {
        const int k = b + j;
        bool in = false;
        float4 t = make_float4(0, 0, 0, 0);
        if (k < e) {
          t = g.pts[k];
          in = box_mindist2(lo, hi, t.x, t.y, t.z) <= thr2 && (k == kb || !dominates(best, t, blo, bhi, a.dom_eps));
        }
        const unsigned long long m = GBALLOT(in);
        if (WRITE && in) {
          const int at = pos + kept + __popcll(m & ((1ull << lane) - 1ull));
          a.pts[at] = t;
          if (a.nrm) {
            const int id = __float_as_int(t.w);
            a.nrm[at] = make_float4(a.nx[id], a.ny[id], a.nz[id], 0.f);
          }
        }
        kept += __popcll(m);
      }
    }
    if (!WRITE && kept <= LOCAL_KEEP) kept = LOCAL_KEEP + 1;  // unreachable in practice (L > LOCAL_WCAP); never replay this path
  }
  if (!WRITE && lane == 0) a.count[cidx] = kept;
}
#undef GBALLOT

// Scan of one cell list.  The candidates are ranked by their squared distance to the query IN THE CLOUD'S REST FRAME
// (qg, 8 flops each); the reference's distance expression -- query against the candidate moved by T, in the query's
// frame -- is evaluated only for the winner.  The two differ by float rounding of the two transforms (bounded by
// `tol` below: 2*d*delta + relative 1e-4, delta = a few ulps of the coordinate magnitude), so whenever the runner-up
// is within tol of the winner the lane falls back to the literal scan (exact expression, (d^2, index) order).
// Result: list position of the nearest neighbour (or -1) and the exact squared distance.
// `moved` receives the winner moved by T (the point the exact distance was measured to): callers that need it again
// (ICP residual, computeLCP reciprocal query) do not recompute it.
template <int B>
__device__ __forceinline__ void cells_nn(const CellListDev& c, V3 qg, const float* T, V3 q, float& best, int& bpos, V3& moved) {
  const float fx = (qg.x - c.ox) * c.inv_cell, fy = (qg.y - c.oy) * c.inv_cell, fz = (qg.z - c.oz) * c.inv_cell;
  if (!(fx >= 0.f && fy >= 0.f && fz >= 0.f && fx < (float)c.dx && fy < (float)c.dy && fz < (float)c.dz)) return;
  const int cidx = ((int)fz * c.dy + (int)fy) * c.dx + (int)fx;
  const int2 rg = c.range[cidx];  // one 8-byte load
  const int beg = rg.x, end = rg.y;
  if (beg >= end) return;
  float b1 = 3.0e38f, b2 = 3.0e38f;
  int k1 = beg;
  float wx = 0.f, wy = 0.f, wz = 0.f;  // the winner stays in registers (no second gather)
  // B entries per step: the loads are issued together (one exposed latency per B candidates); entries past the end
  // re-read the last one and are skipped.  B = 4 for the ICP lists (3.7 entries per non-trivial cell), 1 for the
  // 1 mm computeLCP lists (mostly one entry: wider steps only add cache accesses there).
  for (int k = beg; k < end; k += B) {
    float4 t[B];
#pragma unroll
    for (int u = 0; u < B; ++u) t[u] = c.pts[min(k + u, end - 1)];
#pragma unroll
    for (int u = 0; u < B; ++u) {
      const float dx = qg.x - t[u].x, dy = qg.y - t[u].y, dz = qg.z - t[u].z;
      const float d = (u == 0 || k + u < end) ? dx * dx + dy * dy + dz * dz : 3.0e38f;
      b2 = fminf(b2, fmaxf(b1, d));
      const bool better = d < b1;
      k1 = better ? k + u : k1;
      wx = better ? t[u].x : wx, wy = better ? t[u].y : wy, wz = better ? t[u].z : wz;
      b1 = fminf(b1, d);
    }
  }
  const float mag = fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), 0.25f));
  const float delta = mag * 2.0e-6f;
  // the hardware square root (1 ulp) is inflated by 1e-3: tol only has to be an upper bound
  const float tol = 2.002f * __builtin_amdgcn_sqrtf(b1) * delta + 1.0e-4f * b1 + delta * delta;
  if (b2 - b1 > tol) {
    moved = m4_point(T, v3(wx, wy, wz));
    best = sqdist_flann(q, moved);
    bpos = k1;
    return;
  }
  int bj = 0x7fffffff;
  for (int k = beg; k < end; ++k) {
    const float4 t = c.pts[k];
    const V3 tm = m4_point(T, v3(t.x, t.y, t.z));
    const float d2 = sqdist_flann(q, tm);
    const int j = __float_as_int(t.w);
    if (d2 < best || (d2 == best && j < bj)) best = d2, bj = j, bpos = k, moved = tm;
  }
}

// -DHOP_LCP_COUNT (tools/lcp_counters.py): gathers (per-lane 8 / 16-byte loads from the lists) of k_lcp_cells_fast per lookup
__device__ unsigned long long g_lcp_count[4];  // [0] (point, hypothesis) lookups  [1] gathers of the forward lookups  [2] of the reciprocal lookups  [3] reciprocal lookups
#ifdef HOP_LCP_COUNT
#define LCP_COUNT(slot, v) atomicAdd(&g_lcp_count[slot], (unsigned long long)(v))
#else
#define LCP_COUNT(slot, v) do { } while (0)
#endif
void lcp_counters_read(unsigned long long* out4, bool reset) {
  (void)hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_lcp_count), sizeof(unsigned long long) * 4);
  if (reset) {
    unsigned long long z[4] = {0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lcp_count), z, sizeof(z));
  }
}
// exact scan of a cell list whose entries are already in the query's frame (no transform): the distance expression
// is the reference's, ties go to the lower original index
__device__ __forceinline__ void cells_nn_plain(const CellListDev& c, V3 q, float& best, int& bpos) {
  const float fx = (q.x - c.ox) * c.inv_cell, fy = (q.y - c.oy) * c.inv_cell, fz = (q.z - c.oz) * c.inv_cell;
  if (!(fx >= 0.f && fy >= 0.f && fz >= 0.f && fx < (float)c.dx && fy < (float)c.dy && fz < (float)c.dz)) return;
  const int cidx = ((int)fz * c.dy + (int)fy) * c.dx + (int)fx;
  const int2 rg = c.range[cidx];
  LCP_COUNT(2, 1);
  const int beg = rg.x, end = rg.y;
  int bj = 0x7fffffff;
  for (int k = beg; k < end; ++k) {
    const float4 t = c.pts[k];
    LCP_COUNT(2, 1);
    const float d2 = sqdist_flann(q, v3(t.x, t.y, t.z));
    const int j = __float_as_int(t.w);
    if (d2 < best || (d2 == best && j < bj)) best = d2, bj = j, bpos = k;
  }
}

// computeLCP on NN cell lists (nn_mode 2): forward NN through the model's lists (rest frame), reciprocal NN through the
// scene's lists.  A block covers 64 Morton-consecutive scene points x LCP_TH hypotheses (each wave one hypothesis at a
// time: neighbouring lanes touch neighbouring cells); the terms go through LDS into a POINT-major table
// terms[sorted position][hypothesis], LCP_TH x 8 B = one full 128-byte line per point and block.  The ordered sum then
// reads that table row by row in the caller's point order with lanes = hypotheses, which is coalesced whatever the
// order of the rows -- so neither kernel scatters (the hypothesis-major layout cost 4x write amplification plus the
// read-for-ownership traffic of partial lines: 17 GB of HBM traffic per launch instead of 0.8).
constexpr int LCP_TH = 16;
__global__ HOP_PK_F32 __launch_bounds__(256) void k_lcp_cells(LcpArgs a, int hb, int hs, int npt) {
  __shared__ float2 out[64][LCP_TH + 1];
  const int pt = blockIdx.x % npt, ht = blockIdx.x / npt;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: the pose pointers below stay in SGPRs
  const int k = pt * 64 + lane;
  const bool kin = k < a.ns;
  V3 s = v3(0, 0, 0), sn = v3(0, 0, 0);
  if (kin) s = v3(a.qx[k], a.qy[k], a.qz[k]), sn = v3(a.qnx[k], a.qny[k], a.qnz[k]);
  for (int j = 0; j < LCP_TH / 4; ++j) {
    const int hh = wave * (LCP_TH / 4) + j, hl = ht * LCP_TH + hh;
    float f = -1.f, g = -1.f;
    if (kin && hl < hb) {
      const float* __restrict__ T = a.pose + (size_t)(a.h0 + hl) * 16;  // wave-uniform: scalar loads
      const float* __restrict__ Ti = a.pose_inv + (size_t)(a.h0 + hl) * 12;
      float best = 3.0e38f;
      int pos = -1;
      V3 pm;  // the matched model point under the hypothesis
      cells_nn<1>(a.model_cells, m4_point(Ti, s), T, s, best, pos, pm);
      if (pos >= 0 && best < a.dist * a.dist) {
        const float4 mnr = a.model_cells.nrm[pos];
        const V3 nmod = vnormalized(m4_dir(T, v3(mnr.x, mnr.y, mnr.z)));  // normalised once, used by both terms
        f = lcp_term_unit(sn, nmod, best, a.dist, a.cos_thres);             // sn, rn: unit normals (k_unit_normals)
        float rbest = 3.0e38f;
        int rk = -1;
        cells_nn_plain(a.scene_cells, pm, rbest, rk);
        if (rk >= 0) {
          const float4 rn = a.scene_cells.nrm[rk];
          g = lcp_term_unit(nmod, v3(rn.x, rn.y, rn.z), rbest, a.dist, a.cos_thres);
        }
      }
    }
    out[lane][hh] = make_float2(f, g);
  }
  __syncthreads();
  float2* tt = reinterpret_cast<float2*>(a.terms);
  for (int e = threadIdx.x; e < 64 * LCP_TH; e += 256) {
    const int p = e / LCP_TH, hh = e % LCP_TH;
    const int kk = pt * 64 + p;
    if (kk < a.ns) {
      // streaming store: the 1.6 GB table is written once and read once by the sum; it should not evict the lists
      const float2 v = out[p][hh];
      unsigned long long bits;
      memcpy(&bits, &v, 8);
      __builtin_nontemporal_store(bits, reinterpret_cast<unsigned long long*>(tt + (size_t)kk * hs + ht * LCP_TH + hh));
    }
  }
}

// ordered sum over the point-major table: lanes = hypotheses, rows visited in the caller's point order (inv[i] = sorted
// position of caller point i); additions in exactly that order, loads issued 32 rows ahead
__global__ HOP_PK_F32 __launch_bounds__(64) void k_lcp_sum_t(LcpArgs a, int hb, int hs) {
  const int hl = blockIdx.x * blockDim.x + threadIdx.x;
  if (hl >= hb) return;
  const float2* tt = reinterpret_cast<const float2*>(a.terms) + hl;
  float cp = 0.f;
  constexpr int U = 32;
  int i = 0;
  for (; i + U <= a.ns; i += U) {
    float2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = tt[(size_t)a.inv_perm[i + u] * hs];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (v[u].x >= 0.f) cp += v[u].x;
      if (v[u].y >= 0.f) cp += v[u].y;
    }
  }
  for (; i < a.ns; ++i) {
    const float2 v = tt[(size_t)a.inv_perm[i] * hs];
    if (v.x >= 0.f) cp += v.x;
    if (v.y >= 0.f) cp += v.y;
  }
  a.score[a.h0 + hl] = cp;
}

// ------------------------------------------------------------------------------------------------
// K5: batched point-to-plane ICP (Utils::runICP, Utils.cpp:188-229, pcl::IterativeClosestPoint as
// configured there -- restated).  One iteration = k_icp_nn (move the source by the last
// increment, find correspondences, accumulate the 6x6 normal equations in double) + k_icp_solve.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  return v;
}

template <int R>
__global__ HOP_PK_F32 __launch_bounds__(256) void k_icp_nn(IcpArgs a) {
  __shared__ float4 tile[NN_TILE];
  __shared__ double red[4][ICP_NACC];
  const int hl = blockIdx.y, h = a.h0 + hl;
  IcpState& st = a.state[hl];
  if (!st.active) return;
  const float* pose = a.pose + (size_t)h * 16;
  float Tinc[12];
  for (int k = 0; k < 12; ++k) Tinc[k] = st.T_inc[k];
  V3 q[R], qn[R];
  float best[R];
  int bidx[R];
  const bool first = a.iter == 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
    best[r] = 3.0e38f;
    bidx[r] = -1;
    if (i < a.ns) {
      V3 p, n;
      if (first) {
        p = v3(a.sx[i], a.sy[i], a.sz[i]);
        n = v3(a.snx[i], a.sny[i], a.snz[i]);
      } else {
        const size_t o = (size_t)hl * a.ns * 6 + i;
        p = v3(a.moved[o], a.moved[o + a.ns], a.moved[o + 2 * (size_t)a.ns]);
        n = v3(a.moved[o + 3 * (size_t)a.ns], a.moved[o + 4 * (size_t)a.ns], a.moved[o + 5 * (size_t)a.ns]);
        p = m4_point(Tinc, p);
        n = m4_dir(Tinc, n);
      }
      const size_t o = (size_t)hl * a.ns * 6 + i;
      a.moved[o] = p.x, a.moved[o + a.ns] = p.y, a.moved[o + 2 * (size_t)a.ns] = p.z;
      a.moved[o + 3 * (size_t)a.ns] = n.x, a.moved[o + 4 * (size_t)a.ns] = n.y, a.moved[o + 5 * (size_t)a.ns] = n.z;
      q[r] = p, qn[r] = n;
    } else {
      q[r] = v3(-HOP_FAR, -HOP_FAR, -HOP_FAR);
      qn[r] = v3(0, 0, 0);
    }
  }
  for (int start = 0; start < a.nm; start += NN_TILE) {
    const int tn = min(NN_TILE, round_up(a.nm - start, NN_CH));
    __syncthreads();
    stage_tile_tf(tile, a.mx, a.my, a.mz, start, a.nm, tn, pose);
    __syncthreads();
    nn_scan_tile<R, true, true>(tile, tn, start, q, best, bidx);
  }
  double acc[ICP_NACC];
#pragma unroll
  for (int k = 0; k < ICP_NACC; ++k) acc[k] = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
    if (i >= a.ns || bidx[r] < 0 || !(best[r] <= a.max_d2)) continue;
    const int j = bidx[r];
    const V3 nt = m4_dir(pose, v3(a.mnx[j], a.mny[j], a.mnz[j]));
    if (!(vdot(qn[r], nt) >= a.cos_thr)) continue;
    const V3 tq = m4_point(pose, v3(a.mx[j], a.my[j], a.mz[j]));
    const V3 c = vcross(q[r], nt);
    const double J[6] = {c.x, c.y, c.z, nt.x, nt.y, nt.z};
    const double res = (double)vdot(q[r] - tq, nt);
    int k = 0;
    for (int u = 0; u < 6; ++u)
      for (int v = 0; v <= u; ++v) acc[k++] += J[u] * J[v];
    for (int u = 0; u < 6; ++u) acc[21 + u] -= J[u] * res;
    acc[27] += (double)best[r];
    acc[28] += 1.0;
    acc[29] += (double)q[r].x, acc[30] += (double)q[r].y, acc[31] += (double)q[r].z;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < ICP_NACC; ++k) {
    const double s = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < ICP_NACC) {
    const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    a.partial[((size_t)hl * gridDim.x + blockIdx.x) * ICP_NACC + threadIdx.x] = s;
  }
}
template __global__ void k_icp_nn<4>(IcpArgs);

template <int R>
__global__ HOP_PK_F32 __launch_bounds__(256) void k_icp_nn_grid(IcpArgs a) {
  __shared__ float sT[12], sTi[12];
  __shared__ double red[4][ICP_NACC];
  const int hl = blockIdx.y, h = a.h0 + hl;
  IcpState& st = a.state[hl];
  if (!st.active) return;
  const float* pose = a.pose + (size_t)h * 16;
  float Tinc[12];
  for (int k = 0; k < 12; ++k) Tinc[k] = st.T_inc[k];
  V3 q[R], qn[R];
  float best[R];
  int bidx[R];
  const bool first = a.iter == 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
    best[r] = 3.0e38f;
    bidx[r] = -1;
    if (i < a.ns) {
      V3 p, n;
      if (first) {
        p = v3(a.sx[i], a.sy[i], a.sz[i]);
        n = v3(a.snx[i], a.sny[i], a.snz[i]);
      } else {
        const size_t o = (size_t)hl * a.ns * 6 + i;
        p = v3(a.moved[o], a.moved[o + a.ns], a.moved[o + 2 * (size_t)a.ns]);
        n = v3(a.moved[o + 3 * (size_t)a.ns], a.moved[o + 4 * (size_t)a.ns], a.moved[o + 5 * (size_t)a.ns]);
        p = m4_point(Tinc, p);
        n = m4_dir(Tinc, n);
      }
      const size_t o = (size_t)hl * a.ns * 6 + i;
      a.moved[o] = p.x, a.moved[o + a.ns] = p.y, a.moved[o + 2 * (size_t)a.ns] = p.z;
      a.moved[o + 3 * (size_t)a.ns] = n.x, a.moved[o + 4 * (size_t)a.ns] = n.y, a.moved[o + 5 * (size_t)a.ns] = n.z;
      q[r] = p, qn[r] = n;
    } else {
      q[r] = v3(-HOP_FAR, -HOP_FAR, -HOP_FAR);
      qn[r] = v3(0, 0, 0);
    }
  }
  block_pose_and_inverse(pose, sT, sTi);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
    if (i < a.ns) grid_nn_rings<true>(a.model_grid, m4_point(sTi, q[r]), sT, q[r], a.max_ring, a.max_d2, best[r], bidx[r]);
  }
  double acc[ICP_NACC];
#pragma unroll
  for (int k = 0; k < ICP_NACC; ++k) acc[k] = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
    if (i >= a.ns || bidx[r] < 0 || !(best[r] <= a.max_d2)) continue;
    const int j = bidx[r];
    const V3 nt = m4_dir(pose, v3(a.mnx[j], a.mny[j], a.mnz[j]));
    if (!(vdot(qn[r], nt) >= a.cos_thr)) continue;
    const V3 tq = m4_point(pose, v3(a.mx[j], a.my[j], a.mz[j]));
    const V3 c = vcross(q[r], nt);
    const double J[6] = {c.x, c.y, c.z, nt.x, nt.y, nt.z};
    const double res = (double)vdot(q[r] - tq, nt);
    int k = 0;
    for (int u = 0; u < 6; ++u)
      for (int v = 0; v <= u; ++v) acc[k++] += J[u] * J[v];
    for (int u = 0; u < 6; ++u) acc[21 + u] -= J[u] * res;
    acc[27] += (double)best[r];
    acc[28] += 1.0;
    acc[29] += (double)q[r].x, acc[30] += (double)q[r].y, acc[31] += (double)q[r].z;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < ICP_NACC; ++k) {
    const double s = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < ICP_NACC) {
    const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    a.partial[((size_t)hl * gridDim.x + blockIdx.x) * ICP_NACC + threadIdx.x] = s;
  }
}
template __global__ void k_icp_nn_grid<4>(IcpArgs);

// nn_mode 2: the correspondence search (few registers, high occupancy, gather bound) and the normal-equation
// accumulation (32 f64 accumulators per lane) are separate launches.  Neither keeps a moved copy of the source:
// a point's position at iteration k is the chain T_k(...T_1(p0)) of the increments solved so far, re-applied from the
// per-hypothesis history (same float operations in the same order as moving the stored cloud once per iteration,
// 33 flops per step instead of 48 B of HBM traffic).  The only per-point state between the two launches is the list
// position of the correspondence (4 B).
__device__ __forceinline__ void icp_chain_point(const float* __restrict__ hist, int iter, V3& p) {
  for (int k = 0; k < iter; ++k) p = m4_point(hist + 12 * k, p);
}
__device__ __forceinline__ void icp_chain_point_normal(const float* __restrict__ hist, int iter, V3& p, V3& n) {
  for (int k = 0; k < iter; ++k) {
    p = m4_point(hist + 12 * k, p);
    n = m4_dir(hist + 12 * k, n);
  }
}

__global__ HOP_PK_F32 __launch_bounds__(256) void k_icp_corr_cells(IcpArgs a) {
  const int hl = blockIdx.y, h = a.h0 + hl;
  const IcpState& st = a.state[hl];
  if (!st.active) return;
  const float* __restrict__ sT = a.pose + (size_t)h * 16;         // wave-uniform: scalar loads
  const float* __restrict__ sTi = a.pose_inv + (size_t)h * 12;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.ns) return;
  V3 p = v3(a.sx[i], a.sy[i], a.sz[i]);
  icp_chain_point(a.hist + (size_t)hl * a.max_iter * 12, a.iter, p);
  float best = 3.0e38f;
  int pos = -1;
  V3 moved;
  cells_nn<4>(a.cells, m4_point(sTi, p), sT, p, best, pos, moved);
  a.corr_idx[(size_t)hl * a.ns + i] = (pos >= 0 && best <= a.max_d2) ? pos : -1;
}

__global__ HOP_PK_F32 __launch_bounds__(256) void k_icp_accum(IcpArgs a, int R) {
  __shared__ double red[4][ICP_NACC];
  const int hl = blockIdx.y, h = a.h0 + hl;
  const IcpState& st = a.state[hl];
  if (!st.active) return;
  const float* pose = a.pose + (size_t)h * 16;
  const float* hist = a.hist + (size_t)hl * a.max_iter * 12;
  double acc[ICP_NACC];
#pragma unroll
  for (int k = 0; k < ICP_NACC; ++k) acc[k] = 0.0;
  for (int r = 0; r < R; ++r) {
    const int i = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
    if (i >= a.ns) continue;
    const int pos = a.corr_idx[(size_t)hl * a.ns + i];
    if (pos < 0) continue;
    V3 q = v3(a.sx[i], a.sy[i], a.sz[i]), qn = v3(a.snx[i], a.sny[i], a.snz[i]);
    icp_chain_point_normal(hist, a.iter, q, qn);
    const float4 tp = a.cells.pts[pos], tn = a.cells.nrm[pos];
    const V3 nt = m4_dir(pose, v3(tn.x, tn.y, tn.z));
    if (!(vdot(qn, nt) >= a.cos_thr)) continue;
    const V3 tq = m4_point(pose, v3(tp.x, tp.y, tp.z));
    const float d2 = sqdist_flann(q, tq);
    const V3 c = vcross(q, nt);
    const double J[6] = {c.x, c.y, c.z, nt.x, nt.y, nt.z};
    const double res = (double)vdot(q - tq, nt);
    // J and res are float-valued, so every product below is exact in double and the fused multiply-add rounds
    // exactly like the separate multiply and add of the reference accumulation
    int k = 0;
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
      for (int v = 0; v <= u; ++v) {
        acc[k] = fma(J[u], J[v], acc[k]);
        ++k;
      }
#pragma unroll
    for (int u = 0; u < 6; ++u) acc[21 + u] = fma(-J[u], res, acc[21 + u]);
    acc[27] += (double)d2;
    acc[28] += 1.0;
    acc[29] += (double)q.x, acc[30] += (double)q.y, acc[31] += (double)q.z;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < ICP_NACC; ++k) {
    const double s = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < ICP_NACC) {
    const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    a.partial[((size_t)hl * gridDim.x + blockIdx.x) * ICP_NACC + threadIdx.x] = s;
  }
}

// nn_mode 3: correspondence search and accumulation in one launch (no correspondence array in between).  Same
// arithmetic as the split pair; R points per lane share one block-level reduction of the 32 accumulators.
template <int R>
__global__ HOP_PK_F32 __launch_bounds__(256) void k_icp_fused(IcpArgs a) {
  __shared__ double red[4][ICP_NACC];
  const int hl = blockIdx.y, h = a.h0 + hl;
  const IcpState& st = a.state[hl];
  if (!st.active) return;
  const float* __restrict__ pose = a.pose + (size_t)h * 16;
  const float* __restrict__ sTi = a.pose_inv + (size_t)h * 12;
  const float* __restrict__ hist = a.hist + (size_t)hl * a.max_iter * 12;
  double acc[ICP_NACC];
#pragma unroll
  for (int k = 0; k < ICP_NACC; ++k) acc[k] = 0.0;
  for (int r = 0; r < R; ++r) {
    const int i = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
    if (i >= a.ns) continue;
    V3 q = v3(a.sx[i], a.sy[i], a.sz[i]), qn = v3(a.snx[i], a.sny[i], a.snz[i]);
    icp_chain_point_normal(hist, a.iter, q, qn);
    float d2 = 3.0e38f;
    int pos = -1;
    V3 tq;  // the correspondence moved by the pose, as the distance was measured
    cells_nn<4>(a.cells, m4_point(sTi, q), pose, q, d2, pos, tq);
    if (pos < 0 || !(d2 <= a.max_d2)) continue;
    const float4 tn = a.cells.nrm[pos];
    const V3 nt = m4_dir(pose, v3(tn.x, tn.y, tn.z));
    if (!(vdot(qn, nt) >= a.cos_thr)) continue;
    const V3 c = vcross(q, nt);
    const double J[6] = {c.x, c.y, c.z, nt.x, nt.y, nt.z};
    const double res = (double)vdot(q - tq, nt);
    int k = 0;
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
      for (int v = 0; v <= u; ++v) {
        acc[k] = fma(J[u], J[v], acc[k]);
        ++k;
      }
#pragma unroll
    for (int u = 0; u < 6; ++u) acc[21 + u] = fma(-J[u], res, acc[21 + u]);
    acc[27] += (double)d2;
    acc[28] += 1.0;
    acc[29] += (double)q.x, acc[30] += (double)q.y, acc[31] += (double)q.z;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < ICP_NACC; ++k) {
    const double s = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < ICP_NACC) {
    const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    a.partial[((size_t)hl * gridDim.x + blockIdx.x) * ICP_NACC + threadIdx.x] = s;
  }
}
template __global__ void k_icp_fused<ICP_ACCUM_R>(IcpArgs);

// ------------------------------------------------------------------------------------------------
// nn_mode 3 / 4: the fused ICP kernel on PACKED model lists (CellListDev::rec / qlist).
//
// PMC of the round-1 kernel (profiles/r01_pmc_sq_v9.txt): SIMDs ~80 % busy issuing VALU and ~0.75 L1 tag accesses per
// clock and CU -- 6.3 cache accesses per lookup (every lane's load is its own line: range record, four 16-byte
// candidates, the winner's normal), HBM traffic 0.13x the algorithmic bytes.  Both limits are attacked here:
//   * source point and normal come from an AoS copy of the Morton-ordered scene (two 16-byte loads instead of six);
//   * rest-frame query and its grid coordinate with v_fma (the ranking part is an aid; the exact expression decides);
//   * ONE 16-byte record per cell carries up to two candidates inline (8 bytes each: coordinates quantised to 16 bits
//     in the cell's own frame, step 0.33 um at the ICP sizes, + the original index): no range indirection, and a
//     list of <= 2 costs one access; longer lists are chunks of two entries per access;
//   * a candidate costs 3 cvt + 3 sub + mul + 2 fma + and_or + med3 + min: the key is the squared distance in step
//     units with its lowest mantissa bit replaced by the slot, v_min_u32 / v_med3_u32 keep the best two keys;
//   * the winner's exact point (AoS by original index) is moved by the pose with the reference's expression;
//   * lanes whose runner-up is within `tol` of the winner (ranking cannot decide: transform rounding + quantisation)
//     re-scan their list and evaluate the exact expression for the entries within `tol` of the winner; ties go to the
//     lower index.
// The result is the same correspondence the linear scan finds (same argument as cells_nn above, with the
// quantisation error q_eq of a candidate position added to `tol`).
// ------------------------------------------------------------------------------------------------
// One gather instead of two.  A lookup reads a small record, tests one field (an empty cell ends the lookup) and only then uses the others; left alone,
// the compiler loads the tested field, branches, and fetches the rest in a SECOND load behind the branch -- two passes of the L1 tag pipe over the
// very same lines (a wavefront's gather costs 64.5 cycles of that pipe whatever its width: profiles/r02_l1_load_rates.txt; seen in the gfx950
// assembly of the packed ICP lookups' cell record and of computeLCP's head records).  Declaring every field live right after the load keeps it ONE
// 8- / 16-byte load.  The asm statement is empty: no instruction is emitted, no value changes.  (Not `volatile`: that would count as a store to
// unknown memory and turn every later uniform load -- the poses of computeLCP -- from a scalar load into a vector one.)
__device__ __forceinline__ void keep_whole(uint2& r) { asm("" : "+v"(r.x), "+v"(r.y)); }
__device__ __forceinline__ void keep_whole(uint4& r) { asm("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w)); }
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ V3 m4_point_fma(const float* T, V3 p) {
  return v3(__builtin_fmaf(T[0], p.x, __builtin_fmaf(T[1], p.y, __builtin_fmaf(T[2], p.z, T[3]))),
            __builtin_fmaf(T[4], p.x, __builtin_fmaf(T[5], p.y, __builtin_fmaf(T[6], p.z, T[7]))),
            __builtin_fmaf(T[8], p.x, __builtin_fmaf(T[9], p.y, __builtin_fmaf(T[10], p.z, T[11]))));
}
__device__ __forceinline__ V3 m4_dir_fma(const float* T, V3 n) {
  return v3(__builtin_fmaf(T[0], n.x, __builtin_fmaf(T[1], n.y, T[2] * n.z)), __builtin_fmaf(T[4], n.x, __builtin_fmaf(T[5], n.y, T[6] * n.z)),
            __builtin_fmaf(T[8], n.x, __builtin_fmaf(T[9], n.y, T[10] * n.z)));
}
__device__ __forceinline__ float rank_d2(V3 q, const float4& t) {
  const float dx = q.x - t.x, dy = q.y - t.y, dz = q.z - t.z;
  return __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
}
// -DHOP_ICP_COUNT (tools/icp_counters.py builds such a library): per-query statistics of the packed lookups
__device__ unsigned long long g_icp_count[8];
#ifdef HOP_ICP_COUNT
#define ICP_COUNT(slot, v) atomicAdd(&g_icp_count[slot], (unsigned long long)(v))
#define ICP_COUNT_WAVE(slot) do { if (__builtin_amdgcn_mbcnt_hi(__builtin_amdgcn_read_exec_hi(), __builtin_amdgcn_mbcnt_lo(__builtin_amdgcn_read_exec_lo(), 0)) == 0) atomicAdd(&g_icp_count[slot], 1ull); } while (0)
#else
#define ICP_COUNT(slot, v) do { } while (0)
#define ICP_COUNT_WAVE(slot) do { } while (0)
#endif
void icp_counters_read(unsigned long long* out8, bool reset) {
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_icp_count), sizeof(unsigned long long) * 8);
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_icp_count), z, sizeof(z));
  }
}
// squared distance (whole steps) of the local query l (rounded to whole steps) to a packed entry (lo = x | y << 16, hi = z | index << 16).
// Every coordinate difference fits a signed 16-bit integer (the step of the cell frame is chosen for that, build_cell_lists), so the x and y
// differences are ONE v_pk_sub, the z difference the low half of a second v_pk_sub (its high half, minus the index, is never read), its
// square a v_mad_i32_i16 and the x, y squares ONE v_dot2_i32_i16 added to it: 4 instructions per candidate (the three field extractions, three subtractions
// and three 24-bit multiply-adds of the 32-bit form: 9).  The sum stays below 3 * 2^30 < 2^32.
typedef unsigned short q_v2u16 __attribute__((vector_size(4)));
typedef short q_v2i16 __attribute__((vector_size(4)));
struct Q3 {
  unsigned xy, z;  // x | y << 16, z
};
__device__ __forceinline__ unsigned q_rank(Q3 l, unsigned lo, unsigned hi) {
  const q_v2u16 dxy = __builtin_bit_cast(q_v2u16, l.xy) - __builtin_bit_cast(q_v2u16, lo);
  const unsigned dzw = __builtin_bit_cast(unsigned, __builtin_bit_cast(q_v2u16, l.z) - __builtin_bit_cast(q_v2u16, hi));
  int dz2;
  asm("v_mad_i32_i16 %0, %1, %1, 0" : "=v"(dz2) : "v"(dzw));  // (first: v_dot2c accumulates into its destination, which then needs no zero)
  return (unsigned)__builtin_amdgcn_sdot2(__builtin_bit_cast(q_v2i16, dxy), __builtin_bit_cast(q_v2i16, dxy), dz2, false);
}
#define Q_L(v) (int)(v)
#define Q_KEYF(k) (float)(k)
#define Q_INF 0xFFFFFFFFu
// one chunk of two entries: keys = squared distance with the lowest bit replaced by the slot.
// Empty slots carry coordinates 0xFFFF: on every axis the 16-bit difference to a query (28 000 .. 32 767 steps into the frame) wraps to
// >= 28 000 steps, farther than any real entry of a list can be (those lie within gate + margin + the cell diagonal), so they need no test here.
__device__ __forceinline__ void q_chunk(Q3 l, const uint4& ch, unsigned& b1, unsigned& b2, unsigned& why, unsigned& whw) {
  const unsigned k0 = q_rank(l, ch.x, ch.y) & ~1u, k1 = q_rank(l, ch.z, ch.w) | 1u;
  const unsigned prev = b1;
  b2 = umed3(b1, b2, k0);
  b1 = min(b1, k0);
  b2 = umed3(b1, b2, k1);
  b1 = min(b1, k1);
  const bool changed = b1 != prev;
  why = changed ? ch.y : why, whw = changed ? ch.w : whw;
}
// exact evaluation of the entries of a chunk that the ranking cannot separate from the winner (rare path).  Entries carry
// the Morton rank of their point (the index into pts_idx / nrm_idx); ties go to the lowest ORIGINAL index (pts_idx[].w).
__device__ __forceinline__ void q_chunk_exact(const CellListDev& c, Q3 l, const uint4& ch, float lim, int widx, const float* T, V3 q, float& best,
                                              int& bidx, int& borig, V3& moved) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const unsigned lo = e ? ch.z : ch.x, hi = e ? ch.w : ch.y;
    const int j = (int)(hi >> 16);
    if (hi >= Q_EMPTY_HI || j == widx || !(Q_KEYF(q_rank(l, lo, hi)) <= lim)) continue;
    const float4 t = c.pts_idx[j];
    const V3 tm = m4_point(T, v3(t.x, t.y, t.z));
    const float d2 = sqdist_flann(q, tm);
    const int jo = __float_as_int(t.w);
    if (d2 < best || (d2 == best && jo < borig)) best = d2, bidx = j, borig = jo, moved = tm;
  }
}

// DEFER: a lookup whose two best keys the ranking cannot separate is not re-scanned here; the caller is told (true) and
// repeats it later with DEFER = false (k_icp_fusedq<true> collects such lookups per block and runs them densely packed).
// ONE_GATHER: the cell record as one 8-byte load (keep_whole).  Only k_icp_fusedq_momm asks for it: the kernels that have run on hardware (nn_mode 4
// at 7 waves, nn_mode 6 at 4 -- the forms bench.py falls back to) have no registers for the longer life of r.x (72 VGPRs + 84 B of scratch, 128 + 48 B
// with it) and are to compile to the assembly that ran.
template <bool DEFER, bool ONE_GATHER = false>
__device__ __forceinline__ bool cells_nnq(const CellListDev& c, V3 qg, const float* T, V3 q, float& best, int& bidx, V3& moved) {
  const float hx = __builtin_fmaf(qg.x, c.inv_cell, c.gox), hy = __builtin_fmaf(qg.y, c.inv_cell, c.goy), hz = __builtin_fmaf(qg.z, c.inv_cell, c.goz);
  const float gx = floorf(hx), gy = floorf(hy), gz = floorf(hz);
  const int ix = (int)gx, iy = (int)gy, iz = (int)gz;
  if ((unsigned)ix >= (unsigned)c.dx || (unsigned)iy >= (unsigned)c.dy || (unsigned)iz >= (unsigned)c.dz) return false;
  uint2 r = c.rec[(iz * c.dy + iy) * c.dx + ix];
  if constexpr (ONE_GATHER) keep_whole(r);
  const int nch = (int)r.y;
  ICP_COUNT(0, 1);
  if (nch == 0) return false;
  // local coordinate in whole steps (q_rs carries the + 0.5 of the rounding)
  const Q3 l = {(unsigned)Q_L(__builtin_fmaf(hx - gx, c.q_cs, c.q_rs)) | ((unsigned)Q_L(__builtin_fmaf(hy - gy, c.q_cs, c.q_rs)) << 16),
                (unsigned)Q_L(__builtin_fmaf(hz - gz, c.q_cs, c.q_rs))};
  unsigned b1 = Q_INF, b2 = Q_INF;
  unsigned why = 0xFFFFFFFFu, whw = 0xFFFFFFFFu;  // high words (index) of the chunk that holds the winner
  const uint4* __restrict__ lp = c.qlist + r.x;
  // (32-bit byte offsets from the uniform base: one register and a 64-bit add less than a per-lane pointer; the lists of a
  // cloud stay far below 4 GB)
  const char* __restrict__ qbase = (const char*)c.qlist;
  unsigned qoff = r.x * 16u;
  ICP_COUNT(2, nch);
  for (int k = 0; k < nch; k += 2, qoff += 32u) {  // two chunks (four candidates, two loads in flight) per trip
    ICP_COUNT_WAVE(3);
    const uint4 ca = *(const uint4*)(qbase + qoff), cb = *(const uint4*)(qbase + qoff + 16u);  // (lists hold an even number of chunks)
    q_chunk(l, ca, b1, b2, why, whw);
    q_chunk(l, cb, b1, b2, why, whw);
  }
  ICP_COUNT(1, 1);
  const unsigned whi = (b1 & 1u) ? whw : why;
  const float f1 = Q_KEYF(b1 & ~1u) * c.q_step2, f2 = Q_KEYF(b2 & ~1u) * c.q_step2;
  const int widx = (int)(whi >> 16);
  const float4 w = c.pts_idx[widx];
  moved = m4_point(T, v3(w.x, w.y, w.z));
  best = sqdist_flann(q, moved);
  bidx = widx;
  const float mag = fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), 0.25f));
  const float delta = mag * 2.0e-6f;
  // transform rounding as in cells_nn (1.02e-4 also covers the conversion of the keys to float) + the quantisation of the
  // query and of the two candidate positions being compared (q_eq bounds their sum) + the key bit each of them gave up:
  //   2.002 s1 delta + 1.02e-4 f1 + delta^2 + 2.1 q_eq (s1 + s2) + 2.1 q_eq^2 + 2 step^2,   s1 = sqrt(f1) <= s2 = sqrt(f2).
  // Only an upper bound is needed (a larger tol defers a few more lookups to the exact scan, it never changes an answer): both roots are
  // bounded by S = f2 / (2 c) + c / 2 >= sqrt(f2) for any c > 0 (c = a third of the gate: exact at the typical runner-up distance) -- five
  // multiply-adds instead of two square roots (quarter-rate instructions) and eight operations.
  const float S = __builtin_fmaf(f2, c.q_sa, c.q_sb);
  const float tol = __builtin_fmaf(S, __builtin_fmaf(2.002f, delta, c.q_tk), __builtin_fmaf(1.02e-4f, f1, __builtin_fmaf(delta, delta, c.q_t0)));
  if (f2 - f1 <= tol) {
    if (DEFER) return true;
    ICP_COUNT(4, 1);
    ICP_COUNT_WAVE(5);
    const float lim = (f1 + tol) / c.q_step2;  // back to step units
    int borig = __float_as_int(w.w);
    for (int k = 0; k < nch; ++k) q_chunk_exact(c, l, lp[k], lim, widx, T, q, best, bidx, borig, moved);
  }
  return false;
}

// COMPOSED = false (nn_mode 3): the source point at iteration k is the chain T_k(...T_1(p0)) of the solved increments,
//   the float operations of moving a stored cloud once per iteration (what the oracle / PCL do): same bits as modes 0-2.
// COMPOSED = true (nn_mode 4, the default of the bench): one application of the accumulated transform
//   final_tf = T_k * ... * T_1 (IcpState, float products) with fused multiply-adds.  Positions differ from the chain by
//   float rounding (<= 1e-7 relative), so a correspondence at an exact tie or a residual at the gate can differ:
//   same iteration counts, poses within the tolerances tests/test_gpu_parity.py::test_icp_composed_increments states.
// The per-lane float sums of a 256-thread block added in double, NV of them: a transposition through LDS instead of NV butterfly
// reductions (74 of those cost 888 ds_bpermute + 444 v_add_f64 per wavefront -- a fifth of k_icp_fusedq_mom's time).  Sixteen sums at a
// time: every thread writes its sixteen floats to a row of the tile (stride 17: conflict-free), thread (g, k) adds the sixteen rows of
// group g for sum k in double, two shuffles join the four groups of a wavefront, a 4 x 16 table the wavefronts.  out[k] = the block's sum.
struct BlockSumLds {
  float tile[256][17];
  double part[4][16];
};
template <int NV>
__device__ __forceinline__ void block_sum_floats(const float (&acc)[NV], BlockSumLds& L, double* __restrict__ out) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int k = t & 15, g = t >> 4;
#pragma unroll
  for (int c = 0; c < (NV + 15) / 16; ++c) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 16; ++v)
      if (c * 16 + v < NV) L.tile[t][v] = acc[c * 16 + v];
    __syncthreads();
    double pd = 0.0;
    if (c * 16 + k < NV) {
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) pd += (double)L.tile[g * 16 + s2][k];
    }
    pd += __shfl_xor(pd, 16);
    pd += __shfl_xor(pd, 32);
    if (lane < 16) L.part[wave][lane] = pd;
    __syncthreads();
    if (t < 16 && c * 16 + t < NV) out[c * 16 + t] = (L.part[0][t] + L.part[1][t]) + (L.part[2][t] + L.part[3][t]);
  }
}

#ifndef HOP_ICP_W
#define HOP_ICP_W 7
#endif
template <bool F32>
struct IcpAcc {
  typedef double type;
};
template <>
struct IcpAcc<true> {
  typedef float type;
};
__device__ __forceinline__ double icp_wave_sum(double v) { return wave_sum(v); }
__device__ __forceinline__ double icp_fma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float icp_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// one source point of a hypothesis: move it, find its correspondence, gate it, add its terms to the lane's sums.
// Returns ICP_PT_DEFERRED if the lookup was deferred (DEFER only; nothing was added), ICP_PT_ACCEPTED if terms were added.
enum { ICP_PT_REJECTED = 0, ICP_PT_DEFERRED = 1, ICP_PT_ACCEPTED = 2 };
template <bool COMPOSED, bool DEFER, typename acc_t>
__device__ __forceinline__ int icp_fusedq_point(const IcpArgs& a, int i, const float* __restrict__ pose, const float* __restrict__ sTi,
                                                 const float* __restrict__ hist, const float* __restrict__ F, acc_t (&acc)[ICP_NACC]) {
  const float4 p4 = a.s_pts4[i];
  V3 q = v3(p4.x, p4.y, p4.z), qn;
  if (COMPOSED) {
    if (a.iter > 0) q = m4_point_fma(F, q);
  } else {
    const float4 n4 = a.s_nrm4[i];
    qn = v3(n4.x, n4.y, n4.z);
    icp_chain_point_normal(hist, a.iter, q, qn);
  }
  float d2 = 3.0e38f;
  int j = -1;
  V3 tq;  // the correspondence moved by the pose, as the distance was measured
  if (cells_nnq<DEFER>(a.cells, m4_point_fma(sTi, q), pose, q, d2, j, tq)) return ICP_PT_DEFERRED;
  if (j < 0 || !(d2 <= a.max_d2)) return ICP_PT_REJECTED;
  const float4 tn = a.cells.nrm_idx[j];
  if (COMPOSED) {  // the source normal only now: three registers less across the scan of the list
    const float4 n4 = a.s_nrm4[i];
    qn = v3(n4.x, n4.y, n4.z);
    if (a.iter > 0) qn = m4_dir_fma(F, qn);
  }
  const V3 nt = m4_dir(pose, v3(tn.x, tn.y, tn.z));
  if (!(vdot(qn, nt) >= a.cos_thr)) return ICP_PT_REJECTED;
  ICP_COUNT(6, 1);
  ICP_COUNT_WAVE(7);
  const V3 c = vcross(q, nt);
  const acc_t J[6] = {c.x, c.y, c.z, nt.x, nt.y, nt.z};
  const acc_t res = (acc_t)vdot(q - tq, nt);
  int k = 0;
#pragma unroll
  for (int u = 0; u < 6; ++u)
#pragma unroll
    for (int v = 0; v <= u; ++v) {
      acc[k] = icp_fma(J[u], J[v], acc[k]);
      ++k;
    }
#pragma unroll
  for (int u = 0; u < 6; ++u) acc[21 + u] = icp_fma(-J[u], res, acc[21 + u]);
  acc[27] += (acc_t)d2;
  // the count of accepted correspondences: per lane in double (nn_mode 3), by the caller per wavefront (nn_mode 4)
  if (sizeof(acc_t) == 8) acc[28] += (acc_t)1;
  acc[29] += (acc_t)q.x, acc[30] += (acc_t)q.y, acc[31] += (acc_t)q.z;
  return ICP_PT_ACCEPTED;
}
template <bool COMPOSED>
__global__ HOP_PK_F32 __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(COMPOSED ? HOP_ICP_W : 4))) void k_icp_fusedq(IcpArgs a, int R) {
  __shared__ double red[COMPOSED ? 1 : 4][ICP_NACC];
  __shared__ int n_cnt[4];
  // nn_mode 4: lookups that need the exact re-scan (0.6 % of them, but a fifth of the wavefronts held one) are queued per
  // wavefront and run afterwards densely packed, instead of every such wavefront walking its lists again for one or two lanes.
  // The queue is filled by ballot (no atomics): its order, and with it every float sum, is the same in every run.
  // (the queue and the tile of the final block sum share their bytes: the queue is drained before the sums leave the registers)
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[COMPOSED ? sizeof(BlockSumLds) : 16];
  static_assert(sizeof(BlockSumLds) >= sizeof(unsigned short) * 4 * 64 * ICP_ACCUM_R, "the queue fits the tile");
  unsigned short(*defer_i)[64 * ICP_ACCUM_R] = reinterpret_cast<unsigned short(*)[64 * ICP_ACCUM_R]>(lds_raw);
  BlockSumLds& bs = *reinterpret_cast<BlockSumLds*>(lds_raw);
  const int hl = blockIdx.y, h = a.h0 + hl;
  const IcpState& st = a.state[hl];
  if (!st.active) return;
  const float* __restrict__ pose = a.pose + (size_t)h * 16;
  const float* __restrict__ sTi = a.pose_inv + (size_t)h * 12;
  const float* __restrict__ hist = a.hist + (size_t)hl * a.max_iter * 12;
  const float* __restrict__ F = st.final_tf;
  // nn_mode 4 keeps the 32 per-lane sums in float (a lane adds at most ~30 terms; lanes, waves and blocks are then summed in
  // double as before): half the accumulator registers (71 instead of 107 VGPRs: 7 instead of 4 waves per SIMD).  nn_mode 3
  // accumulates in double like the oracle.
  typedef typename IcpAcc<COMPOSED>::type acc_t;
  acc_t acc[ICP_NACC];
#pragma unroll
  for (int k = 0; k < ICP_NACC; ++k) acc[k] = 0;
  int n_wave = 0, n_def = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = blockIdx.x * (256 * R);
  for (int r = 0; r < R; ++r) {
    const int li = r * 256 + threadIdx.x, i = base + li;
    if (i >= a.ns) continue;
    const int res = icp_fusedq_point<COMPOSED, COMPOSED, acc_t>(a, i, pose, sTi, hist, F, acc);
    // (lane 0 of a wavefront is among the lanes that get here whenever any of them does: its copies are the wavefront's counts)
    if (COMPOSED) {
      const unsigned long long dm = __ballot(res == ICP_PT_DEFERRED);
      if (res == ICP_PT_DEFERRED) defer_i[wave][n_def + __popcll(dm & ((1ull << lane) - 1ull))] = (unsigned short)li;
      n_def += __popcll(dm);
      n_wave += __popcll(__ballot(res == ICP_PT_ACCEPTED));
    }
  }
  if (COMPOSED) {
    const int nd = __builtin_amdgcn_readfirstlane(n_def);
    for (int t = lane; t < nd; t += 64)
      n_wave += __popcll(__ballot(icp_fusedq_point<COMPOSED, false, acc_t>(a, base + defer_i[wave][t], pose, sTi, hist, F, acc) == ICP_PT_ACCEPTED));
  }
  if constexpr (COMPOSED) {  // float per-lane sums: the block sum through LDS (block_sum_floats below the moment kernel's notes)
    double* __restrict__ out = a.partial + ((size_t)hl * gridDim.x + blockIdx.x) * ICP_NACC;
    if (lane == 0) n_cnt[wave] = n_wave;
    block_sum_floats<ICP_NACC>(acc, bs, out);
    __syncthreads();
    if (threadIdx.x == 0) out[28] = (double)((n_cnt[0] + n_cnt[1]) + (n_cnt[2] + n_cnt[3]));
  } else {
#pragma unroll
    for (int k = 0; k < ICP_NACC; ++k) {
      const double s = icp_wave_sum(acc[k]);
      if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < ICP_NACC) {
      const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
      a.partial[((size_t)hl * gridDim.x + blockIdx.x) * ICP_NACC + threadIdx.x] = s;
    }
  }
}
template __global__ void k_icp_fusedq<false>(IcpArgs, int);
template __global__ void k_icp_fusedq<true>(IcpArgs, int);

// ------------------------------------------------------------------------------------------------
// nn_mode 6: the reference's Levenberg-Marquardt minimiser from ONE pass per ICP iteration.
// PCL's residual f_i(x) = n_i . (R(x) p_i + t - q_i) (transformation_estimation_point_to_plane.h:77-84) is linear in the 12 entries
// of [R | t]: with u_i = (n_a p'_b [9], n_a [3], r0_i), p' = p - c (c = the hypothesis' translation, a point inside the object:
// conditioning only), r0_i = n_i . (p_i - q_i) and w(x) = (R - I [9], t + (R - I) c [3], 1),  f_i(x) = w(x) . u_i.  So the sum of
// squares, J^T J and J^T f of EVERY parameter vector the minimiser evaluates -- including NumericalDiff's six forward-difference
// columns -- are quadratic forms in the 13 x 13 moment matrix M = sum_i u_i u_i^T, which this kernel accumulates next to the
// lookups (73 distinct sums: (n_a n_c)(p_b p_d) is symmetric in both index pairs).  k_icp_lm6_solve (hop_icp_lm.hip) then runs
// Eigen's minimizeOneStep / lmpar2 loop to its stopping rule per hypothesis without touching the points again.
// This evaluates the reference's algorithm in exact arithmetic where PCL rounds every residual to float; the difference is the
// rounding noise of that float run, which two builds of the reference differ by as well (profiles/r03_icp_lm_deltas.json:
// closer to the reference's default build than its -march=native build is).  Per-lane sums in float over <= ~30 terms of equal
// magnitude (1e-6 relative; the reference's forward-difference Jacobian carries 1e-3), lanes / waves / blocks in double.
// Gates as PCL writes them (strict normal test against the double threshold, double distance gate: see hop_icp_refine).
// ------------------------------------------------------------------------------------------------
template <bool DEFER>
__device__ __forceinline__ int icp_fusedq_point_mom(const IcpArgs& a, int i, const float* __restrict__ pose, const float* __restrict__ sTi,
                                                     const float* __restrict__ F, V3 ctr, float (&acc)[ICP_NMOM]) {
  const float4 p4 = a.s_pts4[i];
  V3 q = v3(p4.x, p4.y, p4.z);
  if (a.iter > 0) q = m4_point_fma(F, q);
  float d2 = 3.0e38f;
  int j = -1;
  V3 tq;
  if (cells_nnq<DEFER>(a.cells, m4_point_fma(sTi, q), pose, q, d2, j, tq)) return ICP_PT_DEFERRED;
  if (j < 0 || !(d2 <= a.max_d2)) return ICP_PT_REJECTED;
  const float4 tn = a.cells.nrm_idx[j];
  const float4 n4 = a.s_nrm4[i];
  V3 qn = v3(n4.x, n4.y, n4.z);
  if (a.iter > 0) qn = m4_dir_fma(F, qn);
  const V3 nt = m4_dir(pose, v3(tn.x, tn.y, tn.z));
  if (!(((qn.x * nt.x + qn.y * nt.y) + qn.z * nt.z) > a.cos_thr)) return ICP_PT_REJECTED;
  const V3 pc = q - ctr;
  const float r0 = vdot(q - tq, nt);
  const float nn[6] = {nt.x * nt.x, nt.x * nt.y, nt.x * nt.z, nt.y * nt.y, nt.y * nt.z, nt.z * nt.z};
  const float pp[6] = {pc.x * pc.x, pc.x * pc.y, pc.x * pc.z, pc.y * pc.y, pc.y * pc.z, pc.z * pc.z};
  const float pv[3] = {pc.x, pc.y, pc.z}, nv[3] = {nt.x, nt.y, nt.z};
#pragma unroll
  for (int u = 0; u < 6; ++u) {
#pragma unroll
    for (int v = 0; v < 6; ++v) acc[u * 6 + v] = __builtin_fmaf(nn[u], pp[v], acc[u * 6 + v]);
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[36 + u * 3 + b] = __builtin_fmaf(nn[u], pv[b], acc[36 + u * 3 + b]);
    acc[54 + u] += nn[u];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float nr = nv[c] * r0;
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[60 + c * 3 + b] = __builtin_fmaf(nr, pv[b], acc[60 + c * 3 + b]);
    acc[69 + c] += nr;
  }
  acc[72] = __builtin_fmaf(r0, r0, acc[72]);
  acc[73] += d2;
  return ICP_PT_ACCEPTED;
}
#ifndef HOP_ICP_MOM_W
#define HOP_ICP_MOM_W 4
#endif
__global__ HOP_PK_F32 __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HOP_ICP_MOM_W))) void k_icp_fusedq_mom(IcpArgs a, int R) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[sizeof(BlockSumLds)];  // the queue, then the tile of the block sum
  __shared__ int n_cnt[4];
  unsigned short(*defer_i)[64 * ICP_ACCUM_R] = reinterpret_cast<unsigned short(*)[64 * ICP_ACCUM_R]>(lds_raw);
  BlockSumLds& bs = *reinterpret_cast<BlockSumLds*>(lds_raw);
  const int hl = blockIdx.y, h = a.h0 + hl;
  const IcpState& st = a.state[hl];
  if (!st.active) return;
  const float* __restrict__ pose = a.pose + (size_t)h * 16;
  const float* __restrict__ sTi = a.pose_inv + (size_t)h * 12;
  const float* __restrict__ F = st.final_tf;
  const V3 ctr = v3(pose[3], pose[7], pose[11]);
  float acc[ICP_NMOM];
#pragma unroll
  for (int k = 0; k < ICP_NMOM; ++k) acc[k] = 0.f;
  int n_wave = 0, n_def = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = blockIdx.x * (256 * R);
  for (int r = 0; r < R; ++r) {
    const int li = r * 256 + threadIdx.x, i = base + li;
    if (i >= a.ns) continue;
    const int res = icp_fusedq_point_mom<true>(a, i, pose, sTi, F, ctr, acc);
    const unsigned long long dm = __ballot(res == ICP_PT_DEFERRED);
    if (res == ICP_PT_DEFERRED) defer_i[wave][n_def + __popcll(dm & ((1ull << lane) - 1ull))] = (unsigned short)li;
    n_def += __popcll(dm);
    n_wave += __popcll(__ballot(res == ICP_PT_ACCEPTED));
  }
  const int nd = __builtin_amdgcn_readfirstlane(n_def);
  for (int t = lane; t < nd; t += 64)
    n_wave += __popcll(__ballot(icp_fusedq_point_mom<false>(a, base + defer_i[wave][t], pose, sTi, F, ctr, acc) == ICP_PT_ACCEPTED));
  double* __restrict__ out = a.partial + ((size_t)hl * gridDim.x + blockIdx.x) * ICP_NMOM_STRIDE;
  if (lane == 0) n_cnt[wave] = n_wave;
  block_sum_floats<ICP_NMOM>(acc, bs, out);  // (its barriers order n_cnt as well)
  if (threadIdx.x == 0) out[ICP_NMOM] = (double)((n_cnt[0] + n_cnt[1]) + (n_cnt[2] + n_cnt[3]));
}
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// nn_mode 7: the moment form with INTEGER-EXACT sums -- the result no longer depends on which lane, wavefront or workgroup adds which
// correspondence, so the CPU statement of the algorithm (oracle minimiser 7) returns the same bits.
// The 13 components of u = (n_a p'_b, n_a, r0) are put on a power-of-two grid chosen from the model's radius and the gate (IcpArgs::mom_*,
// hop_icp_refine): U = the EXACT value of n_a (p'_b 2^k) (or n_a 2^12, r0 2^k_r) rounded to the nearest integer, ties to even, clamped to
// +-2^ICP_MOM_BITS.  One v_fma_f32 does the product and the rounding: x y + 1.5 2^23 is rounded ONCE, to a float whose unit is 1 (|x y| < 2^22),
// so its mantissa bits are the integer (the oracle: nearbyint of the product formed exactly in double).
// M = sum U U^T (91 entries, lower triangle): a lane handles its points two at a time, packs the two vectors into 16-bit halves
// (v_cvt_pk_i16_i32) and adds both outer products with 91 v_dot2_i32_i16 -- products <= 2^24, a lane adds at most 2 ICP_ACCUM_R = 64 of them in
// 32 bits -- lanes / wavefronts / workgroups are then added in 64 bits.  The squared correspondence distances of the MSE stop rule go the
// same way.  Lookups, gates and the deferred-lookup queue are those of k_icp_fusedq_mom.
// Per accepted correspondence (ISA count): 3 scalings + 13 x (fma, sub, med3) + 6.5 packs + 45.5 dot2 = 94 vector instructions; the float form
// of nn_mode 6 needs ~60 (its 73 FMAs go two to a v_pk_fma_f32); the first integer form of this round (one point at a time, mul / rndne / med3 /
// cvt, 91 v_mad_i32_i24) needed 152.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int momi_q(float v, float s, float lim) {  // rint(v s) for values beyond the fma form's range (the squared distance)
  const float t = __builtin_rintf(v * s);  // s is a power of two: the product is exact
  return (int)__builtin_amdgcn_fmed3f(t, -lim, lim);
}
constexpr float MOMI_MAGIC = 12582912.0f;      // 1.5 * 2^23: floats in [2^23, 2^24) have unit 1
constexpr int MOMI_MAGIC_BITS = 0x4B400000;
__device__ __forceinline__ int momi_qp(float x, float y, int lim) {  // nearest integer (ties to even) of the exact product x y, |x y| < 2^22; clamped
  const float t = __builtin_fmaf(x, y, MOMI_MAGIC);
  const int v = __float_as_int(t) - MOMI_MAGIC_BITS;
  return min(max(v, -lim), lim);
}
// two signed 16-bit integers in one register / the sum of the two products of the halves added to c
typedef short momi_v2i16 __attribute__((vector_size(4)));
__device__ __forceinline__ unsigned momi_pack(int a, int b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(a, b)); }  // (|a|, |b| <= 4096: no saturation)
__device__ __forceinline__ int momi_dot2(unsigned a, unsigned b, int c) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(momi_v2i16, a), __builtin_bit_cast(momi_v2i16, b), c, false);
}
// one source point of a hypothesis: lookup, PCL's two gates, and -- if accepted -- its gridded u in U[13] and its gridded squared distance in dq
// (both left untouched otherwise: the caller zeroes them)
template <bool DEFER>
__device__ __forceinline__ int icp_fusedq_point_momi(const IcpArgs& a, int i, const float* __restrict__ pose, const float* __restrict__ sTi,
                                                      const float* __restrict__ F, V3 ctr, int (&U)[13], int& dq) {
  const float4 p4 = a.s_pts4[i];
  V3 q = v3(p4.x, p4.y, p4.z);
  if (a.iter > 0) q = m4_point_fma(F, q);
  float d2 = 3.0e38f;
  int j = -1;
  V3 tq;
  if (cells_nnq<DEFER>(a.cells, m4_point_fma(sTi, q), pose, q, d2, j, tq)) return ICP_PT_DEFERRED;
  if (j < 0 || !(d2 <= a.max_d2)) return ICP_PT_REJECTED;
  const float4 tn = a.cells.nrm_idx[j];
  const float4 n4 = a.s_nrm4[i];
  V3 qn = v3(n4.x, n4.y, n4.z);
  if (a.iter > 0) qn = m4_dir_fma(F, qn);
  const V3 nt = m4_dir(pose, v3(tn.x, tn.y, tn.z));
  if (!(((qn.x * nt.x + qn.y * nt.y) + qn.z * nt.z) > a.cos_thr)) return ICP_PT_REJECTED;
  const V3 pc = q - ctr;
  const float r0 = vdot(q - tq, nt);
  const float ps[3] = {pc.x * a.mom_s_np, pc.y * a.mom_s_np, pc.z * a.mom_s_np};  // (powers of two: exact)
  const float nv[3] = {nt.x, nt.y, nt.z};
  const int lim = 1 << ICP_MOM_BITS;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int b = 0; b < 3; ++b) U[3 * c + b] = momi_qp(nv[c], ps[b], lim);
    U[9 + c] = momi_qp(nv[c], a.mom_s_n, lim);
  }
  U[12] = momi_qp(r0, a.mom_s_r, lim);
  dq = momi_q(d2, a.mom_s_d, a.mom_lim_d);
  return ICP_PT_ACCEPTED;
}
// the outer products of two gridded vectors (either may be zero: a point that was not accepted) added to the lane's 91 sums
__device__ __forceinline__ void momi_accumulate_pair(const int (&UA)[13], const int (&UB)[13], int (&acc)[ICP_NMOMI]) {
  unsigned P[13];
#pragma unroll
  for (int u = 0; u < 13; ++u) P[u] = momi_pack(UA[u], UB[u]);
  int k = 0;
#pragma unroll
  for (int u = 0; u < 13; ++u)
#pragma unroll
    for (int v = 0; v <= u; ++v) {
      acc[k] = momi_dot2(P[u], P[v], acc[k]);
      ++k;
    }
}
// the per-lane 32-bit sums of a 256-thread block added in 64 bits: block_sum_floats with integers (same tile, same steps)
template <int NV>
__device__ __forceinline__ void block_sum_ints(const int (&acc)[NV], BlockSumLds& L, long long* __restrict__ out) {
  int(*tile)[17] = reinterpret_cast<int(*)[17]>(L.tile);
  long long(*part)[16] = reinterpret_cast<long long(*)[16]>(L.part);
  static_assert(sizeof(L.tile) == sizeof(int) * 256 * 17 && sizeof(L.part) == sizeof(long long) * 4 * 16, "same bytes");
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int k = t & 15, g = t >> 4;
#pragma unroll
  for (int c = 0; c < (NV + 15) / 16; ++c) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 16; ++v)
      if (c * 16 + v < NV) tile[t][v] = acc[c * 16 + v];
    __syncthreads();
    long long pd = 0;
    if (c * 16 + k < NV) {
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) pd += (long long)tile[g * 16 + s2][k];
    }
    pd += __shfl_xor(pd, 16);
    pd += __shfl_xor(pd, 32);
    if (lane < 16) part[wave][lane] = pd;
    __syncthreads();
    if (t < 16 && c * 16 + t < NV) out[c * 16 + t] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
  }
}
#ifndef HOP_ICP_MOMI_W
#define HOP_ICP_MOMI_W 3
#endif
__global__ HOP_PK_F32 __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HOP_ICP_MOMI_W))) void k_icp_fusedq_momi(IcpArgs a, int R) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[sizeof(BlockSumLds)];  // the queue, then the tile of the block sum
  __shared__ int n_cnt[4];
  unsigned short(*defer_i)[64 * ICP_ACCUM_R] = reinterpret_cast<unsigned short(*)[64 * ICP_ACCUM_R]>(lds_raw);
  BlockSumLds& bs = *reinterpret_cast<BlockSumLds*>(lds_raw);
  const int hl = blockIdx.y, h = a.h0 + hl;
  const IcpState& st = a.state[hl];
  if (!st.active) return;
  const float* __restrict__ pose = a.pose + (size_t)h * 16;
  const float* __restrict__ sTi = a.pose_inv + (size_t)h * 12;
  const float* __restrict__ F = st.final_tf;
  const V3 ctr = v3(pose[3], pose[7], pose[11]);
  int acc[ICP_NMOMI];
#pragma unroll
  for (int k = 0; k < ICP_NMOMI; ++k) acc[k] = 0;
  int n_wave = 0, n_def = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = blockIdx.x * (256 * R);
  // the lane's points two at a time (trips r and r + 1): one lookup after the other, both outer products in one pass of 91 v_dot2.
  // Every lane of the wavefront makes every trip (a point past the end of the cloud counts as rejected): the ballots see all 64 lanes.
  for (int r = 0; r < R; r += 2) {
    int UA[13], UB[13], dA = 0, dB = 0;
#pragma unroll
    for (int u = 0; u < 13; ++u) UA[u] = 0, UB[u] = 0;
    int res[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int li = (r + half) * 256 + threadIdx.x, i = base + li;
      int rs = ICP_PT_REJECTED;
      if (r + half < R && i < a.ns) rs = half ? icp_fusedq_point_momi<true>(a, i, pose, sTi, F, ctr, UB, dB) : icp_fusedq_point_momi<true>(a, i, pose, sTi, F, ctr, UA, dA);
      const unsigned long long dm = __ballot(rs == ICP_PT_DEFERRED);
      if (rs == ICP_PT_DEFERRED) defer_i[wave][n_def + __popcll(dm & ((1ull << lane) - 1ull))] = (unsigned short)li;
      n_def += __popcll(dm);
      n_wave += __popcll(__ballot(rs == ICP_PT_ACCEPTED));
      res[half] = rs;
    }
    if (res[0] == ICP_PT_ACCEPTED || res[1] == ICP_PT_ACCEPTED) {
      momi_accumulate_pair(UA, UB, acc);
      acc[91] += dA + dB;
    }
  }
  const int nd = __builtin_amdgcn_readfirstlane(n_def);
  for (int t0 = 0; t0 < nd; t0 += 128) {  // the deferred lookups densely packed, again two to a lane
    int UA[13], UB[13], dA = 0, dB = 0;
#pragma unroll
    for (int u = 0; u < 13; ++u) UA[u] = 0, UB[u] = 0;
    int res[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int t = t0 + half * 64 + lane;
      int rs = ICP_PT_REJECTED;
      if (t < nd) rs = half ? icp_fusedq_point_momi<false>(a, base + defer_i[wave][t], pose, sTi, F, ctr, UB, dB) : icp_fusedq_point_momi<false>(a, base + defer_i[wave][t], pose, sTi, F, ctr, UA, dA);
      n_wave += __popcll(__ballot(rs == ICP_PT_ACCEPTED));
      res[half] = rs;
    }
    if (res[0] == ICP_PT_ACCEPTED || res[1] == ICP_PT_ACCEPTED) {
      momi_accumulate_pair(UA, UB, acc);
      acc[91] += dA + dB;
    }
  }
  long long* __restrict__ out = reinterpret_cast<long long*>(a.partial) + ((size_t)hl * gridDim.x + blockIdx.x) * ICP_NMOMI_STRIDE;
  if (lane == 0) n_cnt[wave] = n_wave;
  block_sum_ints<ICP_NMOMI>(acc, bs, out);  // (its barriers order n_cnt as well)
  if (threadIdx.x == 0) out[ICP_NMOMI] = (long long)((n_cnt[0] + n_cnt[1]) + (n_cnt[2] + n_cnt[3]));
}
void launch_icp_fusedq_momi(const IcpArgs& a, int hb, hipStream_t s) {
  const int nb = icp_blocks_per_hyp(a.ns, true);
  const int R = (a.ns + 256 * nb - 1) / (256 * nb);
  static_assert(2 * ICP_ACCUM_R * (1 << (2 * ICP_MOM_BITS)) < (1ll << 31), "a lane's 32-bit sums cannot overflow");
  hipLaunchKernelGGL(k_icp_fusedq_momi, dim3(nb, hb), dim3(256), 0, s, a, R);
}

// ------------------------------------------------------------------------------------------------
// nn_mode 7 with the moment matrix ON THE MATRIX CORES (k_icp_fusedq_momm; hop_icp_refine takes it unless HOP_ICP_MFMA=0): the same integers as
// k_icp_fusedq_momi -- M = sum U U^T is a contraction over the correspondences (PCL's TransformationEstimationPointToPlane, Utils.cpp:201-202,
// in the moment form), and with 13-bit integer operands it is an exact one for v_mfma_i32_16x16x64_i8: K = 64 correspondences per
// instruction, the 13 components fill 13 of the 16 rows.
//   * the split.  U = 256 H + L with L in [-128, 127], H = (U + 128) >> 8 in [-16, 16]: both signed bytes.  The gridding fma adds the 128 with
//     its magic constant: t = fma(x, y, 1.5 2^23 + 128) is rounded ONCE to a float of unit 1 (the same nearest-even integer of the exact
//     product as momi_qp: the constant is even), v_med3_f32 clamps it to +-2^12 about the constant, and then the low 16 bits of t's encoding
//     are W = U + 128 -- its high byte IS H, its low byte L + 128 = L ^ 0x80.  Two instructions per component, no integer arithmetic.
//   * the transposition.  The MFMA wants, in lane (i, kb), component i of 16 correspondences; the lookups produce one correspondence per
//     lane.  Each accepted lane writes its 13 half-words to the wavefront's own ring in LDS, component-major (13 ds_write_b16), at a slot
//     counted by ballot -- the ring packs the ACCEPTED correspondences densely, whatever trips or deferred batches they come from (the sums
//     are integers: any order).  Whenever 64 slots are complete every lane reads its component's 16 half-words (2 ds_read_b128),
//     de-interleaves them (8 byte gathers: v_perm_b32, + 4 v_xor) into the H and L operands, and three MFMAs add H H^T, H L^T and L L^T.
//   * the sums.  |H| <= 16, |L| <= 128: a wavefront may add 2^31 / 2^14 = 131 072 correspondences per tile entry in 32 bits (it adds at most
//     64 ICP_ACCUM_R = 2 048).  12 accumulator registers replace momi's 92 per-lane sums and their block transposition; the workgroup adds its
//     four wavefronts' tiles and recombines M = 65536 HH + 256 (HL + HL^T) + LL in 64 bits, into the layout k_icp_lm7_solve reads.
// Per accepted correspondence: 3 scalings + 13 x (fma, med3) + 13 LDS writes; per 64 of them: 2 LDS reads, 12 VALU, 3 MFMA (the form above:
// 94 VALU each).  The deferred lookups are drained in the loop as dense batches of 64 (queue of 128 instead of 64 R entries per wavefront).
// ------------------------------------------------------------------------------------------------
typedef int momm_i32x4 __attribute__((vector_size(16)));
constexpr int MOMM_ROW = 144;  // half-words per component row of a wavefront's ring: 128 slots + 16 (row stride 72 words: the 32-byte reads of the 16 rows spread over the banks)
constexpr float MOMM_MAGIC = 12583040.0f;                                  // 1.5 * 2^23 + 128
constexpr float MOMM_LO = MOMM_MAGIC - (float)(1 << ICP_MOM_BITS), MOMM_HI = MOMM_MAGIC + (float)(1 << ICP_MOM_BITS);
constexpr unsigned short MOMM_ZERO = 0x0080;                               // U = 0 in the ring: low byte L + 128, high byte H
static_assert(ICP_MOM_BITS == 12, "the byte split U = 256 H + L, |H| <= 16, is written for 13-bit integers");
struct MommLds {
  unsigned short ring[4][13][MOMM_ROW];  // (after the last flush: the four wavefronts' three tiles, 4 x 3 x 256 ints)
  unsigned short defer_i[4][128];
  long long dsum[4];
  int n_cnt[4];
  __attribute__((aligned(16))) float tf[3][12];  // pose, inverse pose, accumulated ICP transform of the workgroup's hypothesis (rows of 4)
};
static_assert(sizeof(unsigned short) * 4 * 13 * MOMM_ROW >= sizeof(int) * 4 * 3 * 256, "the tiles fit the rings");
struct MommTab {
  unsigned char u[ICP_NMOMI - 1], v[ICP_NMOMI - 1];  // entry k of the packed lower triangle is M[u][v]
  constexpr MommTab() : u{}, v{} {
    int k = 0;
    for (int a = 0; a < 13; ++a)
      for (int b = 0; b <= a; ++b) u[k] = (unsigned char)a, v[k] = (unsigned char)b, ++k;
  }
};
__constant__ MommTab c_momm_tab = MommTab();
__device__ __forceinline__ float momm_qp(float x, float y) {  // the float whose low 16 encoding bits are rint(x y) (clamped to +-2^12) + 128
  return __builtin_amdgcn_fmed3f(__builtin_fmaf(x, y, MOMM_MAGIC), MOMM_LO, MOMM_HI);
}
// lookup, PCL's two gates and -- if accepted -- the ingredients of u (target normal, scaled p', residual: momm_push grids the 13 components on
// their way to the ring, no 13-register vector is ever live) and the gridded squared distance in dq
struct MommU {
  V3 n, ps;
  float r0, s_n, s_r;
  // component C of the gridded vector in the magic form (momm_qp): C = 3 a + b -> n_a p'_b, 9 + a -> n_a, 12 -> r0
  template <int C>
  __device__ __forceinline__ float get() const {
    const float nv[3] = {n.x, n.y, n.z}, pv[3] = {ps.x, ps.y, ps.z};
    if (C < 9) return momm_qp(nv[C / 3], pv[C % 3]);
    if (C < 12) return momm_qp(nv[C - 9 < 0 ? 0 : C - 9], s_n);
    return momm_qp(r0, s_r);
  }
};
template <bool DEFER>
__device__ __forceinline__ int icp_fusedq_point_momm(const IcpArgs& a, int i, const float* __restrict__ pose, const float* __restrict__ sTi,
                                                      const float* __restrict__ F, V3 ctr, MommU& u, int& dq) {
  const float4 p4 = a.s_pts4[i];
  V3 q = v3(p4.x, p4.y, p4.z);
  if (a.iter > 0) q = m4_point_fma(F, q);
  float d2 = 3.0e38f;
  int j = -1;
  V3 tq;
  if (cells_nnq<DEFER, true>(a.cells, m4_point_fma(sTi, q), pose, q, d2, j, tq)) return ICP_PT_DEFERRED;
  if (j < 0 || !(d2 <= a.max_d2)) return ICP_PT_REJECTED;
  const float4 tn = a.cells.nrm_idx[j];
  const float4 n4 = a.s_nrm4[i];
  V3 qn = v3(n4.x, n4.y, n4.z);
  if (a.iter > 0) qn = m4_dir_fma(F, qn);
  const V3 nt = m4_dir(pose, v3(tn.x, tn.y, tn.z));
  if (!(((qn.x * nt.x + qn.y * nt.y) + qn.z * nt.z) > a.cos_thr)) return ICP_PT_REJECTED;
  const V3 pc = q - v3(pose[3], pose[7], pose[11]);  // (= ctr, read from the LDS copy where it is used: three registers less across the lookup -- what lets the kernel run 6 waves per SIMD without scratch)
  const float r0 = vdot(q - tq, nt);
  u.n = nt, u.ps = v3(pc.x * a.mom_s_np, pc.y * a.mom_s_np, pc.z * a.mom_s_np), u.r0 = r0, u.s_n = a.mom_s_n, u.s_r = a.mom_s_r;  // (powers of two: exact)
  dq = momi_q(d2, a.mom_s_d, a.mom_lim_d);
  return ICP_PT_ACCEPTED;
}
template <int C, class UT>
__device__ __forceinline__ void momm_store(unsigned short (*__restrict__ ring)[MOMM_ROW], int pos, const UT& u) {
  ring[C][pos] = (unsigned short)__float_as_uint(u.template get<C>());
  if constexpr (C + 1 < 13) momm_store<C + 1>(ring, pos, u);
}
// bytes ODD, ODD + 2 of `lo` then of `hi` as one word: the H (ODD = 1) or L ^ 0x80 (ODD = 0) bytes of four consecutive ring half-words.
// One v_perm_b32 (selector bytes 0..3 address S1 = lo, 4..7 address S0 = hi); the shift-and-mask statement of the same costs 5 instructions,
// and the read-out runs once per 64 accepted correspondences of every wavefront.  hop_icp_refine checks the whole read-out -- ring layout,
// these gathers, the MFMA operand layout, the tile map -- on the device before it first uses this kernel (k_dev_selftest_momm).
template <int ODD>
__device__ __forceinline__ unsigned momm_bytes(unsigned lo, unsigned hi) {
  return __builtin_amdgcn_perm(hi, lo, ODD ? 0x07050301u : 0x06040200u);
}
// 64 complete slots of the ring (half 0 or 1) -> operands -> three MFMAs.  Every lane of the wavefront is here (v_mfma ignores EXEC).
__device__ __forceinline__ void momm_flush(const unsigned short (*__restrict__ ring)[MOMM_ROW], int lane, int half, momm_i32x4& HH, momm_i32x4& HL,
                                           momm_i32x4& LL) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int row = min(lane & 15, 12);  // (tile rows 13..15 are never read back: those lanes repeat row 12)
  const uint4* __restrict__ p = reinterpret_cast<const uint4*>(&ring[row][half * 64 + (lane >> 4) * 16]);
  const uint4 w0 = p[0], w1 = p[1];  // 16 half-words (L ^ 0x80 | H << 8) = this component of 16 correspondences
  momm_i32x4 Hv, Lv;
  Hv[0] = (int)momm_bytes<1>(w0.x, w0.y), Lv[0] = (int)(momm_bytes<0>(w0.x, w0.y) ^ 0x80808080u);
  Hv[1] = (int)momm_bytes<1>(w0.z, w0.w), Lv[1] = (int)(momm_bytes<0>(w0.z, w0.w) ^ 0x80808080u);
  Hv[2] = (int)momm_bytes<1>(w1.x, w1.y), Lv[2] = (int)(momm_bytes<0>(w1.x, w1.y) ^ 0x80808080u);
  Hv[3] = (int)momm_bytes<1>(w1.z, w1.w), Lv[3] = (int)(momm_bytes<0>(w1.z, w1.w) ^ 0x80808080u);
  HH = __builtin_amdgcn_mfma_i32_16x16x64_i8(Hv, Hv, HH, 0, 0, 0);
  HL = __builtin_amdgcn_mfma_i32_16x16x64_i8(Hv, Lv, HL, 0, 0, 0);
  LL = __builtin_amdgcn_mfma_i32_16x16x64_i8(Lv, Lv, LL, 0, 0, 0);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (later writes to this half stay behind these reads)
  __builtin_amdgcn_wave_barrier();
}
// the accepted lanes of one batch of lookups appended to the ring (fill: slots in use, 0..127, wave-uniform); a half that becomes complete is flushed
template <class UT>
__device__ __forceinline__ void momm_push(unsigned short (*__restrict__ ring)[MOMM_ROW], int lane, bool ok, const UT& u, int& fill, momm_i32x4& HH,
                                          momm_i32x4& HL, momm_i32x4& LL) {
  const unsigned long long m = __ballot(ok);
  if (m == 0ull) return;
  if (ok) momm_store<0>(ring, (fill + __popcll(m & ((1ull << lane) - 1ull))) & 127, u);
  const int nf = fill + __popcll(m);
  if ((fill ^ nf) & 64) momm_flush(ring, lane, (fill >> 6) & 1, HH, HL, LL);
  fill = nf & 127;
}
#ifndef HOP_ICP_MOMM_W
#define HOP_ICP_MOMM_W 6  // 80 VGPRs, no scratch (round 6; 5 waves before: 90-94 VGPRs).  tools/gpu_round_check.sh times 5 and 8 beside it
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HOP_ICP_MOMM_W))) void k_icp_fusedq_momm(IcpArgs a, int R) {
  __shared__ __attribute__((aligned(16))) MommLds L;
  const int hl = blockIdx.y, h = a.h0 + hl;
  const IcpState& st = a.state[hl];
  if (!st.active) return;
  // the three 3 x 4 transforms of this workgroup's hypothesis, staged in LDS once: every lane of every trip reads the same 48-byte rows, and the
  // fences of the ring and of the deferred queue make the compiler re-load them from global memory at each use (15 uniform vector loads per trip
  // through the L1 tag pipe, next to the trip's gathers); a uniform ds_read_b128 is a broadcast and leaves the tags alone
  if (threadIdx.x < 36) {
    const int m = threadIdx.x / 12, k = threadIdx.x % 12;
    L.tf[m][k] = m == 0 ? a.pose[(size_t)h * 16 + k] : m == 1 ? a.pose_inv[(size_t)h * 12 + k] : st.final_tf[k];
  }
  __syncthreads();
  const float* __restrict__ pose = L.tf[0];
  const float* __restrict__ sTi = L.tf[1];
  const float* __restrict__ F = L.tf[2];
  const V3 ctr = v3(pose[3], pose[7], pose[11]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned short(*__restrict__ ring)[MOMM_ROW] = L.ring[wave];
  unsigned short* __restrict__ defer_i = L.defer_i[wave];
  momm_i32x4 HH = {0, 0, 0, 0}, HL = {0, 0, 0, 0}, LL = {0, 0, 0, 0};
  int dsum = 0, n_wave = 0, n_def = 0, fill = 0;
  const int base = blockIdx.x * (256 * R);
  // every lane of the wavefront makes every trip (a point past the end of the cloud counts as rejected): ballots, pushes and MFMAs see all 64 lanes
  for (int r = 0; r < R; ++r) {
    const int li = r * 256 + threadIdx.x, i = base + li;
    MommU u;
    int dq = 0;
    const int res = i < a.ns ? icp_fusedq_point_momm<true>(a, i, pose, sTi, F, ctr, u, dq) : ICP_PT_REJECTED;
    const unsigned long long dm = __ballot(res == ICP_PT_DEFERRED);
    if (res == ICP_PT_DEFERRED) defer_i[n_def + __popcll(dm & ((1ull << lane) - 1ull))] = (unsigned short)li;
    n_def += __popcll(dm);
    n_wave += __popcll(__ballot(res == ICP_PT_ACCEPTED));
    dsum += dq;
    momm_push(ring, lane, res == ICP_PT_ACCEPTED, u, fill, HH, HL, LL);
    while (n_def >= 64 || (r == R - 1 && n_def > 0)) {  // a dense batch of the queued lookups (wave-uniform condition; the queue never holds 128)
      const int take = n_def < 64 ? n_def : 64;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int mine = lane < take ? (int)defer_i[lane] : -1;
      const int rest = lane + 64 < n_def ? (int)defer_i[lane + 64] : -1;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (rest >= 0) defer_i[lane] = (unsigned short)rest;
      n_def -= take;
      MommU u2;
      int dq2 = 0;
      const int res2 = mine >= 0 ? icp_fusedq_point_momm<false>(a, base + mine, pose, sTi, F, ctr, u2, dq2) : ICP_PT_REJECTED;
      n_wave += __popcll(__ballot(res2 == ICP_PT_ACCEPTED));
      dsum += dq2;
      momm_push(ring, lane, res2 == ICP_PT_ACCEPTED, u2, fill, HH, HL, LL);
    }
  }
  if (fill & 63) {  // the incomplete half: its free slots as zeros
    const int half = (fill >> 6) & 1;
    if (lane >= (fill & 63)) {
#pragma unroll
      for (int c = 0; c < 13; ++c) ring[c][half * 64 + lane] = MOMM_ZERO;
    }
    momm_flush(ring, lane, half, HH, HL, LL);
  }
  // the squared distances of the MSE stop rule: a lane added <= R + its share of the deferred batches (each <= 2^24) in 32 bits
  long long dw = (long long)dsum;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) dw += __shfl_xor(dw, o);
  __syncthreads();  // every wavefront is done with its ring: the tiles take their place
  int* __restrict__ tile = reinterpret_cast<int*>(&L.ring[0][0][0]);
  // tile entry (row, col) sits in lane col + 16 (row / 4), register row % 4
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = ((lane >> 4) * 4 + k) * 16 + (lane & 15);
    tile[(wave * 3 + 0) * 256 + e] = HH[k], tile[(wave * 3 + 1) * 256 + e] = HL[k], tile[(wave * 3 + 2) * 256 + e] = LL[k];
  }
  if (lane == 0) L.dsum[wave] = dw, L.n_cnt[wave] = n_wave;
  __syncthreads();
  long long* __restrict__ out = reinterpret_cast<long long*>(a.partial) + ((size_t)hl * gridDim.x + blockIdx.x) * ICP_NMOMI_STRIDE;
  const int t = threadIdx.x;
  if (t < ICP_NMOMI - 1) {
    const int e = (int)c_momm_tab.u[t] * 16 + (int)c_momm_tab.v[t], et = (int)c_momm_tab.v[t] * 16 + (int)c_momm_tab.u[t];
    long long hh = 0, hl2 = 0, ll = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      hh += (long long)tile[(w * 3 + 0) * 256 + e];
      hl2 += (long long)tile[(w * 3 + 1) * 256 + e] + (long long)tile[(w * 3 + 1) * 256 + et];
      ll += (long long)tile[(w * 3 + 2) * 256 + e];
    }
    out[t] = hh * 65536 + hl2 * 256 + ll;
  } else if (t == ICP_NMOMI - 1) {
    out[t] = (L.dsum[0] + L.dsum[1]) + (L.dsum[2] + L.dsum[3]);
  } else if (t == ICP_NMOMI) {
    out[t] = (long long)((L.n_cnt[0] + L.n_cnt[1]) + (L.n_cnt[2] + L.n_cnt[3]));
  }
}
void launch_icp_fusedq_momm(const IcpArgs& a, int hb, hipStream_t s) {
  const int nb = icp_blocks_per_hyp(a.ns, true);
  const int R = (a.ns + 256 * nb - 1) / (256 * nb);
  hipLaunchKernelGGL(k_icp_fusedq_momm, dim3(nb, hb), dim3(256), 0, s, a, R);
}

// development aid (hop_debug_selftest, tests/test_gpu_zzzz_dev_selftest.py): the gfx950-specific primitives of the packed lookups and of the two
// moment kernels on caller-given operands, one element per thread -- what the instructions return on a device is compared with their
// documented semantics restated in numpy (the CPU model of tests/emu states them a third time).
//   out[0] momi_qp(x, y, 2^12)   [1] encoding of momm_qp(x, y)   [2] momi_pack(ia, ib)   [3] momi_dot2(ia, ib, ic)   [4] umed3(ia, ib, ic)
//   [5] q_rank({xy = ia, z = ic >> 16}, lo = ib, hi = ic)   [6] momm_bytes<1>(ib, ia)   [7] momm_bytes<0>(ib, ia)
//   [8] momi_q(x, y, 2^24)
__global__ HOP_PK_F32 void k_dev_selftest_scalar(int n, const float* __restrict__ x, const float* __restrict__ y, const int* __restrict__ ia, const int* __restrict__ ib,
                                      const int* __restrict__ ic, unsigned* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned ua = (unsigned)ia[i], ub = (unsigned)ib[i], uc = (unsigned)ic[i];
  out[0 * (size_t)n + i] = (unsigned)momi_qp(x[i], y[i], 1 << ICP_MOM_BITS);
  out[1 * (size_t)n + i] = __float_as_uint(momm_qp(x[i], y[i]));
  out[2 * (size_t)n + i] = momi_pack(ia[i], ib[i]);
  out[3 * (size_t)n + i] = (unsigned)momi_dot2(ua, ub, ic[i]);
  out[4 * (size_t)n + i] = umed3(ua, ub, uc);
  out[5 * (size_t)n + i] = q_rank(Q3{ua, ia[i] < 0 ? (uc >> 16) : ((uc >> 16) | 0xABCD0000u)}, ub, uc);   // (the high half of l.z must not matter)
  out[6 * (size_t)n + i] = momm_bytes<1>(ub, ua);
  out[7 * (size_t)n + i] = momm_bytes<0>(ub, ua);
  out[8 * (size_t)n + i] = (unsigned)momi_q(x[i], y[i], 16777216.0f);
}
// one v_mfma_i32_16x16x64_i8 per wavefront: a, b, c, d as [tiles][64 lanes][4 registers]
__global__ HOP_PK_F32 __launch_bounds__(64) void k_dev_selftest_mfma(const int* __restrict__ a, const int* __restrict__ b, const int* __restrict__ c, int* __restrict__ d) {
  const size_t o = ((size_t)blockIdx.x * 64 + threadIdx.x) * 4;
  const momm_i32x4 A = {a[o], a[o + 1], a[o + 2], a[o + 3]}, B = {b[o], b[o + 1], b[o + 2], b[o + 3]}, Cc = {c[o], c[o + 1], c[o + 2], c[o + 3]};
  const momm_i32x4 D = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, Cc, 0, 0, 0);
  d[o] = D[0], d[o + 1] = D[1], d[o + 2] = D[2], d[o + 3] = D[3];
}
// the read-out path of k_icp_fusedq_momm end to end on one wavefront: `batches` x 64 gridded vectors (13 integers each, |U| <= 2^12; batch b
// is accepted on the lanes of mask[b]) go through momm_qp-encoded floats -> momm_push -> ring -> momm_flush -> tiles; out = the three tiles
// [3][64 lanes][4 registers].  The caller recombines M = 65536 HH + 256 (HL + HL^T) + LL and compares with the sum of U U^T.
struct MommTestU {  // (13 given integers as the gridded vector)
  float v[13];
  template <int C>
  __device__ __forceinline__ float get() const {
    return momm_qp(v[C], 1.0f);
  }
};
__global__ HOP_PK_F32 __launch_bounds__(64) void k_dev_selftest_momm(int batches, const int* __restrict__ U, const unsigned long long* __restrict__ mask, int* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned short ring[13][MOMM_ROW];
  const int lane = threadIdx.x;
  momm_i32x4 HH = {0, 0, 0, 0}, HL = {0, 0, 0, 0}, LL = {0, 0, 0, 0};
  int fill = 0;
  for (int b = 0; b < batches; ++b) {
    MommTestU t;
#pragma unroll
    for (int c = 0; c < 13; ++c) t.v[c] = (float)U[((size_t)b * 64 + lane) * 13 + c];
    momm_push(ring, lane, ((mask[b] >> lane) & 1ull) != 0ull, t, fill, HH, HL, LL);
  }
  if (fill & 63) {
    const int half = (fill >> 6) & 1;
    if (lane >= (fill & 63)) {
#pragma unroll
      for (int c = 0; c < 13; ++c) ring[c][half * 64 + lane] = MOMM_ZERO;
    }
    momm_flush(ring, lane, half, HH, HL, LL);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) out[(0 * 64 + lane) * 4 + k] = HH[k], out[(1 * 64 + lane) * 4 + k] = HL[k], out[(2 * 64 + lane) * 4 + k] = LL[k];
}
void launch_dev_selftest_momm(int batches, const int* U, const unsigned long long* mask, int* out, hipStream_t s) {
  hipLaunchKernelGGL(k_dev_selftest_momm, dim3(1), dim3(64), 0, s, batches, U, mask, out);
}
void launch_dev_selftest_scalar(int n, const float* x, const float* y, const int* ia, const int* ib, const int* ic, unsigned* out, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_dev_selftest_scalar, dim3((n + 255) / 256), dim3(256), 0, s, n, x, y, ia, ib, ic, out);
}
void launch_dev_selftest_mfma(int tiles, const int* a, const int* b, const int* c, int* d, hipStream_t s) {
  if (tiles > 0) hipLaunchKernelGGL(k_dev_selftest_mfma, dim3(tiles), dim3(64), 0, s, a, b, c, d);
}

void launch_icp_fusedq_mom(const IcpArgs& a, int hb, hipStream_t s) {
  const int nb = icp_blocks_per_hyp(a.ns, true);
  const int R = (a.ns + 256 * nb - 1) / (256 * nb);
  hipLaunchKernelGGL(k_icp_fusedq_mom, dim3(nb, hb), dim3(256), 0, s, a, R);
}

__global__ HOP_PK_F32 void k_soa_to_aos4(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n, float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = make_float4(x[i], y[i], z[i], 0.f);
}
void launch_soa_to_aos4(const float* x, const float* y, const float* z, int n, float4* out, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_soa_to_aos4, dim3((n + 255) / 256), dim3(256), 0, s, x, y, z, n, out);
}

__device__ bool chol6(double A[6][6], const double b[6], double x[6]) {
  double L[6][6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) L[i][j] = 0.0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[i][j];
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

__global__ HOP_PK_F32 __launch_bounds__(64) void k_icp_solve(IcpArgs a, int hb, int nblocks) {
  const int hl = blockIdx.x * blockDim.x + threadIdx.x;
  if (hl >= hb) return;
  IcpState& st = a.state[hl];
  if (!st.active) return;
  double acc[ICP_NACC];
  for (int k = 0; k < ICP_NACC; ++k) acc[k] = 0.0;
  for (int blk = 0; blk < nblocks; ++blk)
    for (int k = 0; k < ICP_NACC; ++k) acc[k] += a.partial[((size_t)hl * nblocks + blk) * ICP_NACC + k];
  const int cnt = (int)acc[28];
  if (cnt < 3) {  // not converged: caller substitutes identity (Utils.cpp:218-225)
    st.active = 0;
    st.converged = 0;
    return;
  }
  double A[6][6], b[6], x[6];
  int k = 0;
  for (int u = 0; u < 6; ++u)
    for (int v = 0; v <= u; ++v) {
      A[u][v] = acc[k];
      A[v][u] = acc[k];
      ++k;
    }
  for (int u = 0; u < 6; ++u) b[u] = acc[21 + u];
  double tr = 0;
  for (int u = 0; u < 6; ++u) tr += A[u][u];
  for (int u = 0; u < 6; ++u) A[u][u] += 1e-9 * tr + 1e-30;
  if (!chol6(A, b, x)) {
    st.active = 0;
    st.converged = 0;
    return;
  }
  const double th = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  double Rm[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  if (th > 1e-12) {
    const double kx = x[0] / th, ky = x[1] / th, kz = x[2] / th, s = sin(th), c = cos(th), v = 1 - c;
    Rm[0][0] = c + kx * kx * v, Rm[0][1] = kx * ky * v - kz * s, Rm[0][2] = kx * kz * v + ky * s;
    Rm[1][0] = ky * kx * v + kz * s, Rm[1][1] = c + ky * ky * v, Rm[1][2] = ky * kz * v - kx * s;
    Rm[2][0] = kz * kx * v - ky * s, Rm[2][1] = kz * ky * v + kx * s, Rm[2][2] = c + kz * kz * v;
  }
  // the increment is applied as the exact rotation about the centroid of the matched source points plus the
  // translation the linear model gives that centroid (see the oracle's run_icp for the reasoning)
  const double c0 = acc[29] / cnt, c1 = acc[30] / cnt, c2 = acc[31] / cnt;
  const double tc[3] = {x[3] + (x[1] * c2 - x[2] * c1), x[4] + (x[2] * c0 - x[0] * c2), x[5] + (x[0] * c1 - x[1] * c0)};
  const double cc[3] = {c0, c1, c2};
  M4 T = m4_identity();
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T.m[4 * i + j] = (float)Rm[i][j];
    T.m[4 * i + 3] = (float)(cc[i] - (Rm[i][0] * c0 + Rm[i][1] * c1 + Rm[i][2] * c2) + tc[i]);
  }
  for (int i = 0; i < 12; ++i) st.T_inc[i] = T.m[i];
  if (a.hist)
    for (int i = 0; i < 12; ++i) a.hist[((size_t)hl * a.max_iter + st.iterations) * 12 + i] = T.m[i];
  M4 F;
  for (int i = 0; i < 16; ++i) F.m[i] = st.final_tf[i];
  F = m4_mul(T, F);
  for (int i = 0; i < 16; ++i) st.final_tf[i] = F.m[i];
  st.iterations += 1;
  const double mse = acc[27] / cnt;
  bool stop = false;
  if (st.iterations >= a.max_iter) stop = true;
  else if (fabs(mse - st.mse_prev) < 1e-6) stop = true;
  else if (fabs(mse - st.mse_prev) / st.mse_prev < 1e-10) stop = true;
  st.mse_prev = mse;
  if (stop) {
    st.active = 0;
    st.converged = 1;
  }
}

// pose <- T_icp^-1 * pose (PoseEstimator.cpp:267), identity if not converged
__global__ HOP_PK_F32 void k_icp_finish(IcpArgs a, int hb, int* iters_out, int* conv_out) {
  const int hl = blockIdx.x * blockDim.x + threadIdx.x;
  if (hl >= hb) return;
  const IcpState& st = a.state[hl];
  M4 F = m4_identity();
  if (st.converged)
    for (int i = 0; i < 16; ++i) F.m[i] = st.final_tf[i];
  M4 P;
  float* pose = a.pose + (size_t)(a.h0 + hl) * 16;
  for (int i = 0; i < 16; ++i) P.m[i] = pose[i];
  const M4 out = m4_mul(m4_inverse_affine(F), P);
  for (int i = 0; i < 16; ++i) pose[i] = out.m[i];
  if (iters_out) iters_out[a.h0 + hl] = st.iterations;
  if (conv_out) conv_out[a.h0 + hl] = st.converged;
}

// inverse of every hypothesis pose (affine), once per stage: the cell-list kernels map queries into the model's rest
// frame with it and read it through scalar loads instead of inverting per block
__global__ HOP_PK_F32 void k_pose_inverse(const float* __restrict__ pose, int n, float* __restrict__ inv12) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n) return;
  M4 P;
  for (int k = 0; k < 16; ++k) P.m[k] = pose[(size_t)h * 16 + k];
  const M4 inv = m4_inverse_affine(P);
  for (int k = 0; k < 12; ++k) inv12[(size_t)h * 12 + k] = inv.m[k];
}
void launch_pose_inverse(const float* pose, int n, float* inv12, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_pose_inverse, dim3((n + 63) / 64), dim3(64), 0, s, pose, n, inv12);
}

__global__ HOP_PK_F32 void k_icp_init(IcpState* st, int hb) {
  const int hl = blockIdx.x * blockDim.x + threadIdx.x;
  if (hl >= hb) return;
  IcpState s;
  for (int i = 0; i < 12; ++i) s.T_inc[i] = (i % 5 == 0) ? 1.f : 0.f;
  for (int i = 0; i < 16; ++i) s.final_tf[i] = (i % 5 == 0) ? 1.f : 0.f;
  s.mse_prev = 1.7976931348623157e308;
  s.iterations = 0;
  s.active = 1;
  s.converged = 0;
  st[hl] = s;
}


// ------------------------------------------------------------------------------------------------
// computeLCP nn_mode 3: the same lookups, but the 2N terms of a hypothesis are reduced inside the wavefront and the
// per-(64-point tile, hypothesis) partial sums go to a small table partial[tile][hypothesis] (N/64 x H floats, 13 MB at
// C2 instead of the 1.6 GB term table); k_lcp_sum_partial adds the tiles of a hypothesis in tile order in double.
// The score is the reference's sum in a different association: equal to ~1e-6 relative, not bit for bit
// (SURVEY.md 8(c) L2 asks 1e-4).  nn_mode 2 keeps the reference's order for the bit tests.
// Forward lookup: cells_nn with fused multiply-adds in the ranking part (the exact expression still decides).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cells_nn1f(const CellListDev& c, V3 qg, const float* T, V3 q, float& best, int& bpos, V3& moved) {
  const float gx = floorf(__builtin_fmaf(qg.x, c.inv_cell, c.gox)), gy = floorf(__builtin_fmaf(qg.y, c.inv_cell, c.goy)),
              gz = floorf(__builtin_fmaf(qg.z, c.inv_cell, c.goz));
  const int ix = (int)gx, iy = (int)gy, iz = (int)gz;
  if ((unsigned)ix >= (unsigned)c.dx || (unsigned)iy >= (unsigned)c.dy || (unsigned)iz >= (unsigned)c.dz) return;
  const int2 rg = c.range[(iz * c.dy + iy) * c.dx + ix];
  LCP_COUNT(1, 1);
  const int beg = rg.x, end = rg.y;
  if (beg >= end) return;
  float b1 = 3.0e38f, b2 = 3.0e38f;
  int k1 = beg;
  float wx = 0.f, wy = 0.f, wz = 0.f;
  for (int k = beg; k < end; ++k) {
    const float4 t = c.pts[k];
    LCP_COUNT(1, 1);
    const float d = rank_d2(qg, t);
    b2 = __builtin_amdgcn_fmed3f(b1, b2, d);
    const bool better = d < b1;
    k1 = better ? k : k1;
    wx = better ? t.x : wx, wy = better ? t.y : wy, wz = better ? t.z : wz;
    b1 = fminf(b1, d);
  }
  const float mag = fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), 0.25f));
  const float delta = mag * 2.0e-6f;
  // (an upper bound of cells_nn's tolerance without the square root: sqrt(b1) <= b1 q_sa + q_sb, see cells_nnq; a larger tol only sends a few
  // more lookups through the exact scan below)
  const float tol = __builtin_fmaf(2.002f * __builtin_fmaf(b1, c.q_sa, c.q_sb), delta, __builtin_fmaf(1.0e-4f, b1, delta * delta));
  moved = m4_point(T, v3(wx, wy, wz));
  best = sqdist_flann(q, moved);
  bpos = k1;
  if (b2 - b1 <= tol) {
    int bj = __float_as_int(c.pts[k1].w);
    const float lim = b1 + tol;
    for (int k = beg; k < end; ++k) {
      if (k == k1) continue;
      const float4 t = c.pts[k];
      if (!(rank_d2(qg, t) <= lim)) continue;
      const V3 tm = m4_point(T, v3(t.x, t.y, t.z));
      const float d2 = sqdist_flann(q, tm);
      const int j = __float_as_int(t.w);
      if (d2 < best || (d2 == best && j < bj)) best = d2, bj = j, bpos = k, moved = tm;
    }
  }
}

// cells_nn1f / cells_nn_plain through the inline-head records (CellListDev::head): the same candidates in the same order with the same
// expressions and tie rules -- the same bits -- but a one-entry list is ONE access (the head) instead of two (range record + entry).
__device__ __forceinline__ void cells_nn1f_head(const CellListDev& c, V3 qg, const float* T, V3 q, float& best, int& bpos, V3& moved) {
  const float gx = floorf(__builtin_fmaf(qg.x, c.inv_cell, c.gox)), gy = floorf(__builtin_fmaf(qg.y, c.inv_cell, c.goy)),
              gz = floorf(__builtin_fmaf(qg.z, c.inv_cell, c.goz));
  const int ix = (int)gx, iy = (int)gy, iz = (int)gz;
  if ((unsigned)ix >= (unsigned)c.dx || (unsigned)iy >= (unsigned)c.dy || (unsigned)iz >= (unsigned)c.dz) return;
  const int cidx = (iz * c.dy + iy) * c.dx + ix;
  uint4 hd = c.head[cidx];
  keep_whole(hd);
  LCP_COUNT(1, 1);
  const int n = (int)(hd.w >> 24);
  if (n == 0) return;
  const int beg = (int)(hd.w & 0xFFFFFFu);
  float wx = __uint_as_float(hd.x), wy = __uint_as_float(hd.y), wz = __uint_as_float(hd.z);
  float b1 = rank_d2(qg, make_float4(wx, wy, wz, 0.f)), b2 = 3.0e38f;
  int k1 = beg, end = beg + n;
  if (n > 1) {
    if (n == 255) {
      end = c.range[cidx].y;
      LCP_COUNT(1, 1);
    }
    for (int k = beg + 1; k < end; ++k) {
      const float4 t = c.pts[k];
      LCP_COUNT(1, 1);
      const float d = rank_d2(qg, t);
      b2 = __builtin_amdgcn_fmed3f(b1, b2, d);
      const bool better = d < b1;
      k1 = better ? k : k1;
      wx = better ? t.x : wx, wy = better ? t.y : wy, wz = better ? t.z : wz;
      b1 = fminf(b1, d);
    }
  }
  const float mag = fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), 0.25f));
  const float delta = mag * 2.0e-6f;
  // (an upper bound of cells_nn's tolerance without the square root: sqrt(b1) <= b1 q_sa + q_sb, see cells_nnq; a larger tol only sends a few
  // more lookups through the exact scan below)
  const float tol = __builtin_fmaf(2.002f * __builtin_fmaf(b1, c.q_sa, c.q_sb), delta, __builtin_fmaf(1.0e-4f, b1, delta * delta));
  moved = m4_point(T, v3(wx, wy, wz));
  best = sqdist_flann(q, moved);
  bpos = k1;
  if (b2 - b1 <= tol) {
    int bj = __float_as_int(c.pts[k1].w);
    const float lim = b1 + tol;
    for (int k = beg; k < end; ++k) {
      if (k == k1) continue;
      const float4 t = c.pts[k];
      if (!(rank_d2(qg, t) <= lim)) continue;
      const V3 tm = m4_point(T, v3(t.x, t.y, t.z));
      const float d2 = sqdist_flann(q, tm);
      const int j = __float_as_int(t.w);
      if (d2 < best || (d2 == best && j < bj)) best = d2, bj = j, bpos = k, moved = tm;
    }
  }
}
__device__ __forceinline__ void cells_nn_plain_head(const CellListDev& c, V3 q, float& best, int& bpos) {
  const float fx = (q.x - c.ox) * c.inv_cell, fy = (q.y - c.oy) * c.inv_cell, fz = (q.z - c.oz) * c.inv_cell;
  if (!(fx >= 0.f && fy >= 0.f && fz >= 0.f && fx < (float)c.dx && fy < (float)c.dy && fz < (float)c.dz)) return;
  const int cidx = ((int)fz * c.dy + (int)fy) * c.dx + (int)fx;
  uint4 hd = c.head[cidx];
  keep_whole(hd);
  LCP_COUNT(2, 1);
  const int n = (int)(hd.w >> 24);
  if (n == 0) return;
  const int beg = (int)(hd.w & 0xFFFFFFu);
  // entry 0 from the head; its original index (the tie rule) is fetched only if another entry ties with it exactly
  const float d0 = sqdist_flann(q, v3(__uint_as_float(hd.x), __uint_as_float(hd.y), __uint_as_float(hd.z)));
  int bj = -1;  // -1: entry 0 is the best so far and its index has not been read
  if (d0 < best) best = d0, bpos = beg;
  else return cells_nn_plain(c, q, best, bpos);  // (callers pass best = 3e38: not taken; kept so that the function is right for any caller)
  if (n > 1) {
    int end = beg + n;
    if (n == 255) {
      end = c.range[cidx].y;
      LCP_COUNT(2, 1);
    }
    for (int k = beg + 1; k < end; ++k) {
      const float4 t = c.pts[k];
      LCP_COUNT(2, 1);
      const float d2 = sqdist_flann(q, v3(t.x, t.y, t.z));
      const int j = __float_as_int(t.w);
      // (selects on the common path; the exact tie -- two scene points at one float distance from the query -- is the only branch)
      const float b0 = best;
      const bool lt = d2 < b0;
      best = lt ? d2 : best, bj = lt ? j : bj, bpos = lt ? k : bpos;
      if (d2 == b0) {
        if (bj < 0) {
          bj = __float_as_int(c.pts[beg].w);
          LCP_COUNT(2, 1);
        }
        if (j < bj) bj = j, bpos = k;
      }
    }
  }
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  return v;
}

// hypotheses per block of the reduced-sum kernel: 4 = one per wavefront (4 / 8 / 16 / 32: 3.29 / 3.62 / 3.75 / 3.91 ms at C2 --
// the blocks of an XCD walk neighbouring point tiles of the same few hypotheses and share more of the model's lines)
constexpr int LCP_FTH = 4;
// K = 64-point tiles per wavefront and hypothesis (lcp_tiles_per_wave): with many hypotheses a wavefront walks K consecutive tiles of the
// Morton-ordered scene for its hypothesis, adds a lane's terms in a register and reduces ONCE -- the pose and inverse pose (28 scalar loads), the
// wavefront reduction and the partial-sum store are paid per 64 K lookups instead of per 64.
template <bool HEAD>
__global__ __launch_bounds__(256) void k_lcp_cells_fast(LcpArgs a, int hb, int hs, int npt, int K) {
  // XCD-aware mapping: workgroups go to the 8 XCDs round-robin (blockIdx.x & 7) and each XCD has its own 4 MB L2; the
  // scene lists + model lists (~18 MB at C2) do not fit one L2, an eighth of the Morton-ordered scene with the model
  // region it meets does.  XCD k walks the point tiles [k * per, (k + 1) * per) for every hypothesis tile.  (npt: tiles of 64 K points)
  const int per = (npt + 7) >> 3;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int pt = xcd * per + j % per, ht = j / per;
  if (pt >= npt) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float inv_dist = 1.f / a.dist;
  for (int j = 0; j < LCP_FTH / 4; ++j) {
    const int hh = wave * (LCP_FTH / 4) + j, hl = ht * LCP_FTH + hh;
    if (hl >= hb) break;  // wave-uniform
    const float* __restrict__ T = a.pose + (size_t)(a.h0 + hl) * 16;
    const float* __restrict__ Ti = a.pose_inv + (size_t)(a.h0 + hl) * 12;
    float v = 0.f;
    for (int u = 0; u < K; ++u) {
      const int k = (pt * K + u) * 64 + lane;
      if (k >= a.ns) continue;
      const V3 s = v3(a.qx[k], a.qy[k], a.qz[k]), sn = v3(a.qnx[k], a.qny[k], a.qnz[k]);
      float best = 3.0e38f;
      int pos = -1;
      V3 pm;
      LCP_COUNT(0, 1);
      if (HEAD) cells_nn1f_head(a.model_cells, m4_point_fma(Ti, s), T, s, best, pos, pm);
      else cells_nn1f(a.model_cells, m4_point_fma(Ti, s), T, s, best, pos, pm);
      if (pos >= 0 && best < a.dist * a.dist) {
        // this mode's contract is 1e-4 relative on the score: the normalisation and the (1 - d / dist) factor use the hardware's
        // reciprocal square root / square root (1 ulp) instead of the IEEE division and square-root sequences (~100 of the ~590 vector
        // instructions of a lookup, profiles/r03_pmc_sq.txt); nn_mode 2 keeps the reference's operations
        const float4 mnr = a.model_cells.nrm[pos];
        LCP_COUNT(1, 1);
        LCP_COUNT(3, 1);
        const V3 nraw = m4_dir(T, v3(mnr.x, mnr.y, mnr.z));
        V3 nmod = nraw * __builtin_amdgcn_rsqf(vsqn(nraw));
        float d1 = vdot(sn, nmod);
        // a dot product within 2e-6 of the threshold is decided by the reference's own operations (the fast value is within 3e-7 of it):
        // which terms enter the sum is exactly the reference's choice
        if (fabsf(d1 - a.cos_thres) < 2.0e-6f) nmod = vnormalized(nraw), d1 = vdot(sn, nmod);
        if (d1 > a.cos_thres) v += d1 * (1.f - __builtin_amdgcn_sqrtf(best) * inv_dist);
        float rbest = 3.0e38f;
        int rk = -1;
        if (HEAD) cells_nn_plain_head(a.scene_cells, pm, rbest, rk);
        else cells_nn_plain(a.scene_cells, pm, rbest, rk);
        if (rk >= 0) {
          const float4 rn = a.scene_cells.nrm[rk];
          LCP_COUNT(2, 1);
          float d2r = vdot(nmod, v3(rn.x, rn.y, rn.z));
          if (fabsf(d2r - a.cos_thres) < 2.0e-6f) d2r = vdot(vnormalized(nraw), v3(rn.x, rn.y, rn.z));
          if (d2r > a.cos_thres) v += d2r * (1.f - __builtin_amdgcn_sqrtf(rbest) * inv_dist);
        }
      }
    }
    v = wave_sum_f(v);
    if (lane == 0) a.terms[(size_t)pt * hs + hl] = v;
  }
}

__global__ HOP_PK_F32 __launch_bounds__(64) void k_lcp_sum_partial(LcpArgs a, int hb, int hs, int npt) {
  const int hl = blockIdx.x * blockDim.x + threadIdx.x;
  if (hl >= hb) return;
  double cp = 0.0;
  for (int t = 0; t < npt; ++t) cp += (double)a.terms[(size_t)t * hs + hl];
  a.score[a.h0 + hl] = (float)cp;
}
// 64-point tiles per wavefront of k_lcp_cells_fast: 4 when that still leaves >= 16 workgroups per CU (C2: 69 x 2560), else 1 (the as-shipped
// sizes -- 100 hypotheses x 2 000 points -- need every workgroup they can get).  HOP_LCP_TILES=<k> overrides (A/B runs).
// Decided ONCE per hop_lcp_select_best call from the whole set's H (LcpArgs::tiles_per_wave), never per batch: K fixes the association of a
// hypothesis' float sum, and a score must not depend on which batch of a call its hypothesis fell into.
int lcp_tiles_per_wave(int ns, int H) {
  if (const char* e = getenv("HOP_LCP_TILES")) return std::max(1, std::min(16, atoi(e)));
  const long long blocks1 = (long long)((ns + 63) / 64) * ((H + LCP_FTH - 1) / LCP_FTH);
  return blocks1 >= 4 * 4096 ? 4 : 1;
}
void launch_lcp_cells_fast(const LcpArgs& a, int hb, hipStream_t s) {
  const int K = std::max(1, a.tiles_per_wave);
  const int npt = (a.ns + 64 * K - 1) / (64 * K), nht = (hb + LCP_FTH - 1) / LCP_FTH;
  const int hs = ((hb + LCP_FTH - 1) / LCP_FTH) * LCP_FTH;
  const dim3 grid((unsigned)(8 * ((npt + 7) / 8) * nht));
  if (a.model_cells.head && a.scene_cells.head) hipLaunchKernelGGL(k_lcp_cells_fast<true>, grid, dim3(256), 0, s, a, hb, hs, npt, K);
  else hipLaunchKernelGGL(k_lcp_cells_fast<false>, grid, dim3(256), 0, s, a, hb, hs, npt, K);
}
void launch_lcp_sum_partial(const LcpArgs& a, int hb, hipStream_t s) {
  const int K = std::max(1, a.tiles_per_wave);
  const int npt = (a.ns + 64 * K - 1) / (64 * K);
  const int hs = ((hb + LCP_FTH - 1) / LCP_FTH) * LCP_FTH;
  hipLaunchKernelGGL(k_lcp_sum_partial, dim3((hb + 63) / 64), dim3(64), 0, s, a, hb, hs, npt);
}

// ------------------------------------------------------------------------------------------------
// K1: hand-state objective, data-parallel part of objFuncPSO (Hand.cpp:67-152) for a batch of particles.
//   k_pso_match : per (particle, finger point) nearest scene point + the match test -> match count
//   k_pso_outer : per (particle, scene point) outer-side distance term (or -1)
//   k_pso_outer_sum : sequential float sum per particle, in scene order, as the reference does
// The host finishes the scalar part of the objective (gripper gap, penalties), hop_host.cpp.
// ------------------------------------------------------------------------------------------------
template <int R>
__global__ HOP_PK_F32 __launch_bounds__(256) void k_pso_match(PsoArgs a) {
  __shared__ float4 tile[NN_TILE];
  const int p = blockIdx.y;
  const PsoParticle& pp = a.particles[p];
  if (pp.skip) return;
  V3 q[R], qn[R];
  float best[R];
  int bidx[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int j = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
    best[r] = 3.0e38f;
    bidx[r] = -1;
    if (j < a.nm) {
      q[r] = m4_point(pp.T, v3(a.mx[j], a.my[j], a.mz[j]));
      qn[r] = m4_dir(pp.T, v3(a.mnx[j], a.mny[j], a.mnz[j]));
    } else {
      q[r] = v3(-HOP_FAR, -HOP_FAR, -HOP_FAR);
      qn[r] = v3(0, 0, 0);
    }
  }
  if (a.use_grid == 2) {
    // NN cell lists of the hand scene (max_dist = dist_thres): the list position is turned back into the scene index
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int j = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
      int pos = -1;
      if (j < a.nm) cells_nn_plain(a.scene_cells, q[r], best[r], pos);
      if (pos >= 0) bidx[r] = __float_as_int(a.scene_cells.pts[pos].w);
    }
  } else if (a.use_grid) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int j = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
      if (j < a.nm) grid_nn_rings<false>(a.scene_grid, q[r], nullptr, q[r], a.max_ring, a.dist_thres * a.dist_thres, best[r], bidx[r]);
    }
  } else {
    for (int start = 0; start < a.ns; start += NN_TILE) {
      const int tn = min(NN_TILE, round_up(a.ns - start, NN_CH));
      __syncthreads();
      stage_tile_raw(tile, a.sx, a.sy, a.sz, start, a.ns, tn);
      __syncthreads();
      nn_scan_tile<R, true, true>(tile, tn, start, q, best, bidx);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int j = blockIdx.x * (256 * R) + r * 256 + threadIdx.x;
    bool match = false;
    if (j < a.nm && bidx[r] >= 0 && best[r] <= a.dist_thres * a.dist_thres) {
      if (!a.check_normal) match = true;
      else {
        // Hand.cpp:91: normal fetched from the UNFILTERED cloud with the filtered cloud's index
        V3 nn = v3(0, 0, 0);
        if (bidx[r] < a.n_lookup) nn = v3(a.lnx[bidx[r]], a.lny[bidx[r]], a.lnz[bidx[r]]);
        if (nn.x == 0.f && nn.y == 0.f && nn.z == 0.f) match = true;
        else if (isfinite(nn.x) && isfinite(nn.y) && isfinite(nn.z)) match = vdot(qn[r], nn) >= a.cos_normal_thres;
      }
    }
    const unsigned long long m = __ballot(match);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&a.match_count[p], __popcll(m));
  }
}
template __global__ void k_pso_match<2>(PsoArgs);

__global__ HOP_PK_F32 __launch_bounds__(256) void k_pso_outer(PsoArgs a) {
  // lanes = particles (each keeps its inverse transform in registers), the block walks a tile of scene points;
  // terms are stored as float4 groups of four consecutive points per particle, [point/4][particle][point%4], so that
  // this kernel writes and the sequential sum reads 16 bytes per lane, coalesced.
  const int p = blockIdx.y * blockDim.x + threadIdx.x;
  const bool live = p < a.n_particles && !a.particles[min(p, a.n_particles - 1)].skip;
  float Ti[12];
  for (int k = 0; k < 12; ++k) Ti[k] = live ? a.particles[p].Tinv[k] : 0.f;
  const int groups = (a.n_swivel + 3) / 4;
  const int per = (groups + gridDim.x - 1) / gridDim.x;
  const int g0 = blockIdx.x * per, g1 = min(groups, g0 + per);
  for (int gi = g0; gi < g1; ++gi) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = 4 * gi + u;
      v[u] = -1.f;  // padding behind the last point: "no contribution"
      if (i < a.n_swivel) {
        const V3 pt = m4_point(Ti, v3(a.wx[i], a.wy[i], a.wz[i]));
        int bin = (int)(fmaxf(pt.z - a.fp_min_z, 0.0f) / a.fp_stride_z);  // FingerProperty::getBinAlongZ, Hand.cpp:244-250
        bin = max(bin, 0);
        bin = min(bin, a.fp_num_division - 1);
        const float lim = a.hist_min_y[bin];
        v[u] = (pt.y >= lim) ? -1.f : fabsf(pt.y - lim);
      }
    }
    if (live) reinterpret_cast<float4*>(a.outer_terms)[(size_t)gi * a.n_particles + p] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

__global__ HOP_PK_F32 __launch_bounds__(64) void k_pso_outer_sum(PsoArgs a, int n_particles) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_particles) return;
  if (a.particles[p].skip) return;
  float sum = 0.f;
  int cnt = 0;
  // The additions stay in scene order (bit-equal to the CPU loop).  Skipped terms (marker -1, or NaN) are added as
  // +0, which leaves the non-negative running sum unchanged, so the dependent chain is one v_add per point.  Four
  // waves are all this launch has, so memory latency is its whole cost: 48 x 16-byte loads are kept in flight per lane.
  constexpr int U = 48;
  const float4* t = reinterpret_cast<const float4*>(a.outer_terms) + p;
  const int groups = (a.n_swivel + 3) / 4;
  auto add = [&](float v) {
    const bool in = v >= 0.f;
    sum += in ? v : 0.f;
    cnt += in ? 1 : 0;
  };
  int gi = 0;
  for (; gi + U <= groups; gi += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = t[(size_t)(gi + u) * n_particles];
#pragma unroll
    for (int u = 0; u < U; ++u) add(v[u].x), add(v[u].y), add(v[u].z), add(v[u].w);
  }
  for (; gi < groups; ++gi) {
    const float4 v = t[(size_t)gi * n_particles];
    add(v.x), add(v.y), add(v.z), add(v.w);
  }
  a.outer_sum[p] = sum;
  a.outer_cnt[p] = cnt;
}


// sum_mode 1: the outer-side term of objFuncPSO (Hand.cpp:141-152) reduced inside the block instead of added in scene
// order: one block per particle, threads stride over the no-swivel scene, float tree reduction in a fixed order
// (deterministic; equal to the reference's sequential float sum to ~1e-6 relative -- tests/test_gpu_parity.py).  No term
// table, no 4-wave sequential pass.
__global__ HOP_PK_F32 __launch_bounds__(256) void k_pso_outer_reduce(PsoArgs a) {
  __shared__ float rs[4];
  __shared__ int rc[4];
  const int p = blockIdx.x;
  const PsoParticle& pp = a.particles[p];
  if (pp.skip) return;
  const float* __restrict__ Ti = pp.Tinv;
  float sum = 0.f;
  int cnt = 0;
  for (int i = threadIdx.x; i < a.n_swivel; i += 256) {
    const V3 pt = m4_point(Ti, v3(a.wx[i], a.wy[i], a.wz[i]));
    int bin = (int)(fmaxf(pt.z - a.fp_min_z, 0.0f) / a.fp_stride_z);
    bin = min(max(bin, 0), a.fp_num_division - 1);
    const float lim = a.hist_min_y[bin];
    const float v = (pt.y >= lim) ? -1.f : fabsf(pt.y - lim);
    if (v >= 0.f) sum += v, cnt += 1;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off), cnt += __shfl_down(cnt, off);
  if ((threadIdx.x & 63) == 0) rs[threadIdx.x >> 6] = sum, rc[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    a.outer_sum[p] = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    a.outer_cnt[p] = (rc[0] + rc[1]) + (rc[2] + rc[3]);
  }
}

// ------------------------------------------------------------------------------------------------
// N4: the pair loop of the offline computePPF tool (computePPF.cpp:17-38,88-100): key of every pair i < j of the model
// cloud, collected in the same direct-address bitmap the generator looks keys up in.
// ------------------------------------------------------------------------------------------------
__global__ HOP_PK_F32 __launch_bounds__(256) void k_model_ppf_keys(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                       const float* __restrict__ nx, const float* __restrict__ ny, const float* __restrict__ nz,
                                                       int n, int dist_bins, unsigned* __restrict__ bitmap, int* __restrict__ overflow) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= n || j <= i) return;
  const V3 p1 = v3(x[i], y[i], z[i]), p2 = v3(x[j], y[j], z[j]);
  const V3 n1 = vnormalized(v3(nx[i], ny[i], nz[i])), n2 = vnormalized(v3(nx[j], ny[j], nz[j]));  // n.normalize(), once
  int key[4];
  if (!ppf_key(p1, n1, p2, n2, key)) return;  // NaN angle: the tool stores an INT_MIN key that can never be looked up
  const int d = key[0] / 5, a1 = key[1] / 10, a2 = key[2] / 10, a3 = key[3] / 10;
  if (key[0] < 0 || (unsigned)a1 >= 19u || (unsigned)a2 >= 19u || (unsigned)a3 >= 19u) return;
  if (d >= dist_bins) {
    *overflow = 1;
    return;
  }
  const unsigned bit = ((unsigned)(d * 19 + a1) * 19u + (unsigned)a2) * 19u + (unsigned)a3;
  atomicOr(&bitmap[bit >> 5], 1u << (bit & 31));
}
void launch_model_ppf_keys(const float* x, const float* y, const float* z, const float* nx, const float* ny, const float* nz, int n,
                           int dist_bins, unsigned* bitmap, int* overflow, hipStream_t s) {
  if (n > 1) hipLaunchKernelGGL(k_model_ppf_keys, dim3((n + 255) / 256, n), dim3(256), 0, s, x, y, z, nx, ny, nz, n, dist_bins, bitmap, overflow);
}

// ------------------------------------------------------------------------------------------------
// N3a: HandT42::removeSurroundingPointsAndAssignProbability (Hand.cpp:779-888).  One thread per scene point; the link
// clouds (a few hundred points each) are scanned by brute force -- every lane of a wave reads the same link point, so
// a load serves 64 queries.  Links are visited in the reference's map order with its early exit.
// ------------------------------------------------------------------------------------------------
__global__ HOP_PK_F32 __launch_bounds__(256) void k_hand_surround(SurroundArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const V3 p = m4_point(a.cam2hb, v3(a.sx[i], a.sy[i], a.sz[i]));
  const V3 nn = m4_dir(a.cam2hb, v3(a.snx[i], a.sny[i], a.snz[i]));
  bool is_near = false;
  float min_dist = 1.0f;
  for (int l = 0; l < a.n_links && !is_near; ++l) {
    const int b = a.link_start[l], e = a.link_start[l + 1];
    if (b >= e) continue;
    float best = 3.402823466e+38f;
    int bj = b;
    for (int j = b; j < e; ++j) {
      const float4 q = a.link_pts[j];
      const float d2 = sqdist_flann(p, v3(q.x, q.y, q.z));
      if (d2 < best) best = d2, bj = j;
    }
    min_dist = fminf(min_dist, sqrtf(best));
    const float thr = a.link_thres[l];
    if (best <= thr) {
      is_near = true;
      break;
    }
    const float4 nei = a.link_pts[bj];
    const float sq_planar = (p.x - nei.x) * (p.x - nei.x) + (p.y - nei.y) * (p.y - nei.y);
    if (sq_planar <= thr && (double)fabsf(p.z - nei.z) <= 0.005) is_near = true;
  }
  bool keep = !is_near;
  if (keep) {
    const V3 p1 = m4_point(a.f1inv, p), p2 = m4_point(a.f2inv, p);
    if ((p1.y < 0 && p1.z >= a.min_z) || (p2.y < 0 && p2.z >= a.min_z)) keep = false;
  }
  const size_t n = (size_t)a.n;
  a.hbp[i] = p.x, a.hbp[n + i] = p.y, a.hbp[2 * n + i] = p.z;
  a.hbp[3 * n + i] = nn.x, a.hbp[4 * n + i] = nn.y, a.hbp[5 * n + i] = nn.z;
  // expf of the CPU library is correctly rounded for all practical purposes: the double exponential rounded to
  // float reproduces it
  a.conf[i] = 1 - (float)exp((double)(-231.04906018664843f * min_dist));
  a.keep[i] = keep ? 1 : 0;
}
__global__ HOP_PK_F32 __launch_bounds__(256) void k_hand_surround_out(SurroundOutArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n || !a.keep[i]) return;
  const size_t n = (size_t)a.n;
  const int k = a.pos[i];
  const V3 pc = m4_point(a.hb2cam, v3(a.hbp[i], a.hbp[n + i], a.hbp[2 * n + i]));
  const V3 nc = m4_dir(a.hb2cam, v3(a.hbp[3 * n + i], a.hbp[4 * n + i], a.hbp[5 * n + i]));
  a.ox[k] = pc.x, a.oy[k] = pc.y, a.oz[k] = pc.z;
  a.onx[k] = nc.x, a.ony[k] = nc.y, a.onz[k] = nc.z;
  a.oconf[k] = a.conf[i];
  a.oindex[k] = i;
}
void launch_hand_surround(const SurroundArgs& a, hipStream_t s) {
  if (a.n > 0) hipLaunchKernelGGL(k_hand_surround, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
}
void launch_hand_surround_out(const SurroundOutArgs& a, hipStream_t s) {
  if (a.n > 0) hipLaunchKernelGGL(k_hand_surround_out, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// voxel-grid construction over a fixed cloud (counting sort by cell): count, scan on host, fill.
// ------------------------------------------------------------------------------------------------
__global__ HOP_PK_F32 void k_grid_cell_ids(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n,
                                GridDev gd, int* __restrict__ cell_of, int* __restrict__ cell_count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cx = (int)floorf((x[i] - gd.ox) * gd.inv_cell), cy = (int)floorf((y[i] - gd.oy) * gd.inv_cell),
      cz = (int)floorf((z[i] - gd.oz) * gd.inv_cell);
  cx = min(max(cx, 0), gd.dx - 1), cy = min(max(cy, 0), gd.dy - 1), cz = min(max(cz, 0), gd.dz - 1);
  const int c = (cz * gd.dy + cy) * gd.dx + cx;
  cell_of[i] = c;
  atomicAdd(&cell_count[c], 1);
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers
// ------------------------------------------------------------------------------------------------
void launch_ppf_matrix(const PpfMatrixArgs& a, hipStream_t s) {
  dim3 grid((a.words + 3) / 4, (a.n + PPF_ROWS - 1) / PPF_ROWS);
  const bool sym = !getenv("HOP_PPF_NO_SYM");  // (tests compare the kernels)
  // (a tiled form that staged the reverse words in LDS and halved the HBM writes, 172 -> 88 MB, was SLOWER at C2, 1.48 vs 1.18 ms: the kernel is
  // bound by its vector instructions and the bitmap probes, not by its stores -- profiles/HISTORY.md)
  if (a.angle_thr && sym) hipLaunchKernelGGL(k_ppf_matrix_sym, grid, dim3(256), 0, s, a);
  else if (a.angle_thr) hipLaunchKernelGGL(k_ppf_matrix<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_ppf_matrix<false>, grid, dim3(256), 0, s, a);
}
void launch_pairs(const PairArgs& a, int nbases, hipStream_t s) {
  const long long total = (long long)a.nq * a.nq;
  dim3 grid((unsigned)((total + 255) / 256), nbases);
  hipLaunchKernelGGL(k_pairs, grid, dim3(256), 0, s, a);
}
void launch_quad_prep(const QuadPrepArgs& a, int nbases, int max_items, hipStream_t s) {
  (void)max_items;
  dim3 grid(16, nbases);  // 64 waves per base: ~12 second pairs per wave at C2 sizes
  hipLaunchKernelGGL(k_quad_prep, grid, dim3(256), 0, s, a);
}
void launch_quads(const QuadArgs& a, int nbases, int blocks_per_base, hipStream_t s) {
  dim3 grid(blocks_per_base, nbases);
  static const bool hashed = getenv("HOP_QUADS_HASH") != nullptr && atoi(getenv("HOP_QUADS_HASH")) != 0;
  if (hashed) {
    hipLaunchKernelGGL(k_quads_hash, dim3(4, nbases), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_quads, grid, dim3(256), 0, s, a, (int)QH_MAX);  // the bases it left (more first pairs than its table holds)
  } else {
    hipLaunchKernelGGL(k_quads, grid, dim3(256), 0, s, a, -1);
  }
  hipLaunchKernelGGL(k_quad_fit, dim3(8, FIT_QUEUES), dim3(256), 0, s, a);
}
void launch_verify_cells(const VerifyArgs& a, const CellListDev& cl, int blocks, hipStream_t s) {
  hipLaunchKernelGGL(k_verify_cells, dim3(blocks), dim3(256), 0, s, a, cl);
}
void launch_verify(const VerifyArgs& a, int mode, const GridDev* gd, int blocks, hipStream_t s) {
  if (mode == 1 && gd) hipLaunchKernelGGL(k_verify_grid, dim3(blocks), dim3(256), 0, s, a, *gd);
  else hipLaunchKernelGGL(k_verify_brute<4>, dim3(blocks), dim3(256), 0, s, a);
}
void launch_emit(const EmitArgs& a, int blocks, hipStream_t s) { hipLaunchKernelGGL(k_emit, dim3(blocks), dim3(256), 0, s, a); }
void launch_gather_hypos(const unsigned* perm, int n, const float* pose_in, const float* score_in, float* pose_out,
                         float* score_out, int* id_out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather_hypos, dim3((n * 16 + 255) / 256), dim3(256), 0, s, perm, n, pose_in, score_in, pose_out, score_out, id_out);
}
void launch_iota(unsigned* p, int n, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, s, p, n);
}
void launch_score_keys(const float* score, const int* ids, int n, unsigned long long* key, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_score_keys, dim3((n + 255) / 256), dim3(256), 0, s, score, ids, n, key);
}
void launch_lcp_reverse(const LcpArgs& a, int hb, hipStream_t s) {
  const int R = 4;
  hipLaunchKernelGGL(k_lcp_reverse<4>, dim3((a.nm + 256 * R - 1) / (256 * R), hb), dim3(256), 0, s, a);
}
void launch_lcp_forward(const LcpArgs& a, int hb, hipStream_t s) {
  const int R = 4;
  hipLaunchKernelGGL(k_lcp_forward<4>, dim3((a.ns + 256 * R - 1) / (256 * R), hb), dim3(256), 0, s, a);
}
void launch_lcp_sum(const LcpArgs& a, int hb, hipStream_t s) {
  hipLaunchKernelGGL(k_lcp_sum, dim3((hb + 63) / 64), dim3(64), 0, s, a, hb);
}
int icp_blocks_per_hyp(int ns, bool cells) {
  const int per_block = 256 * (cells ? ICP_ACCUM_R : 4);
  return (ns + per_block - 1) / per_block;
}
void launch_icp_init(IcpState* st, int hb, hipStream_t s) { hipLaunchKernelGGL(k_icp_init, dim3((hb + 63) / 64), dim3(64), 0, s, st, hb); }
void launch_icp_nn(const IcpArgs& a, int hb, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_nn<4>, dim3(icp_blocks_per_hyp(a.ns, false), hb), dim3(256), 0, s, a);
}
void launch_icp_nn_grid(const IcpArgs& a, int hb, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_nn_grid<4>, dim3(icp_blocks_per_hyp(a.ns, false), hb), dim3(256), 0, s, a);
}
void launch_icp_corr_cells(const IcpArgs& a, int hb, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_corr_cells, dim3((a.ns + 255) / 256, hb), dim3(256), 0, s, a);
}
void launch_icp_fused(const IcpArgs& a, int hb, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_fused<ICP_ACCUM_R>, dim3(icp_blocks_per_hyp(a.ns, true), hb), dim3(256), 0, s, a);
}
void launch_icp_fusedq(const IcpArgs& a, int hb, bool composed, hipStream_t s) {
  // the same number of blocks per hypothesis as the other cell-list kernels (the partial sums are laid out for it), the
  // points spread evenly over them: 20 000 points = 3 blocks of 27 x 256 instead of 32 x 256, 32 x 256, 14 x 256
  const int nb = icp_blocks_per_hyp(a.ns, true);
  const int R = (a.ns + 256 * nb - 1) / (256 * nb);
  if (composed) hipLaunchKernelGGL((k_icp_fusedq<true>), dim3(nb, hb), dim3(256), 0, s, a, R);
  else hipLaunchKernelGGL((k_icp_fusedq<false>), dim3(nb, hb), dim3(256), 0, s, a, R);
}
void launch_icp_accum(const IcpArgs& a, int hb, hipStream_t s) {
  const int nb = icp_blocks_per_hyp(a.ns, true);  // same partition of the points as launch_icp_fusedq: same partial sums
  hipLaunchKernelGGL(k_icp_accum, dim3(nb, hb), dim3(256), 0, s, a, (a.ns + 256 * nb - 1) / (256 * nb));
}
void launch_cell_list_bounds(const CellListBuildArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_cell_list_bounds, dim3(a.dx * a.dy * a.dz), dim3(256), 0, s, a);
}
void launch_cell_list_count(const CellListBuildArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_cell_list_build<false>, dim3(a.dx * a.dy * a.dz), dim3(64), 0, s, a);
}
void launch_cell_list_fill(const CellListBuildArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_cell_list_build<true>, dim3(a.dx * a.dy * a.dz), dim3(64), 0, s, a);
}
void launch_cell_list_local_flag(const CellListBuildArgs& a, const GridDev& g, int* flag, hipStream_t s) {
  const int n = a.dx * a.dy * a.dz + 1;
  hipLaunchKernelGGL(k_cell_list_local_flag, dim3((n + 255) / 256), dim3(256), 0, s, a, g, flag);
}
void launch_cell_list_local_work(const int* flag, const int* flag_scan, int ncell, int* work, hipStream_t s) {
  hipLaunchKernelGGL(k_cell_list_local_work, dim3((ncell + 255) / 256), dim3(256), 0, s, flag, flag_scan, ncell, work);
}
void launch_cell_list_local(const CellListBuildArgs& a, const GridDev& g, bool write, int exist_mode, const int* work, int nwork, int* keep_buf,
                            int lanes, hipStream_t s) {
  if (nwork <= 0) return;
  // lanes per voxel: 16 for the short lists (a few candidate rows of a few points: scene and Verify lists, 1.7x faster), 32 or a
  // whole wavefront where the lists are long (the hand scene: 25 candidate rows per voxel, lists of up to a few hundred entries)
  if (lanes >= 64) {
    if (write) hipLaunchKernelGGL((k_cell_list_local<true, 64>), dim3((nwork + 3) / 4), dim3(256), 0, s, a, g, exist_mode, work, nwork, keep_buf);
    else hipLaunchKernelGGL((k_cell_list_local<false, 64>), dim3((nwork + 3) / 4), dim3(256), 0, s, a, g, exist_mode, work, nwork, keep_buf);
  } else if (lanes == 32) {
    if (write) hipLaunchKernelGGL((k_cell_list_local<true, 32>), dim3((nwork + 7) / 8), dim3(256), 0, s, a, g, exist_mode, work, nwork, keep_buf);
    else hipLaunchKernelGGL((k_cell_list_local<false, 32>), dim3((nwork + 7) / 8), dim3(256), 0, s, a, g, exist_mode, work, nwork, keep_buf);
  } else {
    if (write) hipLaunchKernelGGL((k_cell_list_local<true, 16>), dim3((nwork + 15) / 16), dim3(256), 0, s, a, g, exist_mode, work, nwork, keep_buf);
    else hipLaunchKernelGGL((k_cell_list_local<false, 16>), dim3((nwork + 15) / 16), dim3(256), 0, s, a, g, exist_mode, work, nwork, keep_buf);
  }
}
int cell_list_local_keep() { return LOCAL_KEEP; }
__global__ HOP_PK_F32 __launch_bounds__(256) void k_cell_ranges(const int* __restrict__ start, int ncell, int2* __restrict__ range) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < ncell) range[c] = make_int2(start[c], start[c + 1]);
}
__global__ HOP_PK_F32 __launch_bounds__(256) void k_cell_heads(const int2* __restrict__ range, const float4* __restrict__ pts, int ncell, uint4* __restrict__ head) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const int2 r = range[c];
  const int n = r.y - r.x;
  uint4 h = make_uint4(0u, 0u, 0u, 0u);
  if (n > 0) {
    const float4 t = pts[r.x];
    h = make_uint4(__float_as_uint(t.x), __float_as_uint(t.y), __float_as_uint(t.z), (unsigned)r.x | ((unsigned)min(n, 255) << 24));
  }
  head[c] = h;
}
void launch_cell_heads(const int2* range, const float4* pts, int ncell, uint4* head, hipStream_t s) {
  hipLaunchKernelGGL(k_cell_heads, dim3((ncell + 255) / 256), dim3(256), 0, s, range, pts, ncell, head);
}
void launch_cell_ranges(const int* start, int ncell, int2* range, hipStream_t s) {
  hipLaunchKernelGGL(k_cell_ranges, dim3((ncell + 255) / 256), dim3(256), 0, s, start, ncell, range);
}
int lcp_cells_row_stride(int hb) { return ((hb + LCP_TH - 1) / LCP_TH) * LCP_TH; }
void launch_lcp_cells(const LcpArgs& a, int hb, hipStream_t s) {
  const int npt = (a.ns + 63) / 64, nht = (hb + LCP_TH - 1) / LCP_TH;
  hipLaunchKernelGGL(k_lcp_cells, dim3((unsigned)(npt * nht)), dim3(256), 0, s, a, hb, lcp_cells_row_stride(hb), npt);
}
void launch_lcp_sum_t(const LcpArgs& a, int hb, hipStream_t s) {
  hipLaunchKernelGGL(k_lcp_sum_t, dim3((hb + 63) / 64), dim3(64), 0, s, a, hb, lcp_cells_row_stride(hb));
}
void launch_lcp_grid(const LcpArgs& a, int hb, hipStream_t s) {
  hipLaunchKernelGGL(k_lcp_grid, dim3((a.ns + 255) / 256, hb), dim3(256), 0, s, a);
}
void launch_icp_solve(const IcpArgs& a, int hb, int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_solve, dim3((hb + 63) / 64), dim3(64), 0, s, a, hb, nblocks);
}
void launch_icp_finish(const IcpArgs& a, int hb, int* iters, int* conv, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_finish, dim3((hb + 63) / 64), dim3(64), 0, s, a, hb, iters, conv);
}
void launch_pso(const PsoArgs& a, int n_particles, hipStream_t s) {
  const int R = 2;
  hipLaunchKernelGGL(k_pso_match<2>, dim3((a.nm + 256 * R - 1) / (256 * R), n_particles), dim3(256), 0, s, a);
  if (a.sum_mode == 1) {
    hipLaunchKernelGGL(k_pso_outer_reduce, dim3(n_particles), dim3(256), 0, s, a);
    return;
  }
  hipLaunchKernelGGL(k_pso_outer, dim3(max(1, min(256, (a.n_swivel + 63) / 64)), (n_particles + 255) / 256), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_pso_outer_sum, dim3((n_particles + 63) / 64), dim3(64), 0, s, a, n_particles);
}
void launch_grid_cell_ids(const float* x, const float* y, const float* z, int n, const GridDev& gd, int* cell_of, int* cell_count,
                          hipStream_t s) {
  hipLaunchKernelGGL(k_grid_cell_ids, dim3((n + 255) / 256), dim3(256), 0, s, x, y, z, n, gd, cell_of, cell_count);
}

}  // namespace hop
