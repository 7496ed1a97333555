// hop_sdf.h -- triangle meshes for the physics row: what SDFchecker keeps per registered mesh
// (SDFchecker.cpp:36-78) plus the precomputed pieces of igl::signed_distance's pseudonormal branch
// (igl/signed_distance.cpp:88-98: per_face_normals, per_vertex_normals with angle weights, per_edge_normals with
// uniform weights -- libigl recomputes them on every call), and a 4-wide bounding-volume tree that replaces
// libigl's AABB tree (igl/AABB.cpp:360-452) for the closest-face search.
//
// Layout on the device (one registered mesh):
//   tri   [slots][3] float4   corner A, B, C of the face in tree-leaf order; A.w = 1 when doublearea > 1e-4
//                             (pseudonormal_test.cpp:50-62), B.w = original face index (as int bits)
//   nrm   [slots][7] float4   face normal, the three edge pseudonormals (edge opposite corner 0,1,2), the three
//                             vertex pseudonormals (corner 0,1,2); .w unused
//   nodes [n_nodes]           4 children per node: boxes as SoA (lo/hi x 3 axes x 4 lanes), child ids
//   leaf_first [n_leaves+1]   slot range of every leaf
//   leaf_obb [n_leaves][3]    float4: (u, half thickness), (v, lo_u), (hi_u, lo_v, hi_v, centre along u x v); see sdf_walk_step
//   order [2^(depth+1)]       libigl's own AABB tree (AABB.cpp:106-200: one face per leaf), boxes only, in heap order
//   face_leaf [n_faces]       heap index of the leaf of `order` that holds each face
// A query evaluates, for every face whose box lower bound does not exceed the best squared distance so far (plus the
// float error of that expression), exactly the float expression of igl/point_simplex_squared_distance.cpp:43-113 and
// keeps the minimum.  Faces at exactly the same float distance (a closest point on a shared edge or vertex: 20-30 % of
// all queries) are ordered as libigl's depth-first walk would meet them (AABB.cpp:391-449) -- it reports the first --
// by looking at the two children of their lowest common ancestor in `order`: the choice matters because the sign test
// of pseudonormal_test.cpp falls back to that face's normal.
#ifndef HOP_SDF_H_
#define HOP_SDF_H_

#include "hop_math.h"

#include <stdint.h>

namespace hop {

constexpr int SDF_LEAF = 8;       // faces per leaf at most (measured 1/2/4/6/8/12/16: 6 and 8 are the fastest)
constexpr int SDF_STACK = 32;     // traversal stack entries per thread: a 4-wide tree of depth D (leaves included) needs at most 3 D + 1 (the build checks it)
constexpr int SDF_NO_CHILD = -1;

struct SdfNode {
  float lo[3][4];
  float hi[3][4];
  int child[4];  // stack id of the child: inner node index, SDF_LEAF_BASE + leaf index, or SDF_NO_CHILD (lo = +inf)
  int pad[4];
};
static_assert(sizeof(SdfNode) == 128, "one node is one 128-byte line");

struct SdfOrderNode {  // stored in heap order: root at 1, children of i at 2 i (left) and 2 i + 1 (right)
  float lo[3];
  int pad0;
  float hi[3];
  int pad1;
};
static_assert(sizeof(SdfOrderNode) == 32, "two nodes per 64 bytes");

struct SdfMeshDev {
  const float4* tri;
  const float4* nrm;
  const SdfNode* nodes;
  const SdfOrderNode* order;
  const int* face_leaf;
  const int* leaf_first;  // n_leaves + 1: slots of leaf l are [leaf_first[l], leaf_first[l + 1])
  const float4* leaf_obb;  // n_leaves x 3: an oriented box around the leaf's faces (see sdf_walk_step), w of the first = 0: none
  int n_faces, n_nodes;
  float coord_eps;  // 4e-7 * largest |coordinate| of the mesh: float error scale of a closest point
  // face cells (meshes of 256 .. 65536 faces): a voxel grid around the mesh; cell_start[v] .. cell_start[v + 1] index
  // cell_faces, the slots of every face that can be the closest one (or tie for it) for some query inside voxel v.
  // cell_start[v] < 0 marks a voxel whose list was not built (too many candidates): the tree walk answers there.
  const int* cell_start;
  const int* cell_faces;
  float cell_ox, cell_oy, cell_oz, cell_inv;
  int cell_nx, cell_ny, cell_nz;
  // SDFchecker::transformMesh without touching the vertices: queries are moved by the inverse of the mesh's current pose
  // (hop_sdf_set_mesh_pose), so a mesh that only moves rigidly keeps its tree and its face cells
  int has_pose;
  float pose_inv[12];
};

#if defined(__HIPCC__)
// ---------------------------------------------------------------------------------------------- device side
// Ericson's closest point on a triangle, the float expression of point_simplex_squared_distance.cpp:43-113.
// (Evaluating all seven region tests up front and sharing one division between the edge regions was tried: most tests
// leave through the first two exits after two to four dot products, and the early exits are cheaper.)
__device__ __forceinline__ V3 sdf_closest_point(V3 p, V3 a, V3 b, V3 c) {
  const V3 ab = b - a, ac = c - a, ap = p - a;
  const float d1 = vdot(ab, ap), d2 = vdot(ac, ap);
  if (d1 <= 0.0f && d2 <= 0.0f) return a;
  const V3 bp = p - b;
  const float d3 = vdot(ab, bp), d4 = vdot(ac, bp);
  if (d3 >= 0.0f && d4 <= d3) return b;
  const float vc = d1 * d4 - d3 * d2;
  if (!(a.x == b.x && a.y == b.y && a.z == b.z)) {
    if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
      const float v = d1 / (d1 - d3);
      return a + v * ab;
    }
  }
  const V3 cp = p - c;
  const float d5 = vdot(ab, cp), d6 = vdot(ac, cp);
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
  const float denom = (float)(1.0 / (double)((va + vb) + vc));  // `Scalar denom = 1.0 / (va + vb + vc)`
  const float v = vb * denom;
  const float w = vc * denom;
  return (a + ab * v) + ac * w;
}

// doublearea.cpp:75-109,144-199: Kahan's Heron formula on the sorted float edge lengths, in double
HOP_HD double sdf_doublearea(V3 A, V3 B, V3 C) {
  double l0 = (double)vnorm(B - C), l1 = (double)vnorm(C - A), l2 = (double)vnorm(A - B), t;
  if (l0 < l1) t = l0, l0 = l1, l1 = t;
  if (l1 < l2) t = l1, l1 = l2, l2 = t;
  if (l0 < l1) t = l0, l0 = l1, l1 = t;
  const double arg = (l0 + (l1 + l2)) * (l2 - (l0 - l1)) * (l2 + (l0 - l1)) * (l0 + (l1 - l2));
  return 2.0 * 0.25 * sqrt(arg);
}

// pseudonormal_test.cpp:24-128 for the face in `slot` with closest point c; returns +1 / -1
__device__ inline float sdf_sign(const SdfMeshDev& m, int slot, V3 q, V3 c) {
  const float4 A4 = m.tri[3 * slot], B4 = m.tri[3 * slot + 1], C4 = m.tri[3 * slot + 2];
  const V3 A = v3(A4.x, A4.y, A4.z), B = v3(B4.x, B4.y, B4.z), C = v3(C4.x, C4.y, C4.z);
  const float4* N = m.nrm + 7 * (size_t)slot;
  int pick = 0;  // 0 face, 1..3 edge opposite corner pick-1, 4..6 vertex pick-4
  const float epsf = (float)1e-12;
  if (A4.w != 0.f) {
    // barycentric_coordinates.cpp:88-100
    const V3 v0 = B - A, v1 = C - A, v2 = c - A;
    const float d00 = vdot(v0, v0), d01 = vdot(v0, v1), d11 = vdot(v1, v1), d20 = vdot(v2, v0), d21 = vdot(v2, v1);
    const float denom = d00 * d11 - d01 * d01;
    float b[3];
    b[1] = (d11 * d20 - d01 * d21) / denom;
    b[2] = (d00 * d21 - d01 * d20) / denom;
    b[0] = 1.0f - (b[1] + b[2]);
    const int type = (b[0] <= epsf) + (b[1] <= epsf) + (b[2] <= epsf);
    if (type == 2) {
      pick = b[0] > epsf ? 4 : (b[1] > epsf ? 5 : (b[2] > epsf ? 6 : 0));
    } else if (type == 1) {
      pick = b[0] <= epsf ? 1 : (b[1] <= epsf ? 2 : 3);
    }
  } else {
    const V3 P[3] = {A, B, C};
    bool found = false;
    for (int v = 0; v < 3 && !found; ++v)
      if ((double)vnorm(c - P[v]) < 1e-12) found = true, pick = 4 + v;
    for (int e = 0; e < 3 && !found; ++e) {
      // project_to_line.cpp:36-55, project_to_line_segment.cpp:27-42 (t, sqrD double; vectors float)
      const V3 s = P[(e + 1) % 3], d = P[(e + 2) % 3];
      const V3 DmS = d - s;
      const double v_sqrlen = (double)vsqn(DmS);
      const V3 SmP = s - c;
      const float px = DmS.x * SmP.x, py = DmS.y * SmP.y, pz = DmS.z * SmP.z;
      const double t = (double)(-(px + (py + pz))) / v_sqrlen;
      const V3 projP = ((float)(1 - t)) * s + ((float)t) * d;
      double sqrD = (double)vsqn(c - projP);
      if (t < 0)
        sqrD = (double)vsqn(c - s);
      else if (t > 1)
        sqrD = (double)vsqn(c - d);
      if (sqrt(sqrD) < 1e-12) found = true, pick = 1 + e;
    }
  }
  const float4 n = N[pick];
  return vdot(q - c, v3(n.x, n.y, n.z)) >= 0 ? 1.f : -1.f;
}

// Eigen::AlignedBox::contains / squaredExteriorDistance as AABB.cpp:413-426 evaluates them
__device__ __forceinline__ bool sdf_box_contains(const SdfOrderNode& n, V3 p) {
  return n.lo[0] <= p.x && n.lo[1] <= p.y && n.lo[2] <= p.z && p.x <= n.hi[0] && p.y <= n.hi[1] && p.z <= n.hi[2];
}
__device__ __forceinline__ float sdf_box_ext(const SdfOrderNode& n, V3 p) {
  float d2 = 0.f;
  const float pv[3] = {p.x, p.y, p.z};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (n.lo[k] > pv[k]) {
      const float a = n.lo[k] - pv[k];
      d2 += a * a;
    } else if (pv[k] > n.hi[k]) {
      const float a = pv[k] - n.hi[k];
      d2 += a * a;
    }
  }
  return d2;
}
// true when libigl's walk for query p reaches face f before face g (f != g): decided at their lowest common ancestor,
// found from the heap indices of the two leaves
__device__ inline bool sdf_precedes(const SdfMeshDev& m, V3 p, int f, int g) {
  unsigned a = (unsigned)m.face_leaf[f], b = (unsigned)m.face_leaf[g];
  const int da = 31 - __clz(a), db = 31 - __clz(b);
  if (da > db) a >>= (da - db);
  else b >>= (db - da);
  const int k = 32 - __clz(a ^ b);  // low bits below the common ancestor
  a >>= (k - 1);
  const bool a_left = (a & 1u) == 0u;
  const SdfOrderNode L = m.order[a & ~1u], R = m.order[a | 1u];
  bool left_first;
  if (sdf_box_contains(L, p)) left_first = true;
  else if (sdf_box_contains(R, p)) left_first = false;
  else left_first = sdf_box_ext(L, p) < sdf_box_ext(R, p);
  return left_first == a_left;
}

#ifdef SDF_COUNT  // measurement build (tools/physics_profile.py --counters): nodes visited, faces tested, exact ties, queries
__device__ unsigned long long g_sdf_cnt[4];
#endif
struct SdfHit {
  float sqr_d;
  int slot;  // -1: no face (empty mesh)
  int face;
  V3 c;
};

// ---- exact closest face (see the header comment) as a resumable walk: one call of sdf_walk_step handles one stack
// entry, either an inner node (four child boxes) or a leaf (up to SDF_LEAF faces), so that a kernel can keep every lane
// busy with its own query and hand a lane its next query as soon as its stack runs empty.
// `stack`: SDF_STACK entries of this thread's own storage, entry k at stack[k * stride] (LDS, lane-interleaved so that
// equal depths of a wavefront fall into different banks).  An entry packs an id (low SDF_ID_BITS bits: inner node index,
// or SDF_LEAF_BASE + leaf index) with the top 14 bits of the entry's lower bound (8 exponent + 6 mantissa bits, rounded
// towards zero, i.e. still a lower bound).
constexpr int SDF_ID_BITS = 18;
constexpr unsigned SDF_LEAF_BASE = 1u << (SDF_ID_BITS - 1);
__device__ __forceinline__ unsigned sdf_pack(float lb, unsigned id) {
  return ((__float_as_uint(lb) >> 17) << SDF_ID_BITS) | id;
}
__device__ __forceinline__ float sdf_unpack_bound(unsigned e) { return __uint_as_float((e >> SDF_ID_BITS) << 17); }

struct SdfWalk {
  V3 q;
  SdfHit h;
  float thr, de;
  int sp;
};
__device__ __forceinline__ void sdf_walk_begin(SdfWalk& w, const SdfMeshDev& m, V3 q, unsigned* stack, int stride) {
  w.q = q;
  w.h.sqr_d = __builtin_inff(), w.h.slot = -1, w.h.face = 0x7fffffff, w.h.c = v3(0, 0, 0);
  w.thr = __builtin_inff();  // prune bound: best + float slack of the triangle expression
  w.de = m.coord_eps + 4e-7f * fmaxf(fabsf(q.x), fmaxf(fabsf(q.y), fabsf(q.z)));
  w.sp = 0;
  if (m.n_faces > 0) stack[(w.sp++) * stride] = sdf_pack(0.f, 0u);
#ifdef SDF_COUNT
  atomicAdd(&g_sdf_cnt[3], 1ull);
#endif
}
__device__ inline void sdf_walk_step(SdfWalk& w, const SdfMeshDev& m, unsigned* stack, int stride) {
  const unsigned e = stack[(--w.sp) * stride];
  if (sdf_unpack_bound(e) > w.thr) return;
  const unsigned id = e & ((1u << SDF_ID_BITS) - 1u);
  if (id >= SDF_LEAF_BASE) {
    // A leaf is a small, nearly flat patch; its axis-aligned box is several times thicker than the patch unless the
    // patch happens to be axis aligned, and lets the leaf through for queries a few centimetres away (measured: 63 faces
    // tested per query where 3 are at the minimum distance).  A second test against a box in the patch's own frame
    // (unit axes u, v and their cross product; extents of the vertices) rejects most of those before any face is read.
    const float4 o0 = m.leaf_obb[3 * (id - SDF_LEAF_BASE)];
    if (o0.w != 0.f) {
      const float4 o1 = m.leaf_obb[3 * (id - SDF_LEAF_BASE) + 1], o2 = m.leaf_obb[3 * (id - SDF_LEAF_BASE) + 2];
      const V3 u = v3(o0.x, o0.y, o0.z), v = v3(o1.x, o1.y, o1.z);
      const float su = vdot(u, w.q), sv = vdot(v, w.q), sn = vdot(vcross(u, v), w.q);
      const float du = fmaxf(fmaxf(o1.w - su, su - o2.x), 0.f);
      const float dv = fmaxf(fmaxf(o2.y - sv, sv - o2.z), 0.f);
      const float dn = fmaxf(fabsf(sn - o2.w) - o0.w, 0.f);  // |sn - centre_n| - half thickness
      if (du * du + dv * dv + dn * dn > w.thr) return;
    }
    const int first = m.leaf_first[id - SDF_LEAF_BASE], end = m.leaf_first[id - SDF_LEAF_BASE + 1];
    for (int s = first; s < end; ++s) {
      const float4 A4 = m.tri[3 * s], B4 = m.tri[3 * s + 1], C4 = m.tri[3 * s + 2];
      const V3 c = sdf_closest_point(w.q, v3(A4.x, A4.y, A4.z), v3(B4.x, B4.y, B4.z), v3(C4.x, C4.y, C4.z));
      const float d = vsqn(w.q - c);
      const int face = __float_as_int(B4.w);
#ifdef SDF_COUNT
      atomicAdd(&g_sdf_cnt[1], 1ull);
      if (d == w.h.sqr_d) atomicAdd(&g_sdf_cnt[2], 1ull);
#endif
      if (d < w.h.sqr_d || (d == w.h.sqr_d && sdf_precedes(m, w.q, face, w.h.face))) {
        w.h.sqr_d = d, w.h.slot = s, w.h.face = face, w.h.c = c;
        w.thr = d + (2.f * sqrtf(d) * w.de + w.de * w.de) + d * 1e-5f;
      }
    }
    return;
  }
  const SdfNode& nd = m.nodes[id];
#ifdef SDF_COUNT
  atomicAdd(&g_sdf_cnt[0], 1ull);
#endif
  const int child[4] = {nd.child[0], nd.child[1], nd.child[2], nd.child[3]};
  float lb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dx = fmaxf(fmaxf(nd.lo[0][k] - w.q.x, w.q.x - nd.hi[0][k]), 0.f);
    const float dy = fmaxf(fmaxf(nd.lo[1][k] - w.q.y, w.q.y - nd.hi[1][k]), 0.f);
    const float dz = fmaxf(fmaxf(nd.lo[2][k] - w.q.z, w.q.z - nd.hi[2][k]), 0.f);
    lb[k] = dx * dx + dy * dy + dz * dz;  // +inf for an empty child (lo = +inf)
  }
  // children that can still matter, pushed farthest first so that the nearest is popped first
  // (5-exchange sorting network on registers; key -1 marks a child that is not pushed)
  float k0 = (child[0] != SDF_NO_CHILD && lb[0] <= w.thr) ? lb[0] : -1.f;
  float k1 = (child[1] != SDF_NO_CHILD && lb[1] <= w.thr) ? lb[1] : -1.f;
  float k2 = (child[2] != SDF_NO_CHILD && lb[2] <= w.thr) ? lb[2] : -1.f;
  float k3 = (child[3] != SDF_NO_CHILD && lb[3] <= w.thr) ? lb[3] : -1.f;
  int c0 = child[0], c1 = child[1], c2 = child[2], c3 = child[3];
#define SDF_CX(ka, ca, kb, cb)      \
  if (ka < kb) {                    \
    const float tk = ka;            \
    ka = kb, kb = tk;               \
    const int tc = ca;              \
    ca = cb, cb = tc;               \
  }
  SDF_CX(k0, c0, k1, c1) SDF_CX(k2, c2, k3, c3) SDF_CX(k0, c0, k2, c2) SDF_CX(k1, c1, k3, c3) SDF_CX(k1, c1, k2, c2)
#undef SDF_CX
  if (k0 >= 0.f) stack[(w.sp++) * stride] = sdf_pack(k0, (unsigned)c0);
  if (k1 >= 0.f) stack[(w.sp++) * stride] = sdf_pack(k1, (unsigned)c1);
  if (k2 >= 0.f) stack[(w.sp++) * stride] = sdf_pack(k2, (unsigned)c2);
  if (k3 >= 0.f) stack[(w.sp++) * stride] = sdf_pack(k3, (unsigned)c3);
}

// igl::signed_distance for one point with the bounds SDFchecker passes (lower = -FLT_MAX, upper = FLT_MAX:
// up_sqr_d = +inf, low_sqr_d = 0, signed_distance.cpp:117-156): NaN for a point at distance zero.
__device__ inline float sdf_walk_finish(const SdfWalk& w, const SdfMeshDev& m, int* face_out) {
  const SdfHit& h = w.h;
  if (h.slot < 0 || !(h.sqr_d > 0.f) || !(h.sqr_d < __builtin_inff())) {
    if (face_out) *face_out = m.n_faces + 1;
    return __builtin_nanf("");
  }
  if (face_out) *face_out = h.face;
  return sdf_sign(m, h.slot, w.q, h.c) * sqrtf(h.sqr_d);
}
// the same answer from the face cells when the query lies in a voxel with a list: every face that can attain the minimum
// for a query of that voxel is in the list (see k_face_cells in hop_physics.hip), so the scan below is the exhaustive scan
// restricted to the faces that matter, with the same tie rule
__device__ inline bool sdf_cells_closest(SdfWalk& w, const SdfMeshDev& m) {
  if (m.cell_start == nullptr) return false;
  const float fx = (w.q.x - m.cell_ox) * m.cell_inv, fy = (w.q.y - m.cell_oy) * m.cell_inv, fz = (w.q.z - m.cell_oz) * m.cell_inv;
  if (!(fx >= 0.f && fy >= 0.f && fz >= 0.f && fx < (float)m.cell_nx && fy < (float)m.cell_ny && fz < (float)m.cell_nz)) return false;
  const int v = ((int)fz * m.cell_ny + (int)fy) * m.cell_nx + (int)fx;
  const int b = m.cell_start[v];
  if (b < 0) return false;
  const int e = m.cell_start[v + 1] < 0 ? ~m.cell_start[v + 1] : m.cell_start[v + 1];
  for (int i = b; i < e; ++i) {
    const int s = m.cell_faces[i];
    const float4 A4 = m.tri[3 * s], B4 = m.tri[3 * s + 1], C4 = m.tri[3 * s + 2];
    const V3 c = sdf_closest_point(w.q, v3(A4.x, A4.y, A4.z), v3(B4.x, B4.y, B4.z), v3(C4.x, C4.y, C4.z));
    const float d = vsqn(w.q - c);
    const int face = __float_as_int(B4.w);
#ifdef SDF_COUNT
    atomicAdd(&g_sdf_cnt[1], 1ull);
    if (d == w.h.sqr_d) atomicAdd(&g_sdf_cnt[2], 1ull);
#endif
    if (d < w.h.sqr_d || (d == w.h.sqr_d && sdf_precedes(m, w.q, face, w.h.face))) w.h.sqr_d = d, w.h.slot = s, w.h.face = face, w.h.c = c;
  }
  return true;
}
__device__ inline float sdf_signed_distance(const SdfMeshDev& m, V3 q, unsigned* stack, int stride, int* face_out) {
  if (m.has_pose) q = m4_point(m.pose_inv, q);
  SdfWalk w;
  sdf_walk_begin(w, m, q, stack, stride);
  if (!sdf_cells_closest(w, m))
    while (w.sp > 0) sdf_walk_step(w, m, stack, stride);
  return sdf_walk_finish(w, m, face_out);
}
#endif  // __HIPCC__

}  // namespace hop
#endif  // HOP_SDF_H_
