// hop_physics.hip -- SURVEY.md 8(f) row N1: PoseEstimator::rejectByCollisionOrNonTouching on the GPU
// (src/perception/src/PoseEstimator.cpp:524-735) with its SDFchecker (SDFchecker.cpp:36-134, libigl's
// signed_distance with pseudonormal signs) and the voxel grid it applies to the scene (Utils.cpp:334-340).
//
// What runs where
//   * mesh registration (once per object, once per frame for the four finger links): the host transforms the vertices,
//     computes the face / edge / vertex pseudonormals in libigl's operation order and builds the 4-wide box tree
//     (hop_sdf.h); a few thousand faces, the counterpart of reading the OBJ file in the reference.
//   * everything per point and per hypothesis runs in the kernels below: the voxel grid, the two nearest-point
//     searches of the object centre, finger-cloud-to-object and object-to-finger-mesh signed distances, the decisions.
// The reference moves the object mesh into the hand-base frame for every hypothesis (and libigl rebuilds its tree and
// normals on every call); here the query points are moved into the mesh's frame instead, so every mesh is static.
// Signed distances therefore agree with the oracle to float rounding of the rigid motion (about 1e-7 m), not bit for
// bit; with a mesh and points given in the same frame (hop_sdf_signed_distance) they are bit-equal.
#include "../../include/hop.h"
#include "hop_ctx_ext.h"
#include "hop_sdf.h"

#include "hop_prim.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

using namespace hop;

namespace {

#define PHCHK(ctx, call)                                                              \
  do {                                                                                \
    hipError_t _e = (call);                                                           \
    if (_e != hipSuccess) {                                                           \
      hop_ctx_set_error((ctx), std::string(#call) + ": " + hipGetErrorString(_e));    \
      return HOP_E_HIP;                                                               \
    }                                                                                 \
  } while (0)

constexpr int SDF_BLOCK = 128;
constexpr int MAX_MESHES = HOP_SDF_MAX_MESHES;

// ------------------------------------------------------------------------------------------------ host: mesh preparation
struct HostMesh {
  int nv = 0, nf = 0;
  std::vector<V3> V, FN, VN, EN;
  std::vector<int> F, EMAP;
  float max_abs = 0;
};

// SDFchecker::transformVertices (SDFchecker.cpp:21-33) then the normals of igl/signed_distance.cpp:88-98
void prepare_mesh(HostMesh& m, const float* V, int nv, const int32_t* F, int nf, const float* pose) {
  m.nv = nv, m.nf = nf;
  m.V.resize(nv);
  m.max_abs = 0;
  for (int i = 0; i < nv; ++i) {
    const V3 p = v3(V[3 * i], V[3 * i + 1], V[3 * i + 2]);
    m.V[i] = pose ? m4_point(pose, p) : p;
    m.max_abs = std::max(m.max_abs, std::max(std::fabs(m.V[i].x), std::max(std::fabs(m.V[i].y), std::fabs(m.V[i].z))));
  }
  m.F.assign(F, F + 3 * (size_t)nf);
  // per_face_normals.cpp:22-37
  m.FN.resize(nf);
  for (int i = 0; i < nf; ++i) {
    const V3 a = m.V[F[3 * i]], b = m.V[F[3 * i + 1]], c = m.V[F[3 * i + 2]];
    const V3 n = vcross(b - a, c - a);
    const float r = vnorm(n);
    m.FN[i] = r == 0 ? v3(0, 0, 0) : n / r;
  }
  // per_vertex_normals.cpp:69-107, angle weights (internal_angles.cpp:75-86 on squared_edge_lengths.cpp:36-40)
  m.VN.assign(nv, v3(0, 0, 0));
  for (int i = 0; i < nf; ++i) {
    const V3 a = m.V[F[3 * i]], b = m.V[F[3 * i + 1]], c = m.V[F[3 * i + 2]];
    const float L[3] = {vsqn(b - c), vsqn(c - a), vsqn(a - b)};
    for (int j = 0; j < 3; ++j) {
      const float s1 = L[j], s2 = L[(j + 1) % 3], s3 = L[(j + 2) % 3];
      const float w = (float)std::acos((double)((s3 + s2) - s1) / (2. * std::sqrt(s3 * s2)));
      V3& n = m.VN[F[3 * i + j]];
      n = n + w * m.FN[i];
    }
  }
  for (int v = 0; v < nv; ++v) m.VN[v] = vnormalized(m.VN[v]);
  // per_edge_normals.cpp:36-78, uniform weights, not normalised
  std::map<std::pair<int, int>, int> ids;
  m.EMAP.assign(3 * (size_t)nf, 0);
  for (int c = 0; c < 3; ++c)
    for (int f = 0; f < nf; ++f) {
      int u = F[3 * f + (c + 1) % 3], v = F[3 * f + (c + 2) % 3];
      if (u > v) std::swap(u, v);
      auto it = ids.emplace(std::make_pair(u, v), (int)ids.size()).first;
      m.EMAP[(size_t)c * nf + f] = it->second;
    }
  m.EN.assign(ids.size(), v3(0, 0, 0));
  for (int f = 0; f < nf; ++f)
    for (int c = 0; c < 3; ++c) {
      V3& n = m.EN[m.EMAP[(size_t)c * nf + f]];
      n = n + m.FN[f];
    }
}

struct TreeBuilder {
  const HostMesh& m;
  std::vector<int> order;      // faces, permuted in place
  std::vector<V3> cen, lo, hi; // per face
  std::vector<SdfNode> nodes;
  std::vector<int> slot_face;  // leaf order
  std::vector<int> leaf_first; // slot range of every leaf (n_leaves + 1 entries)
  std::vector<float4> leaf_obb; // 3 per leaf (hop_sdf.h)
  int max_depth = 0;

  explicit TreeBuilder(const HostMesh& mesh) : m(mesh) {
    const int nf = m.nf;
    order.resize(nf), cen.resize(nf), lo.resize(nf), hi.resize(nf);
    for (int f = 0; f < nf; ++f) {
      order[f] = f;
      const V3 a = m.V[m.F[3 * f]], b = m.V[m.F[3 * f + 1]], c = m.V[m.F[3 * f + 2]];
      lo[f] = v3(std::min(a.x, std::min(b.x, c.x)), std::min(a.y, std::min(b.y, c.y)), std::min(a.z, std::min(b.z, c.z)));
      hi[f] = v3(std::max(a.x, std::max(b.x, c.x)), std::max(a.y, std::max(b.y, c.y)), std::max(a.z, std::max(b.z, c.z)));
      cen[f] = v3(0.5f * (lo[f].x + hi[f].x), 0.5f * (lo[f].y + hi[f].y), 0.5f * (lo[f].z + hi[f].z));
    }
  }
  static float comp(V3 v, int ax) { return ax == 0 ? v.x : (ax == 1 ? v.y : v.z); }
  // median split of order[b,e) along the longest axis of the centroid bounds; returns the split position
  int split(int b, int e) {
    V3 mn = cen[order[b]], mx = mn;
    for (int i = b + 1; i < e; ++i) {
      const V3 c = cen[order[i]];
      mn = v3(std::min(mn.x, c.x), std::min(mn.y, c.y), std::min(mn.z, c.z));
      mx = v3(std::max(mx.x, c.x), std::max(mx.y, c.y), std::max(mx.z, c.z));
    }
    const V3 ext = mx - mn;
    const int ax = ext.x >= ext.y && ext.x >= ext.z ? 0 : (ext.y >= ext.z ? 1 : 2);
    const int mid = b + (e - b) / 2;
    std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e, [&](int f, int g) {
      const float cf = comp(cen[f], ax), cg = comp(cen[g], ax);
      return cf < cg || (cf == cg && f < g);
    });
    return mid;
  }
  void range_box(int b, int e, V3* blo, V3* bhi) const {
    V3 mn = lo[order[b]], mx = hi[order[b]];
    for (int i = b + 1; i < e; ++i) {
      const V3 l = lo[order[i]], h = hi[order[i]];
      mn = v3(std::min(mn.x, l.x), std::min(mn.y, l.y), std::min(mn.z, l.z));
      mx = v3(std::max(mx.x, h.x), std::max(mx.y, h.y), std::max(mx.z, h.z));
    }
    *blo = mn, *bhi = mx;
  }
  // oriented box of a leaf: n = area-weighted mean normal, u = direction of the longest edge made orthogonal to n,
  // v = n x u rebuilt in float as the kernel does (vcross(u, v) is then the normal axis); extents of the vertices,
  // widened by the float error of the projections.  No box (w = 0) when the patch has no normal or when the oriented box
  // is not clearly smaller than the axis-aligned one (flat, axis-aligned patches).
  void leaf_box(int b, int e, V3 blo, V3 bhi) {
    float4 o[3] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    double nx = 0, ny = 0, nz = 0, best_len = -1;
    V3 edge = v3(1, 0, 0);
    for (int i = b; i < e; ++i) {
      const int f = order[i];
      const V3 P[3] = {m.V[m.F[3 * f]], m.V[m.F[3 * f + 1]], m.V[m.F[3 * f + 2]]};
      const V3 n = vcross(P[1] - P[0], P[2] - P[0]);
      nx += n.x, ny += n.y, nz += n.z;
      for (int c = 0; c < 3; ++c) {
        const V3 d = P[(c + 1) % 3] - P[c];
        const double l = (double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z;
        if (l > best_len) best_len = l, edge = d;
      }
    }
    const double nl = std::sqrt(nx * nx + ny * ny + nz * nz);
    if (nl > 0 && best_len > 0) {
      const double n[3] = {nx / nl, ny / nl, nz / nl};
      double u[3] = {edge.x, edge.y, edge.z};
      const double un = u[0] * n[0] + u[1] * n[1] + u[2] * n[2];
      for (int k = 0; k < 3; ++k) u[k] -= un * n[k];
      const double ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
      if (ul > 1e-12) {
        const V3 uf = v3((float)(u[0] / ul), (float)(u[1] / ul), (float)(u[2] / ul));
        // v = n x u in double, then float; the kernel's third axis is vcross(uf, vf)
        const double vd[3] = {n[1] * u[2] / ul - n[2] * u[1] / ul, n[2] * u[0] / ul - n[0] * u[2] / ul, n[0] * u[1] / ul - n[1] * u[0] / ul};
        const V3 vf = v3((float)vd[0], (float)vd[1], (float)vd[2]);
        const V3 nf = vcross(uf, vf);
        float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = b; i < e; ++i)
          for (int c = 0; c < 3; ++c) {
            const V3 p = m.V[m.F[3 * order[i] + c]];
            const float s3[3] = {vdot(uf, p), vdot(vf, p), vdot(nf, p)};
            for (int a = 0; a < 3; ++a) lo3[a] = std::min(lo3[a], s3[a]), hi3[a] = std::max(hi3[a], s3[a]);
          }
        const float pad = 4e-6f * (m.max_abs + 1e-3f);
        for (int a = 0; a < 3; ++a) lo3[a] -= pad, hi3[a] += pad;
        const double vol_o = (double)(hi3[0] - lo3[0]) * (hi3[1] - lo3[1]) * (hi3[2] - lo3[2]);
        const double vol_a = (double)(bhi.x - blo.x + 2 * pad) * (bhi.y - blo.y + 2 * pad) * (bhi.z - blo.z + 2 * pad);
        if (vol_o < 0.6 * vol_a) {
          const float half = 0.5f * (hi3[2] - lo3[2]) + pad, centre = 0.5f * (hi3[2] + lo3[2]);
          o[0] = make_float4(uf.x, uf.y, uf.z, half);
          o[1] = make_float4(vf.x, vf.y, vf.z, lo3[0]);
          o[2] = make_float4(hi3[0], lo3[1], hi3[1], centre);
        }
      }
    }
    leaf_obb.push_back(o[0]), leaf_obb.push_back(o[1]), leaf_obb.push_back(o[2]);
  }
  // builds the node for order[b,e) (e - b > SDF_LEAF or the root) and returns its index
  int build(int b, int e, int depth) {
    max_depth = std::max(max_depth, depth);
    const int idx = (int)nodes.size();
    nodes.emplace_back();
    int cuts[5], nc = 0;
    cuts[0] = b;
    if (e - b <= SDF_LEAF) {
      cuts[1] = e, nc = 1;
    } else {
      const int mid = split(b, e);
      int parts[3] = {b, mid, e};
      nc = 0;
      for (int h = 0; h < 2; ++h) {
        const int pb = parts[h], pe = parts[h + 1];
        if (pe - pb > SDF_LEAF) {
          const int q = split(pb, pe);
          cuts[nc++] = pb, cuts[nc++] = q;
        } else
          cuts[nc++] = pb;
      }
      cuts[nc] = e;
    }
    SdfNode nd;
    std::memset(&nd, 0, sizeof(nd));
    for (int k = 0; k < 4; ++k) {
      for (int a = 0; a < 3; ++a) nd.lo[a][k] = INFINITY, nd.hi[a][k] = -INFINITY;
      nd.child[k] = SDF_NO_CHILD;
    }
    for (int k = 0; k < nc; ++k) {
      const int cb = cuts[k], ce = cuts[k + 1];
      if (ce <= cb) continue;
      V3 blo, bhi;
      range_box(cb, ce, &blo, &bhi);
      nd.lo[0][k] = blo.x, nd.lo[1][k] = blo.y, nd.lo[2][k] = blo.z;
      nd.hi[0][k] = bhi.x, nd.hi[1][k] = bhi.y, nd.hi[2][k] = bhi.z;
      if (ce - cb <= SDF_LEAF) {
        max_depth = std::max(max_depth, depth + 1);
        nd.child[k] = (int)(SDF_LEAF_BASE + (unsigned)leaf_first.size());
        leaf_first.push_back((int)slot_face.size());
        std::sort(order.begin() + cb, order.begin() + ce);
        leaf_box(cb, ce, blo, bhi);
        for (int i = cb; i < ce; ++i) slot_face.push_back(order[i]);
      } else {
        nd.child[k] = build(cb, ce, depth + 1);
      }
    }
    nodes[idx] = nd;
    return idx;
  }
};

// libigl's AABB tree over the faces (AABB.cpp:73-200), nodes in heap order (root 1, children 2 i and 2 i + 1; the
// median split leaves ceil(n/2) faces on the left, so the depth is ceil(log2 n)): per-axis ranks of the face barycentres (igl::sort is a std::sort of
// an index map, sort.cpp:287-316), nodes split at the median rank along the longest box axis, one face per leaf.  Only
// what sdf_precedes needs is kept.
struct OrderTree {
  const HostMesh& m;
  std::vector<SdfOrderNode> nodes;
  std::vector<int> face_leaf, rank;  // rank[3 f + d]
  explicit OrderTree(const HostMesh& mesh) : m(mesh) {
    const int nf = m.nf;
    face_leaf.assign(nf, -1);
    rank.resize(3 * (size_t)nf);
    std::vector<float> bc(nf);
    std::vector<size_t> idx(nf);
    for (int d = 0; d < 3; ++d) {
      for (int f = 0; f < nf; ++f) {
        const V3 a = m.V[m.F[3 * f]], b = m.V[m.F[3 * f + 1]], c = m.V[m.F[3 * f + 2]];
        bc[f] = d == 0 ? ((a.x + b.x) + c.x) / 3.0f : (d == 1 ? ((a.y + b.y) + c.y) / 3.0f : ((a.z + b.z) + c.z) / 3.0f);
        idx[f] = f;
      }
      std::sort(idx.begin(), idx.end(), [&](size_t i, size_t j) { return bc[i] < bc[j]; });
      for (int i = 0; i < nf; ++i) rank[3 * idx[i] + d] = i;
    }
    if (nf > 0) {
      std::vector<int> all(nf);
      for (int f = 0; f < nf; ++f) all[f] = f;
      grow(all, 1);
    }
  }
  void grow(const std::vector<int>& I, size_t heap) {
    if (heap >= nodes.size()) nodes.resize(std::max(heap + 1, 2 * nodes.size()), SdfOrderNode{});
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int f : I)
      for (int c = 0; c < 3; ++c) {
        const V3 v = m.V[m.F[3 * f + c]];
        const float pv[3] = {v.x, v.y, v.z};
        for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], pv[k]), hi[k] = std::max(hi[k], pv[k]);
      }
    SdfOrderNode nd{};
    for (int k = 0; k < 3; ++k) nd.lo[k] = lo[k], nd.hi[k] = hi[k];
    nodes[heap] = nd;
    if (I.size() == 1) {
      face_leaf[I[0]] = (int)heap;
      return;
    }
    int ax = 0;
    for (int k = 1; k < 3; ++k)
      if (hi[k] - lo[k] > hi[ax] - lo[ax]) ax = k;
    std::vector<int> r(I.size());
    for (size_t i = 0; i < I.size(); ++i) r[i] = rank[3 * (size_t)I[i] + ax];
    std::vector<int> t = r;
    const size_t n = (t.size() - 1) / 2;
    std::nth_element(t.begin(), t.begin() + n, t.end());
    const int med = t[n];
    std::vector<int> LI, RI;
    for (size_t i = 0; i < I.size(); ++i) (r[i] <= med ? LI : RI).push_back(I[i]);
    grow(LI, 2 * heap);
    grow(RI, 2 * heap + 1);
  }
};

struct MeshStore {
  DevBuf tri_d, nrm_d, nodes_d, order_d, leaf_d, leaf_first_d, leaf_obb_d, cell_start_d, cell_faces_d;
  SdfMeshDev dev{};
  bool valid = false;
  // what the structures were built from: registering the same triangles again (the object mesh, once per frame in the reference's
  // loop, PoseEstimator.cpp:505-508) finds them standing
  std::vector<float> key_V, key_pose;
  std::vector<int32_t> key_F;
  bool same_as(const float* V, int nv, const int32_t* F, int nf, const float* pose16) const {
    return valid && key_V.size() == 3 * (size_t)nv && key_F.size() == 3 * (size_t)nf && key_pose.size() == (pose16 ? 16u : 0u) &&
           (nv == 0 || std::memcmp(key_V.data(), V, sizeof(float) * 3 * (size_t)nv) == 0) &&
           (nf == 0 || std::memcmp(key_F.data(), F, sizeof(int32_t) * 3 * (size_t)nf) == 0) &&
           (!pose16 || std::memcmp(key_pose.data(), pose16, sizeof(float) * 16) == 0);
  }
  void release() { key_V.clear(), key_F.clear(), key_pose.clear(), tri_d.release(), nrm_d.release(), nodes_d.release(), order_d.release(), leaf_d.release(), leaf_first_d.release(), leaf_obb_d.release(), cell_start_d.release(), cell_faces_d.release(), valid = false; }
};

struct Cloud3 {
  DevBuf buf;  // 3 planes of n
  int n = 0;
  const float* x() const { return buf.as<float>(); }
  const float* y() const { return buf.as<float>() + n; }
  const float* z() const { return buf.as<float>() + 2 * (size_t)n; }
};

struct PhysParams {
  float cam2handbase[16];
  float center_init[3];
  float ob_diameter, collision_dist, inside_ob_dist, non_touch_dist, collision_finger_dist, volume_ratio;
  int finger_status[4];
  int finger_active[4];  // finger cloud tested in the third check
  int finger_off[5];     // ranges of the concatenated finger points
  int object_mesh, finger_mesh[4];
  int n_model;
};

struct Physics : HopExt {
  MeshStore mesh[MAX_MESHES];
  Cloud3 fingers, cwh_ds, hand, model, tmp_cloud, tmp_cloud2, tmp_nrm, tmp_nrm2;
  DevBuf keys, keys_alt, vals, vals_alt, flags, pos, starts, sort_tmp, scalars, xf, stage, fmin, diag, gather, tmp_pose, tmp_score, tmp_id, mats;
  PhysParams P{};
  bool have_frame = false;
  hipEvent_t ev[2] = {nullptr, nullptr};
  double ms_frame = 0, ms_reject = 0;
  long long last_cells_total = 0;
  int last_cells_voxels = 0;
  ~Physics() override {
    for (auto& m : mesh) m.release();
    DevBuf* bufs[] = {&fingers.buf, &cwh_ds.buf, &hand.buf, &model.buf, &tmp_cloud.buf, &tmp_cloud2.buf, &tmp_nrm.buf, &tmp_nrm2.buf, &keys, &keys_alt, &vals, &vals_alt, &flags, &pos,
                      &starts, &sort_tmp, &scalars, &xf, &stage, &fmin, &diag, &gather, &tmp_pose, &tmp_score, &tmp_id, &mats};
    for (DevBuf* b : bufs) b->release();
    for (auto& e : ev)
      if (e) (void)hipEventDestroy(e);
  }
};

Physics* physics(hop_ctx* c) {
  HopExt*& e = hop_ctx_ext(c, HOP_EXT_PHYSICS);
  if (!e) e = new Physics;
  return static_cast<Physics*>(e);
}

// ------------------------------------------------------------------------------------------------ kernels
__device__ __forceinline__ unsigned ordered_bits(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
constexpr unsigned ORDERED_NONE = 0xffffffffu;  // above every finite value and +inf: "no sample"

// S[i] = igl::signed_distance of point i (optionally moved by T first) to the mesh
__global__ void __launch_bounds__(SDF_BLOCK) k_sdf_query(SdfMeshDev m, const float* __restrict__ px, const float* __restrict__ py,
                                                         const float* __restrict__ pz, int n, const float* __restrict__ T,
                                                         float* __restrict__ S, int* __restrict__ face) {
  __shared__ unsigned stk[SDF_STACK * SDF_BLOCK];
  const int i = blockIdx.x * SDF_BLOCK + threadIdx.x;
  if (i >= n) return;
  V3 q = v3(px[i], py[i], pz[i]);
  if (T) q = m4_point(T, q);
  int f;
  S[i] = sdf_signed_distance(m, q, stk + threadIdx.x, SDF_BLOCK, &f);
  if (face) face[i] = f;
}

// out = T * in (pcl::transformPointCloud order), SoA planes
__global__ void k_transform_cloud(const float* __restrict__ ix, const float* __restrict__ iy, const float* __restrict__ iz, int n,
                                  const float* __restrict__ T, float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 p = m4_point(T, v3(ix[i], iy[i], iz[i]));
  ox[i] = p.x, oy[i] = p.y, oz[i] = p.z;
}

// ---- pcl::VoxelGrid (voxel_grid.hpp:214-440, xyz only)
// scal[0..2] = min as ordered bits, [3..5] = max, [6] = finite points
__global__ void k_vox_minmax(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n, unsigned* scal) {
  __shared__ unsigned s[7];
  if (threadIdx.x < 7) s[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
  __syncthreads();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float a = x[i], b = y[i], c = z[i];
    if (!isfinite(a) || !isfinite(b) || !isfinite(c)) continue;
    atomicMin(&s[0], ordered_bits(a)), atomicMin(&s[1], ordered_bits(b)), atomicMin(&s[2], ordered_bits(c));
    atomicMax(&s[3], ordered_bits(a)), atomicMax(&s[4], ordered_bits(b)), atomicMax(&s[5], ordered_bits(c));
    atomicAdd(&s[6], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&scal[threadIdx.x], s[threadIdx.x]);
  else if (threadIdx.x < 6) atomicMax(&scal[threadIdx.x], s[threadIdx.x]);
  else if (threadIdx.x == 6) atomicAdd(&scal[6], s[6]);
}
struct VoxGeom {
  float inv;
  int minb[3];
  int mul[3];
};
__global__ void k_vox_keys(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n, VoxGeom g,
                           unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = x[i], b = y[i], c = z[i];
  unsigned k = 0xffffffffu;
  if (isfinite(a) && isfinite(b) && isfinite(c)) {
    const int i0 = (int)floorf(a * g.inv) - g.minb[0], i1 = (int)floorf(b * g.inv) - g.minb[1], i2 = (int)floorf(c * g.inv) - g.minb[2];
    k = (unsigned)(i0 * g.mul[0] + i1 * g.mul[1] + i2 * g.mul[2]);
  }
  keys[i] = k, vals[i] = (unsigned)i;
}
__global__ void k_vox_heads(const unsigned* __restrict__ keys, int n, unsigned* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
__global__ void k_vox_starts(const unsigned* __restrict__ flags, const unsigned* __restrict__ pos, int n, unsigned* __restrict__ starts,
                             unsigned* __restrict__ n_seg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (flags[i]) starts[pos[i]] = (unsigned)i;
  if (i == n - 1) *n_seg = pos[i] + flags[i];
}
// one thread per voxel: CentroidPoint sums xyz in float in the (stable) sorted order and divides by the count
// nx..: optional normal planes: AccumulatorNormal sums them and returns normal.normalized()
__global__ void k_vox_centroids(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const unsigned* __restrict__ vals,
                                const unsigned* __restrict__ starts, int n_seg, int n_finite, float* __restrict__ ox, float* __restrict__ oy,
                                float* __restrict__ oz, const float* __restrict__ nx, const float* __restrict__ ny, const float* __restrict__ nz,
                                float* __restrict__ onx, float* __restrict__ ony, float* __restrict__ onz) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seg) return;
  const int b = (int)starts[s], e = s + 1 < n_seg ? (int)starts[s + 1] : n_finite;
  float sx = 0, sy = 0, sz = 0;
  V3 sn = v3(0, 0, 0);
  for (int j = b; j < e; ++j) {
    const unsigned p = vals[j];
    sx += x[p], sy += y[p], sz += z[p];
    if (nx) sn = sn + v3(nx[p], ny[p], nz[p]);
  }
  const float cnt = (float)(e - b);
  ox[s] = sx / cnt, oy[s] = sy / cnt, oz[s] = sz / cnt;
  if (nx) {
    sn = vnormalized(sn);
    onx[s] = sn.x, ony[s] = sn.y, onz[s] = sn.z;
  }
}
// main_realdata_auto.cpp:160-177: normal towards the viewpoint (0,0,0), confidence of the nearest point of the dense
// cloud (FLANN's squared distance, lowest index on ties); one wavefront per output point
__global__ void __launch_bounds__(256) k_segment_finish(const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz, int m,
                                                        float* __restrict__ nx, float* __restrict__ ny, float* __restrict__ nz, const float* __restrict__ dx,
                                                        const float* __restrict__ dy, const float* __restrict__ dz, const float* __restrict__ dconf, int n,
                                                        float* __restrict__ conf) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= m) return;
  const V3 p = v3(px[i], py[i], pz[i]);
  unsigned long long best = ~0ull;
  for (int j = lane; j < n; j += 64) {
    const unsigned long long v = ((unsigned long long)__float_as_uint(sqdist_flann(p, v3(dx[j], dy[j], dz[j]))) << 32) | (unsigned)j;
    best = v < best ? v : best;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_down(best, o);
    best = t < best ? t : best;
  }
  if (lane != 0) return;
  conf[i] = n > 0 ? dconf[(int)(best & 0xffffffffu)] : 0.f;
  const float vx = 0.f - p.x, vy = 0.f - p.y, vz = 0.f - p.z;
  const float cos_theta = (vx * nx[i] + vy * ny[i]) + vz * nz[i];
  if (cos_theta < 0) nx[i] *= -1, ny[i] *= -1, nz[i] *= -1;
}

// ---- rejectByCollisionOrNonTouching
// per hypothesis: xf[h] = {model2handbase (12), its inverse (12), object centre in the hand-base frame (3), pad}
constexpr int XF = 28;
__global__ void k_phys_prepare(const float* __restrict__ pose, int H, PhysParams P, float* __restrict__ xf, int* __restrict__ stage,
                               unsigned* __restrict__ fmin, float* __restrict__ diag) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  M4 c2h, m2s;
  for (int k = 0; k < 16; ++k) c2h.m[k] = P.cam2handbase[k], m2s.m[k] = pose[16 * (size_t)h + k];
  const M4 m2h = m4_mul(c2h, m2s);  // PoseEstimator.cpp:587
  const M4 inv = m4_inverse_affine(m2h);
  float* o = xf + (size_t)XF * h;
  for (int k = 0; k < 12; ++k) o[k] = m2h.m[k], o[12 + k] = inv.m[k];
  // cur_center = model2handbase * (centre, 1) (:589-590), a 4x4 by 4-vector product
  const float* m = m2h.m;
  o[24] = ((m[0] * P.center_init[0] + m[1] * P.center_init[1]) + m[2] * P.center_init[2]) + m[3] * 1.0f;
  o[25] = ((m[4] * P.center_init[0] + m[5] * P.center_init[1]) + m[6] * P.center_init[2]) + m[7] * 1.0f;
  o[26] = ((m[8] * P.center_init[0] + m[9] * P.center_init[1]) + m[10] * P.center_init[2]) + m[11] * 1.0f;
  stage[h] = 0;
  for (int k = 0; k < 12; ++k) fmin[12 * (size_t)h + k] = k < 8 ? ORDERED_NONE : 0u;  // 4 finger-cloud mins, 4 model mins, 4 inside counts
  diag[8 * (size_t)h] = 0.f;
  for (int k = 1; k < 8; ++k) diag[8 * (size_t)h + k] = __builtin_nanf("");
}

__device__ __forceinline__ unsigned long long nn_pack(float d, int i) { return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i; }

// checks 1 and 2 (:598-642): nearest scene / hand point to the object centre, one block per hypothesis
__global__ void __launch_bounds__(256) k_phys_center(SdfMeshDev obj, const float* __restrict__ xf, int H, PhysParams P, const float* __restrict__ sx,
                                                     const float* __restrict__ sy, const float* __restrict__ sz, int ns, const float* __restrict__ hx,
                                                     const float* __restrict__ hy, const float* __restrict__ hz, int nh, int* __restrict__ stage,
                                                     float* __restrict__ diag) {
  __shared__ unsigned long long best[2];
  __shared__ unsigned stk[SDF_STACK];
  const int h = blockIdx.x;
  const float* X = xf + (size_t)XF * h;
  const V3 c = v3(X[24], X[25], X[26]);
  if (threadIdx.x < 2) best[threadIdx.x] = ~0ull;
  __syncthreads();
  unsigned long long b0 = ~0ull, b1 = ~0ull;
  for (int i = threadIdx.x; i < ns; i += blockDim.x) {
    const unsigned long long v = nn_pack(sqdist_flann(c, v3(sx[i], sy[i], sz[i])), i);
    b0 = v < b0 ? v : b0;
  }
  for (int i = threadIdx.x; i < nh; i += blockDim.x) {
    const unsigned long long v = nn_pack(sqdist_flann(c, v3(hx[i], hy[i], hz[i])), i);
    b1 = v < b1 ? v : b1;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t0 = __shfl_down(b0, o), t1 = __shfl_down(b1, o);
    b0 = t0 < b0 ? t0 : b0, b1 = t1 < b1 ? t1 : b1;
  }
  if ((threadIdx.x & 63) == 0) atomicMin(&best[0], b0), atomicMin(&best[1], b1);
  __syncthreads();
  if (threadIdx.x != 0) return;
  const float* inv = X + 12;
  int st = 0;
  if (ns > 0) {
    const int i = (int)(best[0] & 0xffffffffu);
    const float s = sdf_signed_distance(obj, m4_point(inv, v3(sx[i], sy[i], sz[i])), stk, 1, nullptr);
    diag[8 * (size_t)h + 1] = s;
    if (s <= P.inside_ob_dist) st = 1;
  }
  if (st == 0 && nh > 0) {
    const int i = (int)(best[1] & 0xffffffffu);
    const float sq = __uint_as_float((unsigned)(best[1] >> 32));
    if (sqrtf(sq) < P.ob_diameter / 2) {
      const float s = sdf_signed_distance(obj, m4_point(inv, v3(hx[i], hy[i], hz[i])), stk, 1, nullptr);
      diag[8 * (size_t)h + 2] = s;
      if (s < P.collision_dist) st = 2;
    }
  }
  stage[h] = st;
  diag[8 * (size_t)h] = (float)st;
}

// check 3 (:646-668): the active finger clouds (hand-base frame, concatenated) against the object; grid (chunks, H)
__global__ void __launch_bounds__(SDF_BLOCK) k_phys_fingers(SdfMeshDev obj, const float* __restrict__ xf, PhysParams P, const float* __restrict__ fx,
                                                            const float* __restrict__ fy, const float* __restrict__ fz, const int* __restrict__ stage,
                                                            unsigned* __restrict__ fmin) {
  __shared__ unsigned stk[SDF_STACK * SDF_BLOCK];
  __shared__ unsigned mn[4];
  const int h = blockIdx.y;
  if (stage[h] != 0) return;
  if (threadIdx.x < 4) mn[threadIdx.x] = ORDERED_NONE;
  __syncthreads();
  const int n = P.finger_off[4];
  const int i = blockIdx.x * SDF_BLOCK + threadIdx.x;
  if (i < n) {
    const int k = (i >= P.finger_off[1]) + (i >= P.finger_off[2]) + (i >= P.finger_off[3]);
    const float* inv = xf + (size_t)XF * h + 12;
    const float s = sdf_signed_distance(obj, m4_point(inv, v3(fx[i], fy[i], fz[i])), stk + threadIdx.x, SDF_BLOCK, nullptr);
    if (s == s) atomicMin(&mn[k], ordered_bits(s));
  }
  __syncthreads();
  if (threadIdx.x < 4 && mn[threadIdx.x] != ORDERED_NONE) atomicMin(&fmin[12 * (size_t)h + threadIdx.x], mn[threadIdx.x]);
}

__global__ void k_phys_decide_fingers(int H, PhysParams P, const unsigned* __restrict__ fmin, int* __restrict__ stage, float* __restrict__ diag) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H || stage[h] != 0) return;
  bool rejected = false, non_touch[4] = {false, false, false, false};
  for (int k = 0; k < 4; ++k) {
    if (!P.finger_active[k]) continue;
    const unsigned u = fmin[12 * (size_t)h + k];
    const float m = u == ORDERED_NONE ? __builtin_inff() : from_ordered_bits(u);
    diag[8 * (size_t)h + 3 + k] = m;
    if (m <= P.collision_dist) rejected = true;
    if (m > P.non_touch_dist && P.finger_status[k]) non_touch[k] = true;
  }
  int st = 0;
  if (rejected) st = 3;
  else if ((non_touch[0] && non_touch[1]) || (non_touch[2] && non_touch[3])) st = 4;
  stage[h] = st;
  diag[8 * (size_t)h] = (float)st;
}

// check 4 (:684-722): the object's points, moved into the hand-base frame, against the four finger meshes
struct FingerMeshes {
  SdfMeshDev m[4];
};
__global__ void __launch_bounds__(SDF_BLOCK) k_phys_model(FingerMeshes fm, const float* __restrict__ xf, int n_model, const float* __restrict__ mx,
                                                          const float* __restrict__ my, const float* __restrict__ mz, const int* __restrict__ stage,
                                                          unsigned* __restrict__ fmin) {
  __shared__ unsigned stk[SDF_STACK * SDF_BLOCK];
  __shared__ unsigned mn[4], inside[4];
  const int h = blockIdx.y;
  if (stage[h] != 0) return;
  if (threadIdx.x < 4) mn[threadIdx.x] = ORDERED_NONE, inside[threadIdx.x] = 0;
  __syncthreads();
  const int i = blockIdx.x * SDF_BLOCK + threadIdx.x;
  if (i < n_model) {
    const V3 p = m4_point(xf + (size_t)XF * h, v3(mx[i], my[i], mz[i]));
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      const float s = sdf_signed_distance(fm.m[k], p, stk + threadIdx.x, SDF_BLOCK, nullptr);
      if (s == s) atomicMin(&mn[k], ordered_bits(s));
      if (s < 0) atomicAdd(&inside[k], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    if (mn[threadIdx.x] != ORDERED_NONE) atomicMin(&fmin[12 * (size_t)h + 4 + threadIdx.x], mn[threadIdx.x]);
    if (inside[threadIdx.x]) atomicAdd(&fmin[12 * (size_t)h + 8 + threadIdx.x], inside[threadIdx.x]);
  }
}

__global__ void k_phys_decide_model(int H, PhysParams P, const unsigned* __restrict__ fmin, int* __restrict__ stage, float* __restrict__ diag) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H || stage[h] != 0) return;
  bool rejected = false;
  float smallest = __builtin_inff();
  for (int k = 0; k < 4 && !rejected; ++k) {
    const unsigned u = fmin[12 * (size_t)h + 4 + k];
    const float m = u == ORDERED_NONE ? __builtin_inff() : from_ordered_bits(u);
    smallest = fminf(smallest, m);
    if (m < P.collision_finger_dist) rejected = true;
    // `num_inside/P.rows() > ratio` is an integer division (:715)
    const long q = (long)fmin[12 * (size_t)h + 8 + k] / (long)max(P.n_model, 1);
    if ((float)q > P.volume_ratio) rejected = true;
  }
  diag[8 * (size_t)h + 7] = smallest;
  stage[h] = rejected ? 5 : 0;
  diag[8 * (size_t)h] = rejected ? 5.f : 0.f;
}

__global__ void k_phys_gather(const int* __restrict__ src, int n, const float* __restrict__ pose, const float* __restrict__ score, const int* __restrict__ id,
                              float* __restrict__ opose, float* __restrict__ oscore, int* __restrict__ oid) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 16) return;
  const int k = t >> 4, e = t & 15;
  const int s = src[k];
  opose[t] = pose[16 * (size_t)s + e];
  if (e == 0) oscore[k] = score[s], oid[k] = id[s];
}

// ---- scene front end (main_realdata_auto.cpp:54-96 without colours and normals)
// readDepthImage + convert3dOrganizedRGB + the z pass-through: one thread per pixel, NaN for a dropped pixel (the voxel
// grid skips non-finite points, as pcl::VoxelGrid does); valid pixels are counted
__global__ void k_depth_to_cloud(const unsigned short* __restrict__ depth, int H, int W, double unit, float cx, float fx, float cy, float fy,
                                 float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz, unsigned* __restrict__ n_valid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int u = i / W, v = i - u * W;
  float d = (float)((double)(float)depth[i] * unit);  // Utils.cpp:44
  if (d > 2.0 || d < 0.1) d = 0.0f;                   // Utils.cpp:45 (double literals)
  const float nan = __builtin_nanf("");
  float x = nan, y = nan, z = nan;
  if (d > 0.1 && d < 2.0) {  // Utils.cpp:96-100
    const float px = (float)(((float)v - cx) * d / fx), py = (float)(((float)u - cy) * d / fy);
    if (!(d < 0.1f || d > 2.0f)) x = px, y = py, z = d;  // PassThrough "z" in [0.1, 2.0] (main :64-70)
  }
  ox[i] = x, oy[i] = y, oz[i] = z;
  if (z == z) atomicAdd(n_valid, 1u);
}
// transform into the hand-base frame, the three pass-through filters (z, x, y; inclusive limits), transform back; flag
__global__ void k_crop_handbase(const float* __restrict__ ix, const float* __restrict__ iy, const float* __restrict__ iz, int n, const float* __restrict__ A,
                                const float* __restrict__ B, float3 lo, float3 hi, float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz,
                                unsigned* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 q = m4_point(A, v3(ix[i], iy[i], iz[i]));
  const bool keep = isfinite(q.x) && isfinite(q.y) && isfinite(q.z) && !(q.z < lo.z || q.z > hi.z) && !(q.x < lo.x || q.x > hi.x) && !(q.y < lo.y || q.y > hi.y);
  const V3 r = m4_point(B, q);
  ox[i] = r.x, oy[i] = r.y, oz[i] = r.z;
  flag[i] = keep ? 1u : 0u;
}
__global__ void k_compact3(const float* __restrict__ ix, const float* __restrict__ iy, const float* __restrict__ iz, const unsigned* __restrict__ flag,
                           const unsigned* __restrict__ pos, int n, float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz, int stride,
                           unsigned* __restrict__ n_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (flag[i]) {
    const unsigned k = pos[i];
    ox[k] = ix[i], oy[k] = iy[i], oz[k] = iz[i];
  }
  if (i == n - 1) *n_out = pos[i] + flag[i];
}

// ---- Hand::setCurScene filters (Hand.cpp:289-321), hand-base frame
__global__ void k_transform_cloud_nrm(const float* __restrict__ ix, const float* __restrict__ iy, const float* __restrict__ iz, const float* __restrict__ inx,
                                      const float* __restrict__ iny, const float* __restrict__ inz, int n, const float* __restrict__ T, float* __restrict__ o,
                                      float* __restrict__ on) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 p = m4_point(T, v3(ix[i], iy[i], iz[i])), m = m4_dir(T, v3(inx[i], iny[i], inz[i]));
  o[i] = p.x, o[n + i] = p.y, o[2 * (size_t)n + i] = p.z;
  on[i] = m.x, on[n + i] = m.y, on[2 * (size_t)n + i] = m.z;
}
// pcl::RadiusOutlierRemoval: a live point stays when more than min_pts live points (itself included) are strictly
// within the radius (FLANN RadiusResultSet: dist < r^2).  One wavefront per point, lanes stride over the cloud, so that a
// cloud of a few thousand points still fills the GPU; a point is settled as soon as its count passes min_pts.
__global__ void __launch_bounds__(256) k_radius_outlier(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n,
                                                        const unsigned char* __restrict__ live, float r2, int min_pts, unsigned char* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  if (!live[i]) {
    if (lane == 0) out[i] = 0;
    return;
  }
  const V3 p = v3(x[i], y[i], z[i]);
  int k = 0;
  for (int base = 0; base < n; base += 64) {
    const int j = base + lane;
    const bool hit = j < n && live[j] && sqdist_flann(p, v3(x[j], y[j], z[j])) < r2;
    k += __popcll(__ballot(hit));
    if (k > min_pts) break;  // uniform across the wavefront
  }
  if (lane == 0) out[i] = k > min_pts ? 1 : 0;
}
// pcl::StatisticalOutlierRemoval, first pass: mean distance to the mean_k nearest live neighbours (the 21 smallest
// squared distances include the point itself, which is skipped).  One wavefront per point: every lane keeps the sorted
// 21 smallest of its stripe in LDS, then the wavefront extracts the 21 smallest overall, in ascending order.
constexpr int SOR_K = 20;
__global__ void __launch_bounds__(256) k_sor_mean(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n,
                                                  const unsigned char* __restrict__ live, float* __restrict__ dist) {
  __shared__ float top[(SOR_K + 1) * 256];
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  if (!live[i]) {
    if (lane == 0) dist[i] = 0.f;
    return;
  }
  const V3 p = v3(x[i], y[i], z[i]);
  float* mine = top + threadIdx.x;
  for (int k = 0; k <= SOR_K; ++k) mine[k * 256] = __builtin_inff();
  float worst = __builtin_inff();
  for (int j = lane; j < n; j += 64) {
    if (!live[j]) continue;
    const float d = sqdist_flann(p, v3(x[j], y[j], z[j]));
    if (d < worst) {  // insert into this lane's ascending list
      int k = SOR_K;
      while (k > 0 && mine[(k - 1) * 256] > d) mine[k * 256] = mine[(k - 1) * 256], --k;
      mine[k * 256] = d;
      worst = mine[SOR_K * 256];
    }
  }
  // merge: 21 rounds, each takes the smallest head among the 64 lists (equal values: any order gives the same sum)
  int head = 0;
  double sum = 0.0;
  for (int round = 0; round <= SOR_K; ++round) {
    const float mine_head = head <= SOR_K ? mine[head * 256] : __builtin_inff();
    unsigned long long key = ((unsigned long long)__float_as_uint(mine_head) << 32) | (unsigned)lane;
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long t = __shfl_xor(key, o);
      key = t < key ? t : key;
    }
    if ((int)(key & 63u) == lane) ++head;
    if (round > 0) sum += (double)sqrtf(__uint_as_float((unsigned)(key >> 32)));  // round 0 is the query point itself
  }
  if (lane == 0) dist[i] = (float)(sum / SOR_K);
}
__global__ void k_sor_apply(const float* __restrict__ x, int n, const unsigned char* __restrict__ live, const float* __restrict__ dist, double thr, int use_sor,
                            unsigned char* __restrict__ keep_noise, unsigned char* __restrict__ keep_swivel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool k = live[i] && (!use_sor || !((double)dist[i] > thr));
  keep_noise[i] = k;
  keep_swivel[i] = k && !(x[i] < -0.25f || x[i] > -0.1f);  // pass-through x (Hand.cpp:316-320)
}

// Hand::handbaseICP source cloud (Hand.cpp:685-729): hand-base transform, two pass-throughs, finger connections removed
__global__ void k_handbase_region(const float* __restrict__ ix, const float* __restrict__ iy, const float* __restrict__ iz, const float* __restrict__ inx,
                                  const float* __restrict__ iny, const float* __restrict__ inz, int n, const float* __restrict__ T, float y1, float z1,
                                  float y2, float z2, float* __restrict__ o, float* __restrict__ on, unsigned char* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 pt = m4_point(T, v3(ix[i], iy[i], iz[i])), m = m4_dir(T, v3(inx[i], iny[i], inz[i]));
  o[i] = pt.x, o[n + i] = pt.y, o[2 * (size_t)n + i] = pt.z;
  on[i] = m.x, on[n + i] = m.y, on[2 * (size_t)n + i] = m.z;
  bool k = isfinite(pt.x) && isfinite(pt.y) && isfinite(pt.z) && !(pt.x < -0.07f || pt.x > 0.03f) && !(pt.z < -0.18f || pt.z > 0.01f);
  if (k) {
    const float sq_dist1 = (pt.z - z1) * (pt.z - z1) + (pt.y - y1) * (pt.y - y1);
    const float sq_dist2 = (pt.z - z2) * (pt.z - z2) + (pt.y - y2) * (pt.y - y2);
    if (sq_dist1 <= 0.015 * 0.015) k = false;
    else if (sq_dist2 <= 0.015 * 0.015) k = false;
    else if (((pt.y >= y1 && pt.y <= y2) || (pt.y >= y2 && pt.y <= y1)) && fabsf(pt.z - z1) <= 0.01) k = false;  // :718-724 tests z1 twice
  }
  keep[i] = k;
}

// ---- face cells: for every voxel of a grid around a mesh, the faces that can be closest for a query inside it
// One wavefront per voxel (centre c, half diagonal r of the voxel grown by the rounding of the voxel index):
//   1. d0 = distance from c to the mesh (all faces, the float expression the queries use);
//   2. band: a face can be closest for some x of the voxel only if dist(c, f) <= d0 + 2 r  (distance to a set is
//      1-Lipschitz: dist(x, f) >= dist(c, f) - r and dist(x, mesh) <= d0 + r);
//   3. a band face f is dropped when one of the FC_PIVOTS band faces nearest to c, f', is closer over the whole voxel:
//      dist(x, f') <= |x - y'| with y' the point of f' closest to c, and dist(x, f) >= n . (x - c_f) with c_f the point
//      of f closest to c and n = (c - c_f) / |c - c_f| (f is convex: the plane through c_f normal to n supports it).
//      |x - y'| - n . (x - c_f) is convex in x, so its maximum over the voxel is at a corner: eight evaluations; f goes
//      when the maximum is below -FC_MARGIN (a strict margin above the float error of every distance involved: a dropped
//      face can neither win nor tie).
// Survivors are written in ascending slot order.  A voxel whose band exceeds FC_CAND gets no list (tree walk instead).
constexpr int FC_CAND = 768, FC_PIVOTS = 8;
constexpr float FC_MARGIN = 2e-6f;
struct FaceGrid {
  float ox, oy, oz, s;
  int nx, ny, nz;
};
template <bool WRITE>
__global__ void __launch_bounds__(64) k_face_cells(SdfMeshDev m, FaceGrid g, float margin, int* __restrict__ count, const int* __restrict__ start,
                                                   int* __restrict__ faces) {
  __shared__ int c_slot[FC_CAND];
  __shared__ float c_d[FC_CAND];
  __shared__ float c_x[FC_CAND], c_y[FC_CAND], c_z[FC_CAND];
  __shared__ int n_cand;
  __shared__ int piv[FC_PIVOTS];
  const int v = blockIdx.x, lane = threadIdx.x;
  const int ix = v % g.nx, iy = (v / g.nx) % g.ny, iz = v / (g.nx * g.ny);
  const V3 c = v3(g.ox + (ix + 0.5f) * g.s, g.oy + (iy + 0.5f) * g.s, g.oz + (iz + 0.5f) * g.s);
  const float h = 0.5f * g.s + margin;  // half edge, grown: the voxel index of a query is computed in float
  const float r = 1.7320508f * h;
  // 1. distance of the centre
  float dmin = __builtin_inff();
  for (int s = lane; s < m.n_faces; s += 64) {
    const float4 A4 = m.tri[3 * s], B4 = m.tri[3 * s + 1], C4 = m.tri[3 * s + 2];
    const V3 p = sdf_closest_point(c, v3(A4.x, A4.y, A4.z), v3(B4.x, B4.y, B4.z), v3(C4.x, C4.y, C4.z));
    dmin = fminf(dmin, vsqn(c - p));
  }
  for (int o = 32; o > 0; o >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, o));
  const float d0 = sqrtf(dmin);
  const float band = d0 + 2.f * r + margin;
  const float band2 = band * band * 1.00001f;
  // 2. the band, in slot order
  if (lane == 0) n_cand = 0;
  __syncthreads();
  bool overflow = false;
  for (int base = 0; base < m.n_faces; base += 64) {
    const int s = base + lane;
    bool in = false;
    V3 p = v3(0, 0, 0);
    float d2 = 0;
    if (s < m.n_faces) {
      const float4 A4 = m.tri[3 * s], B4 = m.tri[3 * s + 1], C4 = m.tri[3 * s + 2];
      p = sdf_closest_point(c, v3(A4.x, A4.y, A4.z), v3(B4.x, B4.y, B4.z), v3(C4.x, C4.y, C4.z));
      d2 = vsqn(c - p);
      in = d2 <= band2;
    }
    const unsigned long long mask = __ballot(in);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    const int at = n_cand + before;
    if (in && at < FC_CAND) c_slot[at] = s, c_d[at] = sqrtf(d2), c_x[at] = p.x, c_y[at] = p.y, c_z[at] = p.z;
    __syncthreads();
    if (lane == 0) n_cand += __popcll(mask);
    __syncthreads();
    if (n_cand > FC_CAND) {
      overflow = true;
      break;
    }
  }
  if (overflow) {
    if (!WRITE && lane == 0) count[v] = -1;
    return;
  }
  const int n = n_cand;
  // 3a. the FC_PIVOTS band faces nearest to the centre (ties: lowest position)
  for (int k = 0; k < FC_PIVOTS; ++k) {
    unsigned long long best = ~0ull;
    for (int i = lane; i < n; i += 64) {
      bool taken = false;
      for (int j = 0; j < k; ++j) taken = taken || piv[j] == i;
      if (taken) continue;
      const unsigned long long key = ((unsigned long long)__float_as_uint(c_d[i]) << 32) | (unsigned)i;
      best = key < best ? key : best;
    }
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long t = __shfl_xor(best, o);
      best = t < best ? t : best;
    }
    if (lane == 0) piv[k] = best == ~0ull ? -1 : (int)(best & 0xffffffffu);
    __syncthreads();
  }
  // 3b. domination
  int kept_before = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    bool keep = false;
    if (i < n) {
      keep = true;
      const V3 cf = v3(c_x[i], c_y[i], c_z[i]);
      const float di = c_d[i];
      if (di > 0.f) {
        const V3 nf = (c - cf) / di;
        for (int k = 0; k < FC_PIVOTS && keep; ++k) {
          const int j = piv[k];
          if (j < 0 || j == i) continue;
          if (!(c_d[j] < di)) continue;  // only a face that is nearer at the centre can win everywhere
          const V3 y = v3(c_x[j], c_y[j], c_z[j]);
          float worst = -__builtin_inff();
#pragma unroll
          for (int corner = 0; corner < 8; ++corner) {
            const V3 x = v3(c.x + ((corner & 1) ? h : -h), c.y + ((corner & 2) ? h : -h), c.z + ((corner & 4) ? h : -h));
            worst = fmaxf(worst, vnorm(x - y) - vdot(nf, x - cf));
          }
          if (worst < -FC_MARGIN - margin) keep = false;
        }
      }
    }
    const unsigned long long mask = __ballot(keep);
    if (WRITE) {
      if (keep) faces[start[v] + kept_before + __popcll(mask & ((1ull << lane) - 1ull))] = c_slot[i];
    }
    kept_before += __popcll(mask);
  }
  if (!WRITE && lane == 0) count[v] = kept_before;
}
// turns the counts (-1: no list) into cell_start: exclusive scan of max(count, 0), complemented where there is no list
__global__ void k_face_cells_pack(const int* __restrict__ count, const unsigned* __restrict__ scan, int nvox, int* __restrict__ cell_start) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v > nvox) return;
  if (v == nvox) {
    cell_start[v] = (int)scan[v];
    return;
  }
  cell_start[v] = count[v] < 0 ? ~(int)scan[v] : (int)scan[v];
}
__global__ void k_face_cells_clamp(const int* __restrict__ count, int nvox, unsigned* __restrict__ pos) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v <= nvox) pos[v] = (v < nvox && count[v] > 0) ? (unsigned)count[v] : 0u;
}

// HandT42::adjustHandHeight matching loop (Hand.cpp:1010-1049): grid (hand points / 4, trial heights), one wavefront per
// (hand point, height): nearest scene point (FLANN's squared distance, lowest index on ties), 5 mm gate, 45 degree gate
__global__ void __launch_bounds__(256) k_hand_height(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                                                     const float* __restrict__ snx, const float* __restrict__ sny, const float* __restrict__ snz, int ns,
                                                     const float* __restrict__ hx, const float* __restrict__ hy, const float* __restrict__ hz,
                                                     const float* __restrict__ hnx, const float* __restrict__ hny, const float* __restrict__ hnz, int nh,
                                                     const float* __restrict__ heights, int* __restrict__ counts) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, t = blockIdx.y;
  if (i >= nh) return;
  const V3 p = v3(hx[i], hy[i], hz[i] + heights[t]);
  unsigned long long best = ~0ull;
  for (int j = lane; j < ns; j += 64) {
    const unsigned long long v = ((unsigned long long)__float_as_uint(sqdist_flann(p, v3(sx[j], sy[j], sz[j]))) << 32) | (unsigned)j;
    best = v < best ? v : best;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long u = __shfl_xor(best, o);
    best = u < best ? u : best;
  }
  if (lane != 0 || ns == 0) return;
  const float sq = __uint_as_float((unsigned)(best >> 32));
  if ((double)sq > 0.005 * 0.005) return;
  const int j = (int)(best & 0xffffffffu);
  if ((double)vdot(v3(hnx[i], hny[i], hnz[i]), v3(snx[j], sny[j], snz[j])) >= cos(45 / 180.0 * M_PI)) atomicAdd(&counts[t], 1);
}

// ------------------------------------------------------------------------------------------------ host helpers
int upload_planes(hop_ctx* c, Cloud3& dst, const float* planes, int n) {
  PHCHK(c, dst.buf.ensure(sizeof(float) * 3 * (size_t)std::max(n, 1)));
  dst.n = n;
  if (n > 0) PHCHK(c, hop_ctx_h2d(c, dst.buf.p, planes, sizeof(float) * 3 * (size_t)n));
  return HOP_OK;
}

// pcl::VoxelGrid on a device cloud (planes x,y,z of n); result into `out`
int voxel_downsample_device(hop_ctx* c, Physics* ph, const float* x, const float* y, const float* z, int n, float leaf, Cloud3& out,
                            const float* nrm = nullptr, Cloud3* out_nrm = nullptr) {
  hipStream_t st = hop_ctx_stream(c);
  out.n = 0;
  PHCHK(c, out.buf.ensure(sizeof(float) * 3));
  if (n <= 0) return HOP_OK;
  if (!(leaf > 0)) return HOP_E_INVALID;
  PHCHK(c, ph->scalars.ensure(sizeof(unsigned) * 16));
  const unsigned init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
  PHCHK(c, hop_ctx_h2d(c, ph->scalars.p, init, sizeof(init)));
  k_vox_minmax<<<std::min((n + 255) / 256, 1024), 256, 0, st>>>(x, y, z, n, ph->scalars.as<unsigned>());
  unsigned sc[8];
  PHCHK(c, hop_ctx_d2h(c, sc, ph->scalars.p, sizeof(sc)));
  PHCHK(c, hipStreamSynchronize(st));
  const int n_finite = (int)sc[6];
  if (n_finite == 0) return HOP_OK;
  auto dec = [](unsigned u) {
    const unsigned b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
  };
  VoxGeom g;
  g.inv = 1.0f / leaf;
  int64_t div[3];
  for (int a = 0; a < 3; ++a) {
    const float mn = dec(sc[a]), mx = dec(sc[3 + a]);
    // voxel_grid.hpp:243-251: the index of a cell must fit an int
    const double cells = ((double)mx - (double)mn) * (double)g.inv + 1;
    if (cells > (double)INT_MAX) {
      hop_ctx_set_error(c, "voxel grid: leaf size too small for the input (integer indices would overflow)");
      return HOP_E_CAPACITY;
    }
    g.minb[a] = (int)std::floor(mn * g.inv);
    div[a] = (int64_t)(int)std::floor(mx * g.inv) - g.minb[a] + 1;
  }
  if (div[0] * div[1] * div[2] > (int64_t)INT_MAX) {
    hop_ctx_set_error(c, "voxel grid: leaf size too small for the input (integer indices would overflow)");
    return HOP_E_CAPACITY;
  }
  g.mul[0] = 1, g.mul[1] = (int)div[0], g.mul[2] = (int)(div[0] * div[1]);
  const size_t nb = sizeof(unsigned) * (size_t)n;
  PHCHK(c, ph->keys.ensure(nb));
  PHCHK(c, ph->keys_alt.ensure(nb));
  PHCHK(c, ph->vals.ensure(nb));
  PHCHK(c, ph->vals_alt.ensure(nb));
  PHCHK(c, ph->flags.ensure(nb));
  PHCHK(c, ph->pos.ensure(nb));
  PHCHK(c, ph->starts.ensure(nb));
  const int blocks = (n + 255) / 256;
  k_vox_keys<<<blocks, 256, 0, st>>>(x, y, z, n, g, ph->keys.as<unsigned>(), ph->vals.as<unsigned>());
  size_t tmp1 = 0, tmp2 = 0;
  PHCHK(c, prim_sort_pairs(nullptr, tmp1, ph->keys.as<unsigned>(), ph->keys_alt.as<unsigned>(), ph->vals.as<unsigned>(),
                                              ph->vals_alt.as<unsigned>(), n, 0, 32, st));
  PHCHK(c, prim_exclusive_sum(nullptr, tmp2, ph->flags.as<unsigned>(), ph->pos.as<unsigned>(), n_finite, st));
  size_t tmp = std::max(tmp1, tmp2);
  PHCHK(c, ph->sort_tmp.ensure(tmp + 16));
  PHCHK(c, prim_sort_pairs(ph->sort_tmp.p, tmp1, ph->keys.as<unsigned>(), ph->keys_alt.as<unsigned>(), ph->vals.as<unsigned>(),
                                              ph->vals_alt.as<unsigned>(), n, 0, 32, st));
  // non-finite points carry the largest key and sit behind the n_finite sorted ones
  const int fb = (n_finite + 255) / 256;
  k_vox_heads<<<fb, 256, 0, st>>>(ph->keys_alt.as<unsigned>(), n_finite, ph->flags.as<unsigned>());
  PHCHK(c, prim_exclusive_sum(ph->sort_tmp.p, tmp2, ph->flags.as<unsigned>(), ph->pos.as<unsigned>(), n_finite, st));
  k_vox_starts<<<fb, 256, 0, st>>>(ph->flags.as<unsigned>(), ph->pos.as<unsigned>(), n_finite, ph->starts.as<unsigned>(), ph->scalars.as<unsigned>() + 8);
  unsigned n_seg = 0;
  PHCHK(c, hop_ctx_d2h(c, &n_seg, ph->scalars.as<unsigned>() + 8, sizeof(unsigned)));
  PHCHK(c, hipStreamSynchronize(st));
  PHCHK(c, out.buf.ensure(sizeof(float) * 3 * (size_t)n_seg));
  out.n = (int)n_seg;
  float* o = out.buf.as<float>();
  float* on = nullptr;
  if (nrm && out_nrm) {
    PHCHK(c, out_nrm->buf.ensure(sizeof(float) * 3 * (size_t)n_seg));
    out_nrm->n = (int)n_seg;
    on = out_nrm->buf.as<float>();
  }
  k_vox_centroids<<<((int)n_seg + 127) / 128, 128, 0, st>>>(x, y, z, ph->vals_alt.as<unsigned>(), ph->starts.as<unsigned>(), (int)n_seg, n_finite, o,
                                                             o + n_seg, o + 2 * (size_t)n_seg, on ? nrm : nullptr, on ? nrm + n : nullptr,
                                                             on ? nrm + 2 * (size_t)n : nullptr, on, on ? on + n_seg : nullptr,
                                                             on ? on + 2 * (size_t)n_seg : nullptr);
  PHCHK(c, hipGetLastError());
  return HOP_OK;
}

}  // namespace

// ================================================================================================ C-ABI
extern "C" {

int hop_sdf_register_mesh(hop_ctx* c, int mesh_id, const float* V, int nv, const int32_t* F, int nf, const float* pose16) {
  if (!c || mesh_id < 0 || mesh_id >= MAX_MESHES || nv < 0 || nf < 0 || (nv > 0 && !V) || (nf > 0 && !F)) return HOP_E_INVALID;
  for (int i = 0; i < 3 * nf; ++i)
    if (F[i] < 0 || F[i] >= nv) return HOP_E_INVALID;
  PHCHK(c, hipSetDevice(hop_ctx_device(c)));
  Physics* ph = physics(c);
  hipStream_t st = hop_ctx_stream(c);
  {
    MeshStore& standing = ph->mesh[mesh_id];
    static const bool no_reuse = getenv("HOP_SDF_NO_REUSE") != nullptr;
    if (!no_reuse && standing.same_as(V, nv, F, nf, pose16)) {
      standing.dev.has_pose = 0;  // (a fresh registration clears what hop_sdf_set_mesh_pose set)
      return HOP_OK;
    }
    standing.valid = false, standing.key_V.clear(), standing.key_F.clear(), standing.key_pose.clear();  // until the rebuild below completes
  }
  HostMesh hm;
  prepare_mesh(hm, V, nv, F, nf, pose16);
  TreeBuilder tb(hm);
  if (nf > 0) tb.build(0, nf, 0);
  tb.leaf_first.push_back((int)tb.slot_face.size());
  if (3 * tb.max_depth + 1 > SDF_STACK || tb.nodes.size() >= (size_t)SDF_LEAF_BASE || tb.leaf_first.size() > (size_t)SDF_LEAF_BASE || nf > (1 << 22)) {
    hop_ctx_set_error(c, "mesh too large for the traversal stack");
    return HOP_E_CAPACITY;
  }
  OrderTree ot(hm);
  std::vector<float4> tri(3 * (size_t)nf), nrm(7 * (size_t)nf);
  for (int s = 0; s < nf; ++s) {
    const int f = tb.slot_face[s];
    const V3 P[3] = {hm.V[F[3 * f]], hm.V[F[3 * f + 1]], hm.V[F[3 * f + 2]]};
    const double area = sdf_doublearea(P[0], P[1], P[2]);
    int fbits = f;
    float fw;
    std::memcpy(&fw, &fbits, 4);
    tri[3 * (size_t)s] = make_float4(P[0].x, P[0].y, P[0].z, area > 1e-4 ? 1.f : 0.f);
    tri[3 * (size_t)s + 1] = make_float4(P[1].x, P[1].y, P[1].z, fw);
    tri[3 * (size_t)s + 2] = make_float4(P[2].x, P[2].y, P[2].z, 0.f);
    auto put = [&](int k, V3 n) { nrm[7 * (size_t)s + k] = make_float4(n.x, n.y, n.z, 0.f); };
    put(0, hm.FN[f]);
    for (int e = 0; e < 3; ++e) put(1 + e, hm.EN[hm.EMAP[(size_t)e * nf + f]]);
    for (int v = 0; v < 3; ++v) put(4 + v, hm.VN[F[3 * f + v]]);
  }
  MeshStore& ms = ph->mesh[mesh_id];
  PHCHK(c, hipStreamSynchronize(st));  // a previous frame may still read the old buffers
  PHCHK(c, ms.tri_d.ensure(std::max<size_t>(sizeof(float4) * tri.size(), 16)));
  PHCHK(c, ms.nrm_d.ensure(std::max<size_t>(sizeof(float4) * nrm.size(), 16)));
  PHCHK(c, ms.nodes_d.ensure(std::max<size_t>(sizeof(SdfNode) * tb.nodes.size(), 256)));
  PHCHK(c, ms.order_d.ensure(std::max<size_t>(sizeof(SdfOrderNode) * ot.nodes.size(), 32)));
  PHCHK(c, ms.leaf_d.ensure(std::max<size_t>(sizeof(int) * (size_t)nf, 16)));
  PHCHK(c, ms.leaf_first_d.ensure(sizeof(int) * tb.leaf_first.size()));
  PHCHK(c, hop_ctx_h2d(c, ms.leaf_first_d.p, tb.leaf_first.data(), sizeof(int) * tb.leaf_first.size()));
  PHCHK(c, ms.leaf_obb_d.ensure(std::max<size_t>(sizeof(float4) * tb.leaf_obb.size(), 16)));
  if (!tb.leaf_obb.empty()) PHCHK(c, hop_ctx_h2d(c, ms.leaf_obb_d.p, tb.leaf_obb.data(), sizeof(float4) * tb.leaf_obb.size()));
  if (nf > 0) {
    PHCHK(c, hop_ctx_h2d(c, ms.tri_d.p, tri.data(), sizeof(float4) * tri.size()));
    PHCHK(c, hop_ctx_h2d(c, ms.nrm_d.p, nrm.data(), sizeof(float4) * nrm.size()));
    PHCHK(c, hop_ctx_h2d(c, ms.nodes_d.p, tb.nodes.data(), sizeof(SdfNode) * tb.nodes.size()));
    PHCHK(c, hop_ctx_h2d(c, ms.order_d.p, ot.nodes.data(), sizeof(SdfOrderNode) * ot.nodes.size()));
    PHCHK(c, hop_ctx_h2d(c, ms.leaf_d.p, ot.face_leaf.data(), sizeof(int) * (size_t)nf));
  }
  PHCHK(c, hipStreamSynchronize(st));
  ms.dev.tri = ms.tri_d.as<float4>(), ms.dev.nrm = ms.nrm_d.as<float4>(), ms.dev.nodes = ms.nodes_d.as<SdfNode>();
  ms.dev.order = ms.order_d.as<SdfOrderNode>(), ms.dev.face_leaf = ms.leaf_d.as<int>(), ms.dev.leaf_first = ms.leaf_first_d.as<int>(), ms.dev.leaf_obb = ms.leaf_obb_d.as<float4>();
  ms.dev.n_faces = nf, ms.dev.n_nodes = (int)tb.nodes.size();
  ms.dev.coord_eps = 4e-7f * hm.max_abs;
  ms.dev.cell_start = nullptr, ms.dev.cell_faces = nullptr;
  ms.dev.has_pose = 0;
  ms.valid = true;
  // face cells for meshes that are worth it (measured: no gain on the finger links' ~100-face hulls, which are also
  // re-registered every frame); HOP_SDF_CELLS=0 turns them off
  const char* env = getenv("HOP_SDF_CELLS");
  if (nf >= 256 && nf <= 65536 && !(env && env[0] == '0')) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const V3& p : hm.V) {
      lo[0] = std::min(lo[0], p.x), lo[1] = std::min(lo[1], p.y), lo[2] = std::min(lo[2], p.z);
      hi[0] = std::max(hi[0], p.x), hi[1] = std::max(hi[1], p.y), hi[2] = std::max(hi[2], p.z);
    }
    const float pad = 0.06f;  // queries farther than this from the mesh's box use the tree
    const float ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2])) + 2 * pad;
    FaceGrid g;
    g.s = std::max(0.0025f, ext / 96.f);  // measured on the bench case: 1.25 / 1.5 / 2.5 / 4 mm -> 8.0 / 8.2 / 7.4 / 9.6 ms
    g.ox = lo[0] - pad, g.oy = lo[1] - pad, g.oz = lo[2] - pad;
    g.nx = (int)std::ceil((hi[0] - lo[0] + 2 * pad) / g.s), g.ny = (int)std::ceil((hi[1] - lo[1] + 2 * pad) / g.s), g.nz = (int)std::ceil((hi[2] - lo[2] + 2 * pad) / g.s);
    const int nvox = g.nx * g.ny * g.nz;
    const float margin = 4e-6f * (hm.max_abs + pad + 1e-3f);
    PHCHK(c, ph->keys.ensure(sizeof(int) * ((size_t)nvox + 1)));
    PHCHK(c, ph->pos.ensure(sizeof(unsigned) * ((size_t)nvox + 1)));
    PHCHK(c, ph->starts.ensure(sizeof(unsigned) * ((size_t)nvox + 1)));
    PHCHK(c, ms.cell_start_d.ensure(sizeof(int) * ((size_t)nvox + 1)));
    int* count = ph->keys.as<int>();
    k_face_cells<false><<<nvox, 64, 0, st>>>(ms.dev, g, margin, count, nullptr, nullptr);
    k_face_cells_clamp<<<(nvox + 256) / 256, 256, 0, st>>>(count, nvox, ph->pos.as<unsigned>());
    size_t tmp = 0;
    PHCHK(c, prim_exclusive_sum(nullptr, tmp, ph->pos.as<unsigned>(), ph->starts.as<unsigned>(), nvox + 1, st));
    PHCHK(c, ph->sort_tmp.ensure(tmp + 16));
    PHCHK(c, prim_exclusive_sum(ph->sort_tmp.p, tmp, ph->pos.as<unsigned>(), ph->starts.as<unsigned>(), nvox + 1, st));
    k_face_cells_pack<<<(nvox + 256) / 256, 256, 0, st>>>(count, ph->starts.as<unsigned>(), nvox, ms.cell_start_d.as<int>());
    unsigned total = 0;
    PHCHK(c, hop_ctx_d2h(c, &total, ph->starts.as<unsigned>() + nvox, sizeof(unsigned)));
    PHCHK(c, hipStreamSynchronize(st));
    PHCHK(c, ms.cell_faces_d.ensure(sizeof(int) * std::max<size_t>(total, 4)));
    k_face_cells<true><<<nvox, 64, 0, st>>>(ms.dev, g, margin, nullptr, reinterpret_cast<const int*>(ph->starts.as<unsigned>()), ms.cell_faces_d.as<int>());
    PHCHK(c, hipGetLastError());
    PHCHK(c, hipStreamSynchronize(st));
    ms.dev.cell_start = ms.cell_start_d.as<int>(), ms.dev.cell_faces = ms.cell_faces_d.as<int>();
    ms.dev.cell_ox = g.ox, ms.dev.cell_oy = g.oy, ms.dev.cell_oz = g.oz, ms.dev.cell_inv = 1.0f / g.s;
    ms.dev.cell_nx = g.nx, ms.dev.cell_ny = g.ny, ms.dev.cell_nz = g.nz;
    ph->last_cells_total = (long long)total, ph->last_cells_voxels = nvox;
  }
  ms.key_V.assign(V, V + 3 * (size_t)nv), ms.key_F.assign(F, F + 3 * (size_t)nf);
  if (pose16) ms.key_pose.assign(pose16, pose16 + 16);
  else ms.key_pose.clear();
  return HOP_OK;
}

int hop_sdf_set_mesh_pose(hop_ctx* c, int mesh_id, const float* pose16) {
  if (!c || mesh_id < 0 || mesh_id >= MAX_MESHES) return HOP_E_INVALID;
  Physics* ph = physics(c);
  MeshStore& ms = ph->mesh[mesh_id];
  if (!ms.valid) return HOP_E_STATE;
  PHCHK(c, hipStreamSynchronize(hop_ctx_stream(c)));  // kernels in flight carry the old pose by value; nothing to wait for on the device, kept for symmetry with register
  if (!pose16) {
    ms.dev.has_pose = 0;
    return HOP_OK;
  }
  M4 T;
  for (int i = 0; i < 16; ++i) T.m[i] = pose16[i];
  const M4 inv = m4_inverse_affine(T);
  for (int i = 0; i < 12; ++i) ms.dev.pose_inv[i] = inv.m[i];
  ms.dev.has_pose = 1;
  return HOP_OK;
}

int hop_sdf_signed_distance(hop_ctx* c, int mesh_id, const float* pts_xyz, int n, float* dists, int32_t* faces, float* min_dist, float* max_dist) {
  if (!c || mesh_id < 0 || mesh_id >= MAX_MESHES || n < 0 || (n > 0 && (!pts_xyz || !dists))) return HOP_E_INVALID;
  PHCHK(c, hipSetDevice(hop_ctx_device(c)));
  Physics* ph = physics(c);
  if (!ph->mesh[mesh_id].valid) return HOP_E_STATE;
  hipStream_t st = hop_ctx_stream(c);
  float mn = FLT_MAX, mx = -FLT_MAX;  // SDFchecker.cpp:119-120
  if (n > 0) {
    int rc = upload_planes(c, ph->tmp_cloud, pts_xyz, n);
    if (rc) return rc;
    PHCHK(c, ph->tmp_cloud2.buf.ensure(sizeof(float) * 2 * (size_t)n));
    float* S = ph->tmp_cloud2.buf.as<float>();
    int* I = reinterpret_cast<int*>(S + n);
    k_sdf_query<<<(n + SDF_BLOCK - 1) / SDF_BLOCK, SDF_BLOCK, 0, st>>>(ph->mesh[mesh_id].dev, ph->tmp_cloud.x(), ph->tmp_cloud.y(), ph->tmp_cloud.z(), n,
                                                                       nullptr, S, I);
    PHCHK(c, hipGetLastError());
    PHCHK(c, hop_ctx_d2h(c, dists, S, sizeof(float) * (size_t)n));
    if (faces) PHCHK(c, hop_ctx_d2h(c, faces, I, sizeof(int) * (size_t)n));
    PHCHK(c, hipStreamSynchronize(st));
    mn = INFINITY, mx = -INFINITY;  // S.minCoeff() / S.maxCoeff(), NaN (points on the surface) skipped
    for (int i = 0; i < n; ++i)
      if (dists[i] == dists[i]) mn = std::min(mn, dists[i]), mx = std::max(mx, dists[i]);
  }
  if (min_dist) *min_dist = mn;
  if (max_dist) *max_dist = mx;
  return HOP_OK;
}

int hop_voxel_downsample(hop_ctx* c, const float* xyz, int n, float leaf, float* out_xyz, int cap, int* n_out) {
  if (!c || n < 0 || (n > 0 && !xyz) || !n_out || cap < 0) return HOP_E_INVALID;
  PHCHK(c, hipSetDevice(hop_ctx_device(c)));
  Physics* ph = physics(c);
  hipStream_t st = hop_ctx_stream(c);
  int rc = upload_planes(c, ph->tmp_cloud, xyz, n);
  if (rc) return rc;
  rc = voxel_downsample_device(c, ph, ph->tmp_cloud.x(), ph->tmp_cloud.y(), ph->tmp_cloud.z(), n, leaf, ph->tmp_cloud2);
  if (rc) return rc;
  const int m = ph->tmp_cloud2.n;
  *n_out = m;
  if (m > cap) return HOP_E_CAPACITY;
  if (m > 0 && out_xyz)
    for (int a = 0; a < 3; ++a)
      PHCHK(c, hop_ctx_d2h(c, out_xyz + (size_t)a * cap, ph->tmp_cloud2.buf.as<float>() + (size_t)a * m, sizeof(float) * (size_t)m));
  PHCHK(c, hipStreamSynchronize(st));
  return HOP_OK;
}

// normals through the crop: pcl::transformPointCloudWithNormals into the hand-base frame and back (main :72,94) rotates
// them twice in float
__global__ void k_rotate_twice(const float* __restrict__ inx, const float* __restrict__ iny, const float* __restrict__ inz, int n, const float* __restrict__ A,
                               const float* __restrict__ B, float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 r = m4_dir(B, m4_dir(A, v3(inx[i], iny[i], inz[i])));
  ox[i] = r.x, oy[i] = r.y, oz[i] = r.z;
}

static int scene_from_depth_impl(hop_ctx* c, const uint16_t* depth_raw, int H, int W, double depth_unit, const float K9[9], const float cam_in_handbase[16],
                                 const float handbase_in_cam[16], float leaf, const float crop_min[3], const float crop_max[3], bool with_normals,
                                 float max_depth_change_factor, float normal_smoothing_size, float* out_xyz, float* out_nrm, int cap, int* n_out, int* counts3);

int hop_scene_from_depth(hop_ctx* c, const uint16_t* depth_raw, int H, int W, double depth_unit, const float K9[9], const float cam_in_handbase[16],
                         const float handbase_in_cam[16], float leaf, const float crop_min[3], const float crop_max[3], float* out_xyz, int cap,
                         int* n_out, int* counts3) {
  return scene_from_depth_impl(c, depth_raw, H, W, depth_unit, K9, cam_in_handbase, handbase_in_cam, leaf, crop_min, crop_max, false, 0.f, 0.f, out_xyz, nullptr,
                               cap, n_out, counts3);
}
int hop_scene_from_depth_normals(hop_ctx* c, const uint16_t* depth_raw, int H, int W, double depth_unit, const float K9[9], const float cam_in_handbase[16],
                                 const float handbase_in_cam[16], float leaf, const float crop_min[3], const float crop_max[3], float max_depth_change_factor,
                                 float normal_smoothing_size, float* out_xyz, float* out_nrm, int cap, int* n_out, int* counts3) {
  if (!(normal_smoothing_size >= 1.f)) return HOP_E_INVALID;
  return scene_from_depth_impl(c, depth_raw, H, W, depth_unit, K9, cam_in_handbase, handbase_in_cam, leaf, crop_min, crop_max, true, max_depth_change_factor,
                               normal_smoothing_size, out_xyz, out_nrm, cap, n_out, counts3);
}

static int scene_from_depth_impl(hop_ctx* c, const uint16_t* depth_raw, int H, int W, double depth_unit, const float K9[9], const float cam_in_handbase[16],
                                 const float handbase_in_cam[16], float leaf, const float crop_min[3], const float crop_max[3], bool with_normals,
                                 float max_depth_change_factor, float normal_smoothing_size, float* out_xyz, float* out_nrm, int cap, int* n_out, int* counts3) {
  if (!c || !depth_raw || H <= 0 || W <= 0 || !K9 || !cam_in_handbase || !handbase_in_cam || !crop_min || !crop_max || !n_out || cap < 0) return HOP_E_INVALID;
  PHCHK(c, hipSetDevice(hop_ctx_device(c)));
  Physics* ph = physics(c);
  hipStream_t st = hop_ctx_stream(c);
  const int n = H * W;
  PHCHK(c, ph->keys_alt.ensure(sizeof(uint16_t) * (size_t)n));  // staging of the raw image
  PHCHK(c, hop_ctx_h2d(c, ph->keys_alt.p, depth_raw, sizeof(uint16_t) * (size_t)n));
  PHCHK(c, ph->tmp_cloud.buf.ensure(sizeof(float) * 3 * (size_t)n));
  ph->tmp_cloud.n = n;
  PHCHK(c, ph->scalars.ensure(sizeof(unsigned) * 16));
  PHCHK(c, hipMemsetAsync(ph->scalars.as<unsigned>() + 12, 0, sizeof(unsigned) * 2, st));
  float* t = ph->tmp_cloud.buf.as<float>();
  k_depth_to_cloud<<<(n + 255) / 256, 256, 0, st>>>(ph->keys_alt.as<unsigned short>(), H, W, depth_unit, K9[2], K9[0], K9[5], K9[4], t, t + n, t + 2 * (size_t)n,
                                                     ph->scalars.as<unsigned>() + 12);
  PHCHK(c, hipStreamSynchronize(st));  // keys_alt is reused by the voxel grid
  int rc = HOP_OK;
  if (with_normals) {
    // main :61: integral-image normals on the organised cloud (dropped pixels count as (0,0,0) there), before the pass-through
    PHCHK(c, ph->tmp_nrm.buf.ensure(sizeof(float) * 3 * (size_t)n));
    ph->tmp_nrm.n = n;
    float* q = ph->tmp_nrm.buf.as<float>();
    rc = hop_normals_ii_device(c, t, t + n, t + 2 * (size_t)n, H, W, max_depth_change_factor, normal_smoothing_size, 1, q, q + n, q + 2 * (size_t)n);
    if (rc) return rc;
    rc = voxel_downsample_device(c, ph, ph->tmp_cloud.x(), ph->tmp_cloud.y(), ph->tmp_cloud.z(), n, leaf, ph->tmp_cloud2, q, &ph->tmp_nrm2);
  } else {
    rc = voxel_downsample_device(c, ph, ph->tmp_cloud.x(), ph->tmp_cloud.y(), ph->tmp_cloud.z(), n, leaf, ph->tmp_cloud2);
  }
  if (rc) return rc;
  const int m = ph->tmp_cloud2.n;
  unsigned n_valid = 0, kept = 0;
  PHCHK(c, hop_ctx_d2h(c, &n_valid, ph->scalars.as<unsigned>() + 12, sizeof(unsigned)));
  if (m > 0) {
    PHCHK(c, ph->mats.ensure(sizeof(float) * 16 * 5));
    PHCHK(c, hop_ctx_h2d(c, ph->mats.as<float>(), cam_in_handbase, sizeof(float) * 16));
    PHCHK(c, hop_ctx_h2d(c, ph->mats.as<float>() + 16, handbase_in_cam, sizeof(float) * 16));
    PHCHK(c, ph->tmp_cloud.buf.ensure(sizeof(float) * 6 * (size_t)m));
    PHCHK(c, ph->flags.ensure(sizeof(unsigned) * (size_t)m));
    PHCHK(c, ph->pos.ensure(sizeof(unsigned) * (size_t)m));
    float* a = ph->tmp_cloud.buf.as<float>();
    float* b = a + 3 * (size_t)m;
    k_crop_handbase<<<(m + 255) / 256, 256, 0, st>>>(ph->tmp_cloud2.x(), ph->tmp_cloud2.y(), ph->tmp_cloud2.z(), m, ph->mats.as<float>(), ph->mats.as<float>() + 16,
                                                    make_float3(crop_min[0], crop_min[1], crop_min[2]), make_float3(crop_max[0], crop_max[1], crop_max[2]), a,
                                                    a + m, a + 2 * (size_t)m, ph->flags.as<unsigned>());
    size_t tmp = 0;
    PHCHK(c, prim_exclusive_sum(nullptr, tmp, ph->flags.as<unsigned>(), ph->pos.as<unsigned>(), m, st));
    PHCHK(c, ph->sort_tmp.ensure(tmp + 16));
    PHCHK(c, prim_exclusive_sum(ph->sort_tmp.p, tmp, ph->flags.as<unsigned>(), ph->pos.as<unsigned>(), m, st));
    k_compact3<<<(m + 255) / 256, 256, 0, st>>>(a, a + m, a + 2 * (size_t)m, ph->flags.as<unsigned>(), ph->pos.as<unsigned>(), m, b, b + m, b + 2 * (size_t)m, m,
                                               ph->scalars.as<unsigned>() + 13);
    PHCHK(c, hop_ctx_d2h(c, &kept, ph->scalars.as<unsigned>() + 13, sizeof(unsigned)));
    PHCHK(c, hipStreamSynchronize(st));
    if ((int)kept <= cap && out_xyz)
      for (int k = 0; k < 3; ++k)
        PHCHK(c, hop_ctx_d2h(c, out_xyz + (size_t)k * cap, b + (size_t)k * m, sizeof(float) * (size_t)kept));
    if (with_normals && out_nrm && (int)kept <= cap) {
      PHCHK(c, hipStreamSynchronize(st));
      const float* vn = ph->tmp_nrm2.buf.as<float>();
      k_rotate_twice<<<(m + 255) / 256, 256, 0, st>>>(vn, vn + m, vn + 2 * (size_t)m, m, ph->mats.as<float>(), ph->mats.as<float>() + 16, a, a + m, a + 2 * (size_t)m);
      k_compact3<<<(m + 255) / 256, 256, 0, st>>>(a, a + m, a + 2 * (size_t)m, ph->flags.as<unsigned>(), ph->pos.as<unsigned>(), m, b, b + m, b + 2 * (size_t)m, m,
                                                 ph->scalars.as<unsigned>() + 14);
      for (int k = 0; k < 3; ++k)
        PHCHK(c, hop_ctx_d2h(c, out_nrm + (size_t)k * cap, b + (size_t)k * m, sizeof(float) * (size_t)kept));
    }
  }
  PHCHK(c, hipStreamSynchronize(st));
  *n_out = (int)kept;
  if (counts3) counts3[0] = (int)n_valid, counts3[1] = m, counts3[2] = (int)kept;
  return (int)kept > cap ? HOP_E_CAPACITY : HOP_OK;
}

int hop_object_segment(hop_ctx* c, const float* xyz, const float* nrm, const float* conf, int n, float leaf, float* out_xyz, float* out_nrm,
                       float* out_conf, int cap, int* n_out) {
  if (!c || n < 0 || (n > 0 && (!xyz || !nrm || !conf)) || !n_out || cap < 0) return HOP_E_INVALID;
  PHCHK(c, hipSetDevice(hop_ctx_device(c)));
  Physics* ph = physics(c);
  hipStream_t st = hop_ctx_stream(c);
  *n_out = 0;
  if (n == 0) return HOP_OK;
  int rc = upload_planes(c, ph->tmp_cloud, xyz, n);
  if (rc) return rc;
  rc = upload_planes(c, ph->tmp_nrm, nrm, n);
  if (rc) return rc;
  PHCHK(c, ph->gather.ensure(sizeof(float) * (size_t)n));
  PHCHK(c, hop_ctx_h2d(c, ph->gather.p, conf, sizeof(float) * (size_t)n));
  rc = voxel_downsample_device(c, ph, ph->tmp_cloud.x(), ph->tmp_cloud.y(), ph->tmp_cloud.z(), n, leaf, ph->tmp_cloud2, ph->tmp_nrm.buf.as<float>(), &ph->tmp_nrm2);
  if (rc) return rc;
  const int m = ph->tmp_cloud2.n;
  *n_out = m;
  if (m == 0) return HOP_OK;
  if (m > cap) return HOP_E_CAPACITY;
  PHCHK(c, ph->tmp_pose.ensure(sizeof(float) * (size_t)m));
  float* on = ph->tmp_nrm2.buf.as<float>();
  k_segment_finish<<<(m + 3) / 4, 256, 0, st>>>(ph->tmp_cloud2.x(), ph->tmp_cloud2.y(), ph->tmp_cloud2.z(), m, on, on + m, on + 2 * (size_t)m, ph->tmp_cloud.x(),
                                               ph->tmp_cloud.y(), ph->tmp_cloud.z(), ph->gather.as<float>(), n, ph->tmp_pose.as<float>());
  PHCHK(c, hipGetLastError());
  for (int k = 0; k < 3; ++k) {
    if (out_xyz) PHCHK(c, hop_ctx_d2h(c, out_xyz + (size_t)k * cap, ph->tmp_cloud2.buf.as<float>() + (size_t)k * m, sizeof(float) * (size_t)m));
    if (out_nrm) PHCHK(c, hop_ctx_d2h(c, out_nrm + (size_t)k * cap, on + (size_t)k * m, sizeof(float) * (size_t)m));
  }
  if (out_conf) PHCHK(c, hop_ctx_d2h(c, out_conf, ph->tmp_pose.p, sizeof(float) * (size_t)m));
  PHCHK(c, hipStreamSynchronize(st));
  return HOP_OK;
}

int hop_hand_scene_filters(hop_ctx* c, const float* xyz, const float* nrm, int n, const float cam_in_handbase[16], float* hb_xyz, float* hb_nrm,
                           unsigned char* keep_noise, unsigned char* keep_swivel) {
  if (!c || n < 0 || (n > 0 && (!xyz || !nrm || !hb_xyz || !hb_nrm || !keep_noise || !keep_swivel)) || !cam_in_handbase) return HOP_E_INVALID;
  PHCHK(c, hipSetDevice(hop_ctx_device(c)));
  Physics* ph = physics(c);
  hipStream_t st = hop_ctx_stream(c);
  if (n == 0) return HOP_OK;
  int rc = upload_planes(c, ph->tmp_cloud, xyz, n);
  if (rc) return rc;
  rc = upload_planes(c, ph->tmp_nrm, nrm, n);
  if (rc) return rc;
  PHCHK(c, ph->mats.ensure(sizeof(float) * 16 * 5));
  PHCHK(c, hop_ctx_h2d(c, ph->mats.p, cam_in_handbase, sizeof(float) * 16));
  PHCHK(c, ph->tmp_cloud2.buf.ensure(sizeof(float) * 3 * (size_t)n));
  PHCHK(c, ph->tmp_nrm2.buf.ensure(sizeof(float) * 3 * (size_t)n));
  ph->tmp_cloud2.n = n;
  float* hb = ph->tmp_cloud2.buf.as<float>();
  float* hbn = ph->tmp_nrm2.buf.as<float>();
  k_transform_cloud_nrm<<<(n + 255) / 256, 256, 0, st>>>(ph->tmp_cloud.x(), ph->tmp_cloud.y(), ph->tmp_cloud.z(), ph->tmp_nrm.buf.as<float>(),
                                                        ph->tmp_nrm.buf.as<float>() + n, ph->tmp_nrm.buf.as<float>() + 2 * (size_t)n, n, ph->mats.as<float>(), hb, hbn);
  PHCHK(c, ph->flags.ensure(4 * (size_t)n + 16));  // four byte masks
  unsigned char* live0 = ph->flags.as<unsigned char>();
  unsigned char *live1 = live0 + n, *live2 = live1 + n, *sw = live2 + n;
  PHCHK(c, hipMemsetAsync(live0, 1, (size_t)n, st));
  k_radius_outlier<<<(n + 3) / 4, 256, 0, st>>>(hb, hb + n, hb + 2 * (size_t)n, n, live0, 0.02f * 0.02f, 30, live1);   // Hand.cpp:293-299
  k_radius_outlier<<<(n + 3) / 4, 256, 0, st>>>(hb, hb + n, hb + 2 * (size_t)n, n, live1, 0.04f * 0.04f, 100, live2);  // :300-306
  PHCHK(c, ph->pos.ensure(sizeof(float) * (size_t)n));
  float* dist = ph->pos.as<float>();
  k_sor_mean<<<(n + 3) / 4, 256, 0, st>>>(hb, hb + n, hb + 2 * (size_t)n, n, live2, dist);  // :307-313
  std::vector<float> dh(n);
  std::vector<unsigned char> lh(n);
  PHCHK(c, hop_ctx_d2h(c, dh.data(), dist, sizeof(float) * (size_t)n));
  PHCHK(c, hop_ctx_d2h(c, lh.data(), live2, (size_t)n));
  PHCHK(c, hipStreamSynchronize(st));
  // mean and standard deviation of the per-point mean distances, in double, in point order (statistical_outlier_removal.hpp)
  double sum = 0, sq_sum = 0;
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (lh[i]) sum += dh[i], sq_sum += (double)dh[i] * dh[i], ++m;
  const int use_sor = m > SOR_K;
  double thr = 0;
  if (use_sor) {
    const double mean = sum / (double)m;
    const double variance = (sq_sum - sum * sum / (double)m) / ((double)m - 1);
    thr = mean + 2.0 * std::sqrt(variance);
  }
  k_sor_apply<<<(n + 255) / 256, 256, 0, st>>>(hb, n, live2, dist, thr, use_sor, live0, sw);
  PHCHK(c, hipGetLastError());
  PHCHK(c, hop_ctx_d2h(c, hb_xyz, hb, sizeof(float) * 3 * (size_t)n));
  PHCHK(c, hop_ctx_d2h(c, hb_nrm, hbn, sizeof(float) * 3 * (size_t)n));
  PHCHK(c, hop_ctx_d2h(c, keep_noise, live0, (size_t)n));
  PHCHK(c, hop_ctx_d2h(c, keep_swivel, sw, (size_t)n));
  PHCHK(c, hipStreamSynchronize(st));
  return HOP_OK;
}

int hop_voxel_downsample_normals(hop_ctx* c, const float* xyz, const float* nrm, int n, float leaf, float* out_xyz, float* out_nrm, int cap, int* n_out) {
  if (!c || n < 0 || (n > 0 && (!xyz || !nrm)) || !n_out || cap < 0) return HOP_E_INVALID;
  PHCHK(c, hipSetDevice(hop_ctx_device(c)));
  Physics* ph = physics(c);
  hipStream_t st = hop_ctx_stream(c);
  *n_out = 0;
  if (n == 0) return HOP_OK;
  int rc = upload_planes(c, ph->tmp_cloud, xyz, n);
  if (rc) return rc;
  rc = upload_planes(c, ph->tmp_nrm, nrm, n);
  if (rc) return rc;
  rc = voxel_downsample_device(c, ph, ph->tmp_cloud.x(), ph->tmp_cloud.y(), ph->tmp_cloud.z(), n, leaf, ph->tmp_cloud2, ph->tmp_nrm.buf.as<float>(), &ph->tmp_nrm2);
  if (rc) return rc;
  const int m = ph->tmp_cloud2.n;
  *n_out = m;
  if (m > cap) return HOP_E_CAPACITY;
  for (int k = 0; k < 3 && m > 0; ++k) {
    if (out_xyz) PHCHK(c, hop_ctx_d2h(c, out_xyz + (size_t)k * cap, ph->tmp_cloud2.buf.as<float>() + (size_t)k * m, sizeof(float) * (size_t)m));
    if (out_nrm) PHCHK(c, hop_ctx_d2h(c, out_nrm + (size_t)k * cap, ph->tmp_nrm2.buf.as<float>() + (size_t)k * m, sizeof(float) * (size_t)m));
  }
  PHCHK(c, hipStreamSynchronize(st));
  return HOP_OK;
}

int hop_handbase_region(hop_ctx* c, const float* xyz, const float* nrm, int n, const float cam_in_handbase[16], float y1, float z1, float y2, float z2,
                        float* hb_xyz, float* hb_nrm, unsigned char* keep) {
  if (!c || n < 0 || (n > 0 && (!xyz || !nrm || !hb_xyz || !hb_nrm || !keep)) || !cam_in_handbase) return HOP_E_INVALID;
  PHCHK(c, hipSetDevice(hop_ctx_device(c)));
  Physics* ph = physics(c);
  hipStream_t st = hop_ctx_stream(c);
  if (n == 0) return HOP_OK;
  int rc = upload_planes(c, ph->tmp_cloud, xyz, n);
  if (rc) return rc;
  rc = upload_planes(c, ph->tmp_nrm, nrm, n);
  if (rc) return rc;
  PHCHK(c, ph->mats.ensure(sizeof(float) * 16 * 5));
  PHCHK(c, hop_ctx_h2d(c, ph->mats.p, cam_in_handbase, sizeof(float) * 16));
  PHCHK(c, ph->tmp_cloud2.buf.ensure(sizeof(float) * 3 * (size_t)n));
  PHCHK(c, ph->tmp_nrm2.buf.ensure(sizeof(float) * 3 * (size_t)n));
  PHCHK(c, ph->flags.ensure((size_t)n + 16));
  const float* nn = ph->tmp_nrm.buf.as<float>();
  k_handbase_region<<<(n + 255) / 256, 256, 0, st>>>(ph->tmp_cloud.x(), ph->tmp_cloud.y(), ph->tmp_cloud.z(), nn, nn + n, nn + 2 * (size_t)n, n, ph->mats.as<float>(), y1, z1,
                                                    y2, z2, ph->tmp_cloud2.buf.as<float>(), ph->tmp_nrm2.buf.as<float>(), ph->flags.as<unsigned char>());
  PHCHK(c, hipGetLastError());
  PHCHK(c, hop_ctx_d2h(c, hb_xyz, ph->tmp_cloud2.buf.p, sizeof(float) * 3 * (size_t)n));
  PHCHK(c, hop_ctx_d2h(c, hb_nrm, ph->tmp_nrm2.buf.p, sizeof(float) * 3 * (size_t)n));
  PHCHK(c, hop_ctx_d2h(c, keep, ph->flags.p, (size_t)n));
  PHCHK(c, hipStreamSynchronize(st));
  return HOP_OK;
}

int hop_hand_height_matches(hop_ctx* c, const float* scene_xyz, const float* scene_nrm, int n_scene, const float* hand_xyz, const float* hand_nrm,
                            int n_hand, const float* heights, int n_heights, int* counts) {
  if (!c || n_scene < 0 || n_hand < 0 || n_heights <= 0 || n_heights > 64 || !heights || !counts || (n_scene > 0 && (!scene_xyz || !scene_nrm)) ||
      (n_hand > 0 && (!hand_xyz || !hand_nrm)))
    return HOP_E_INVALID;
  PHCHK(c, hipSetDevice(hop_ctx_device(c)));
  Physics* ph = physics(c);
  hipStream_t st = hop_ctx_stream(c);
  for (int t = 0; t < n_heights; ++t) counts[t] = 0;
  if (n_hand == 0 || n_scene == 0) return HOP_OK;
  int rc = upload_planes(c, ph->tmp_cloud, scene_xyz, n_scene);
  if (rc) return rc;
  rc = upload_planes(c, ph->tmp_nrm, scene_nrm, n_scene);
  if (rc) return rc;
  rc = upload_planes(c, ph->tmp_cloud2, hand_xyz, n_hand);
  if (rc) return rc;
  rc = upload_planes(c, ph->tmp_nrm2, hand_nrm, n_hand);
  if (rc) return rc;
  PHCHK(c, ph->scalars.ensure(sizeof(unsigned) * 16));
  PHCHK(c, ph->mats.ensure(sizeof(float) * 16 * 5));
  PHCHK(c, ph->gather.ensure(sizeof(int) * 64));
  PHCHK(c, hop_ctx_h2d(c, ph->mats.p, heights, sizeof(float) * (size_t)n_heights));
  PHCHK(c, hipMemsetAsync(ph->gather.p, 0, sizeof(int) * 64, st));
  const float *sn = ph->tmp_nrm.buf.as<float>(), *hn = ph->tmp_nrm2.buf.as<float>();
  k_hand_height<<<dim3((n_hand + 3) / 4, n_heights), 256, 0, st>>>(ph->tmp_cloud.x(), ph->tmp_cloud.y(), ph->tmp_cloud.z(), sn, sn + n_scene, sn + 2 * (size_t)n_scene, n_scene,
                                                                  ph->tmp_cloud2.x(), ph->tmp_cloud2.y(), ph->tmp_cloud2.z(), hn, hn + n_hand, hn + 2 * (size_t)n_hand, n_hand,
                                                                  ph->mats.as<float>(), ph->gather.as<int>());
  PHCHK(c, hipGetLastError());
  PHCHK(c, hop_ctx_d2h(c, counts, ph->gather.p, sizeof(int) * (size_t)n_heights));
  PHCHK(c, hipStreamSynchronize(st));
  return HOP_OK;
}

int hop_physics_set_frame(hop_ctx* c, const hop_physics_args* a) {
  if (!c || !a) return HOP_E_INVALID;
  PHCHK(c, hipSetDevice(hop_ctx_device(c)));
  Physics* ph = physics(c);
  hipStream_t st = hop_ctx_stream(c);
  if (a->object_mesh < 0 || a->object_mesh >= MAX_MESHES || !ph->mesh[a->object_mesh].valid) return HOP_E_STATE;
  for (int k = 0; k < 4; ++k)
    if (a->finger_mesh[k] < 0 || a->finger_mesh[k] >= MAX_MESHES || !ph->mesh[a->finger_mesh[k]].valid) return HOP_E_STATE;
  if (a->n_model < 0 || a->n_hand_cloud < 0 || a->n_cloud_without_hand < 0) return HOP_E_INVALID;
  if (!ph->ev[0]) {
    PHCHK(c, hipEventCreate(&ph->ev[0]));
    PHCHK(c, hipEventCreate(&ph->ev[1]));
  }
  PHCHK(c, hipEventRecord(ph->ev[0], st));
  PhysParams& P = ph->P;
  std::memcpy(P.cam2handbase, a->cam2handbase, sizeof(P.cam2handbase));
  std::memcpy(P.center_init, a->model_center_init, sizeof(P.center_init));
  P.ob_diameter = a->ob_diameter;
  P.collision_dist = std::min(-a->smallest_dim * a->collision_thres, -0.007f);  // PoseEstimator.cpp:563-564
  P.inside_ob_dist = std::min(-a->smallest_dim / 5, -0.01f);
  P.non_touch_dist = a->non_touch_dist;
  P.collision_finger_dist = -a->collision_finger_dist;
  P.volume_ratio = a->collision_finger_volume_ratio;
  P.object_mesh = a->object_mesh;
  P.n_model = a->n_model;
  // finger clouds that take part in the third check (:538-551, :650-652), moved into the hand-base frame
  int total = 0;
  for (int k = 0; k < 4; ++k) {
    P.finger_status[k] = a->finger_status[k] != 0;
    P.finger_mesh[k] = a->finger_mesh[k];
    bool act = P.finger_status[k] != 0;
    if (!a->finger_status[0] && (k == 0 || k == 1)) act = false;
    if (!a->finger_status[2] && (k == 2 || k == 3)) act = false;
    if (act && (a->finger_n[k] < 0 || (a->finger_n[k] > 0 && (!a->finger_xyz[k] || !a->finger2handbase[k])))) return HOP_E_INVALID;
    P.finger_active[k] = act;
    P.finger_off[k] = total;
    if (act) total += a->finger_n[k];
  }
  P.finger_off[4] = total;
  PHCHK(c, ph->fingers.buf.ensure(sizeof(float) * 3 * (size_t)std::max(total, 1)));
  ph->fingers.n = total;
  PHCHK(c, ph->mats.ensure(sizeof(float) * 16 * 5));
  for (int k = 0; k < 4; ++k) {
    if (!P.finger_active[k] || a->finger_n[k] == 0) continue;
    const int n = a->finger_n[k];
    int rc = upload_planes(c, ph->tmp_cloud, a->finger_xyz[k], n);
    if (rc) return rc;
    PHCHK(c, hop_ctx_h2d(c, ph->mats.as<float>() + 16 * k, a->finger2handbase[k], sizeof(float) * 16));
    float* o = ph->fingers.buf.as<float>() + P.finger_off[k];
    k_transform_cloud<<<(n + 255) / 256, 256, 0, st>>>(ph->tmp_cloud.x(), ph->tmp_cloud.y(), ph->tmp_cloud.z(), n, ph->mats.as<float>() + 16 * k, o, o + total,
                                                       o + 2 * (size_t)total);
    PHCHK(c, hipStreamSynchronize(st));  // tmp_cloud is reused by the next finger
  }
  // scene without the hand: hand-base frame, then the 5 mm voxel grid (:553-557)
  {
    const int n = a->n_cloud_without_hand;
    int rc = upload_planes(c, ph->tmp_cloud, a->cloud_without_hand_xyz, n);
    if (rc) return rc;
    PHCHK(c, ph->tmp_cloud2.buf.ensure(sizeof(float) * 3 * (size_t)std::max(n, 1)));
    ph->tmp_cloud2.n = n;
    PHCHK(c, hop_ctx_h2d(c, ph->mats.as<float>() + 64, a->cam2handbase, sizeof(float) * 16));
    if (n > 0) {
      float* o = ph->tmp_cloud2.buf.as<float>();
      k_transform_cloud<<<(n + 255) / 256, 256, 0, st>>>(ph->tmp_cloud.x(), ph->tmp_cloud.y(), ph->tmp_cloud.z(), n, ph->mats.as<float>() + 64, o, o + n,
                                                         o + 2 * (size_t)n);
    }
    rc = voxel_downsample_device(c, ph, ph->tmp_cloud2.x(), ph->tmp_cloud2.y(), ph->tmp_cloud2.z(), n, a->voxel_size, ph->cwh_ds);
    if (rc) return rc;
  }
  int rc = upload_planes(c, ph->hand, a->hand_cloud_xyz, a->n_hand_cloud);
  if (rc) return rc;
  rc = upload_planes(c, ph->model, a->model_xyz, a->n_model);
  if (rc) return rc;
  PHCHK(c, hipEventRecord(ph->ev[1], st));
  PHCHK(c, hipStreamSynchronize(st));
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ph->ev[0], ph->ev[1]);
  ph->ms_frame = ms;
  ph->have_frame = true;
  return HOP_OK;
}

int hop_reject_by_collision(hop_ctx* c, unsigned char* keep_out, float* diag8_out, int* n_in_out) {
  if (!c) return HOP_E_INVALID;
  PHCHK(c, hipSetDevice(hop_ctx_device(c)));
  Physics* ph = physics(c);
  if (!ph->have_frame) return HOP_E_STATE;
  hipStream_t st = hop_ctx_stream(c);
  const HopHypView hv = hop_ctx_hyp(c);
  const int H = hv.n;
  if (n_in_out) *n_in_out = H;
  if (H == 0) return HOP_OK;
  const PhysParams& P = ph->P;
  PHCHK(c, ph->xf.ensure(sizeof(float) * XF * (size_t)H));
  PHCHK(c, ph->stage.ensure(sizeof(int) * (size_t)H));
  PHCHK(c, ph->fmin.ensure(sizeof(unsigned) * 12 * (size_t)H));
  PHCHK(c, ph->diag.ensure(sizeof(float) * 8 * (size_t)H));
  float* xf = ph->xf.as<float>();
  int* stage = ph->stage.as<int>();
  unsigned* fmin = ph->fmin.as<unsigned>();
  float* diag = ph->diag.as<float>();
  const SdfMeshDev obj = ph->mesh[P.object_mesh].dev;
  PHCHK(c, hipEventRecord(ph->ev[0], st));
  k_phys_prepare<<<(H + 63) / 64, 64, 0, st>>>(hv.pose, H, P, xf, stage, fmin, diag);
  k_phys_center<<<H, 256, 0, st>>>(obj, xf, H, P, ph->cwh_ds.x(), ph->cwh_ds.y(), ph->cwh_ds.z(), ph->cwh_ds.n, ph->hand.x(), ph->hand.y(), ph->hand.z(),
                                   ph->hand.n, stage, diag);
  const int nfp = P.finger_off[4];
  if (nfp > 0)
    k_phys_fingers<<<dim3((nfp + SDF_BLOCK - 1) / SDF_BLOCK, H), SDF_BLOCK, 0, st>>>(obj, xf, P, ph->fingers.x(), ph->fingers.y(), ph->fingers.z(), stage, fmin);
  k_phys_decide_fingers<<<(H + 63) / 64, 64, 0, st>>>(H, P, fmin, stage, diag);
  FingerMeshes fm;
  for (int k = 0; k < 4; ++k) fm.m[k] = ph->mesh[P.finger_mesh[k]].dev;
  if (P.n_model > 0)
    k_phys_model<<<dim3((P.n_model + SDF_BLOCK - 1) / SDF_BLOCK, H), SDF_BLOCK, 0, st>>>(fm, xf, P.n_model, ph->model.x(), ph->model.y(), ph->model.z(), stage,
                                                                                         fmin);
  k_phys_decide_model<<<(H + 63) / 64, 64, 0, st>>>(H, P, fmin, stage, diag);
  PHCHK(c, hipGetLastError());
  PHCHK(c, hipEventRecord(ph->ev[1], st));
  std::vector<int> stg(H);
  PHCHK(c, hop_ctx_d2h(c, stg.data(), stage, sizeof(int) * (size_t)H));
  if (diag8_out) PHCHK(c, hop_ctx_d2h(c, diag8_out, diag, sizeof(float) * 8 * (size_t)H));
  PHCHK(c, hipStreamSynchronize(st));
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ph->ev[0], ph->ev[1]);
  ph->ms_reject = ms;
  // survivors keep their order (the reference's order depends on OpenMP scheduling, :730-731)
  std::vector<int> src;
  src.reserve(H);
  for (int h = 0; h < H; ++h) {
    if (keep_out) keep_out[h] = stg[h] == 0;
    if (stg[h] == 0) src.push_back(h);
  }
  const int K = (int)src.size();
  if (K < H && K > 0) {
    PHCHK(c, ph->gather.ensure(sizeof(int) * (size_t)K));
    PHCHK(c, ph->tmp_pose.ensure(sizeof(float) * 16 * (size_t)K));
    PHCHK(c, ph->tmp_score.ensure(sizeof(float) * (size_t)K));
    PHCHK(c, ph->tmp_id.ensure(sizeof(int) * (size_t)K));
    PHCHK(c, hop_ctx_h2d(c, ph->gather.p, src.data(), sizeof(int) * (size_t)K));
    k_phys_gather<<<(K * 16 + 255) / 256, 256, 0, st>>>(ph->gather.as<int>(), K, hv.pose, hv.score, hv.id, ph->tmp_pose.as<float>(), ph->tmp_score.as<float>(),
                                                        ph->tmp_id.as<int>());
    PHCHK(c, hipMemcpyAsync(hv.pose, ph->tmp_pose.p, sizeof(float) * 16 * (size_t)K, hipMemcpyDeviceToDevice, st));
    PHCHK(c, hipMemcpyAsync(hv.score, ph->tmp_score.p, sizeof(float) * (size_t)K, hipMemcpyDeviceToDevice, st));
    PHCHK(c, hipMemcpyAsync(hv.id, ph->tmp_id.p, sizeof(int) * (size_t)K, hipMemcpyDeviceToDevice, st));
    PHCHK(c, hipStreamSynchronize(st));
  }
  hop_ctx_hyp_set_count(c, K);
  return HOP_OK;
}

#ifdef SDF_COUNT
int hop_sdf_counters(unsigned long long* out4, int reset) {
  unsigned long long z[4] = {0, 0, 0, 0};
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_sdf_cnt), sizeof(z));
  if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sdf_cnt), z, sizeof(z));
  return 0;
}
#endif
int hop_physics_timing(hop_ctx* c, double* ms_set_frame, double* ms_reject) {
  if (!c) return HOP_E_INVALID;
  Physics* ph = physics(c);
  if (ms_set_frame) *ms_set_frame = ph->ms_frame;
  if (ms_reject) *ms_reject = ph->ms_reject;
  return HOP_OK;
}

}  // extern "C"
