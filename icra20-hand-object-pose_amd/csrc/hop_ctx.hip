// hop_ctx.hip -- host side of libhop: device memory, streams, the generator's sequential base selection,
// batching of the per-base kernels, and the C-ABI of include/hop.h.
//
// What stays on the host and why
//   * base selection (MatchBase::SelectRandomTriangle / Match4pcsBase::SelectQuadrilateral,
//     matchBase.hpp:111-212, match4pcsBase.hpp:107-189) is a sequential, RNG-driven process whose draws
//     depend on the outcome of the previous draw (probability annealing, std::discrete_distribution
//     rebuilt per draw).  Its only data-parallel part -- PPF key membership of point pairs -- is computed
//     once on the GPU as an N x N bit matrix (k_ppf_matrix); the host then only reads bits.
//   * clusterPoses (PoseEstimator.cpp:106-233) is a greedy pass over a sorted list (host in the reference too).
//   * the PSO bookkeeping (pso.hpp:146-351) -- a few hundred doubles per generation.
// Everything that touches clouds runs in the kernels of hop_kernels.hip.
#include "../../include/hop.h"
#include "hop_device.h"
#include "hop_select.h"
#include "hop_ctx_ext.h"

#include "hop_prim.h"
#include <sys/mman.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <random>
#include <string>
#include <vector>

using namespace hop;

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
namespace {

struct GridStore {
  DevBuf cell_start_d, pts_d;
  hop::GridDev g{};
  bool valid = false;
  float cell = 0;
  void release() {
    cell_start_d.release();
    pts_d.release();
    valid = false;
  }
};

struct CellListStore {
  DevBuf start_d, pts_d, nrm_d, u2_d, count_d, work_d, keep_d, range_d, nrm_idx_d, pts_idx_d, rec_d, qlist_d, head_d;
  hop::CellListDev c{};
  bool valid = false;
  float cell = 0, max_dist = 0, coord_mag = 0;
  bool packed = false, pack_requested = false;
  int sub = 1;  // per-frame lists: subdivision of the ring grid they were built with
  float avg_len = 0;  // entries per voxel with candidates at the last build (chooses the lanes per voxel of the next one)
  size_t total_entries = 0;  // length of pts at the last build
  void release() {
    start_d.release(), pts_d.release(), nrm_d.release(), u2_d.release(), count_d.release(), work_d.release(), keep_d.release(), range_d.release();
    nrm_idx_d.release(), pts_idx_d.release(), rec_d.release(), qlist_d.release(), head_d.release();
    valid = false, c.head = nullptr;
  }
};

struct CloudDevice {
  DevBuf buf;  // 6 planes
  int n = 0;
  const float* plane(int k) const { return buf.as<float>() + (size_t)k * n; }
};

enum TimeCat { T_VERIFY = 0, T_GEN_OTHER, T_ICP_NN, T_ICP_SOLVE, T_LCP_FWD, T_LCP_REV, T_PSO, T_PPF, T_ICP_ACCUM, T_LCP_SUM, T_QUADS, T_BUILD, T_NCAT };

struct TimedSpan {
  hipEvent_t a, b;
  int cat;
};

struct BaseTraceHost {
  int ids[4];
  float inv1, inv2;
  int n1 = 0, n2 = 0, nq = 0;
};

}  // namespace

struct hop_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string last_error;

  // host-side clouds and generator state (scene = _scene_high_confidence)
  GenState gen;
  CloudDevice scene_d;
  CloudDevice scene_unit_d, scene_sorted_unit_d;  // planes 3..5: unit normals (computeLCP on cell lists), caller / Morton order
  CloudDevice scene_sorted_d;  // scoring copy in Morton order (grid paths), + permutation back to caller order
  DevBuf scene_perm_d;
  DevBuf scene_sorted_aos_d;  // the same Morton-ordered copy as AoS: [m] float4 points, then [m] float4 normals (fused ICP kernel)
  CloudDevice model_d[2];

  // PPF key set
  std::vector<unsigned> key_bitmap;
  int key_dist_bins = 0;
  DevBuf key_bitmap_d;
  bool have_keys = false;

  // generator state of the last run
  CloudDevice gp_d;          // planes: x y z + ppf normals
  CloudDevice gq_d;          // planes: x y z nx ny nz
  DevBuf gq_unit_d;          // 3 planes
  DevBuf ppf_matrix_d;
  PinnedBuf ppf_matrix_h;
  // pageable copy of the membership matrix for the host selection (2 MiB aligned, transparent huge pages requested:
  // the selection makes one random access into it per iteration)
  unsigned long long* ppf_matrix_cached = nullptr;
  size_t ppf_matrix_cached_bytes = 0;
  bool ppf_matrix_registered = false;
  float coord_mag = 4.f;  // power of two >= 4 m bounding every coordinate handed over so far (float error scale of the lists)
  DevBuf fit_queue_d, fit_count_d, angle_thr_d, sur_in, sur_ws, sur_links, sur_out;
  bool angle_thr_tried = false, angle_thr_ok = false;
  int ppf_words = 0;
  std::vector<BaseTraceHost> trace;
  bool have_gen_state = false;
  // verify clouds set explicitly (hop_verify_set_clouds)
  CloudDevice vp_d, vq_d;
  std::vector<float> vp_h[3];
  bool have_verify_clouds = false;
  // voxel grids: centred P for verify_mode 1; model rest frames and scene for nn_mode 1
  GridStore verify_grid, model_grid[2], scene_grid, hand_grid;
  std::vector<float> hand_scene_h[3];
  CellListStore model_cells[2], scene_cells, verify_cells, hand_cells;
  float grid_delta = 0;

  // batch workspaces of the generator
  DevBuf bases_d, pairs1_d, pairs2_d, cnt_d, elems_d, queries_d, cands_d, cand_counts_d, counters_d;
  PinnedBuf bases_h, cnt_h;
  // resident hypothesis set
  DevBuf hyp_pose, hyp_score, hyp_id, hyp_key, hyp_inv, tmp_pose, tmp_score, tmp_id, sort_keys_alt, sort_vals, sort_vals_alt, sort_tmp;
  int n_hyp = 0;

  // scoring workspaces
  DevBuf lcp_rev_idx, lcp_rev_d2, lcp_terms, icp_moved, icp_partial, icp_state, icp_iters, icp_conv, icp_corr_idx, icp_hist, icp_lm, pose_inv, topk_rows;

  // hand
  CloudDevice hand_scene_d, hand_lookup_d, hand_swivel_d, hand_model_d;
  int hand_n_scene = 0, hand_n_lookup = 0, hand_n_swivel = 0;
  hop_finger_args finger{};
  std::vector<float> finger_hist;
  int pso_sum_mode = 0;  // hop_hand_set_sum_mode
  DevBuf finger_hist_d, pso_particles_d, pso_match_d, pso_terms_d, pso_sum_d, pso_cnt_d;
  PinnedBuf pso_particles_h, pso_out_h;
  PinnedBuf stage_up, stage_down;  // hop_ctx_h2d / hop_ctx_d2h
  size_t stage_cur = 0;
  int icp_last_engine = -1;  // nn_mode 7's moment kernel of the last hop_icp_refine: 1 matrix cores (k_icp_fusedq_momm), 0 vector units (k_icp_fusedq_momi)
  bool have_finger = false, have_hand_scene = false;

  // row modules (hop_physics.hip)
  HopExt* ext[HOP_EXT_SLOTS] = {};

  // timing
  bool timing_on = false;
  std::vector<TimedSpan> spans;
  std::vector<hipEvent_t> event_pool;
  hop_timing timing{};
};

namespace {

#define HIPCHK(ctx, call)                                                                        \
  do {                                                                                           \
    hipError_t _e = (call);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(_e);                     \
      return HOP_E_HIP;                                                                          \
    }                                                                                            \
  } while (0)

hipEvent_t get_event(hop_ctx* c) {
  if (!c->event_pool.empty()) {
    hipEvent_t e = c->event_pool.back();
    c->event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
struct SpanGuard {
  hop_ctx* c;
  int cat;
  hipEvent_t a{}, b{};
  SpanGuard(hop_ctx* ctx, int category) : c(ctx), cat(category) {
    if (c->timing_on) {
      a = get_event(c);
      b = get_event(c);
      (void)hipEventRecord(a, c->stream);
    }
  }
  ~SpanGuard() {
    if (c->timing_on) {
      (void)hipEventRecord(b, c->stream);
      c->spans.push_back({a, b, cat});
    }
  }
};
void resolve_spans(hop_ctx* c) {
  if (c->spans.empty()) return;
  (void)hipStreamSynchronize(c->stream);
  for (auto& s : c->spans) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, s.a, s.b);
    double* slot = nullptr;
    switch (s.cat) {
      case T_VERIFY: slot = &c->timing.ms_verify; break;
      case T_GEN_OTHER: slot = &c->timing.ms_gen_other; break;
      case T_ICP_NN: slot = &c->timing.ms_icp_nn; break;
      case T_ICP_SOLVE: slot = &c->timing.ms_icp_solve; break;
      case T_LCP_FWD: slot = &c->timing.ms_lcp_fwd; break;
      case T_LCP_REV: slot = &c->timing.ms_lcp_rev; break;
      case T_PSO: slot = &c->timing.ms_pso; break;
      case T_PPF: slot = &c->timing.ms_ppf_matrix; break;
      case T_ICP_ACCUM: slot = &c->timing.ms_icp_accum; break;
      case T_LCP_SUM: slot = &c->timing.ms_lcp_sum; break;
      case T_QUADS: slot = &c->timing.ms_quads; break;
      case T_BUILD: slot = &c->timing.ms_build; break;
    }
    if (slot) *slot += ms;
    c->event_pool.push_back(s.a);
    c->event_pool.push_back(s.b);
  }
  c->spans.clear();
}

int upload_cloud(hop_ctx* c, CloudDevice& d, const CloudHost& h) {
  d.n = h.n;
  HIPCHK(c, d.buf.ensure(sizeof(float) * 6 * (size_t)std::max(h.n, 1)));
  if (h.n == 0) return HOP_OK;
  const std::vector<float>* pl[6] = {&h.x, &h.y, &h.z, &h.nx, &h.ny, &h.nz};
  for (int k = 0; k < 6; ++k)
    HIPCHK(c, hop_ctx_h2d(c, d.buf.as<float>() + (size_t)k * h.n, pl[k]->data(), sizeof(float) * h.n));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

NsetGeom make_nset_geom(float eps) {
  NsetGeom g;
  g.nepsilon = (float)((double)(1.f / 7.f) + 0.00001);
  const int depth = (int)(-std::log2(eps));
  g.eg_size = (int)std::pow(2, depth);
  g.epsilon = 1.f / g.eg_size;
  return g;
}

// Voxel grid over a fixed cloud (counting sort by cell on the host: one pass over n points, then upload).
// pts[k].w carries the original index of the point.
int build_grid(hop_ctx* c, GridStore& gs, const float* x, const float* y, const float* z, int n, float cell) {
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = 0; i < n; ++i) {
    mn[0] = std::min(mn[0], x[i]), mn[1] = std::min(mn[1], y[i]), mn[2] = std::min(mn[2], z[i]);
    mx[0] = std::max(mx[0], x[i]), mx[1] = std::max(mx[1], y[i]), mx[2] = std::max(mx[2], z[i]);
  }
  GridDev g{};
  g.ox = mn[0], g.oy = mn[1], g.oz = mn[2];
  g.cell = cell;
  g.inv_cell = 1.0f / cell;
  g.dx = std::max(1, (int)std::floor((mx[0] - mn[0]) * g.inv_cell) + 1);
  g.dy = std::max(1, (int)std::floor((mx[1] - mn[1]) * g.inv_cell) + 1);
  g.dz = std::max(1, (int)std::floor((mx[2] - mn[2]) * g.inv_cell) + 1);
  const size_t ncell = (size_t)g.dx * g.dy * g.dz;
  if (ncell > (size_t)1 << 28) return HOP_E_CAPACITY;
  std::vector<int> start(ncell + 1, 0), cell_of(n);
  for (int i = 0; i < n; ++i) {
    int cx = (int)std::floor((x[i] - g.ox) * g.inv_cell), cy = (int)std::floor((y[i] - g.oy) * g.inv_cell),
        cz = (int)std::floor((z[i] - g.oz) * g.inv_cell);
    cx = std::min(std::max(cx, 0), g.dx - 1), cy = std::min(std::max(cy, 0), g.dy - 1), cz = std::min(std::max(cz, 0), g.dz - 1);
    cell_of[i] = (cz * g.dy + cy) * g.dx + cx;
    start[cell_of[i] + 1]++;
  }
  for (size_t k = 0; k < ncell; ++k) start[k + 1] += start[k];
  std::vector<int> fill(start.begin(), start.end() - 1);
  std::vector<float4> pts(std::max(n, 1));
  for (int i = 0; i < n; ++i) {
    float w;
    std::memcpy(&w, &i, 4);
    pts[fill[cell_of[i]]++] = make_float4(x[i], y[i], z[i], w);
  }
  HIPCHK(c, gs.cell_start_d.ensure(sizeof(int) * (ncell + 1)));
  HIPCHK(c, gs.pts_d.ensure(sizeof(float4) * (size_t)std::max(n, 1)));
  HIPCHK(c, hop_ctx_h2d(c, gs.cell_start_d.p, start.data(), sizeof(int) * (ncell + 1)));
  HIPCHK(c, hop_ctx_h2d(c, gs.pts_d.p, pts.data(), sizeof(float4) * (size_t)std::max(n, 1)));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  g.cell_start = gs.cell_start_d.as<int>();
  g.pts = gs.pts_d.as<float4>();
  gs.g = g;
  gs.valid = true;
  gs.cell = cell;
  return HOP_OK;
}

// NN cell lists of a cloud (device-resident planes x,y,z), padded by max_dist: three small kernels and one scan
// packed: additionally the 16-byte cell records / 8-byte quantised entries of CellListDev::rec (ICP lookups, cells_nnq);
// packing runs on the host (once per model and gating distance, like the lists themselves).
int build_cell_lists(hop_ctx* c, CellListStore& cs, const CloudHost& h, const CloudDevice& d, float max_dist, float cell, bool packed = false) {
  // nothing of a previous build may be taken for valid while this one overwrites its geometry and buffers (a failed build must not
  // leave the old cell / max_dist next to new arrays)
  cs.valid = false, cs.packed = false, cs.c.rec = nullptr, cs.c.qlist = nullptr, cs.c.head = nullptr;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = 0; i < h.n; ++i) {
    mn[0] = std::min(mn[0], h.x[i]), mn[1] = std::min(mn[1], h.y[i]), mn[2] = std::min(mn[2], h.z[i]);
    mx[0] = std::max(mx[0], h.x[i]), mx[1] = std::max(mx[1], h.y[i]), mx[2] = std::max(mx[2], h.z[i]);
  }
  const float pad = max_dist + 4 * GRID_MARGIN;
  CellListBuildArgs a{};
  a.x = d.plane(0), a.y = d.plane(1), a.z = d.plane(2), a.n = h.n;
  a.ox = mn[0] - pad, a.oy = mn[1] - pad, a.oz = mn[2] - pad, a.cell = cell;
  a.dx = (int)std::ceil((mx[0] - mn[0] + 2 * pad) / cell) + 1;
  a.dy = (int)std::ceil((mx[1] - mn[1] + 2 * pad) / cell) + 1;
  a.dz = (int)std::ceil((mx[2] - mn[2] + 2 * pad) / cell) + 1;
  a.max_dist = max_dist, a.margin = 4 * GRID_MARGIN;
  a.nx = d.plane(3), a.ny = d.plane(4), a.nz = d.plane(5);
  // domination margin: far above the float error of a squared distance <= max_dist^2 between points of magnitude <= 4 m
  // domination margin: far above the float error of a squared distance <= max_dist^2 between points whose coordinates
  // (in the frame where the reference measures it: the scene's) stay below c->coord_mag metres
  a.dom_eps = 64.f * max_dist * (c->coord_mag * 6.0e-8f) + 1.0e-12f;
  const size_t ncell = (size_t)a.dx * a.dy * a.dz;
  if (ncell > (size_t)1 << 24) return HOP_E_CAPACITY;
  HIPCHK(c, cs.u2_d.ensure(sizeof(float) * ncell));
  HIPCHK(c, cs.count_d.ensure(sizeof(int) * ncell));
  HIPCHK(c, cs.start_d.ensure(sizeof(int) * (ncell + 1)));
  a.u2 = cs.u2_d.as<float>(), a.count = cs.count_d.as<int>();
  launch_cell_list_bounds(a, c->stream);
  launch_cell_list_count(a, c->stream);
  std::vector<int> cnt(ncell), start(ncell + 1, 0);
  HIPCHK(c, hop_ctx_d2h(c, cnt.data(), cs.count_d.p, sizeof(int) * ncell));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (size_t k = 0; k < ncell; ++k) start[k + 1] = start[k] + cnt[k];
  const size_t total = (size_t)start[ncell];
  HIPCHK(c, cs.pts_d.ensure(sizeof(float4) * std::max<size_t>(total, 1)));
  HIPCHK(c, cs.nrm_d.ensure(sizeof(float4) * std::max<size_t>(total, 1)));
  a.nrm = cs.nrm_d.as<float4>();
  HIPCHK(c, hop_ctx_h2d(c, cs.start_d.p, start.data(), sizeof(int) * (ncell + 1)));
  a.start = cs.start_d.as<int>(), a.pts = cs.pts_d.as<float4>();
  launch_cell_list_fill(a, c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  cs.c.ox = a.ox, cs.c.oy = a.oy, cs.c.oz = a.oz, cs.c.cell = cell, cs.c.inv_cell = 1.0f / cell;
  cs.c.dx = a.dx, cs.c.dy = a.dy, cs.c.dz = a.dz;
  cs.c.start = cs.start_d.as<int>(), cs.c.pts = cs.pts_d.as<float4>(), cs.c.nrm = cs.nrm_d.as<float4>();
  HIPCHK(c, cs.range_d.ensure(sizeof(int2) * ncell));
  launch_cell_ranges(cs.c.start, (int)ncell, cs.range_d.as<int2>(), c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  cs.c.range = cs.range_d.as<int2>();
  cs.c.gox = -a.ox * cs.c.inv_cell, cs.c.goy = -a.oy * cs.c.inv_cell, cs.c.goz = -a.oz * cs.c.inv_cell;
  {  // sqrt(d2) <= d2 q_sa + q_sb for every d2 >= 0 (exact at d = max_dist / 2): the root-free tolerance of cells_nn1f; the packed lists set their own
    const double cc = (double)max_dist / 2.0;
    cs.c.q_sa = (float)(0.5 / cc * 1.000001), cs.c.q_sb = (float)(0.5 * cc * 1.000001);
  }
  HIPCHK(c, cs.nrm_idx_d.ensure(sizeof(float4) * (size_t)std::max(h.n, 1)));
  launch_soa_to_aos4(d.plane(3), d.plane(4), d.plane(5), h.n, cs.nrm_idx_d.as<float4>(), c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  cs.c.nrm_idx = cs.nrm_idx_d.as<float4>();
  HIPCHK(c, cs.pts_idx_d.ensure(sizeof(float4) * (size_t)std::max(h.n, 1)));
  launch_soa_to_aos4(d.plane(0), d.plane(1), d.plane(2), h.n, cs.pts_idx_d.as<float4>(), c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  cs.c.pts_idx = cs.pts_idx_d.as<float4>();
  cs.packed = false, cs.pack_requested = packed, cs.c.rec = nullptr, cs.c.qlist = nullptr;
  // (the empty-slot convention of cells_nnq -- coordinates 0xFFFF: per axis the 16-bit difference to a query wraps to >= 28 000 steps, a key above
  // that of any real member of the list -- needs the cell well below the gating distance: cell < 0.3 max_dist)
  if (packed && h.n < 0xFFFF && cell < 0.3f * max_dist) {
    std::vector<float4> hp(std::max<size_t>(total, 1));
    HIPCHK(c, hop_ctx_d2h(c, hp.data(), cs.pts_d.p, sizeof(float4) * total));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // per-cell quantisation frame: every list member lies within max_dist + margin of the cell's box on every axis
    // (cell_list_thr2), so [corner - R, corner + cell + R] holds it
    const double R = (double)max_dist + 2.0 * (double)a.margin;
    // step: the largest difference between a query (inside the cell) and a list member, cell + R, is 32 767 steps -- q_rank subtracts and
    // squares in 16 bits (v_pk_sub_i16 / v_dot2_i32_i16) -- which leaves the frame [0, cell + 2 R] below 65 535 steps as well
    const double step = ((double)cell + R) / 32767.0, qcs = (double)cell / step, qrs = R / step;
    // q_rank's invariant, stated where the lists are made: a query's local coordinate is a whole number in [qrs - 0.5, qrs + qcs + 0.5] with
    // qrs + qcs = 32 767, and every member coordinate w of a list must differ from every such query by at most 32 767 steps (a signed 16-bit
    // difference): 0 <= w <= qrs + 32 766.  A member beyond that would wrap to a SMALL difference and win the ranking silently, so the guard
    // below tests exactly this bound (not just the 16-bit field), and a list that breaks it sends the store to the plain lists.
    bool in_range = qrs + qcs <= 32767.0 + 1.0e-6;
    const double w_max = std::min(65534.0, std::floor(qrs) + 32766.0);
    // The exact point and normal of a winner are fetched by index; neighbouring queries win neighbouring points, so the two
    // arrays are stored in Morton order of the cloud and the entries carry the Morton rank (a wavefront's fetch then touches a
    // handful of 128-byte lines instead of up to 64: tools/ubench/l1_rate.hip).  .w of a point = its original index (ties).
    std::vector<uint32_t> rank(std::max(h.n, 1));
    {
      float lo3[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi3[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
      for (int i = 0; i < h.n; ++i) {
        const float v[3] = {h.x[i], h.y[i], h.z[i]};
        for (int ax = 0; ax < 3; ++ax) lo3[ax] = std::min(lo3[ax], v[ax]), hi3[ax] = std::max(hi3[ax], v[ax]);
      }
      const float ext = std::max({hi3[0] - lo3[0], hi3[1] - lo3[1], hi3[2] - lo3[2], 1.0e-9f});
      auto spread = [](uint32_t v) {  // 10 bits -> every third bit
        v = (v | (v << 16)) & 0x030000FFu, v = (v | (v << 8)) & 0x0300F00Fu, v = (v | (v << 4)) & 0x030C30C3u, v = (v | (v << 2)) & 0x09249249u;
        return v;
      };
      std::vector<std::pair<uint32_t, int>> code(h.n);
      for (int i = 0; i < h.n; ++i) {
        const float v[3] = {h.x[i], h.y[i], h.z[i]};
        uint32_t m = 0;
        for (int ax = 0; ax < 3; ++ax) m |= spread((uint32_t)std::min(1023.f, std::max(0.f, (v[ax] - lo3[ax]) / ext * 1023.f))) << ax;
        code[i] = {m, i};
      }
      std::sort(code.begin(), code.end());
      // (the values come from the device cloud the other modes read: the host copy of the generator keeps normalised normals)
      std::vector<float> dv(6 * (size_t)std::max(h.n, 1));
      HIPCHK(c, hop_ctx_d2h(c, dv.data(), d.plane(0), sizeof(float) * 6 * (size_t)h.n));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      const size_t hn = (size_t)h.n;
      std::vector<float4> pm(std::max(h.n, 1)), nm(std::max(h.n, 1));
      for (int r = 0; r < h.n; ++r) {
        const int i = code[r].second;
        rank[i] = (uint32_t)r;
        float wi;
        memcpy(&wi, &i, 4);
        pm[r] = make_float4(dv[i], dv[hn + i], dv[2 * hn + i], wi), nm[r] = make_float4(dv[3 * hn + i], dv[4 * hn + i], dv[5 * hn + i], 0.f);
      }
      HIPCHK(c, hop_ctx_h2d(c, cs.pts_idx_d.p, pm.data(), sizeof(float4) * (size_t)h.n));
      HIPCHK(c, hop_ctx_h2d(c, cs.nrm_idx_d.p, nm.data(), sizeof(float4) * (size_t)h.n));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    std::vector<uint32_t> rec(2 * ncell), ql;
    ql.reserve(2 * total + 4 * ncell);
    for (size_t k = 0; k < ncell; ++k) {
      const int ic[3] = {(int)(k % a.dx), (int)((k / a.dx) % a.dy), (int)(k / ((size_t)a.dx * a.dy))};
      const float org[3] = {a.ox, a.oy, a.oz};
      const int n = cnt[k], chunks = 2 * ((n + 3) / 4);  // an even number of chunks: the lookup reads two per trip
      rec[2 * k] = (uint32_t)(ql.size() / 4), rec[2 * k + 1] = (uint32_t)chunks;
      for (int e = 0; e < 2 * chunks; ++e) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0xFFFFFFFFu;
        if (e < n) {
          const float4 t = hp[(size_t)start[k] + e];
          const float v[3] = {t.x, t.y, t.z};
          uint32_t u[3], idx;
          for (int ax = 0; ax < 3; ++ax) {
            const double w = std::nearbyint((((double)v[ax] - (double)org[ax]) / (double)cell - (double)ic[ax]) * qcs + qrs);
            if (w < 0.0 || w > w_max) in_range = false;
            u[ax] = (uint32_t)std::min(65534.0, std::max(0.0, w));
          }
          memcpy(&idx, &t.w, 4);
          lo = u[0] | (u[1] << 16), hi = u[2] | (rank[idx] << 16);
        }
        ql.push_back(lo), ql.push_back(hi);
      }
    }
    if (!in_range) {
      // a list does not fit the 16-bit local frame: the plain lists built above are complete and exact on their own -- the callers
      // fall back to them (launch_icp_fused) when rec is null; pts_idx / nrm_idx are read by the packed kernels only
      cs.valid = true, cs.cell = cell, cs.max_dist = max_dist, cs.coord_mag = c->coord_mag, cs.total_entries = total;
      return HOP_OK;
    }
    if (ql.empty()) ql.assign(4, 0xFFFFFFFFu);
    HIPCHK(c, cs.rec_d.ensure(sizeof(uint32_t) * rec.size()));
    HIPCHK(c, cs.qlist_d.ensure(sizeof(uint32_t) * ql.size()));
    HIPCHK(c, hop_ctx_h2d(c, cs.rec_d.p, rec.data(), sizeof(uint32_t) * rec.size()));
    HIPCHK(c, hop_ctx_h2d(c, cs.qlist_d.p, ql.data(), sizeof(uint32_t) * ql.size()));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    cs.c.rec = cs.rec_d.as<uint2>(), cs.c.qlist = cs.qlist_d.as<uint4>();
    cs.c.q_cs = (float)qcs, cs.c.q_rs = (float)(qrs + 0.5), cs.c.q_step2 = (float)(step * step);  // + 0.5: the lookup truncates
    // sqrt(3)/2 of a step for the entry, as much for the query (rounded to whole steps too), the float evaluation of its local coordinate
    cs.c.q_eq = (float)(step * 1.8);
    {
      const double cc = (double)max_dist / 3.0;
      cs.c.q_sa = (float)(0.5 / cc * 1.000001), cs.c.q_sb = (float)(0.5 * cc * 1.000001);   // (rounded up: S stays an upper bound)
      cs.c.q_tk = 4.2f * cs.c.q_eq, cs.c.q_t0 = 2.1f * cs.c.q_eq * cs.c.q_eq + 2.f * cs.c.q_step2;
    }
    cs.packed = true;
    if (getenv("HOP_PROFILE_SELECT")) std::printf("packed lists: %.1f MB records + %.1f MB chunks, step %.3g m\n", rec.size() * 4e-6, ql.size() * 4e-6, step);
  }
  cs.valid = true, cs.cell = cell, cs.max_dist = max_dist, cs.coord_mag = c->coord_mag, cs.total_entries = total;
  if (getenv("HOP_PROFILE_SELECT")) std::printf("cell lists: %zu cells, %zu entries (%.1f per cell), cell %.4f\n", ncell, total, (double)total / (double)ncell, cell);
  return HOP_OK;
}

// NN cell lists of a per-frame cloud from its ring grid (cell >= max_dist + margin): two passes of one kernel around a
// device-side exclusive scan; nothing but the total entry count crosses PCIe.
int build_cell_lists_local(hop_ctx* c, CellListStore& cs, const GridStore& gs, const CloudDevice* normals, float max_dist, int sub,
                           int exist_mode) {
  const GridDev& g = gs.g;
  CellListBuildArgs a{};
  a.margin = 4 * GRID_MARGIN;
  cs.c.head = nullptr;  // (belongs to the lists this call replaces; cell_list_heads rebuilds it on demand)
  if (!(g.cell > 0) || sub < 1) return HOP_E_STATE;
  // the list grid is the ring grid padded by enough ring cells to reach max_dist beyond the cloud's box, subdivided
  const int pad = (int)std::ceil((max_dist + a.margin + 1.0e-6f) / g.cell);
  a.cell = g.cell / (float)sub;
  a.ox = g.ox - pad * g.cell, a.oy = g.oy - pad * g.cell, a.oz = g.oz - pad * g.cell;
  a.dx = (g.dx + 2 * pad) * sub, a.dy = (g.dy + 2 * pad) * sub, a.dz = (g.dz + 2 * pad) * sub;
  a.max_dist = max_dist;
  if (normals) a.n = normals->n, a.nx = normals->plane(3), a.ny = normals->plane(4), a.nz = normals->plane(5);
  a.dom_eps = 64.f * max_dist * (c->coord_mag * 6.0e-8f) + 1.0e-12f;
  a.pair_max = 160;  // (0 / 16 / 32 / 160 build equally fast: the pivot rounds before it remove 97 % of what it would)
  const size_t ncell = (size_t)a.dx * a.dy * a.dz;
  if (ncell > (size_t)1 << 28) return HOP_E_CAPACITY;
  HIPCHK(c, cs.count_d.ensure(sizeof(int) * (ncell + 1)));
  HIPCHK(c, cs.start_d.ensure(sizeof(int) * (ncell + 1)));
  HIPCHK(c, cs.u2_d.ensure(sizeof(int) * (ncell + 1)));  // flags of the cells that have candidates at all
  a.count = cs.count_d.as<int>();
  int* flag = cs.u2_d.as<int>();
  int* scan = cs.start_d.as<int>();
  c->timing.n_build_launches += 1;
  SpanGuard sg_build(c, T_BUILD);
  // stage 1: flag cells with candidates, compact them (in cell order) into a work list
  launch_cell_list_local_flag(a, g, flag, c->stream);
  size_t tmp_bytes = 0;
  HIPCHK(c, prim_exclusive_sum(nullptr, tmp_bytes, flag, scan, (int)(ncell + 1), c->stream));
  HIPCHK(c, c->sort_tmp.ensure(tmp_bytes + 16));
  HIPCHK(c, prim_exclusive_sum(c->sort_tmp.p, tmp_bytes, flag, scan, (int)(ncell + 1), c->stream));
  int nwork = 0;
  HIPCHK(c, hop_ctx_d2h(c, &nwork, scan + ncell, sizeof(int)));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, cs.work_d.ensure(sizeof(int) * (size_t)std::max(nwork, 1)));
  HIPCHK(c, cs.keep_d.ensure(sizeof(int) * (size_t)std::max(nwork, 1) * cell_list_local_keep()));
  int* work = cs.work_d.as<int>();
  int* keep = cs.keep_d.as<int>();
  launch_cell_list_local_work(flag, scan, (int)ncell, work, c->stream);
  // stage 2: one wave per listed cell, count then write around a scan of the counts
  // lanes per voxel from the list lengths this store had last time (first build: short lists assumed)
  static const int lanes_env = getenv("HOP_LOCAL_SUB") ? atoi(getenv("HOP_LOCAL_SUB")) : 0;
  // (counting pass of the long-list store: 32 lanes -- its 25 candidate rows fill them, 805 -> 527 us; its replay pass is fastest
  // with a whole wavefront per voxel)
  const int lanes = lanes_env ? lanes_env : (cs.avg_len > 5.0f ? 32 : 16);
  launch_cell_list_local(a, g, false, exist_mode, work, nwork, keep, lanes, c->stream);
  HIPCHK(c, prim_exclusive_sum(c->sort_tmp.p, tmp_bytes, a.count, cs.start_d.as<int>(), (int)(ncell + 1), c->stream));
  int total = 0;
  HIPCHK(c, hop_ctx_d2h(c, &total, cs.start_d.as<int>() + ncell, sizeof(int)));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, cs.pts_d.ensure(sizeof(float4) * (size_t)std::max(total, 1)));
  if (normals) HIPCHK(c, cs.nrm_d.ensure(sizeof(float4) * (size_t)std::max(total, 1)));
  a.start = cs.start_d.as<int>(), a.pts = cs.pts_d.as<float4>(), a.nrm = normals ? cs.nrm_d.as<float4>() : nullptr;
  cs.avg_len = nwork > 0 ? (float)total / (float)nwork : 0.f;
  launch_cell_list_local(a, g, true, exist_mode, work, nwork, keep, lanes_env ? lanes_env : (cs.avg_len > 5.0f ? 64 : 16), c->stream);
  cs.c.ox = a.ox, cs.c.oy = a.oy, cs.c.oz = a.oz, cs.c.cell = a.cell, cs.c.inv_cell = 1.0f / a.cell;
  cs.c.dx = a.dx, cs.c.dy = a.dy, cs.c.dz = a.dz;
  cs.c.start = cs.start_d.as<int>(), cs.c.pts = cs.pts_d.as<float4>(), cs.c.nrm = normals ? cs.nrm_d.as<float4>() : nullptr;
  HIPCHK(c, cs.range_d.ensure(sizeof(int2) * ncell));
  launch_cell_ranges(cs.c.start, (int)ncell, cs.range_d.as<int2>(), c->stream);
  cs.c.range = cs.range_d.as<int2>();
  cs.valid = true, cs.cell = a.cell, cs.max_dist = max_dist, cs.coord_mag = c->coord_mag, cs.total_entries = (size_t)std::max(total, 0);
  if (getenv("HOP_PROFILE_SELECT")) {
    std::vector<int> cnt(ncell);
    (void)hipMemcpy(cnt.data(), a.count, sizeof(int) * ncell, hipMemcpyDeviceToHost);
    int mx = 0, over64 = 0, over192 = 0;
    for (int v : cnt) mx = std::max(mx, v), over64 += v > 64, over192 += v > 192;
    std::printf("local cell lists: %zu cells (%d with candidates), %d entries, cell %.5f, longest %d, >64: %d, >192: %d\n", ncell, nwork, total,
                a.cell, mx, over64, over192);
  }
  return HOP_OK;
}

// Verify acceleration structures over P for `delta`: mode 1 the ring grid, mode 2 the EXIST-mode cell lists on top of it
int ensure_verify_structures(hop_ctx* c, int mode, float delta, const float* x, const float* y, const float* z, int n) {
  if (mode < 1) return HOP_OK;
  if (!c->verify_grid.valid || c->grid_delta != delta) {
    const int rc = build_grid(c, c->verify_grid, x, y, z, n, delta + 8 * GRID_MARGIN);
    if (rc) return rc;
    c->grid_delta = delta;
    c->verify_cells.valid = false;
  }
  if (mode >= 2 && (!c->verify_cells.valid || c->verify_cells.max_dist != delta)) {
    int sub = 3;
    if (const char* e = getenv("HOP_VERIFY_SUB")) sub = std::max(1, std::min(8, atoi(e)));
    const int rc = build_cell_lists_local(c, c->verify_cells, c->verify_grid, nullptr, delta, sub, 1);
    if (rc) return rc;
  }
  return HOP_OK;
}

// sort the resident set by 64-bit keys (ascending) and gather pose/score; ids become 0..n-1
int sort_resident_by_keys(hop_ctx* c, unsigned long long* keys_d, int n) {
  if (n <= 0) return HOP_OK;
  HIPCHK(c, c->sort_keys_alt.ensure(sizeof(unsigned long long) * (size_t)n));
  HIPCHK(c, c->sort_vals.ensure(sizeof(unsigned) * (size_t)n));
  HIPCHK(c, c->sort_vals_alt.ensure(sizeof(unsigned) * (size_t)n));
  launch_iota(c->sort_vals.as<unsigned>(), n, c->stream);
  size_t tmp_bytes = 0;
  HIPCHK(c, prim_sort_pairs(nullptr, tmp_bytes, keys_d, c->sort_keys_alt.as<unsigned long long>(),
                                               c->sort_vals.as<unsigned>(), c->sort_vals_alt.as<unsigned>(), n, 0, 64, c->stream));
  HIPCHK(c, c->sort_tmp.ensure(tmp_bytes + 16));
  HIPCHK(c, prim_sort_pairs(c->sort_tmp.p, tmp_bytes, keys_d, c->sort_keys_alt.as<unsigned long long>(),
                                               c->sort_vals.as<unsigned>(), c->sort_vals_alt.as<unsigned>(), n, 0, 64, c->stream));
  HIPCHK(c, c->tmp_pose.ensure(sizeof(float) * 16 * (size_t)n));
  HIPCHK(c, c->tmp_score.ensure(sizeof(float) * (size_t)n));
  HIPCHK(c, c->tmp_id.ensure(sizeof(int) * (size_t)n));
  launch_gather_hypos(c->sort_vals_alt.as<unsigned>(), n, c->hyp_pose.as<float>(), c->hyp_score.as<float>(), c->tmp_pose.as<float>(),
                      c->tmp_score.as<float>(), c->tmp_id.as<int>(), c->stream);
  std::swap(c->hyp_pose, c->tmp_pose);
  std::swap(c->hyp_score, c->tmp_score);
  std::swap(c->hyp_id, c->tmp_id);
  return HOP_OK;
}

int ensure_hyp_capacity(hop_ctx* c, int cap) {
  HIPCHK(c, c->hyp_pose.ensure(sizeof(float) * 16 * (size_t)cap));
  HIPCHK(c, c->hyp_score.ensure(sizeof(float) * (size_t)cap));
  HIPCHK(c, c->hyp_id.ensure(sizeof(int) * (size_t)cap));
  HIPCHK(c, c->hyp_key.ensure(sizeof(unsigned long long) * (size_t)cap));
  HIPCHK(c, c->hyp_inv.ensure(sizeof(unsigned) * (size_t)cap));
  return HOP_OK;
}

// clusterPoses core (PoseEstimator.cpp:106-233).  Euler angles of every pose are computed once instead of
// inside the double loop; the comparisons are the reference's.
void euler_zyx(const float* T, float res[3]) {  // Eigen 3.3 eulerAngles(2,1,0), Geometry/EulerAngles.h:36-108
  auto R = [&](int r, int cidx) { return T[4 * r + cidx]; };
  res[0] = std::atan2(R(1, 0), R(0, 0));
  const float c2 = std::sqrt(R(2, 2) * R(2, 2) + R(2, 1) * R(2, 1));
  if (res[0] < 0.f) {
    res[0] += (float)M_PI;
    res[1] = std::atan2(-R(2, 0), -c2);
  } else
    res[1] = std::atan2(-R(2, 0), c2);
  const float s1 = std::sin(res[0]), c1 = std::cos(res[0]);
  res[2] = std::atan2(s1 * R(0, 2) - c1 * R(1, 2), c1 * R(1, 1) - s1 * R(0, 1));
}

// rotationGeodesicDistance (Utils.cpp:29-32): std::acos(((R1 * R2).trace()-1) / 2.0) on the rotation blocks of two row-major 4x4 poses.
// "(R1 * R2).trace()-1" is a float subtraction, "/ 2.0" promotes (pinned against the reference's vendored Eigen: tests/golden/icp_lm_kat.npz)
float rotation_geodesic_distance(const float* cl, const float* cur) {
  float tr[3];
  for (int d = 0; d < 3; ++d) tr[d] = cl[4 * d + 0] * cur[0 + d] + (cl[4 * d + 1] * cur[4 + d] + cl[4 * d + 2] * cur[8 + d]);
  const float trace = tr[0] + (tr[1] + tr[2]);
  return (float)std::acos((double)(trace - 1.0f) / 2.0);
}

int cluster_core(const float* pose16, const float* lcp, const int* ids, int H, float angle_diff, float dist_diff,
                 const float* sym_deg3, std::vector<int>& keep) {
  keep.clear();
  if (H <= 0) return 0;
  std::vector<int> order(H);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) {  // HypoCompare
    if (lcp[a] > lcp[b]) return true;
    if (lcp[a] < lcp[b]) return false;
    return ids[a] < ids[b];
  });
  const float radian_thres = (float)((double)(angle_diff / 180.0f) * M_PI);
  const float sym[3] = {(float)((double)sym_deg3[0] / 180 * M_PI), (float)((double)sym_deg3[1] / 180 * M_PI),
                        (float)((double)sym_deg3[2] / 180 * M_PI)};
  std::vector<std::array<float, 3>> eul(H);
  for (int i = 0; i < H; ++i) euler_zyx(pose16 + 16 * (size_t)i, eul[i].data());
  keep.push_back(order[0]);
  for (int oi = 1; oi < H; ++oi) {
    const int ci = order[oi];
    const float* cur = pose16 + 16 * (size_t)ci;
    bool isnew = true;
    for (int k : keep) {
      const float* cl = pose16 + 16 * (size_t)k;
      const V3 t0 = v3(cl[3], cl[7], cl[11]), t1 = v3(cur[3], cur[7], cur[11]);
      if (vnorm(t0 - t1) >= dist_diff) continue;
      float roll = std::fabs(eul[k][2] - eul[ci][2]), pitch = std::fabs(eul[k][1] - eul[ci][1]), yaw = std::fabs(eul[k][0] - eul[ci][0]);
      if (sym[0] == 0) roll = 0;
      else if (sym[0] > 0) roll = std::min(roll, sym[0] - roll);
      if (sym[1] == 0) pitch = 0;
      else if (sym[1] > 0) pitch = std::min(pitch, sym[1] - pitch);
      if (sym[2] == 0) yaw = 0;
      else if (sym[2] > 0) yaw = std::min(yaw, sym[2] - yaw);
      if (pitch <= radian_thres && roll <= radian_thres && yaw <= radian_thres) {
        isnew = false;
        break;
      }
      const float rot_diff = rotation_geodesic_distance(cl, cur);
      if (rot_diff <= radian_thres) {
        isnew = false;
        break;
      }
    }
    if (isnew) keep.push_back(ci);
  }
  return (int)keep.size();
}

}  // namespace

// ==================================================================================================
// ---------------------------------------------------------------------------------------------- hop_ctx_ext.h
hipStream_t hop_ctx_stream(hop_ctx* c) { return c->stream; }
// Every transfer goes through the context's pinned staging areas by default (threshold 0: what rounds 3-4 measured).
// HOP_STAGE_MIN=<bytes> hands copies below the threshold to the runtime directly; a huge value turns staging off.
static const size_t STAGE_MIN = getenv("HOP_STAGE_MIN") ? (size_t)std::atoll(getenv("HOP_STAGE_MIN")) : (size_t)0;
hipError_t hop_ctx_h2d(hop_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes < STAGE_MIN) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream);
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (c->stage_cur + need > c->stage_up.cap) {  // the area is full: wait for the copies that read it, then start over (or grow)
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return e;
    c->stage_cur = 0;
    if (need > c->stage_up.cap && (e = c->stage_up.ensure(std::max(need, (size_t)16 << 20))) != hipSuccess) return e;
  }
  char* at = static_cast<char*>(c->stage_up.p) + c->stage_cur;
  static const bool prof = getenv("HOP_PROFILE_STAGE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  if (bytes) std::memcpy(at, src, bytes);
  const auto t1 = std::chrono::steady_clock::now();
  c->stage_cur += need;
  const hipError_t e = hipMemcpyAsync(dst, at, bytes, hipMemcpyHostToDevice, c->stream);
  if (prof) {
    const auto t2 = std::chrono::steady_clock::now();
    (void)hipStreamSynchronize(c->stream);
    const auto t3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::printf("stage h2d %zu bytes: memcpy %.3f ms, enqueue %.3f ms, wait %.3f ms\n", bytes, ms(t0, t1), ms(t1, t2), ms(t2, t3));
  }
  return e;
}
hipError_t hop_ctx_d2h(hop_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes < STAGE_MIN) {  // direct: same completion contract as the staged form (dst is valid on return)
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(c->stream);
  }
  hipError_t e = c->stage_down.ensure(bytes);
  if (e != hipSuccess) return e;
  if ((e = hipMemcpyAsync(c->stage_down.p, src, bytes, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return e;
  c->stage_cur = 0;  // everything queued before has completed: the upload area is free again
  std::memcpy(dst, c->stage_down.p, bytes);
  return hipSuccess;
}
int hop_ctx_device(const hop_ctx* c) { return c->device; }
void hop_ctx_set_error(hop_ctx* c, const std::string& msg) { c->last_error = msg; }
HopExt*& hop_ctx_ext(hop_ctx* c, int slot) { return c->ext[slot]; }
HopHypView hop_ctx_hyp(hop_ctx* c) { return HopHypView{c->hyp_pose.as<float>(), c->hyp_score.as<float>(), c->hyp_id.as<int>(), c->n_hyp}; }
void hop_ctx_hyp_set_count(hop_ctx* c, int n) { c->n_hyp = n; }


extern "C" {

int hop_abi_version(void) { return HOP_ABI_VERSION; }

const char* hop_strerror(int s) {
  switch (s) {
    case HOP_OK: return "ok";
    case HOP_E_INVALID: return "invalid argument";
    case HOP_E_NO_DEVICE: return "no usable HIP device (libhop has no CPU path)";
    case HOP_E_HIP: return "HIP runtime error";
    case HOP_E_CAPACITY: return "capacity exceeded";
    case HOP_E_STATE: return "call order / missing input";
    case HOP_E_NO_HYPOTHESIS: return "no hypothesis generated";
    case HOP_E_ALLOC: return "allocation failed";
    case HOP_E_COMM: return "RCCL unavailable or collective failed";
  }
  return "unknown status";
}

const char* hop_last_error(const hop_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int hop_ctx_create(int device, hop_ctx** out) {
  if (!out) return HOP_E_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return HOP_E_NO_DEVICE;
  if (device < 0 || device >= ndev) return HOP_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return HOP_E_NO_DEVICE;
  hop_ctx* c = new (std::nothrow) hop_ctx;
  if (!c) return HOP_E_ALLOC;
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return HOP_E_HIP;
  }
  *out = c;
  return HOP_OK;
}

void hop_ctx_destroy(hop_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto& s : c->spans) {
    (void)hipEventDestroy(s.a);
    (void)hipEventDestroy(s.b);
  }
  for (auto e : c->event_pool) (void)hipEventDestroy(e);
  if (c->ppf_matrix_registered) (void)hipHostUnregister(c->ppf_matrix_cached);
  std::free(c->ppf_matrix_cached);
  c->fit_queue_d.release(), c->fit_count_d.release(), c->angle_thr_d.release(), c->sur_in.release(), c->sur_ws.release(), c->sur_links.release(), c->sur_out.release();
  DevBuf* bufs[] = {&c->scene_sorted_aos_d, &c->scene_d.buf, &c->scene_sorted_d.buf, &c->scene_unit_d.buf, &c->scene_sorted_unit_d.buf, &c->scene_perm_d, &c->model_d[0].buf, &c->model_d[1].buf, &c->key_bitmap_d, &c->gp_d.buf, &c->gq_d.buf, &c->gq_unit_d,
                    &c->ppf_matrix_d, &c->vp_d.buf, &c->vq_d.buf, &c->bases_d, &c->pairs1_d,
                    &c->pairs2_d, &c->cnt_d, &c->elems_d, &c->queries_d, &c->cands_d, &c->cand_counts_d, &c->counters_d, &c->hyp_pose,
                    &c->hyp_score, &c->hyp_id, &c->hyp_key, &c->hyp_inv, &c->tmp_pose, &c->tmp_score, &c->tmp_id, &c->sort_keys_alt,
                    &c->sort_vals, &c->sort_vals_alt, &c->sort_tmp, &c->lcp_rev_idx, &c->lcp_rev_d2, &c->lcp_terms, &c->icp_moved,
                    &c->icp_partial, &c->icp_state, &c->icp_iters, &c->icp_conv, &c->icp_corr_idx, &c->icp_hist, &c->icp_lm, &c->pose_inv, &c->topk_rows, &c->hand_scene_d.buf, &c->hand_lookup_d.buf,
                    &c->hand_swivel_d.buf, &c->hand_model_d.buf, &c->finger_hist_d, &c->pso_particles_d, &c->pso_match_d,
                    &c->pso_terms_d, &c->pso_sum_d, &c->pso_cnt_d};
  for (DevBuf* b : bufs) b->release();
  c->verify_grid.release(), c->model_grid[0].release(), c->model_grid[1].release(), c->scene_grid.release(), c->hand_grid.release();
  c->model_cells[0].release(), c->model_cells[1].release(), c->scene_cells.release(), c->verify_cells.release(), c->hand_cells.release();
  c->ppf_matrix_h.release(), c->bases_h.release(), c->cnt_h.release(), c->pso_particles_h.release(), c->pso_out_h.release(), c->stage_up.release(), c->stage_down.release();
  for (HopExt*& e : c->ext) {
    delete e;
    e = nullptr;
  }
  (void)hipStreamDestroy(c->stream);
  delete c;
}

int hop_synchronize(hop_ctx* c) {
  if (!c) return HOP_E_INVALID;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

// ---------------------------------------------------------------------------------------------- clouds
int hop_set_scene(hop_ctx* c, const float* xyz, const float* nrm, const float* conf, int n, float thres) {
  if (!c || !xyz || !nrm || n < 0) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  CloudHost all;
  load_cloud_host(all, xyz, nrm, n, true);
  CloudHost& s = c->gen.scene_h;
  s.resize(0);
  c->gen.scene_conf.clear();
  std::vector<int> keep;
  for (int i = 0; i < n; ++i) {
    const float cf = conf ? conf[i] : 1.0f;
    if (conf && thres > 0 && cf < thres) continue;  // PoseEstimator.cpp:41-45
    keep.push_back(i);
  }
  s.resize((int)keep.size());
  c->gen.scene_conf.resize(keep.size());
  for (size_t k = 0; k < keep.size(); ++k) {
    const int i = keep[k];
    s.x[k] = all.x[i], s.y[k] = all.y[i], s.z[k] = all.z[i], s.nx[k] = all.nx[i], s.ny[k] = all.ny[i], s.nz[k] = all.nz[i];
    c->gen.scene_conf[k] = conf ? conf[i] : 1.0f;
  }
  // The scoring stages read the raw normals (computeLCP normalises on use; ICP uses them as given):
  // keep the un-normalised normals on the device copy used by ICP/LCP.
  CloudHost raw = s;
  for (size_t k = 0; k < keep.size(); ++k) {
    const int i = keep[k];
    raw.nx[k] = nrm[i], raw.ny[k] = nrm[n + i], raw.nz[k] = nrm[2 * (size_t)n + i];
  }
  {
    float m = 0.f;
    for (size_t k = 0; k < keep.size(); ++k) m = std::max(m, std::max(std::fabs(raw.x[k]), std::max(std::fabs(raw.y[k]), std::fabs(raw.z[k]))));
    while (c->coord_mag < m) c->coord_mag *= 2.f;  // model-side lists built for a smaller scale are rebuilt on next use
  }
  c->have_gen_state = false;
  c->verify_grid.valid = false;
  c->scene_grid.valid = false;
  c->scene_cells.valid = false;
  {
    // Morton order of 2 mm voxels for the grid-based scoring kernels
    const int m = raw.n;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    for (int i = 0; i < m; ++i) mn[0] = std::min(mn[0], raw.x[i]), mn[1] = std::min(mn[1], raw.y[i]), mn[2] = std::min(mn[2], raw.z[i]);
    auto part = [](unsigned long long v) {
      v &= 0x1fffffull;
      v = (v | v << 32) & 0x1f00000000ffffull;
      v = (v | v << 16) & 0x1f0000ff0000ffull;
      v = (v | v << 8) & 0x100f00f00f00f00full;
      v = (v | v << 4) & 0x10c30c30c30c30c3ull;
      v = (v | v << 2) & 0x1249249249249249ull;
      return v;
    };
    std::vector<std::pair<unsigned long long, int>> order(m);
    for (int i = 0; i < m; ++i) {
      const unsigned long long qx = (unsigned long long)std::max(0.f, (raw.x[i] - mn[0]) * 500.f), qy = (unsigned long long)std::max(0.f, (raw.y[i] - mn[1]) * 500.f),
                               qz = (unsigned long long)std::max(0.f, (raw.z[i] - mn[2]) * 500.f);
      order[i] = {part(qx) | part(qy) << 1 | part(qz) << 2, i};
    }
    std::sort(order.begin(), order.end());
    CloudHost sorted;
    sorted.resize(m);
    std::vector<int> perm(std::max(m, 1));
    for (int k = 0; k < m; ++k) {
      const int i = order[k].second;
      perm[k] = i;
      sorted.x[k] = raw.x[i], sorted.y[k] = raw.y[i], sorted.z[k] = raw.z[i];
      sorted.nx[k] = raw.nx[i], sorted.ny[k] = raw.ny[i], sorted.nz[k] = raw.nz[i];
    }
    const int rc = upload_cloud(c, c->scene_sorted_d, sorted);
    if (rc) return rc;
    {
      std::vector<float> aos((size_t)8 * std::max(m, 1), 0.f);
      for (int k = 0; k < m; ++k) {
        float* pp = aos.data() + (size_t)4 * k;
        float* pn = aos.data() + (size_t)4 * (m + k);
        pp[0] = sorted.x[k], pp[1] = sorted.y[k], pp[2] = sorted.z[k];
        pn[0] = sorted.nx[k], pn[1] = sorted.ny[k], pn[2] = sorted.nz[k];
      }
      HIPCHK(c, c->scene_sorted_aos_d.ensure(sizeof(float) * aos.size()));
      HIPCHK(c, hop_ctx_h2d(c, c->scene_sorted_aos_d.p, aos.data(), sizeof(float) * aos.size()));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    HIPCHK(c, c->scene_perm_d.ensure(sizeof(int) * 2 * (size_t)std::max(m, 1)));
    std::vector<int> inv(std::max(m, 1));
    for (int k = 0; k < m; ++k) inv[perm[k]] = k;
    HIPCHK(c, hop_ctx_h2d(c, c->scene_perm_d.p, perm.data(), sizeof(int) * (size_t)std::max(m, 1)));
    HIPCHK(c, hop_ctx_h2d(c, c->scene_perm_d.as<int>() + std::max(m, 1), inv.data(), sizeof(int) * (size_t)std::max(m, 1)));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return upload_cloud(c, c->scene_d, raw);
}

int hop_scene_size(const hop_ctx* c) { return c ? c->gen.scene_h.n : HOP_E_INVALID; }

int hop_set_model(hop_ctx* c, int level, const float* xyz, const float* nrm, int n) {
  if (!c || !xyz || !nrm || n <= 0 || (level != HOP_MODEL_5MM && level != HOP_MODEL_1MM)) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  load_cloud_host(c->gen.model_h[level], xyz, nrm, n, true);  // generator view: Point3D normals
  CloudHost raw;
  load_cloud_host(raw, xyz, nrm, n, false);  // scoring view: as given
  c->have_gen_state = false;
  c->model_grid[level].valid = false;
  c->model_cells[level].valid = false;
  return upload_cloud(c, c->model_d[level], raw);
}

int hop_set_ppf_keys(hop_ctx* c, const int32_t* keys4, int nkeys) {
  if (!c || !keys4 || nkeys < 0) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  build_key_bitmap(keys4, nkeys, c->key_bitmap, c->key_dist_bins);
  HIPCHK(c, c->key_bitmap_d.ensure(sizeof(unsigned) * c->key_bitmap.size()));
  HIPCHK(c, hop_ctx_h2d(c, c->key_bitmap_d.p, c->key_bitmap.data(), sizeof(unsigned) * c->key_bitmap.size()));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_keys = true;
  c->have_gen_state = false;
  return HOP_OK;
}

// ---------------------------------------------------------------------------------------------- generator
void hop_s4pcs_default_opts(hop_s4pcs_opts* o) {
  if (!o) return;
  o->sample_size = 100, o->overlap = 0.2f, o->delta = 0.003f, o->dispersion = 0.5f;
  o->success_quadrilaterals = 10, o->max_time_seconds = 1, o->n_trials = 0, o->random_seed = 5489u;
  o->max_normal_difference = -1.f, o->max_color_distance = -1.f, o->verify_mode = 2;
}

int hop_s4pcs_generate(hop_ctx* c, const hop_s4pcs_opts* opts, float* poses16_out, float* lcp_out, int cap, int* n_out,
                       hop_s4pcs_stats* stats_out) {
  if (!c || !opts) return HOP_E_INVALID;
  if (n_out) *n_out = 0;
  if (opts->max_normal_difference >= 0 || opts->max_color_distance >= 0) return HOP_E_INVALID;
  if (opts->sample_size <= 0 || opts->sample_size > 4096 || !(opts->delta > 0)) return HOP_E_INVALID;
  if (c->gen.scene_h.n <= 0 || c->gen.model_h[HOP_MODEL_5MM].n <= 0 || !c->have_keys) return HOP_E_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  const auto t_begin = std::chrono::steady_clock::now();
  hop_s4pcs_stats st{};
  c->trace.clear();
  c->n_hyp = 0;

  GenHost G(&c->gen, *opts);
  G.init_clouds();
  const int N = c->gen.gp_h.n, NQ = c->gen.gq_h.n;
  if (N > 65536) return HOP_E_CAPACITY;  // bit matrix N^2/8 bytes
  st.n_sampled_q = NQ;
  for (int k = 0; k < 3; ++k) st.centroid_p[k] = c->gen.centroid_p[k], st.centroid_q[k] = c->gen.centroid_q[k];
  st.diameter = c->gen.diameter;

  // ---- upload generator clouds: P planes x y z + PPF normals (two extra normalisations, matchBase.hpp:53-56)
  {
    CloudHost up = c->gen.gp_h;
    for (int i = 0; i < N; ++i) {
      V3 nn = vnormalized(vnormalized(v3(up.nx[i], up.ny[i], up.nz[i])));
      up.nx[i] = nn.x, up.ny[i] = nn.y, up.nz[i] = nn.z;
    }
    int rc = upload_cloud(c, c->gp_d, up);
    if (rc) return rc;
    rc = upload_cloud(c, c->gq_d, c->gen.gq_h);
    if (rc) return rc;
    HIPCHK(c, c->gq_unit_d.ensure(sizeof(float) * 3 * (size_t)NQ));
    for (int k = 0; k < 3; ++k)
      HIPCHK(c, hop_ctx_h2d(c, c->gq_unit_d.as<float>() + (size_t)k * NQ, c->gen.gq_unit[k].data(), sizeof(float) * NQ));
  }
  c->have_gen_state = true;
  c->have_verify_clouds = false;
  c->verify_grid.valid = false;

  // ---- K2: PPF membership matrix
  const int W = (N + 63) / 64;
  c->ppf_words = W;
  const size_t mbytes = sizeof(unsigned long long) * (size_t)N * W;
  HIPCHK(c, c->ppf_matrix_d.ensure(mbytes));
  // The selection reads matrix rows millions of times, at random: the host copy lives in ordinary (CPU-cached) memory,
  // 2 MiB aligned with transparent huge pages requested, registered with the HIP runtime so that the device-to-host DMA
  // lands in it directly (hipHostMalloc memory is mapped uncached on the CPU side and reads from it are several times
  // slower).  If registration is refused, a pinned staging buffer plus one memcpy is used instead.
  if (c->ppf_matrix_cached_bytes < mbytes) {
    if (c->ppf_matrix_registered) (void)hipHostUnregister(c->ppf_matrix_cached);
    c->ppf_matrix_registered = false;
    std::free(c->ppf_matrix_cached);
    const size_t cap = (mbytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    c->ppf_matrix_cached = static_cast<unsigned long long*>(std::aligned_alloc((size_t)2 << 20, cap));
    if (!c->ppf_matrix_cached) {
      c->ppf_matrix_cached_bytes = 0;
      return HOP_E_ALLOC;
    }
    madvise(c->ppf_matrix_cached, cap, MADV_HUGEPAGE);
    std::memset(c->ppf_matrix_cached, 0, cap);  // touch: the pages exist (as huge pages where granted) before pinning
    c->ppf_matrix_cached_bytes = cap;
    if (!getenv("HOP_NO_HOST_REGISTER") && hipHostRegister(c->ppf_matrix_cached, cap, hipHostRegisterDefault) == hipSuccess)
      c->ppf_matrix_registered = true;
    else
      (void)hipGetLastError();
  }
  if (!c->ppf_matrix_registered) HIPCHK(c, c->ppf_matrix_h.ensure(mbytes));
  {
    PpfMatrixArgs a{};
    a.x = c->gp_d.plane(0), a.y = c->gp_d.plane(1), a.z = c->gp_d.plane(2);
    a.nx = c->gp_d.plane(3), a.ny = c->gp_d.plane(4), a.nz = c->gp_d.plane(5);
    a.n = N, a.words = W, a.bitmap = c->key_bitmap_d.as<unsigned>(), a.dist_bins = c->key_dist_bins;
    a.out = c->ppf_matrix_d.as<unsigned long long>();
    if (!c->angle_thr_tried) {  // once per context: cosine thresholds of the angle bins, verified against acosf
      c->angle_thr_tried = true;
      float thr[32];
      if (!getenv("HOP_PPF_LITERAL") && build_angle_thresholds(thr)) {
        HIPCHK(c, c->angle_thr_d.ensure(sizeof(thr)));
        HIPCHK(c, hop_ctx_h2d(c, c->angle_thr_d.p, thr, sizeof(thr)));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->angle_thr_ok = true;
      }
    }
    a.angle_thr = c->angle_thr_ok ? c->angle_thr_d.as<float>() : nullptr;
    {
      SpanGuard sg(c, T_PPF);
      launch_ppf_matrix(a, c->stream);
    }
    void* dst = c->ppf_matrix_registered ? (void*)c->ppf_matrix_cached : c->ppf_matrix_h.p;
    HIPCHK(c, hipMemcpyAsync(dst, c->ppf_matrix_d.p, mbytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!c->ppf_matrix_registered) std::memcpy(c->ppf_matrix_cached, c->ppf_matrix_h.p, mbytes);
  }
  G.M = c->ppf_matrix_cached;
  G.W = W;

  {
    const int rc = ensure_verify_structures(c, opts->verify_mode, opts->delta, c->gen.gp_h.x.data(), c->gen.gp_h.y.data(), c->gen.gp_h.z.data(), N);
    if (rc) return rc;
  }

  // ---- batch workspaces
  const int n_trials = opts->n_trials > 0 ? opts->n_trials : 30;
  if (n_trials > 16384) return HOP_E_CAPACITY;  // 14 bits of the order key
  const bool never_stops_early = opts->success_quadrilaterals >= n_trials && opts->max_time_seconds <= 0;
  int BATCH = never_stops_early ? 128 : std::min(32, n_trials);
  if (const char* e = getenv("HOP_GEN_BATCH")) BATCH = std::max(1, std::min(atoi(e), n_trials));
  const int pair_cap = NQ * (NQ - 1) + 2;
  const int cand_cap = 1 << 22;
  const int hyp_cap = 1 << 23;
  HIPCHK(c, c->bases_d.ensure(sizeof(BaseDev) * (size_t)BATCH));
  HIPCHK(c, c->bases_h.ensure(sizeof(BaseDev) * (size_t)BATCH * 2));
  HIPCHK(c, c->pairs1_d.ensure(sizeof(unsigned) * (size_t)BATCH * pair_cap));
  HIPCHK(c, c->pairs2_d.ensure(sizeof(unsigned) * (size_t)BATCH * pair_cap));
  HIPCHK(c, c->elems_d.ensure(sizeof(QuadElem) * (size_t)BATCH * pair_cap));
  HIPCHK(c, c->queries_d.ensure(sizeof(QuadQuery) * (size_t)BATCH * pair_cap));
  HIPCHK(c, c->cnt_d.ensure(sizeof(int) * 3 * (size_t)n_trials + 64));
  HIPCHK(c, c->cnt_h.ensure(sizeof(int) * 3 * (size_t)n_trials + 64));
  HIPCHK(c, c->cands_d.ensure(sizeof(Candidate) * (size_t)cand_cap));
  HIPCHK(c, c->cand_counts_d.ensure(sizeof(int) * (size_t)cand_cap));
  HIPCHK(c, c->counters_d.ensure(sizeof(int) * 16));
  HIPCHK(c, c->fit_queue_d.ensure(sizeof(int4) * (size_t)cand_cap * 4));  // FIT_QUEUES sub-queues of cand_cap/FIT_QUEUES*4 entries
  // per flushed batch: its candidate count and the FIT_QUEUES sub-queue counts (a flush carries at least one base: <= n_trials of them) --
  // zeroed ONCE per frame below instead of by two fill kernels before every batch (~32 of the ~100 fills of a C2 frame)
  const size_t batch_cnt_stride = 1 + FIT_QUEUES;
  HIPCHK(c, c->fit_count_d.ensure(sizeof(int) * batch_cnt_stride * ((size_t)n_trials + 1)));
  int rc = ensure_hyp_capacity(c, hyp_cap);
  if (rc) return rc;
  int* counters = c->counters_d.as<int>();  // [1] hyp_count, [2] overflow, [3] candidates in total ([0]: unused since the batches count their own)
  HIPCHK(c, hipMemsetAsync(counters, 0, sizeof(int) * 16, c->stream));
  HIPCHK(c, hipMemsetAsync(c->cnt_d.p, 0, sizeof(int) * 3 * (size_t)n_trials, c->stream));
  HIPCHK(c, hipMemsetAsync(c->fit_count_d.p, 0, sizeof(int) * batch_cnt_stride * ((size_t)n_trials + 1), c->stream));
  int n_flushed = 0;
  int* cnt1_all = c->cnt_d.as<int>();
  int* cnt2_all = cnt1_all + n_trials;
  const NsetGeom geom = make_nset_geom(opts->delta / c->gen.ratio);
  const float* qx = c->gq_d.plane(0);
  const float* qy = c->gq_d.plane(1);
  const float* qz = c->gq_d.plane(2);

  std::vector<BaseDev> batch;
  batch.reserve(BATCH);
  int batch_first_trace = 0;
  int flip = 0;
  double ms_select = 0;
  hipEvent_t stage_ev[2] = {get_event(c), get_event(c)};
  bool stage_used[2] = {false, false};
  int* nquads_all = cnt2_all + n_trials;

  auto flush = [&]() -> int {
    const int nb = (int)batch.size();
    if (!nb) return HOP_OK;
    BaseDev* stage = static_cast<BaseDev*>(c->bases_h.p) + (size_t)flip * BATCH;
    // this staging half was last read by the H2D copy issued two flushes ago
    if (stage_used[flip]) HIPCHK(c, hipEventSynchronize(stage_ev[flip]));
    std::memcpy(stage, batch.data(), sizeof(BaseDev) * nb);
    HIPCHK(c, hipMemcpyAsync(c->bases_d.p, stage, sizeof(BaseDev) * nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(stage_ev[flip], c->stream));
    stage_used[flip] = true;
    flip ^= 1;
    if (n_flushed > n_trials) return HOP_E_STATE;
    int* cand_count = c->fit_count_d.as<int>() + batch_cnt_stride * (size_t)n_flushed;  // this batch's own, still zero from the frame's one fill
    int* fit_count = cand_count + 1;
    ++n_flushed;
    int* cnt1 = cnt1_all + batch_first_trace;
    int* cnt2 = cnt2_all + batch_first_trace;
    {
      SpanGuard sg(c, T_GEN_OTHER);
      PairArgs pa{};
      pa.bases = c->bases_d.as<BaseDev>();
      pa.qx = qx, pa.qy = qy, pa.qz = qz, pa.qnx = c->gq_d.plane(3), pa.qny = c->gq_d.plane(4), pa.qnz = c->gq_d.plane(5);
      pa.nq = NQ, pa.eps = 1.0f * opts->delta;
      pa.pairs1 = c->pairs1_d.as<unsigned>(), pa.pairs2 = c->pairs2_d.as<unsigned>();
      pa.cnt1 = cnt1, pa.cnt2 = cnt2, pa.cap = pair_cap, pa.overflow = counters + 2;
      launch_pairs(pa, nb, c->stream);
      QuadPrepArgs qp{};
      qp.bases = pa.bases, qp.qx = qx, qp.qy = qy, qp.qz = qz;
      qp.ux = c->gq_unit_d.as<float>(), qp.uy = qp.ux + NQ, qp.uz = qp.ux + 2 * (size_t)NQ;
      qp.pairs1 = pa.pairs1, qp.pairs2 = pa.pairs2, qp.cnt1 = cnt1, qp.cnt2 = cnt2, qp.cap = pair_cap, qp.geom = geom;
      qp.elems = c->elems_d.as<QuadElem>(), qp.queries = c->queries_d.as<QuadQuery>();
      launch_quad_prep(qp, nb, 2 * pair_cap, c->stream);
      QuadArgs qa{};
      qa.bases = pa.bases, qa.qx = qx, qa.qy = qy, qa.qz = qz, qa.pairs1 = pa.pairs1, qa.pairs2 = pa.pairs2;
      qa.cnt1 = cnt1, qa.cnt2 = cnt2, qa.cap = pair_cap, qa.geom = geom, qa.elems = qp.elems, qa.queries = qp.queries;
      qa.dist_thr2 = 1.0f * opts->delta, qa.delta = 1.0f * opts->delta, qa.base_index0 = batch_first_trace;
      qa.cands = c->cands_d.as<Candidate>(), qa.cand_counts = c->cand_counts_d.as<int>(), qa.cand_count = cand_count, qa.cand_cap = cand_cap;
      qa.nquads = nquads_all + batch_first_trace;
      qa.overflow = counters + 2;
      qa.fit_queue = c->fit_queue_d.as<int4>(), qa.fit_count = fit_count, qa.fit_cap = cand_cap / FIT_QUEUES * 4;
      {
        SpanGuard sq(c, T_QUADS);
        launch_quads(qa, nb, 64, c->stream);
      }
      c->timing.n_quads_launches += 1;
    }
    VerifyArgs va{};
    va.px = c->gp_d.plane(0), va.py = c->gp_d.plane(1), va.pz = c->gp_d.plane(2), va.np = N;
    va.qx = qx, va.qy = qy, va.qz = qz, va.nq = NQ;
    va.T = reinterpret_cast<const float*>(c->cands_d.p), va.t_stride = sizeof(Candidate) / sizeof(float);
    va.n_cand_ptr = cand_count, va.n_cand = 0, va.cand_cap = cand_cap;
    va.sq_eps = opts->delta * opts->delta, va.counts = c->cand_counts_d.as<int>();
    {
      SpanGuard sg(c, T_VERIFY);
      if (opts->verify_mode >= 2) launch_verify_cells(va, c->verify_cells.c, 2048, c->stream);
      else launch_verify(va, opts->verify_mode, c->verify_grid.valid ? &c->verify_grid.g : nullptr, 2048, c->stream);
    }
    c->timing.n_verify_launches += 1;
    {
      SpanGuard sg(c, T_GEN_OTHER);
      EmitArgs ea{};
      ea.cands = c->cands_d.as<Candidate>(), ea.cand_counts = c->cand_counts_d.as<int>(), ea.cand_count = cand_count, ea.cand_cap = cand_cap;
      for (int k = 0; k < 3; ++k) ea.cp[k] = c->gen.centroid_p[k], ea.cq[k] = c->gen.centroid_q[k];
      ea.nq = NQ, ea.pose = c->hyp_pose.as<float>(), ea.score = c->hyp_score.as<float>(), ea.key = c->hyp_key.as<unsigned long long>();
      ea.inv_count = c->hyp_inv.as<unsigned>(), ea.hyp_count = counters + 1, ea.hyp_cap = hyp_cap, ea.cand_total = counters + 3;
      ea.overflow = counters + 2;
      launch_emit(ea, 256, c->stream);
    }
    batch_first_trace += nb;
    batch.clear();
    return HOP_OK;
  };

  // ---- trial loop (Perform_N_steps, cse.hpp:133-194)
  int success = 0;
  int trials_run = 0;
  int checked_trace = 0;  // trace entries whose success has been read back
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n_trials; ++i) {
    trials_run = i + 1;
    float inv1 = 0, inv2 = 0;
    int ids[4];
    const auto ts = std::chrono::steady_clock::now();
    const bool ok = G.SelectQuadrilateral(inv1, inv2, ids);
    ms_select += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count();
    if (ok) {
      BaseDev B;
      G.fill_base(B, inv1, inv2);
      batch.push_back(B);
      BaseTraceHost t;
      for (int k = 0; k < 4; ++k) t.ids[k] = ids[k];
      t.inv1 = inv1, t.inv2 = inv2;
      c->trace.push_back(t);
    }
    const bool batch_full = (int)batch.size() >= BATCH;
    if (batch_full || (!never_stops_early && ok)) {
      // A base "succeeds" (TryOneBase, cse.hpp:201-216) when both pair lists are non-empty and at least one
      // congruent quadrilateral exists; that is known only on the device.  Without early stop nothing has
      // to be read back here; otherwise flush and read the counters of the bases submitted so far.
      rc = flush();
      if (rc) return rc;
      if (!never_stops_early) {
        HIPCHK(c, hipMemcpyAsync(c->cnt_h.p, c->cnt_d.p, sizeof(int) * 3 * (size_t)n_trials, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const int* h1 = static_cast<const int*>(c->cnt_h.p);
        const int* h2 = h1 + n_trials;
        const int* hq = h2 + n_trials;
        for (; checked_trace < (int)c->trace.size(); ++checked_trace)
          if (h1[checked_trace] > 0 && h2[checked_trace] > 0 && hq[checked_trace] > 0) ++success;
      }
    }
    const float fraction_try = float(i) / float(n_trials);
    float fraction_time = 0.f;
    if (opts->max_time_seconds > 0) {
      const long long secs = std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t0).count();
      fraction_time = float(secs / opts->max_time_seconds);
    }
    const float fraction = std::max(fraction_time, fraction_try);
    if (i > n_trials || fraction >= 0.99f || success >= opts->success_quadrilaterals) break;
  }
  rc = flush();
  if (rc) return rc;
  st.n_trials_run = trials_run;

  // ---- collect
  int hc[4] = {0, 0, 0, 0};
  HIPCHK(c, hipMemcpyAsync(c->cnt_h.p, c->cnt_d.p, sizeof(int) * 3 * (size_t)n_trials, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(hc, counters, sizeof(int) * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (hc[2]) return HOP_E_CAPACITY;
  {
    const int* h1 = static_cast<const int*>(c->cnt_h.p);
    const int* h2 = h1 + n_trials;
    const int* hq = h2 + n_trials;
    for (size_t k = 0; k < c->trace.size(); ++k) {
      c->trace[k].n1 = h1[k], c->trace[k].n2 = h2[k], c->trace[k].nq = hq[k];
      st.n_pairs += (long long)h1[k] + h2[k];
      st.n_quads += hq[k];
    }
  }
  st.n_candidates = hc[3];
  c->timing.pairs_verify += (long long)hc[3] * NQ * (long long)N;
  c->event_pool.push_back(stage_ev[0]);
  c->event_pool.push_back(stage_ev[1]);
  const int H = std::min(hc[1], hyp_cap);
  st.n_bases = (int)c->trace.size();
  st.n_hypotheses = H;
  c->n_hyp = H;
  // canonical emission order
  rc = sort_resident_by_keys(c, c->hyp_key.as<unsigned long long>(), H);
  if (rc) return rc;
  if (H > 0 && (poses16_out || lcp_out)) {
    const int m = std::min(H, cap);
    if (poses16_out) HIPCHK(c, hop_ctx_d2h(c, poses16_out, c->hyp_pose.p, sizeof(float) * 16 * (size_t)m));
    if (lcp_out) HIPCHK(c, hop_ctx_d2h(c, lcp_out, c->hyp_score.p, sizeof(float) * (size_t)m));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  st.ms_select = ms_select;
  if (getenv("HOP_PROFILE_SELECT")) G.print_profile();
  st.ms_device = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  if (n_out) *n_out = H;
  if (stats_out) *stats_out = st;
  if (H == 0) return HOP_E_NO_HYPOTHESIS;
  if ((poses16_out || lcp_out) && cap < H) return HOP_E_CAPACITY;
  return HOP_OK;
}

int hop_s4pcs_num_bases(const hop_ctx* c) { return c ? (int)c->trace.size() : HOP_E_INVALID; }
int hop_s4pcs_get_base(const hop_ctx* c, int i, int* base4, float* inv2, int* counts3) {
  if (!c || i < 0 || i >= (int)c->trace.size()) return HOP_E_INVALID;
  const BaseTraceHost& t = c->trace[i];
  if (base4)
    for (int k = 0; k < 4; ++k) base4[k] = t.ids[k];
  if (inv2) inv2[0] = t.inv1, inv2[1] = t.inv2;
  if (counts3) counts3[0] = t.n1, counts3[1] = t.n2, counts3[2] = t.nq;
  return HOP_OK;
}
int hop_s4pcs_get_sampled_q(const hop_ctx* c, float* xyz, float* nrm) {
  if (!c || !c->have_gen_state) return HOP_E_STATE;
  const CloudHost& q = c->gen.gq_h;
  for (int i = 0; i < q.n; ++i) {
    if (xyz) xyz[i] = q.x[i], xyz[q.n + i] = q.y[i], xyz[2 * (size_t)q.n + i] = q.z[i];
    if (nrm) nrm[i] = q.nx[i], nrm[q.n + i] = q.ny[i], nrm[2 * (size_t)q.n + i] = q.nz[i];
  }
  return HOP_OK;
}

int hop_verify_set_clouds(hop_ctx* c, const float* p_xyz, int n_p, const float* q_xyz, int n_q) {
  if (!c || !p_xyz || !q_xyz || n_p <= 0 || n_q <= 0) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  CloudHost P, Q;
  std::vector<float> zeros((size_t)3 * std::max(n_p, n_q), 0.f);
  load_cloud_host(P, p_xyz, zeros.data(), n_p, false);
  load_cloud_host(Q, q_xyz, zeros.data(), n_q, false);
  int rc = upload_cloud(c, c->vp_d, P);
  if (rc) return rc;
  rc = upload_cloud(c, c->vq_d, Q);
  if (rc) return rc;
  c->vp_h[0] = P.x, c->vp_h[1] = P.y, c->vp_h[2] = P.z;
  c->have_verify_clouds = true;
  c->verify_grid.valid = false;
  return HOP_OK;
}

int hop_verify_batch(hop_ctx* c, const float* T16, int H, float delta, int mode, int* count_out) {
  if (!c || !T16 || !count_out || H < 0) return HOP_E_INVALID;
  if (!c->have_verify_clouds && !c->have_gen_state) return HOP_E_STATE;
  if (H == 0) return HOP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const CloudDevice& P = c->have_verify_clouds ? c->vp_d : c->gp_d;
  const CloudDevice& Q = c->have_verify_clouds ? c->vq_d : c->gq_d;
  {
    const int rc = c->have_verify_clouds ? ensure_verify_structures(c, mode, delta, c->vp_h[0].data(), c->vp_h[1].data(), c->vp_h[2].data(), P.n)
                                         : ensure_verify_structures(c, mode, delta, c->gen.gp_h.x.data(), c->gen.gp_h.y.data(), c->gen.gp_h.z.data(), P.n);
    if (rc) return rc;
  }
  HIPCHK(c, c->tmp_pose.ensure(sizeof(float) * 16 * (size_t)H));
  HIPCHK(c, c->cand_counts_d.ensure(sizeof(int) * (size_t)H));
  HIPCHK(c, hop_ctx_h2d(c, c->tmp_pose.p, T16, sizeof(float) * 16 * (size_t)H));
  HIPCHK(c, hipMemsetAsync(c->cand_counts_d.p, 0, sizeof(int) * (size_t)H, c->stream));
  VerifyArgs va{};
  va.px = P.plane(0), va.py = P.plane(1), va.pz = P.plane(2), va.np = P.n;
  va.qx = Q.plane(0), va.qy = Q.plane(1), va.qz = Q.plane(2), va.nq = Q.n;
  va.T = c->tmp_pose.as<float>(), va.t_stride = 16, va.n_cand_ptr = nullptr, va.n_cand = H, va.cand_cap = H;
  va.sq_eps = delta * delta, va.counts = c->cand_counts_d.as<int>();
  const long long total = (long long)H * Q.n;
  const int blocks = (int)std::min<long long>(4096, (total + 1023) / 1024);
  {
    SpanGuard sg(c, T_VERIFY);
    if (mode >= 2) launch_verify_cells(va, c->verify_cells.c, std::max(blocks, 1), c->stream);
    else launch_verify(va, mode, (mode == 1 && c->verify_grid.valid) ? &c->verify_grid.g : nullptr, std::max(blocks, 1), c->stream);
  }
  c->timing.n_verify_launches += 1;
  c->timing.pairs_verify += total * (long long)P.n;
  HIPCHK(c, hop_ctx_d2h(c, count_out, c->cand_counts_d.p, sizeof(int) * (size_t)H));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

// ---------------------------------------------------------------------------------------------- resident set
int hop_hypos_upload(hop_ctx* c, const float* poses16, const float* scores, int H) {
  if (!c || !poses16 || H < 0) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure_hyp_capacity(c, std::max(H, 1));
  if (rc) return rc;
  HIPCHK(c, hop_ctx_h2d(c, c->hyp_pose.p, poses16, sizeof(float) * 16 * (size_t)H));
  if (scores) HIPCHK(c, hop_ctx_h2d(c, c->hyp_score.p, scores, sizeof(float) * (size_t)H));
  else HIPCHK(c, hipMemsetAsync(c->hyp_score.p, 0, sizeof(float) * (size_t)H, c->stream));
  std::vector<int> ids(H);
  std::iota(ids.begin(), ids.end(), 0);
  HIPCHK(c, hop_ctx_h2d(c, c->hyp_id.p, ids.data(), sizeof(int) * (size_t)H));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->n_hyp = H;
  return HOP_OK;
}
int hop_hypos_count(const hop_ctx* c) { return c ? c->n_hyp : HOP_E_INVALID; }
int hop_hypos_download(hop_ctx* c, float* poses16_out, float* scores_out, int* ids_out, int cap, int* n_out) {
  if (!c) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  const int m = std::min(c->n_hyp, cap);
  if (n_out) *n_out = c->n_hyp;
  if (m > 0) {
    if (poses16_out) HIPCHK(c, hop_ctx_d2h(c, poses16_out, c->hyp_pose.p, sizeof(float) * 16 * (size_t)m));
    if (scores_out) HIPCHK(c, hop_ctx_d2h(c, scores_out, c->hyp_score.p, sizeof(float) * (size_t)m));
    if (ids_out) HIPCHK(c, hop_ctx_d2h(c, ids_out, c->hyp_id.p, sizeof(int) * (size_t)m));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return cap < c->n_hyp ? HOP_E_CAPACITY : HOP_OK;
}
int hop_hypos_keep_topk(hop_ctx* c, int k) {
  if (!c || k < 0) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  const int H = c->n_hyp;
  if (H == 0) return HOP_OK;
  HIPCHK(c, c->hyp_key.ensure(sizeof(unsigned long long) * (size_t)H));
  launch_score_keys(c->hyp_score.as<float>(), c->hyp_id.as<int>(), H, c->hyp_key.as<unsigned long long>(), c->stream);
  const int rc = sort_resident_by_keys(c, c->hyp_key.as<unsigned long long>(), H);
  if (rc) return rc;
  c->n_hyp = std::min(H, k);
  return HOP_OK;
}

// ---------------------------------------------------------------------------------------------- ICP
// k_icp_fusedq_momm rests on the operand layout of v_mfma_i32_16x16x64_i8 (rows / columns = lane & 15, the four 16-lane groups = the four
// K blocks, C / D: column lane & 15, rows 4 (lane >> 4) + register) and on v_perm_b32's selector convention.  Both are documented, but this
// code was written without a device to run it on: the first nn_mode-7 refinement on a device sends known vectors through the kernel's own
// read-out path (k_dev_selftest_momm: momm_push -> ring -> momm_flush -> tiles) and compares the recombined matrix with sum U U^T computed
// here.  A device that answers differently gets the vector-unit kernel (k_icp_fusedq_momi: the same integers from v_dot2_i32_i16) and one
// line on stderr -- never a wrong moment matrix.
int hop_debug_selftest(hop_ctx* c, int what, int n, const void* in, void* out);
// The packed lookups (cells_nnq, every ICP mode >= 3) rank candidates with q_rank: v_pk_sub_i16, an inline-asm v_mad_i32_i16 and v_dot2_i32_i16
// -- written, like the matrix-core read-out below, without a device to run them on.  A wrong ranking would return a wrong neighbour that the
// exact re-scan (bounded by the same keys) cannot catch, so the first packed lookup on a device is preceded by this check of q_rank and
// v_med3_u32 against their scalar statements on operands that cover the field boundaries.  A device that disagrees gets NO packed lists
// (the callers then take the plain-list kernels: nn_mode 3 / 4 -> k_icp_fused, 6 -> the per-evaluation form, 7 refuses and the mirrors retry
// with 5), one line on stderr, and the verdict in hop_debug_selfcheck -- never a silently wrong correspondence.
static std::mutex g_selfcheck_mu;
static std::map<int, int> g_selfcheck;  // device -> bits of hop_debug_selfcheck
static unsigned qrank_scalar(unsigned lxy, unsigned lz, unsigned lo, unsigned hi) {
  const int dx = (int)(int16_t)(uint16_t)((lxy & 0xFFFFu) - (lo & 0xFFFFu)), dy = (int)(int16_t)(uint16_t)((lxy >> 16) - (lo >> 16));
  const int dz = (int)(int16_t)(uint16_t)((lz & 0xFFFFu) - (hi & 0xFFFFu));
  return (unsigned)(dx * dx) + (unsigned)(dy * dy) + (unsigned)(dz * dz);
}
static bool qrank_ok(hop_ctx* c) {
  std::lock_guard<std::mutex> lk(g_selfcheck_mu);
  int& bits = g_selfcheck[c->device];
  if (bits & 1) return (bits & 2) != 0;
  constexpr int N = 1024;
  std::vector<int> in(5 * N);
  std::vector<unsigned> out(9 * N, 0u);
  float* xf = reinterpret_cast<float*>(in.data());
  unsigned lcg = 2463534242u;
  auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return lcg >> 8; };
  for (int i = 0; i < N; ++i) {
    xf[i] = 0.25f, xf[N + i] = 1.0f;  // (operands of the moment primitives: not judged here)
    // queries 28 000 .. 32 767 steps into the frame, members anywhere in [0, 65 534], empty slots 0xFFFF: every sign of every difference
    unsigned q[3], w[3];
    for (int ax = 0; ax < 3; ++ax) {
      q[ax] = 28000u + rnd() % 4768u;
      const unsigned lo_w = q[ax] > 32767u ? q[ax] - 32767u : 0u, hi_w = std::min(65534u, q[ax] + 32767u);
      w[ax] = (i & 7) == 7 ? 0xFFFFu : lo_w + rnd() % (hi_w - lo_w + 1u);
      if (i < 6 && (i >> 1) == ax) w[ax] = (i & 1) ? hi_w : lo_w;  // the extreme differences +-32 767 on each axis
    }
    in[2 * N + i] = (int)(q[0] | (q[1] << 16));                         // ia: l.xy (bit 31 decides what garbage the kernel puts above l.z)
    in[3 * N + i] = (int)(w[0] | (w[1] << 16));                         // ib: lo
    in[4 * N + i] = (int)(w[2] | (q[2] << 16));                         // ic: hi = member z | (what k_dev_selftest_scalar takes for l.z) << 16
  }
  const bool ran = hop_debug_selftest(c, 0, N, in.data(), out.data()) == HOP_OK;
  bool ok = ran;
  for (int i = 0; i < N && ok; ++i) {
    const unsigned ua = (unsigned)in[2 * N + i], ub = (unsigned)in[3 * N + i], uc = (unsigned)in[4 * N + i];
    ok = out[5 * (size_t)N + i] == qrank_scalar(ua, uc >> 16, ub, uc);
    const unsigned lo3 = std::min(ua, std::min(ub, uc)), hi3 = std::max(ua, std::max(ub, uc));
    ok = ok && out[4 * (size_t)N + i] == (unsigned)((unsigned long long)ua + ub + uc - lo3 - hi3);  // v_med3_u32
  }
  if (!ok)
    std::fprintf(stderr, "libhop: the 16-bit packed ranking of the ICP lookups (q_rank: v_pk_sub_i16 / v_mad_i32_i16 / v_dot2_i32_i16, v_med3_u32) %s on device %d; "
                 "the packed cell lists are NOT used (plain-list kernels instead; nn_mode 7 refuses, the mirrors retry with nn_mode 5)\n",
                 ran ? "does not reproduce its scalar statement" : "could not be checked", c->device);
  bits |= 1 | (ok ? 2 : 0);
  return ok;
}
static bool mfma_i8_layout_ok(hop_ctx* c) {
  std::lock_guard<std::mutex> lk(g_selfcheck_mu);
  int& bits = g_selfcheck[c->device];
  if (bits & 4) return (bits & 8) != 0;
  // three batches of 64 vectors, the second accepted on 38 lanes only: a full half, a half completed across two pushes, a partial last half
  constexpr int NB = 3;
  std::vector<int> in(NB * 64 * 13 + NB * 2), out(3 * 256, 0);
  const unsigned long long masks[NB] = {~0ull, 0x0000F0F3FFFF1F7Full, ~0ull};
  unsigned lcg = 12345u;
  long long M[13][13] = {};
  for (int b = 0; b < NB; ++b)
    for (int l = 0; l < 64; ++l) {
      int* u = &in[((size_t)b * 64 + l) * 13];
      for (int k = 0; k < 13; ++k) {
        lcg = lcg * 1664525u + 1013904223u;
        u[k] = (int)((lcg >> 8) % 8193u) - 4096;
      }
      if (b == 0 && l < 4) u[l] = l & 1 ? 4096 : -4096;  // the ends of the range
      if ((masks[b] >> l) & 1ull)
        for (int i = 0; i < 13; ++i)
          for (int j = 0; j < 13; ++j) M[i][j] += (long long)u[i] * u[j];
    }
  std::memcpy(&in[NB * 64 * 13], masks, sizeof(masks));
  const bool ran = hop_debug_selftest(c, 2, NB, in.data(), out.data()) == HOP_OK;
  bool ok = ran;
  auto tile = [&](int t, int i, int j) { return (long long)out[(t * 64 + j + 16 * (i / 4)) * 4 + i % 4]; };  // entry (row i, column j)
  for (int i = 0; i < 13 && ok; ++i)
    for (int j = 0; j < 13 && ok; ++j) ok = tile(0, i, j) * 65536 + (tile(1, i, j) + tile(1, j, i)) * 256 + tile(2, i, j) == M[i][j];
  if (!ok)
    std::fprintf(stderr, "libhop: the matrix-core read-out of k_icp_fusedq_momm %s on device %d; nn_mode 7 adds its moment sums on the vector units "
                 "(k_icp_fusedq_momi: the same integers)\n", ran ? "does not reproduce sum U U^T (v_mfma_i32_16x16x64_i8 operand layout / v_perm_b32 selectors)" : "could not be checked", c->device);
  bits |= 4 | (ok ? 8 : 0);
  return ok;
}

int hop_icp_refine(hop_ctx* c, const hop_icp_opts* o, int* iterations_out, int* converged_out) {
  if (!c || !o || o->max_iter <= 0) return HOP_E_INVALID;
  if (c->scene_d.n <= 0 || c->model_d[HOP_MODEL_5MM].n <= 0) return HOP_E_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  int H = c->n_hyp;
  if (o->max_hypotheses > 0 && H > o->max_hypotheses) {  // PoseEstimator.cpp:241: the rest is dropped
    H = o->max_hypotheses;
    c->n_hyp = H;
  }
  if (H == 0) return HOP_OK;
  const CloudDevice& S = c->scene_d;
  const CloudDevice& Mo = c->model_d[HOP_MODEL_5MM];
  const int nb = icp_blocks_per_hyp(S.n, o->nn_mode >= 2);
  // batch so that the per-point workspace (moved source, 24 B/pt; cell-list path: correspondence, 4 B/pt) stays bounded
  const bool cells = o->nn_mode >= 2;
  bool lm_mode = o->nn_mode == 5;   // the reference's Levenberg-Marquardt minimiser (csrc/hop_icp_lm.hip), float-faithful, one pass per evaluation
  bool lm6_mode = o->nn_mode == 6;  // the same minimiser from the moment matrix of the correspondences: one pass per ICP iteration
  bool icp_mfma = true;
  bool lm7_mode = o->nn_mode == 7;  // the moment form with integer-exact sums and IEEE operations only: the bits the oracle's minimiser 7 returns
  if (o->nn_mode < 0 || o->nn_mode > 7) return HOP_E_INVALID;
  const size_t per_h = cells ? sizeof(int) * (size_t)S.n : sizeof(float) * 6 * (size_t)S.n;
  size_t ws_cap = cells ? ((size_t)4 << 30) : ((size_t)1 << 30);
  if (const char* e = getenv("HOP_ICP_WS_CAP_MB")) ws_cap = (size_t)std::max(1, atoi(e)) << 20;  // (tests: several hypothesis batches at small sizes)
  const int HB = (int)std::max<size_t>(1, std::min<size_t>((size_t)H, ws_cap / per_h));
  if (!cells) HIPCHK(c, c->icp_moved.ensure(per_h * HB));
  HIPCHK(c, c->icp_partial.ensure(sizeof(double) * (size_t)std::max(std::max(ICP_NACC, ICP_NMOM_STRIDE), ICP_NMOMI_STRIDE) * (size_t)nb * HB));
  HIPCHK(c, c->icp_state.ensure(sizeof(IcpState) * (size_t)HB));
  HIPCHK(c, c->icp_iters.ensure(sizeof(int) * (size_t)H));
  HIPCHK(c, c->icp_conv.ensure(sizeof(int) * (size_t)H));
  IcpArgs a{};
  a.sx = S.plane(0), a.sy = S.plane(1), a.sz = S.plane(2), a.snx = S.plane(3), a.sny = S.plane(4), a.snz = S.plane(5), a.ns = S.n;
  a.mx = Mo.plane(0), a.my = Mo.plane(1), a.mz = Mo.plane(2), a.mnx = Mo.plane(3), a.mny = Mo.plane(4), a.mnz = Mo.plane(5), a.nm = Mo.n;
  a.pose = c->hyp_pose.as<float>();
  a.max_iter = o->max_iter;
  a.max_d2 = o->max_corr_dist * o->max_corr_dist;
  a.cos_thr = (float)std::cos((double)(o->angle_deg / 180.0f) * M_PI);
  a.moved = c->icp_moved.as<float>(), a.partial = c->icp_partial.as<double>(), a.state = c->icp_state.as<IcpState>();
  if (o->nn_mode == 1) {
    // model grid in its rest frame; cells of a third of the gating distance, rings grow until they cover it
    const float cell = o->max_corr_dist / 3.f + GRID_MARGIN;
    GridStore& gs = c->model_grid[HOP_MODEL_5MM];
    if (!gs.valid || gs.cell != cell) {
      const CloudHost& mh = c->gen.model_h[HOP_MODEL_5MM];
      const int rc = build_grid(c, gs, mh.x.data(), mh.y.data(), mh.z.data(), mh.n, cell);
      if (rc) return rc;
    }
    a.model_grid = gs.g;
    a.max_ring = (int)std::ceil((o->max_corr_dist + 2 * GRID_MARGIN) / cell);
  } else if (cells) {
    // cells of a sixth of the gating distance for the plain lists (nn_mode 2), a seventh for the packed ones (measured:
    // 6 / 7 / 8 / 10 / 12 -> 819 / 784 / 789 / 845 / 925 us per launch at C2)
    const bool packed_mode = o->nn_mode == 3 || o->nn_mode == 4 || lm6_mode || lm7_mode;
    float cell = o->max_corr_dist / (packed_mode ? 7.f : 6.f);
    if (const char* e = getenv("HOP_ICP_CELL_DIV")) cell = o->max_corr_dist / (float)atof(e);
    CellListStore& cs = c->model_cells[HOP_MODEL_5MM];
    const bool want_packed = packed_mode && c->gen.model_h[HOP_MODEL_5MM].n < 0xFFFF && qrank_ok(c);  // (a device that fails the ranking check gets the plain lists)
    if (!cs.valid || cs.cell != cell || cs.max_dist != o->max_corr_dist || cs.coord_mag < c->coord_mag || (want_packed && !cs.pack_requested)) {
      const int rc = build_cell_lists(c, cs, c->gen.model_h[HOP_MODEL_5MM], c->model_d[HOP_MODEL_5MM], o->max_corr_dist, cell, want_packed);
      if (rc) return rc;
    }
    a.cells = cs.c;
    // nn_mode 6 / 7 need the packed lists; a model of >= 65535 points or a list outside the 16-bit frame has none.  nn_mode 6 then runs the
    // same minimiser in its per-evaluation form on the plain lists (nn_mode 5: the float results of that form); nn_mode 7 promises the
    // oracle's bits and has no other form that returns them: it refuses instead of answering with another arithmetic.
    if (lm7_mode && !cs.c.rec) {
      hop_ctx_set_error(c, qrank_ok(c) ? "hop_icp_refine: nn_mode 7 needs the packed cell lists (HOP_MODEL_5MM of < 65535 points, lists inside the 16-bit cell frame); use nn_mode 5"
                                          : "hop_icp_refine: nn_mode 7 needs the packed cell lists, and this device failed the check of their 16-bit ranking (q_rank; see stderr, hop_debug_selfcheck); use nn_mode 5");
      return HOP_E_STATE;
    }
    if (lm6_mode && !cs.c.rec) lm6_mode = false, lm_mode = true;
    if (o->nn_mode == 2 || lm_mode) HIPCHK(c, c->icp_corr_idx.ensure(sizeof(int) * (size_t)S.n * HB));  // the fused kernels keep no correspondence array
    if (lm_mode) {
      HIPCHK(c, c->icp_lm.ensure(sizeof(LmDev) * (size_t)HB + 64));
      a.lm = c->icp_lm.as<LmDev>();
    }
    // nn_mode 7: the moment sums on the matrix cores (k_icp_fusedq_momm) unless HOP_ICP_MFMA=0 (k_icp_fusedq_momi, v_dot2 on the vector units): same integers
    // (read at every call: a test or a tool may switch between two refinements of one process)
    icp_mfma = !(getenv("HOP_ICP_MFMA") != nullptr && atoi(getenv("HOP_ICP_MFMA")) == 0);
    if (lm7_mode && icp_mfma) icp_mfma = mfma_i8_layout_ok(c);
    if (lm7_mode) c->icp_last_engine = icp_mfma ? 1 : 0;
    if (lm7_mode) {
      // the grid of the moment form (oracle: mom_spec): powers of two from the model's radius about its origin and the gate
      const CloudHost& mh = c->gen.model_h[HOP_MODEL_5MM];
      double r2 = 0;
      for (int i = 0; i < mh.n; ++i) r2 = std::max(r2, (double)mh.x[i] * (double)mh.x[i] + (double)mh.y[i] * (double)mh.y[i] + (double)mh.z[i] * (double)mh.z[i]);
      const double radius = std::sqrt(r2), gate = (double)o->max_corr_dist, lim = std::ldexp(1.0, ICP_MOM_BITS), lim_d = 16777216.0;
      a.mom_k_n = ICP_MOM_BITS;
      a.mom_k_np = std::ilogb(lim / ((radius + gate) * 1.01));
      a.mom_k_r = std::ilogb(lim / gate);
      a.mom_k_d = std::ilogb(lim_d / (gate * gate));
      a.mom_s_n = std::ldexp(1.0f, a.mom_k_n), a.mom_s_np = std::ldexp(1.0f, a.mom_k_np), a.mom_s_r = std::ldexp(1.0f, a.mom_k_r), a.mom_s_d = std::ldexp(1.0f, a.mom_k_d);
      a.mom_lim = (float)lim, a.mom_lim_d = (float)lim_d;
    }
    if (lm_mode || lm6_mode || lm7_mode) {
      // PCL's gates (correspondence_estimation.hpp: double max_dist_sqr = max_distance * max_distance, skip if distance > it;
      // correspondence_rejection_surface_normal: double(dot) > std::cos(angle / 180.0 * M_PI), Utils.cpp:205) against float values:
      // d <= m and d > c for a float d and double m, c are d <= (largest float <= m) and d > (largest float <= c)
      const double m_d = (double)o->max_corr_dist * (double)o->max_corr_dist;
      float m_f = (float)m_d;
      if ((double)m_f > m_d) m_f = std::nextafterf(m_f, -INFINITY);
      const double c_d = std::cos((double)o->angle_deg / 180.0 * M_PI);
      float c_f = (float)c_d;
      if ((double)c_f > c_d) c_f = std::nextafterf(c_f, -INFINITY);
      a.max_d2 = m_f, a.cos_thr = c_f;
    }
    HIPCHK(c, c->icp_hist.ensure(sizeof(float) * 12 * (size_t)std::max(o->max_iter, 1) * HB));
    a.corr_idx = c->icp_corr_idx.as<int>(), a.hist = c->icp_hist.as<float>();
    HIPCHK(c, c->pose_inv.ensure(sizeof(float) * 12 * (size_t)H));
    launch_pose_inverse(a.pose, H, c->pose_inv.as<float>(), c->stream);
    a.pose_inv = c->pose_inv.as<float>();
    // these paths walk the scene in Morton order (the per-hypothesis sums are order-insensitive up to f64 rounding)
    const CloudDevice& Q = c->scene_sorted_d;
    a.sx = Q.plane(0), a.sy = Q.plane(1), a.sz = Q.plane(2), a.snx = Q.plane(3), a.sny = Q.plane(4), a.snz = Q.plane(5);
    a.s_pts4 = c->scene_sorted_aos_d.as<float4>(), a.s_nrm4 = a.s_pts4 + std::max(S.n, 1);
  }
  for (int h0 = 0; h0 < H; h0 += HB) {
    const int hb = std::min(HB, H - h0);
    a.h0 = h0;
    launch_icp_init(a.state, hb, c->stream);
    if (lm_mode) {
      // per ICP iteration: correspondences, then one pass per function evaluation Eigen's minimiser asks for, until no hypothesis waits
      unsigned* n_wait_d = reinterpret_cast<unsigned*>(c->icp_lm.as<char>() + sizeof(LmDev) * (size_t)HB);
      const bool lm_profile = getenv("HOP_PROFILE_LM") != nullptr;  // prints the hypotheses still waiting after every pass
      for (int it = 0; it < o->max_iter; ++it) {
        a.iter = it;
        {
          SpanGuard sg(c, T_ICP_NN);
          launch_icp_corr_cells(a, hb, c->stream);
        }
        c->timing.n_icp_nn_launches += 1;
        SpanGuard sg(c, T_ICP_SOLVE);
        launch_icp_lm_begin(a, hb, c->stream);
        for (int pass = 0;; ++pass) {
          launch_icp_lm_pass(a, hb, pass == 0, c->stream);
          HIPCHK(c, hipMemsetAsync(n_wait_d, 0, sizeof(unsigned), c->stream));
          launch_icp_lm_solve(a, hb, nb, pass == 0, n_wait_d, c->stream);
          unsigned n_wait = 0;
          HIPCHK(c, hop_ctx_d2h(c, &n_wait, n_wait_d, sizeof(unsigned)));
          HIPCHK(c, hipStreamSynchronize(c->stream));
          if (lm_profile) std::printf("%u%s", n_wait, n_wait ? " " : "\n");
          if (n_wait == 0) break;
          if (pass > 450) return HOP_E_STATE;  // maxfev = 400 evaluations bound every minimiser
        }
      }
      launch_icp_finish(a, hb, c->icp_iters.as<int>(), c->icp_conv.as<int>(), c->stream);
      continue;
    }
    if (lm6_mode || lm7_mode) {
      for (int it = 0; it < o->max_iter; ++it) {
        a.iter = it;
        {
          SpanGuard sg(c, T_ICP_NN);
          if (lm7_mode && icp_mfma) launch_icp_fusedq_momm(a, hb, c->stream);
          else if (lm7_mode) launch_icp_fusedq_momi(a, hb, c->stream);
          else launch_icp_fusedq_mom(a, hb, c->stream);
        }
        {
          SpanGuard sg(c, T_ICP_SOLVE);
          if (lm7_mode) launch_icp_lm7_solve(a, hb, nb, c->stream);
          else launch_icp_lm6_solve(a, hb, nb, c->stream);
        }
        c->timing.n_icp_nn_launches += 1;
      }
      launch_icp_finish(a, hb, c->icp_iters.as<int>(), c->icp_conv.as<int>(), c->stream);
      continue;
    }
    for (int it = 0; it < o->max_iter; ++it) {
      a.iter = it;
      {
        SpanGuard sg(c, T_ICP_NN);
        if (o->nn_mode >= 3) {
          if (!a.cells.rec) launch_icp_fused(a, hb, c->stream);  // (models of >= 65535 points: unpacked lists)
          else launch_icp_fusedq(a, hb, o->nn_mode == 4, c->stream);
        } else if (cells) {
          launch_icp_corr_cells(a, hb, c->stream);
        } else if (o->nn_mode == 1) launch_icp_nn_grid(a, hb, c->stream);
        else launch_icp_nn(a, hb, c->stream);
      }
      if (o->nn_mode == 2) {
        SpanGuard sg(c, T_ICP_ACCUM);
        launch_icp_accum(a, hb, c->stream);
      }
      {
        SpanGuard sg(c, T_ICP_SOLVE);
        launch_icp_solve(a, hb, nb, c->stream);
      }
      c->timing.n_icp_nn_launches += 1;
    }
    launch_icp_finish(a, hb, c->icp_iters.as<int>(), c->icp_conv.as<int>(), c->stream);
  }
  if (iterations_out) HIPCHK(c, hop_ctx_d2h(c, iterations_out, c->icp_iters.p, sizeof(int) * (size_t)H));
  if (converged_out) HIPCHK(c, hop_ctx_d2h(c, converged_out, c->icp_conv.p, sizeof(int) * (size_t)H));
  if (iterations_out || converged_out) HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

// the inline-head records of a list store (CellListDev::head), built on demand for computeLCP's reduced-sum kernel: one small kernel over the
// cells.  Stores with more than 2^23 cells or 2^24 entries keep head = null (the kernel then reads range records as before).
static int cell_list_heads(hop_ctx* c, CellListStore& cs) {
  if (cs.c.head || !cs.valid) return HOP_OK;
  const size_t ncell = (size_t)cs.c.dx * cs.c.dy * cs.c.dz;
  if (ncell == 0 || ncell > ((size_t)1 << 23)) return HOP_OK;
  if (cs.total_entries >= ((size_t)1 << 24)) return HOP_OK;
  HIPCHK(c, cs.head_d.ensure(sizeof(uint4) * ncell));
  launch_cell_heads(cs.c.range, cs.c.pts, (int)ncell, cs.head_d.as<uint4>(), c->stream);
  cs.c.head = cs.head_d.as<uint4>();
  return HOP_OK;
}

// ---------------------------------------------------------------------------------------------- LCP
int hop_lcp_select_best(hop_ctx* c, const hop_lcp_opts* o, float* best_pose16_out, float* best_score_out, int* best_index_out) {
  if (!c || !o || !(o->dist > 0)) return HOP_E_INVALID;
  if (c->scene_d.n <= 0 || c->model_d[HOP_MODEL_1MM].n <= 0) return HOP_E_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  const int H = c->n_hyp;
  if (H == 0) return HOP_E_NO_HYPOTHESIS;
  const CloudDevice& S = c->scene_d;
  const CloudDevice& Mo = c->model_d[HOP_MODEL_1MM];
  const size_t per_h = sizeof(float) * 2 * (size_t)S.n + sizeof(float) * 2 * (size_t)Mo.n;
  // nn_mode < 0: pick by size.  All modes return the same bits; the cell lists need per-frame scene structures (~1-3 ms
  // to build), which a small hypothesis set (the as-shipped <= 100) does not amortise: brute force is ~0.1 ms there.
  int nn_mode = o->nn_mode;
  if (nn_mode < 0) nn_mode = ((double)H * (double)S.n * (double)Mo.n < 2.0e10 && !c->scene_cells.valid) ? 0 : 2;
  const bool lcp_grid = nn_mode >= 1, lcp_cells = nn_mode >= 2;
  const size_t ws_cap = lcp_grid ? ((size_t)4 << 30) : ((size_t)1 << 30);
  const int HB = (int)std::max<size_t>(1, std::min<size_t>((size_t)H, ws_cap / per_h));
  if (!lcp_grid) {
    HIPCHK(c, c->lcp_rev_idx.ensure(sizeof(int) * (size_t)Mo.n * HB));
    HIPCHK(c, c->lcp_rev_d2.ensure(sizeof(float) * (size_t)Mo.n * HB));
  }
  HIPCHK(c, c->lcp_terms.ensure(sizeof(float) * 2 * (size_t)S.n * (lcp_cells ? lcp_cells_row_stride(HB) : HB) + 64));
  LcpArgs a{};
  a.sx = S.plane(0), a.sy = S.plane(1), a.sz = S.plane(2), a.snx = S.plane(3), a.sny = S.plane(4), a.snz = S.plane(5), a.ns = S.n;
  a.mx = Mo.plane(0), a.my = Mo.plane(1), a.mz = Mo.plane(2), a.mnx = Mo.plane(3), a.mny = Mo.plane(4), a.mnz = Mo.plane(5), a.nm = Mo.n;
  a.pose = c->hyp_pose.as<float>();
  a.dist = o->dist;
  a.cos_thres = (float)std::cos((double)(o->angle_deg / 180.0f) * M_PI);
  a.rev_idx = c->lcp_rev_idx.as<int>(), a.rev_d2 = c->lcp_rev_d2.as<float>(), a.terms = c->lcp_terms.as<float>();
  a.score = c->hyp_score.as<float>();
  a.tiles_per_wave = nn_mode == 3 ? lcp_tiles_per_wave(S.n, H) : 1;  // from the whole set, not from a batch: one association per call
  if (lcp_grid) {
    const float cell = o->dist + GRID_MARGIN * 8;
    GridStore& gm = c->model_grid[HOP_MODEL_1MM];
    if (lcp_cells) {
      CellListStore& cs = c->model_cells[HOP_MODEL_1MM];
      float mcell = o->dist;
      if (const char* e = getenv("HOP_LCP_MODEL_DIV")) mcell = o->dist / (float)atof(e);
      if (!cs.valid || cs.cell != mcell || cs.max_dist != o->dist || cs.coord_mag < c->coord_mag) {
        const int rc = build_cell_lists(c, cs, c->gen.model_h[HOP_MODEL_1MM], c->model_d[HOP_MODEL_1MM], o->dist, mcell);
        if (rc) return rc;
      }
      a.model_cells = cs.c;
      HIPCHK(c, c->pose_inv.ensure(sizeof(float) * 12 * (size_t)H));
      launch_pose_inverse(a.pose, H, c->pose_inv.as<float>(), c->stream);
      a.pose_inv = c->pose_inv.as<float>();
    } else if (!gm.valid || gm.cell != cell) {
      const CloudHost& mh = c->gen.model_h[HOP_MODEL_1MM];
      const int rc = build_grid(c, gm, mh.x.data(), mh.y.data(), mh.z.data(), mh.n, cell);
      if (rc) return rc;
    }
    if (!c->scene_grid.valid || c->scene_grid.cell != cell) {
      const CloudHost& sh = c->gen.scene_h;
      const int rc = build_grid(c, c->scene_grid, sh.x.data(), sh.y.data(), sh.z.data(), sh.n, cell);
      if (rc) return rc;
    }
    a.model_grid = gm.g;
    a.scene_grid = c->scene_grid.g;
    if (lcp_cells) {
      // list cells of half the ring cell when many hypotheses share the lists: 7.2 -> ~4 entries per non-empty cell, the
      // reverse lookups of computeLCP 5.7 -> 3.9 ms at C2 for +0.5 ms of building (1 / 2 / 3 / 4: 7.6 / 6.3 / 7.0 / 8.8 ms)
      int sub = H >= 1024 ? 2 : 1;
      if (const char* e = getenv("HOP_LCP_SCENE_SUB")) sub = std::max(1, atoi(e));
      if (!c->scene_cells.valid || c->scene_cells.max_dist != o->dist || c->scene_cells.sub != sub) {
        // unit normals of the scene, caller order (for the lists) and sorted order (for the walk): computeLCP normalises
        // a normal at every use, which for a scene normal is the same value every time
        const int ns = S.n;
        for (CloudDevice* u : {&c->scene_unit_d, &c->scene_sorted_unit_d}) {
          u->n = ns;
          HIPCHK(c, u->buf.ensure(sizeof(float) * 6 * (size_t)std::max(ns, 1)));
        }
        auto unit = [&](const CloudDevice& src, CloudDevice& dst) {
          float* b = dst.buf.as<float>();
          launch_unit_normals(src.plane(3), src.plane(4), src.plane(5), ns, b + 3 * (size_t)ns, b + 4 * (size_t)ns, b + 5 * (size_t)ns, c->stream);
        };
        unit(c->scene_d, c->scene_unit_d);
        unit(c->scene_sorted_d, c->scene_sorted_unit_d);
        const int rc = build_cell_lists_local(c, c->scene_cells, c->scene_grid, &c->scene_unit_d, o->dist, sub, 0);
        if (rc) return rc;
        c->scene_cells.sub = sub;
      }
      if (nn_mode == 3 && !getenv("HOP_LCP_NO_HEAD")) {  // (HOP_LCP_NO_HEAD=1: the range-record lookups of rounds 2-4, for A/B runs)
        int rc = cell_list_heads(c, c->model_cells[HOP_MODEL_1MM]);
        if (!rc) rc = cell_list_heads(c, c->scene_cells);
        if (rc) return rc;
        a.model_cells = c->model_cells[HOP_MODEL_1MM].c;
      }
      a.scene_cells = c->scene_cells.c;
      if (nn_mode != 3 || getenv("HOP_LCP_NO_HEAD")) a.model_cells.head = a.scene_cells.head = nullptr;  // (records an earlier call may have left in the stores)
    }
    const CloudDevice& Q = c->scene_sorted_d;
    a.qx = Q.plane(0), a.qy = Q.plane(1), a.qz = Q.plane(2), a.qnx = Q.plane(3), a.qny = Q.plane(4), a.qnz = Q.plane(5);
    if (lcp_cells) {
      const CloudDevice& U = c->scene_sorted_unit_d;
      a.qnx = U.plane(3), a.qny = U.plane(4), a.qnz = U.plane(5);
    }
    a.perm = c->scene_perm_d.as<int>();
    a.inv_perm = a.perm + std::max(S.n, 1);
  }
  for (int h0 = 0; h0 < H; h0 += HB) {
    const int hb = std::min(HB, H - h0);
    a.h0 = h0;
    if (lcp_grid) {
      SpanGuard sg(c, T_LCP_FWD);
      if (nn_mode == 3) launch_lcp_cells_fast(a, hb, c->stream);
      else if (lcp_cells) launch_lcp_cells(a, hb, c->stream);
      else launch_lcp_grid(a, hb, c->stream);
    } else {
      {
        SpanGuard sg(c, T_LCP_REV);
        launch_lcp_reverse(a, hb, c->stream);
      }
      {
        SpanGuard sg(c, T_LCP_FWD);
        launch_lcp_forward(a, hb, c->stream);
      }
    }
    {
      SpanGuard sg(c, T_LCP_SUM);
      if (nn_mode == 3) launch_lcp_sum_partial(a, hb, c->stream);
      else if (lcp_cells) launch_lcp_sum_t(a, hb, c->stream);
      else launch_lcp_sum(a, hb, c->stream);
    }
    c->timing.n_lcp_launches += 1;
  }
  c->timing.pairs_lcp += 2ll * H * (long long)S.n * Mo.n;
  // arg-max on the host over H floats: first strict maximum in set order (PoseEstimator.cpp:468-496, best_lcp starts at 0)
  std::vector<float> sc(H);
  HIPCHK(c, hop_ctx_d2h(c, sc.data(), c->hyp_score.p, sizeof(float) * (size_t)H));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int best = 0;
  float best_lcp = 0.f;
  for (int h = 0; h < H; ++h)
    if (sc[h] > best_lcp) best_lcp = sc[h], best = h;
  if (best_index_out) *best_index_out = best;
  if (best_score_out) *best_score_out = sc[best];
  if (best_pose16_out) {
    HIPCHK(c, hop_ctx_d2h(c, best_pose16_out, c->hyp_pose.as<float>() + 16 * (size_t)best, sizeof(float) * 16));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return HOP_OK;
}

// ---------------------------------------------------------------------------------------------- clustering
int hop_cluster_poses_host(const float* poses16, const float* scores, const int* ids, int H, float angle_deg, float dist,
                           const float* sym_deg3, int* keep_out, int* n_keep_out) {
  if (!poses16 || !scores || !ids || !sym_deg3 || !keep_out || H < 0) return HOP_E_INVALID;
  std::vector<int> keep;
  cluster_core(poses16, scores, ids, H, angle_deg, dist, sym_deg3, keep);
  for (size_t i = 0; i < keep.size(); ++i) keep_out[i] = keep[i];
  if (n_keep_out) *n_keep_out = (int)keep.size();
  return HOP_OK;
}

int hop_cluster_pose_terms(const float* pose_a16, const float* pose_b16, float* out5) {
  if (!pose_a16 || !pose_b16 || !out5) return HOP_E_INVALID;
  euler_zyx(pose_a16, out5);
  out5[3] = rotation_geodesic_distance(pose_a16, pose_b16);
  out5[4] = vnorm(v3(pose_a16[3], pose_a16[7], pose_a16[11]) - v3(pose_b16[3], pose_b16[7], pose_b16[11]));
  return HOP_OK;
}

int hop_cluster_poses(hop_ctx* c, float angle_deg, float dist, const float* sym_deg3, int assign_id) {
  if (!c || !sym_deg3) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  const int H = c->n_hyp;
  if (H == 0) return HOP_OK;
  std::vector<float> pose((size_t)H * 16), sc(H);
  std::vector<int> ids(H);
  int n = 0;
  int rc = hop_hypos_download(c, pose.data(), sc.data(), ids.data(), H, &n);
  if (rc) return rc;
  std::vector<int> keep;
  cluster_core(pose.data(), sc.data(), ids.data(), H, angle_deg, dist, sym_deg3, keep);
  std::vector<float> p2(keep.size() * 16), s2(keep.size());
  std::vector<int> i2(keep.size());
  for (size_t k = 0; k < keep.size(); ++k) {
    std::memcpy(&p2[16 * k], &pose[16 * (size_t)keep[k]], sizeof(float) * 16);
    s2[k] = sc[keep[k]];
    i2[k] = assign_id ? (int)k : ids[keep[k]];
  }
  const int K = (int)keep.size();
  HIPCHK(c, hop_ctx_h2d(c, c->hyp_pose.p, p2.data(), sizeof(float) * 16 * (size_t)K));
  HIPCHK(c, hop_ctx_h2d(c, c->hyp_score.p, s2.data(), sizeof(float) * (size_t)K));
  HIPCHK(c, hop_ctx_h2d(c, c->hyp_id.p, i2.data(), sizeof(int) * (size_t)K));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->n_hyp = K;
  return HOP_OK;
}

// ---------------------------------------------------------------------------------------------- top-k exchange
int hop_topk_pack_device(hop_ctx* c, int k, int id_offset, float* rows_dev, int* n_rows_out);
int hop_topk_pack(hop_ctx* c, int k, int id_offset, float* rows_out, int* n_rows_out) {
  if (!c || k <= 0 || !rows_out) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  // sorted and packed on the device (hop_topk_pack_device); only the k rows cross PCIe
  HIPCHK(c, c->topk_rows.ensure(sizeof(float) * (size_t)k * HOP_TOPK_ROW_FLOATS));
  const int rc = hop_topk_pack_device(c, k, id_offset, c->topk_rows.as<float>(), n_rows_out);
  if (rc) return rc;
  HIPCHK(c, hop_ctx_d2h(c, rows_out, c->topk_rows.p, sizeof(float) * (size_t)k * HOP_TOPK_ROW_FLOATS));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HOP_OK;
}

// the same table written on the DEVICE (rows_dev: k rows of HOP_TOPK_ROW_FLOATS floats in device memory): 64-bit HypoCompare keys, one
// radix sort of (key, index) pairs, one pack kernel -- the resident set is neither reordered nor copied to the host.  Returns after
// the work has completed on the context's stream (the caller may use rows_dev from any stream).
int hop_topk_pack_device(hop_ctx* c, int k, int id_offset, float* rows_dev, int* n_rows_out) {
  if (!c || k <= 0 || !rows_dev) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  const int H = c->n_hyp;
  const unsigned* order = nullptr;
  if (H > 0) {
    HIPCHK(c, c->hyp_key.ensure(sizeof(unsigned long long) * (size_t)H));
    HIPCHK(c, c->sort_keys_alt.ensure(sizeof(unsigned long long) * (size_t)H));
    HIPCHK(c, c->sort_vals.ensure(sizeof(unsigned) * (size_t)H));
    HIPCHK(c, c->sort_vals_alt.ensure(sizeof(unsigned) * (size_t)H));
    launch_score_keys(c->hyp_score.as<float>(), c->hyp_id.as<int>(), H, c->hyp_key.as<unsigned long long>(), c->stream);
    launch_iota(c->sort_vals.as<unsigned>(), H, c->stream);
    size_t tmp_bytes = 0;
    HIPCHK(c, prim_sort_pairs(nullptr, tmp_bytes, c->hyp_key.as<unsigned long long>(), c->sort_keys_alt.as<unsigned long long>(), c->sort_vals.as<unsigned>(),
                              c->sort_vals_alt.as<unsigned>(), (size_t)H, 0, 64, c->stream));
    HIPCHK(c, c->sort_tmp.ensure(tmp_bytes + 16));
    HIPCHK(c, prim_sort_pairs(c->sort_tmp.p, tmp_bytes, c->hyp_key.as<unsigned long long>(), c->sort_keys_alt.as<unsigned long long>(), c->sort_vals.as<unsigned>(),
                              c->sort_vals_alt.as<unsigned>(), (size_t)H, 0, 64, c->stream));
    order = c->sort_vals_alt.as<unsigned>();
  }
  launch_topk_pack(order, H, k, id_offset, c->hyp_pose.as<float>(), c->hyp_score.as<float>(), c->hyp_id.as<int>(), rows_dev, c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (n_rows_out) *n_rows_out = std::min(k, H);
  return HOP_OK;
}

int hop_topk_merge(const float* tables, int n_tables, int k, float* rows_out, int* n_rows_out) {
  if (!tables || n_tables <= 0 || k <= 0 || !rows_out) return HOP_E_INVALID;
  std::vector<const float*> rows;
  for (int t = 0; t < n_tables; ++t)
    for (int r = 0; r < k; ++r) {
      const float* row = tables + ((size_t)t * k + r) * HOP_TOPK_ROW_FLOATS;
      int id;
      std::memcpy(&id, &row[1], 4);
      if (id >= 0) rows.push_back(row);
    }
  std::stable_sort(rows.begin(), rows.end(), [](const float* a, const float* b) {
    const uint32_t ka = score_order_key(a[0]), kb = score_order_key(b[0]);  // (the key the device sorts by: -0 = +0, NaN last)
    if (ka != kb) return ka > kb;
    int ia, ib;
    std::memcpy(&ia, &a[1], 4);
    std::memcpy(&ib, &b[1], 4);
    return ia < ib;
  });
  const int m = std::min<int>(k, (int)rows.size());
  for (int r = 0; r < m; ++r) std::memcpy(rows_out + (size_t)r * HOP_TOPK_ROW_FLOATS, rows[r], sizeof(float) * HOP_TOPK_ROW_FLOATS);
  for (int r = m; r < k; ++r) {
    float* row = rows_out + (size_t)r * HOP_TOPK_ROW_FLOATS;
    row[0] = -FLT_MAX;
    const int id = -1;
    std::memcpy(&row[1], &id, 4);
    for (int q = 0; q < 16; ++q) row[2 + q] = 0.f;
  }
  if (n_rows_out) *n_rows_out = m;
  return HOP_OK;
}

// ---------------------------------------------------------------------------------------------- hand
int hop_hand_set_scene(hop_ctx* c, const float* scene_xyz, int n_scene, const float* lookup_nrm, int n_lookup, const float* swivel_xyz,
                       int n_swivel) {
  if (!c || !scene_xyz || n_scene <= 0 || n_lookup < 0 || n_swivel < 0) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  auto up3 = [&](CloudDevice& d, const float* planes, int n) -> int {
    CloudHost h;
    std::vector<float> z((size_t)3 * std::max(n, 1), 0.f);
    load_cloud_host(h, planes ? planes : z.data(), z.data(), n, false);
    return upload_cloud(c, d, h);
  };
  int rc = up3(c->hand_scene_d, scene_xyz, n_scene);
  if (rc) return rc;
  rc = up3(c->hand_lookup_d, lookup_nrm, n_lookup);
  if (rc) return rc;
  rc = up3(c->hand_swivel_d, swivel_xyz, n_swivel);
  if (rc) return rc;
  c->hand_n_scene = n_scene, c->hand_n_lookup = n_lookup, c->hand_n_swivel = n_swivel;
  for (int k = 0; k < 3; ++k) c->hand_scene_h[k].assign(scene_xyz + (size_t)k * n_scene, scene_xyz + (size_t)(k + 1) * n_scene);
  c->hand_grid.valid = false;
  c->hand_cells.valid = false;
  c->have_hand_scene = true;
  return HOP_OK;
}

int hop_hand_set_finger(hop_ctx* c, const hop_finger_args* a) {
  if (!c || !a || !a->model_xyz || !a->model_nrm || a->n_model <= 0 || !a->fp_hist_min_y || a->fp_num_division <= 0) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  c->finger = *a;
  c->finger_hist.assign(a->fp_hist_min_y, a->fp_hist_min_y + a->fp_num_division);
  c->finger.fp_hist_min_y = c->finger_hist.data();
  CloudHost h;
  load_cloud_host(h, a->model_xyz, a->model_nrm, a->n_model, false);
  const int rc = upload_cloud(c, c->hand_model_d, h);
  if (rc) return rc;
  c->finger.model_xyz = nullptr, c->finger.model_nrm = nullptr;
  HIPCHK(c, c->finger_hist_d.ensure(sizeof(float) * a->fp_num_division));
  HIPCHK(c, hop_ctx_h2d(c, c->finger_hist_d.p, c->finger_hist.data(), sizeof(float) * a->fp_num_division));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_finger = true;
  return HOP_OK;
}

static void mat4_vec4(const M4& T, const float v[4], float out[4]) {
  for (int i = 0; i < 4; ++i) out[i] = ((T.m[4 * i] * v[0] + T.m[4 * i + 1] * v[1]) + T.m[4 * i + 2] * v[2]) + T.m[4 * i + 3] * v[3];
}

// objFuncPSO (Hand.cpp:10-178): scalar parts on the host, cloud parts on the device.
int hop_hand_set_sum_mode(hop_ctx* c, int mode) {
  if (!c || (mode != 0 && mode != 1)) return HOP_E_INVALID;
  c->pso_sum_mode = mode;
  return HOP_OK;
}

int hop_hand_pso_eval_batch(hop_ctx* c, const double* angles, int n, double* cost_out) {
  if (!c || !angles || !cost_out || n < 0) return HOP_E_INVALID;
  if (!c->have_finger || !c->have_hand_scene) return HOP_E_STATE;
  if (n == 0) return HOP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const hop_finger_args& a = c->finger;
  HIPCHK(c, c->pso_particles_h.ensure(sizeof(PsoParticle) * (size_t)n));
  HIPCHK(c, c->pso_particles_d.ensure(sizeof(PsoParticle) * (size_t)n));
  HIPCHK(c, c->pso_match_d.ensure(sizeof(int) * 3 * (size_t)n));  // match count | outer count | outer sum, one read-back
  HIPCHK(c, c->pso_terms_d.ensure(sizeof(float) * (size_t)n * (std::max(c->hand_n_swivel, 1) + 4)));
  HIPCHK(c, c->pso_out_h.ensure((sizeof(int) * 2 + sizeof(float)) * (size_t)n));
  PsoParticle* P = static_cast<PsoParticle*>(c->pso_particles_h.p);
  M4 model2handbase, finger_out2parent;
  std::memcpy(model2handbase.m, a.model2handbase, sizeof(float) * 16);
  std::memcpy(finger_out2parent.m, a.finger_out2parent, sizeof(float) * 16);
  std::vector<char> early(n, 0);
  for (int p = 0; p < n; ++p) {
    const double X0 = angles[p];
    M4 tf_self = m4_identity();
    const float ang = (float)X0;
    const float cs = std::cos(ang), sn = std::sin(ang);
    tf_self.m[5] = cs, tf_self.m[6] = -sn, tf_self.m[9] = sn, tf_self.m[10] = cs;
    const M4 cur = m4_mul(model2handbase, tf_self);
    float tip1[4], tip2[4];
    if (a.is_palm_side) {
      const float t1[4] = {a.fo_min[0], a.fo_max[1], a.fo_min[2], 1};
      mat4_vec4(m4_mul(cur, finger_out2parent), t1, tip1);
      const float t2[4] = {a.fp_min[0], a.fp_max[1], a.fp_min[2], 1};
      mat4_vec4(cur, t2, tip2);
    } else {
      const float t1[4] = {a.fp_min[0], a.fp_max[1], a.fp_min[2], 1};
      mat4_vec4(cur, t1, tip1);
      const float t2[4] = {a.fp_min[0], a.fp_max[1], a.fp_max[2], 1};
      mat4_vec4(cur, t2, tip2);
    }
    float g1, g2;
    if (a.is_right_side) g1 = tip1[1] - a.pair_tip1[1], g2 = tip2[1] - a.pair_tip2[1];
    else g1 = -tip1[1] + a.pair_tip1[1], g2 = -tip2[1] + a.pair_tip2[1];
    PsoParticle& pp = P[p];
    std::memset(&pp, 0, sizeof(pp));
    if (g1 < a.gripper_min_dist || g2 < a.gripper_min_dist) {
      float score = 0;
      const float penalty = (float)(1e3 + 1e3 * (double)std::fabs(a.gripper_min_dist - g1));
      score -= penalty;
      cost_out[p] = -score;
      early[p] = 1;
      pp.skip = 1;
      continue;
    }
    const M4 inv = m4_inverse_affine(cur);
    for (int k = 0; k < 12; ++k) pp.T[k] = cur.m[k], pp.Tinv[k] = inv.m[k];
  }
  HIPCHK(c, hipMemcpyAsync(c->pso_particles_d.p, P, sizeof(PsoParticle) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(c->pso_match_d.p, 0, sizeof(int) * 3 * (size_t)n, c->stream));
  PsoArgs pa{};
  pa.particles = c->pso_particles_d.as<PsoParticle>();
  const CloudDevice& Mo = c->hand_model_d;
  pa.mx = Mo.plane(0), pa.my = Mo.plane(1), pa.mz = Mo.plane(2), pa.mnx = Mo.plane(3), pa.mny = Mo.plane(4), pa.mnz = Mo.plane(5), pa.nm = Mo.n;
  pa.sx = c->hand_scene_d.plane(0), pa.sy = c->hand_scene_d.plane(1), pa.sz = c->hand_scene_d.plane(2), pa.ns = c->hand_n_scene;
  pa.lnx = c->hand_lookup_d.plane(0), pa.lny = c->hand_lookup_d.plane(1), pa.lnz = c->hand_lookup_d.plane(2), pa.n_lookup = c->hand_n_lookup;
  pa.wx = c->hand_swivel_d.plane(0), pa.wy = c->hand_swivel_d.plane(1), pa.wz = c->hand_swivel_d.plane(2), pa.n_swivel = c->hand_n_swivel;
  pa.dist_thres = a.dist_thres, pa.cos_normal_thres = a.cos_normal_thres, pa.check_normal = a.check_normal;
  pa.fp_min_z = a.fp_min[2], pa.fp_stride_z = a.fp_stride_z, pa.fp_num_division = a.fp_num_division;
  pa.hist_min_y = c->finger_hist_d.as<float>();
  if (getenv("HOP_PSO_RINGS")) {
    const float cell = a.dist_thres / 2.f + GRID_MARGIN;
    if (!c->hand_grid.valid || c->hand_grid.cell != cell) {
      const int rc = build_grid(c, c->hand_grid, c->hand_scene_h[0].data(), c->hand_scene_h[1].data(), c->hand_scene_h[2].data(), c->hand_n_scene, cell);
      if (rc) return rc;
    }
    pa.scene_grid = c->hand_grid.g;
    pa.max_ring = (int)std::ceil((a.dist_thres + 2 * GRID_MARGIN) / cell);
    pa.use_grid = 1;
  } else {
    // NN cell lists of the hand scene for this gating distance (per frame; shared by the four finger searches)
    const float cell = (a.dist_thres + 8 * GRID_MARGIN) / 2.f;  // finer than the gate: short candidate rows, many of them
    if (!c->hand_grid.valid || c->hand_grid.cell != cell) {
      const int rc = build_grid(c, c->hand_grid, c->hand_scene_h[0].data(), c->hand_scene_h[1].data(), c->hand_scene_h[2].data(), c->hand_n_scene, cell);
      if (rc) return rc;
      c->hand_cells.valid = false;
    }
    if (!c->hand_cells.valid || c->hand_cells.max_dist != a.dist_thres) {
      const int rc = build_cell_lists_local(c, c->hand_cells, c->hand_grid, nullptr, a.dist_thres, 2, 0);
      if (rc) return rc;
    }
    pa.scene_cells = c->hand_cells.c;
    pa.use_grid = 2;
  }
  pa.n_particles = n;
  pa.sum_mode = c->pso_sum_mode;
  pa.match_count = c->pso_match_d.as<int>(), pa.outer_terms = c->pso_terms_d.as<float>();
  pa.outer_cnt = c->pso_match_d.as<int>() + n, pa.outer_sum = reinterpret_cast<float*>(c->pso_match_d.as<int>() + 2 * (size_t)n);
  {
    SpanGuard sg(c, T_PSO);
    launch_pso(pa, n, c->stream);
  }
  c->timing.n_pso_launches += 1;
  c->timing.pairs_pso += (long long)n * Mo.n * c->hand_n_scene;
  int* h_match = static_cast<int*>(c->pso_out_h.p);
  int* h_cnt = h_match + n;
  float* h_sum = reinterpret_cast<float*>(h_cnt + n);
  HIPCHK(c, hipMemcpyAsync(h_match, c->pso_match_d.p, sizeof(int) * 3 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int p = 0; p < n; ++p) {
    if (early[p]) continue;
    const double X0 = angles[p];
    float score = 0;
    // num_match += 1 + X[0] per match, float accumulator (Hand.cpp:99,104,117): replayed exactly
    float num_match = 0;
    for (int k = 0; k < h_match[p]; ++k) num_match = (float)((double)num_match + (1 + X0));
    score += num_match;
    if (num_match == 0) {
      score = (float)(-100 + X0);
      cost_out[p] = -score;
      continue;
    }
    const int num_outer = h_cnt[p];
    const float avg = h_sum[p] / num_outer;  // 0/0 -> NaN when nothing is outside, as in the reference
    if (num_outer >= a.max_outter_pts || avg >= 0.005) {
      const float pen = (float)(1e3 + (double)(a.outter_pt_dist_weight * std::max(avg - a.outter_pt_dist, 0.0f)));
      score -= pen;
    } else if (num_outer >= 0 && avg - a.outter_pt_dist > 0) {
      const float pen = a.outter_pt_dist_weight * std::exp(avg * 1000);
      score -= pen;
    }
    cost_out[p] = -score;
  }
  return HOP_OK;
}

void hop_pso_default_settings(hop_pso_settings* s) {
  if (!s) return;
  s->n_pop = 15, s->n_gen = 3, s->check_freq = 10;
  s->c_cog = 0.1, s->c_soc = 0.9, s->initial_w = 0.0;
  s->w_min = 0.10, s->w_max = 0.99, s->err_tol = 1e-5;
  s->lower_rad = 0.0, s->upper_rad = 120.0 * M_PI / 180.0;
  s->seed = 0;
}

// optim::pso_int (pso.hpp:146-351): vals_bound, centre particle, inertia method 1, velocity method 1.
// Random stream: std::mt19937_64(seed) + uniform_real_distribution<double>(0,1) (what Armadillo's C++11
// backend draws from after arma_rng::set_seed); see DESIGN.md "PSO random stream".
int hop_hand_pso_search(hop_ctx* c, const hop_pso_settings* s, double* best_angle_out, double* objval_out) {
  if (!c || !s || s->n_pop <= 0 || s->n_gen < 0) return HOP_E_INVALID;
  if (!c->have_finger || !c->have_hand_scene) return HOP_E_STATE;
  std::mt19937_64 eng(s->seed);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  const size_t n_pop = (size_t)s->n_pop + 1;
  const size_t n_gen = (size_t)s->n_gen;
  const size_t check_freq = s->check_freq > 0 ? (size_t)s->check_freq : n_gen;
  double par_w = s->initial_w;
  auto inv_tf = [&](double v) { return v * (s->upper_rad - s->lower_rad) + s->lower_rad; };
  std::vector<double> P(n_pop), V(n_pop), objfn(n_pop), ang(n_pop);
  auto eval_all = [&]() -> int {
    for (size_t i = 0; i < n_pop; ++i) ang[i] = inv_tf(P[i]);
    const int rc = hop_hand_pso_eval_batch(c, ang.data(), (int)n_pop, objfn.data());
    if (rc) return rc;
    for (size_t i = 0; i < n_pop; ++i)
      if (!std::isfinite(objfn[i])) objfn[i] = std::numeric_limits<double>::max();
    return HOP_OK;
  };
  for (size_t i = 0; i < n_pop; ++i) P[i] = U(eng);
  auto center = [&]() {
    double sum = 0;
    for (size_t i = 0; i + 1 < n_pop; ++i) sum += P[i];
    P[n_pop - 1] = sum / double(n_pop - 1);
  };
  center();
  int rc = eval_all();
  if (rc) return rc;
  std::vector<double> best_vals = objfn, best_vecs = P;
  size_t gi = std::min_element(objfn.begin(), objfn.end()) - objfn.begin();
  double cur_best = objfn[gi], best_check = cur_best, gbest = P[gi];
  size_t iter = 0;
  double err = 2.0 * s->err_tol;
  for (size_t i = 0; i < n_pop; ++i) V[i] = U(eng);
  std::vector<double> r1(n_pop), r2(n_pop);
  while (err > s->err_tol && iter < n_gen) {
    iter++;
    for (size_t i = 0; i < n_pop; ++i) r1[i] = U(eng);
    for (size_t i = 0; i < n_pop; ++i) r2[i] = U(eng);
    for (size_t i = 0; i < n_pop; ++i) {
      V[i] = par_w * V[i] + s->c_cog * r1[i] * (best_vecs[i] - P[i]) + s->c_soc * r2[i] * (gbest - P[i]);
      P[i] += V[i];
    }
    center();
    for (size_t i = 0; i < n_pop; ++i) P[i] = std::min(std::max(P[i], 0.0), 1.0);
    rc = eval_all();
    if (rc) return rc;
    for (size_t i = 0; i < n_pop; ++i)
      if (objfn[i] < best_vals[i]) best_vals[i] = objfn[i], best_vecs[i] = P[i];
    const size_t mi = std::min_element(best_vals.begin(), best_vals.end()) - best_vals.begin();
    if (best_vals[mi] < cur_best) cur_best = best_vals[mi], gbest = best_vecs[mi];
    if (iter % check_freq == 0) err = std::fabs(cur_best - best_check) / (1e-20 + std::fabs(best_check));
    if (cur_best < best_check) best_check = cur_best;
    par_w = s->w_min + (s->w_max - s->w_min) * double(iter + 1) / double(n_gen);
  }
  if (best_angle_out) *best_angle_out = inv_tf(gbest);
  if (objval_out) *objval_out = (double)(float)best_check;
  return HOP_OK;
}

// ---------------------------------------------------------------------------------------------- N4 (computePPF pair loop)
int hop_model_ppf_keys(hop_ctx* c, const float* xyz, const float* nrm, int n, int32_t* keys4_out, int cap, int* n_keys) {
  if (!c || !xyz || !nrm || n < 0 || !n_keys || cap < 0 || (cap > 0 && !keys4_out)) return HOP_E_INVALID;
  *n_keys = 0;
  if (n < 2) return HOP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  CloudHost h;
  load_cloud_host(h, xyz, nrm, n, false);  // normals as given: the kernel normalises once, as the tool does
  CloudDevice d;
  int rc = upload_cloud(c, d, h);
  if (rc) return rc;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = 0; i < n; ++i) {
    mn[0] = std::min(mn[0], h.x[i]), mn[1] = std::min(mn[1], h.y[i]), mn[2] = std::min(mn[2], h.z[i]);
    mx[0] = std::max(mx[0], h.x[i]), mx[1] = std::max(mx[1], h.y[i]), mx[2] = std::max(mx[2], h.z[i]);
  }
  const double diag = std::sqrt((double)(mx[0] - mn[0]) * (mx[0] - mn[0]) + (double)(mx[1] - mn[1]) * (mx[1] - mn[1]) +
                                (double)(mx[2] - mn[2]) * (mx[2] - mn[2]));
  if (!(diag * 1000.0 < 1.0e6)) {
    d.buf.release();
    return HOP_E_CAPACITY;
  }
  const int dist_bins = (int)(diag * 1000.0 / 5.0) + 3;
  const size_t words = ((size_t)dist_bins * 19 * 19 * 19 + 31) / 32 + 1;
  DevBuf bm;
  rc = HOP_OK;
  std::vector<unsigned> host(words);
  int overflow = 0;
  do {
    if (bm.ensure(sizeof(unsigned) * words + sizeof(int)) != hipSuccess) {
      rc = HOP_E_ALLOC;
      break;
    }
    if (hipMemsetAsync(bm.p, 0, sizeof(unsigned) * words + sizeof(int), c->stream) != hipSuccess) {
      rc = HOP_E_HIP;
      break;
    }
    int* ovf = reinterpret_cast<int*>(bm.as<unsigned>() + words);
    launch_model_ppf_keys(d.plane(0), d.plane(1), d.plane(2), d.plane(3), d.plane(4), d.plane(5), n, dist_bins, bm.as<unsigned>(), ovf, c->stream);
    if (hop_ctx_d2h(c, host.data(), bm.p, sizeof(unsigned) * words) != hipSuccess ||
        hop_ctx_d2h(c, &overflow, ovf, sizeof(int)) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
      rc = HOP_E_HIP;
  } while (false);
  bm.release();
  d.buf.release();
  if (rc) return rc;
  if (overflow) return HOP_E_CAPACITY;
  int count = 0;
  for (int dd = 0; dd < dist_bins; ++dd)
    for (int a1 = 0; a1 < 19; ++a1)
      for (int a2 = 0; a2 < 19; ++a2)
        for (int a3 = 0; a3 < 19; ++a3) {
          const size_t bit = (((size_t)dd * 19 + a1) * 19 + a2) * 19 + a3;
          if (!((host[bit >> 5] >> (bit & 31)) & 1u)) continue;
          if (count < cap) {
            int32_t* k = keys4_out + 4 * (size_t)count;
            k[0] = dd * 5, k[1] = a1 * 10, k[2] = a2 * 10, k[3] = a3 * 10;
          }
          ++count;
        }
  *n_keys = count;
  return count > cap ? HOP_E_CAPACITY : HOP_OK;
}

// ---------------------------------------------------------------------------------------------- N3a
int hop_hand_remove_surrounding(hop_ctx* c, const float* scene_xyz, const float* scene_nrm, int n, const float handbase_in_cam[16],
                                const hop_hand_link* links, int n_links, const float finger12_in_handbase[16],
                                const float finger22_in_handbase[16], float finger12_min_z, float* out_xyz, float* out_nrm, float* out_conf,
                                int* keep_index, int* n_out) {
  if (!c || !scene_xyz || !scene_nrm || n < 0 || !handbase_in_cam || (!links && n_links > 0) || n_links < 0 || !finger12_in_handbase ||
      !finger22_in_handbase || !out_xyz || !out_nrm || !out_conf || !n_out)
    return HOP_E_INVALID;
  *n_out = 0;
  if (n == 0) return HOP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t N = (size_t)n;
  // inputs: scene planes, concatenated link clouds
  std::vector<float4> lp;
  std::vector<int> ls(n_links + 1, 0);
  std::vector<float> lt(std::max(n_links, 1), 0.f);
  for (int l = 0; l < n_links; ++l) {
    if (links[l].n < 0 || (links[l].n > 0 && !links[l].xyz)) return HOP_E_INVALID;
    for (int j = 0; j < links[l].n; ++j)
      lp.push_back(make_float4(links[l].xyz[j], links[l].xyz[(size_t)links[l].n + j], links[l].xyz[2 * (size_t)links[l].n + j], 0.f));
    ls[l + 1] = (int)lp.size();
    lt[l] = links[l].sq_dist_thres;
  }
  HIPCHK(c, c->sur_in.ensure(sizeof(float) * 6 * N));
  HIPCHK(c, c->sur_ws.ensure(sizeof(float) * 6 * N + sizeof(float) * N + sizeof(int) * 2 * (N + 1)));
  HIPCHK(c, c->sur_links.ensure(sizeof(float4) * std::max<size_t>(lp.size(), 1) + sizeof(int) * (n_links + 1) + sizeof(float) * std::max(n_links, 1)));
  HIPCHK(c, c->sur_out.ensure(sizeof(float) * 7 * N + sizeof(int) * N));
  float* in = c->sur_in.as<float>();
  HIPCHK(c, hop_ctx_h2d(c, in, scene_xyz, sizeof(float) * 3 * N));
  HIPCHK(c, hop_ctx_h2d(c, in + 3 * N, scene_nrm, sizeof(float) * 3 * N));
  float4* lpd = c->sur_links.as<float4>();
  int* lsd = reinterpret_cast<int*>(lpd + std::max<size_t>(lp.size(), 1));
  float* ltd = reinterpret_cast<float*>(lsd + n_links + 1);
  if (!lp.empty()) HIPCHK(c, hop_ctx_h2d(c, lpd, lp.data(), sizeof(float4) * lp.size()));
  HIPCHK(c, hop_ctx_h2d(c, lsd, ls.data(), sizeof(int) * (n_links + 1)));
  HIPCHK(c, hop_ctx_h2d(c, ltd, lt.data(), sizeof(float) * std::max(n_links, 1)));
  M4 hb, f1, f2;
  std::memcpy(hb.m, handbase_in_cam, sizeof(float) * 16);
  std::memcpy(f1.m, finger12_in_handbase, sizeof(float) * 16);
  std::memcpy(f2.m, finger22_in_handbase, sizeof(float) * 16);
  const M4 cam2hb = m4_inverse_affine(hb), f1i = m4_inverse_affine(f1), f2i = m4_inverse_affine(f2);
  SurroundArgs a{};
  a.sx = in, a.sy = in + N, a.sz = in + 2 * N, a.snx = in + 3 * N, a.sny = in + 4 * N, a.snz = in + 5 * N, a.n = n;
  for (int k = 0; k < 12; ++k) a.cam2hb[k] = cam2hb.m[k], a.hb2cam[k] = hb.m[k], a.f1inv[k] = f1i.m[k], a.f2inv[k] = f2i.m[k];
  a.link_pts = lpd, a.link_start = lsd, a.link_thres = ltd, a.n_links = n_links, a.min_z = finger12_min_z;
  float* ws = c->sur_ws.as<float>();
  a.hbp = ws, a.conf = ws + 6 * N;
  int* keep = reinterpret_cast<int*>(ws + 7 * N);
  int* pos = keep + (N + 1);
  a.keep = keep;
  HIPCHK(c, hipMemsetAsync(keep + N, 0, sizeof(int), c->stream));
  launch_hand_surround(a, c->stream);
  size_t tmp_bytes = 0;
  HIPCHK(c, prim_exclusive_sum(nullptr, tmp_bytes, keep, pos, n + 1, c->stream));
  HIPCHK(c, c->sort_tmp.ensure(tmp_bytes + 16));
  HIPCHK(c, prim_exclusive_sum(c->sort_tmp.p, tmp_bytes, keep, pos, n + 1, c->stream));
  SurroundOutArgs o{};
  o.hbp = a.hbp, o.conf = a.conf, o.keep = keep, o.pos = pos, o.n = n;
  for (int k = 0; k < 12; ++k) o.hb2cam[k] = hb.m[k];
  float* od = c->sur_out.as<float>();
  o.ox = od, o.oy = od + N, o.oz = od + 2 * N, o.onx = od + 3 * N, o.ony = od + 4 * N, o.onz = od + 5 * N, o.oconf = od + 6 * N;
  o.oindex = reinterpret_cast<int*>(od + 7 * N);
  launch_hand_surround_out(o, c->stream);
  int kept = 0;
  HIPCHK(c, hop_ctx_d2h(c, &kept, pos + N, sizeof(int)));
  HIPCHK(c, hop_ctx_d2h(c, out_xyz, od, sizeof(float) * 3 * N));
  HIPCHK(c, hop_ctx_d2h(c, out_nrm, od + 3 * N, sizeof(float) * 3 * N));
  HIPCHK(c, hop_ctx_d2h(c, out_conf, od + 6 * N, sizeof(float) * N));
  if (keep_index) HIPCHK(c, hop_ctx_d2h(c, keep_index, o.oindex, sizeof(int) * N));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_out = kept;
  return HOP_OK;
}

// ---------------------------------------------------------------------------------------------- timing
int hop_timing_enable(hop_ctx* c, int on) {
  if (!c) return HOP_E_INVALID;
  c->timing_on = on != 0;
  return HOP_OK;
}
int hop_timing_reset(hop_ctx* c) {
  if (!c) return HOP_E_INVALID;
  resolve_spans(c);
  std::memset(&c->timing, 0, sizeof(c->timing));
  return HOP_OK;
}
// development aid, not part of the ABI (include/hop.h): per-query statistics of the packed ICP lookups, all zero unless
// the library was built with -DHOP_ICP_COUNT (tools/icp_counters.py)
int hop_debug_icp_counters(hop_ctx* c, unsigned long long* out8, int reset) {
  if (!c || !out8) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hop::icp_counters_read(out8, reset != 0);
  return HOP_OK;
}

// development aid, not part of the ABI: statistics of the Levenberg-Marquardt minimiser of nn_mode 6 (zero unless built with -DHOP_LM_COUNT)
int hop_debug_lm_counters(hop_ctx* c, unsigned long long* out8, int reset) {
  if (!c || !out8) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hop::lm_counters_read(out8, reset != 0);
  return HOP_OK;
}

// development aid, not part of the ABI: the PPF key membership matrix of the last hop_generate (n rows of `words` 64-bit words)
int hop_debug_ppf_matrix(hop_ctx* c, unsigned long long* out, size_t cap_words, int* n_out, int* words_out) {
  if (!c || !n_out || !words_out) return HOP_E_INVALID;
  const int N = c->gen.gp_h.n, W = (N + 63) / 64;
  *n_out = N, *words_out = W;
  if (!c->ppf_matrix_cached) return HOP_E_STATE;
  if (out) {
    if (cap_words < (size_t)N * W) return HOP_E_CAPACITY;
    std::memcpy(out, c->ppf_matrix_cached, sizeof(unsigned long long) * (size_t)N * W);
  }
  return HOP_OK;
}

// development aid, not part of the ABI: which moment kernel the last nn_mode-7 refinement of this context ran (1 k_icp_fusedq_momm on the
// matrix cores, 0 k_icp_fusedq_momi on the vector units -- HOP_ICP_MFMA=0, or a device that failed the read-out check --, -1 none yet)
int hop_debug_icp_engine(hop_ctx* c) { return c ? c->icp_last_engine : -1; }

// development aid, not part of the ABI: what the first-use checks of the gfx950-specific instructions said on this context's device.
//   bit 0: the packed ranking (q_rank, v_med3_u32) has been checked, bit 1: it reproduced its scalar statement;
//   bit 2: the matrix-core read-out of k_icp_fusedq_momm has been checked, bit 3: it reproduced sum U U^T.
// force != 0 runs the checks now (smoke(), bench.py and the tests ask before they time or judge anything); a checked-and-failed bit pair
// (01 / 0100) means the library is running a substitute kernel on this device and has said so on stderr.
int hop_debug_selfcheck(hop_ctx* c, int force) {
  if (!c) return 0;
  if (force) {
    if (hipSetDevice(c->device) != hipSuccess) return 0;
    (void)qrank_ok(c);
    (void)mfma_i8_layout_ok(c);
  }
  std::lock_guard<std::mutex> lk(g_selfcheck_mu);
  auto it = g_selfcheck.find(c->device);
  return it == g_selfcheck.end() ? 0 : it->second;
}

// development aid, not part of the ABI: list gathers of k_lcp_cells_fast per lookup (zero unless built with -DHOP_LCP_COUNT; tools/lcp_counters.py)
int hop_debug_lcp_counters(hop_ctx* c, unsigned long long* out4, int reset) {
  if (!c || !out4) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hop::lcp_counters_read(out4, reset != 0);
  return HOP_OK;
}

// development aid, not part of the ABI: the gfx950-specific primitives of the ICP kernels on caller-given operands (k_dev_selftest_*)
//   what = 0: n elements; in = x[n] y[n] (float) ia[n] ib[n] ic[n] (int32) consecutively, out = 9 x n uint32
//   what = 1: n tiles of one v_mfma_i32_16x16x64_i8; in = a, b, c as [n][64][4] int32 consecutively, out = d [n][64][4] int32
//   what = 2: the read-out path of k_icp_fusedq_momm on one wavefront (k_dev_selftest_momm): n batches; in = U [n][64][13] int32 then the
//             accepted-lane masks [n] uint64; out = the tiles HH, HL, LL as [3][64][4] int32
int hop_debug_selftest(hop_ctx* c, int what, int n, const void* in, void* out) {
  if (!c || !in || !out || n <= 0 || what < 0 || what > 2) return HOP_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t in_bytes = what == 0 ? sizeof(int) * 5 * (size_t)n : what == 1 ? sizeof(int) * 3 * 256 * (size_t)n : (sizeof(int) * 64 * 13 + 8) * (size_t)n;
  const size_t out_bytes = what == 0 ? sizeof(int) * 9 * (size_t)n : what == 1 ? sizeof(int) * 256 * (size_t)n : sizeof(int) * 3 * 256;
  DevBuf din, dout;
  const auto run = [&]() -> int {
    HIPCHK(c, din.ensure(in_bytes));
    HIPCHK(c, dout.ensure(out_bytes));
    HIPCHK(c, hop_ctx_h2d(c, din.p, in, in_bytes));
    const int* ip = din.as<int>();
    if (what == 0) hop::launch_dev_selftest_scalar(n, din.as<float>(), din.as<float>() + n, ip + 2 * (size_t)n, ip + 3 * (size_t)n, ip + 4 * (size_t)n, dout.as<unsigned>(), c->stream);
    else if (what == 1) hop::launch_dev_selftest_mfma(n, ip, ip + 256 * (size_t)n, ip + 512 * (size_t)n, dout.as<int>(), c->stream);
    else hop::launch_dev_selftest_momm(n, ip, reinterpret_cast<const unsigned long long*>(ip + 64 * 13 * (size_t)n), dout.as<int>(), c->stream);
    HIPCHK(c, hop_ctx_d2h(c, out, dout.p, out_bytes));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HOP_OK;
  };
  const int rc = run();
  (void)hipStreamSynchronize(c->stream);
  din.release(), dout.release();
  return rc;
}

int hop_timing_get(hop_ctx* c, hop_timing* out) {
  if (!c || !out) return HOP_E_INVALID;
  resolve_spans(c);
  *out = c->timing;
  return HOP_OK;
}

}  // extern "C"
