// PoseEstimator.h -- host-side mirror of the reference's PoseEstimator<PointT> for the hot-path members
// (src/perception/include/PoseEstimator.h:12-49), forwarding to the C-ABI of include/hop.h.  PCL cloud types are
// replaced by SoA planes (hop::Cloud), OBJ files by vertex / face arrays (hop::Mesh), cv::Mat depth by a 16-bit image.
#ifndef HOP_HOST_POSEESTIMATOR_H_
#define HOP_HOST_POSEESTIMATOR_H_
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/hop.h"
#include "ConfigParser.h"
#include "PoseHypo.h"

namespace hop {
struct Cloud {            // SoA planes, as the ABI takes them
  std::vector<float> xyz;  // 3*n
  std::vector<float> nrm;  // 3*n
  std::vector<float> conf; // n (optional)
  int n = 0;
};
struct Mesh {             // what igl::readOBJ / pcl::PolygonMesh hold: vertices (nv x 3) and triangles (nf x 3)
  std::vector<float> V;
  std::vector<int32_t> F;
  int nv() const { return (int)(V.size() / 3); }
  int nf() const { return (int)(F.size() / 3); }
};
// hop_icp_opts.nn_mode of the mirrors: 7 (the oracle's bits).  HOP_ICP_NN_MODE=<0..7> overrides it without a rebuild -- e.g. 6, the float-sum
// form that round 3 measured on hardware, should a device disagree with nn_mode 7's CPU-verified bits.
inline int icp_nn_mode_reference() {
  const char* e = std::getenv("HOP_ICP_NN_MODE");
  if (!e || !*e) return 7;
  char* end = nullptr;
  const long m = std::strtol(e, &end, 10);
  while (end && (*end == ' ' || *end == '\t' || *end == '\n')) ++end;
  if (end == e || (end && *end) || m < 0 || m > 7) {  // empty, trailing text or out of range: the default, and a line that says so
    std::fprintf(stderr, "hop: HOP_ICP_NN_MODE='%s' is not an integer in 0..7; using 7\n", e);
    return 7;
  }
  return (int)m;
}
inline void check(int rc, hop_ctx* c, const char* where) {
  if (rc != HOP_OK) throw std::runtime_error(std::string(where) + ": " + hop_strerror(rc) + " " + (c ? hop_last_error(c) : ""));
}
// The mirrors' refinement call.  nn_mode 7 needs the packed model lists (a 5 mm model of < 65535 points inside the 16-bit cell frame) and
// refuses with HOP_E_STATE without them; the reference refines any model, so the mirrors retry once with nn_mode 5 (the same minimiser in its
// per-evaluation float form on the plain lists) and say so on stderr -- a different arithmetic is never taken silently.
inline void icp_refine_reference(hop_ctx* c, hop_icp_opts o, const char* where) {
  int rc = hop_icp_refine(c, &o, nullptr, nullptr);
  if (rc == HOP_E_STATE && o.nn_mode == 7) {
    std::fprintf(stderr, "hop: %s: %s -- retrying with nn_mode 5 (per-evaluation float form of the same minimiser)\n", where, hop_last_error(c));
    o.nn_mode = 5;
    rc = hop_icp_refine(c, &o, nullptr, nullptr);
  }
  check(rc, c, where);
}
}  // namespace hop

class PoseEstimator {
 public:
  // PoseEstimator(cfg, model, model001, K) (PoseEstimator.cpp:8-23)
  PoseEstimator(ConfigParser* cfg1, const hop::Cloud& model, const hop::Cloud& model001, int device = 0) : cfg(cfg1) {
    hop::check(hop_ctx_create(device, &ctx_), nullptr, "hop_ctx_create");
    hop::check(hop_set_model(ctx_, HOP_MODEL_5MM, model.xyz.data(), model.nrm.data(), model.n), ctx_, "hop_set_model(5mm)");
    hop::check(hop_set_model(ctx_, HOP_MODEL_1MM, model001.xyz.data(), model001.nrm.data(), model001.n), ctx_, "hop_set_model(1mm)");
    _model_xyz = model.xyz;
    _n_model = model.n;
    // PoseEstimator.cpp:12-20: centroid (float sums in order, as pcl::computeCentroid) and bounding box of the 1 mm model
    float s[3] = {0, 0, 0}, mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) {
      const float* p = model001.xyz.data() + (size_t)k * model001.n;
      for (int i = 0; i < model001.n; ++i) {
        s[k] += p[i];
        if (i == 0 || p[i] < mn[k]) mn[k] = p[i];
        if (i == 0 || p[i] > mx[k]) mx[k] = p[i];
      }
      _model_center_init[k] = model001.n ? s[k] / (float)model001.n : 0.f;
    }
    const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    _smallest_dim = std::min(std::min(ex, ey), ez);
    _ob_diameter = std::sqrt(ex * ex + ey * ey + ez * ez);
  }
  ~PoseEstimator() { hop_ctx_destroy(ctx_); }
  PoseEstimator(const PoseEstimator&) = delete;
  PoseEstimator& operator=(const PoseEstimator&) = delete;

  // setCurScene (PoseEstimator.cpp:32-46): keeps points with confidence >= pose_estimator_high_confidence_thres
  void setCurScene(const hop::Cloud& object_segment, const hop::Cloud& cloud_withouthand_raw) {
    _cloud_withouthand_raw = cloud_withouthand_raw;
    setCurScene(object_segment);
  }
  void setCurScene(const hop::Cloud& object_segment) {
    hop::check(hop_set_scene(ctx_, object_segment.xyz.data(), object_segment.nrm.data(),
                             object_segment.conf.empty() ? nullptr : object_segment.conf.data(), object_segment.n,
                             cfg->pose_estimator_high_confidence_thres),
               ctx_, "hop_set_scene");
  }

  // reset (PoseEstimator.cpp:49-60): scene clouds and hypotheses of the frame are dropped, the models stay
  void reset() {
    _cloud_withouthand_raw = hop::Cloud();
    const float none[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    hop::check(hop_hypos_upload(ctx_, none, nullptr, 0), ctx_, "hop_hypos_upload");
  }

  // runSuper4pcs (PoseEstimator.cpp:62-100); ppf_keys4 = the key set of the reference's ppf map (4 ints per key)
  bool runSuper4pcs(const std::vector<int32_t>& ppf_keys4) {
    hop::check(hop_set_ppf_keys(ctx_, ppf_keys4.data(), (int)(ppf_keys4.size() / 4)), ctx_, "hop_set_ppf_keys");
    hop_s4pcs_opts o;
    hop_s4pcs_default_opts(&o);
    o.sample_size = cfg->super4pcs_sample_size;
    o.overlap = cfg->super4pcs_overlap;
    o.max_time_seconds = cfg->super4pcs_max_time_seconds;
    o.delta = cfg->super4pcs_delta;
    o.dispersion = cfg->getf("super4pcs_dispersion");
    o.success_quadrilaterals = cfg->geti("super4pcs_success_quadrilaterals");
    o.max_normal_difference = cfg->super4pcs_max_normal_difference;
    o.max_color_distance = cfg->super4pcs_max_color_distance;
    int n = 0;
    const int rc = hop_s4pcs_generate(ctx_, &o, nullptr, nullptr, 0, &n, &last_stats);
    if (rc == HOP_E_NO_HYPOTHESIS) return false;
    hop::check(rc, ctx_, "hop_s4pcs_generate");
    return n > 0;
  }

  // clusterPoses (PoseEstimator.cpp:106-233)
  void clusterPoses(float angle_diff, float dist_diff, bool assign_id) {
    const std::string base = "object_symmetry." + cfg->get("model_name") + ".";
    const float sym[3] = {cfg->getf(base + "x"), cfg->getf(base + "y"), cfg->getf(base + "z")};
    hop::check(hop_cluster_poses(ctx_, angle_diff, dist_diff, sym, assign_id ? 1 : 0), ctx_, "hop_cluster_poses");
  }

  // refineByICP (PoseEstimator.cpp:235-275)
  void refineByICP() {
    // nn_mode 7: the reference's minimiser (PCL's TransformationEstimationPointToPlane = Eigen's Levenberg-Marquardt, Utils.cpp:200-216) from
    // integer-exact moment sums with an IEEE-only solve: the refined poses are the CPU oracle's, bit for bit
    hop_icp_opts o{10, cfg->getf("icp_angle_thres"), cfg->getf("icp_dist_thres"), 100, hop::icp_nn_mode_reference()};
    hop::icp_refine_reference(ctx_, o, "hop_icp_refine");
  }

  // registerMesh (PoseEstimator.cpp:505-508 -> SDFchecker::registerMesh, SDFchecker.cpp:36-78)
  void registerMesh(const hop::Mesh& mesh, const std::string& name, const float* pose16) {
    auto it = _mesh_ids.find(name);
    const int id = it != _mesh_ids.end() ? it->second : (int)_mesh_ids.size();
    hop::check(hop_sdf_register_mesh(ctx_, id, mesh.V.data(), mesh.nv(), mesh.F.data(), mesh.nf(), pose16), ctx_, "hop_sdf_register_mesh");
    _mesh_ids[name] = id;
    if (name == "object") _obj_mesh = mesh;  // what rejectByRender draws under every hypothesis
  }
  // the frame's depth image and the camera (setCurScene's depth argument, the constructor's K): rejectByRender only
  void setDepth(const std::vector<uint16_t>& depth_raw, int H, int W, double depth_unit, const float K9[9]) {
    _depth_raw = depth_raw, _H = H, _W = W, _depth_unit = depth_unit;
    std::copy(K9, K9 + 9, _K);
  }
  // rejectByRender (PoseEstimator.cpp:345-463; projection_thres is unused there as well): HandT provides _meshes (the
  // components' meshes, link frame), _component_status, getTFHandBase and _handbase_in_cam
  template <class HandT>
  void rejectByRender(float /*projection_thres*/, HandT* hand, int sum_mode = 0) {
    printf("before projection check, #hypo=%d\n", numHypos());
    std::vector<float> V;
    std::vector<int32_t> F;
    for (const auto& h : hand->_meshes) {
      if (!hand->_component_status[h.first]) continue;
      typename HandT::Mat tf_in_base;
      hand->getTFHandBase(h.first, tf_in_base);
      const typename HandT::Mat T = hand->_handbase_in_cam * tf_in_base;
      const int32_t off = (int32_t)(V.size() / 3);
      for (int i = 0; i < h.second.nv(); ++i) {
        const float* p = &h.second.V[3 * (size_t)i];
        for (int r = 0; r < 3; ++r) V.push_back(((T.m[4 * r] * p[0] + T.m[4 * r + 1] * p[1]) + T.m[4 * r + 2] * p[2]) + T.m[4 * r + 3]);
      }
      for (int32_t f : h.second.F) F.push_back(f + off);
    }
    hop::check(hop_render_set_frame(ctx_, _depth_raw.data(), _H, _W, _depth_unit, _K, V.data(), (int)(V.size() / 3), F.data(), (int)(F.size() / 3)), ctx_,
               "hop_render_set_frame");
    hop::check(hop_render_set_object(ctx_, _obj_mesh.V.data(), _obj_mesh.nv(), _obj_mesh.F.data(), _obj_mesh.nf()), ctx_, "hop_render_set_object");
    int kept = 0;
    hop::check(hop_reject_by_render(ctx_, cfg->getf("render_roi_weight"), cfg->getf("render_keep_hypo"), sum_mode, nullptr, nullptr, &kept), ctx_,
               "hop_reject_by_render");
    printf("after projection check, #hypo=%d\n", kept);
  }
  // registerHandMesh (PoseEstimator.cpp:510-520): HandT is the host Hand mirror (getTFHandBase, _convex_meshes)
  template <class HandT>
  void registerHandMesh(HandT* hand) {
    for (const auto& h : hand->_convex_meshes) {
      if (!(h.first == "finger_1_1" || h.first == "finger_1_2" || h.first == "finger_2_1" || h.first == "finger_2_2")) continue;
      typename HandT::Mat model2handbase;
      hand->getTFHandBase(h.first, model2handbase);
      registerMesh(h.second, h.first, model2handbase.m);
    }
  }
  // rejectByCollisionOrNonTouching (PoseEstimator.cpp:524-735); cam2handbase = hand->_handbase_in_cam.inverse()
  template <class HandT>
  void rejectByCollisionOrNonTouching(HandT* hand) {
    if (cfg->get("pose_estimator_use_physics") == "false") {
      printf("Not using physics\n");
      return;
    }
    static const char* names[4] = {"finger_1_1", "finger_1_2", "finger_2_1", "finger_2_2"};
    hop_physics_args a{};
    a.object_mesh = _mesh_ids.at("object");
    typename HandT::Mat f2h[4];
    for (int k = 0; k < 4; ++k) {
      a.finger_mesh[k] = _mesh_ids.at(names[k]);
      const hop::Cloud& fc = hand->_clouds.at(names[k]);
      a.finger_xyz[k] = fc.xyz.data(), a.finger_n[k] = fc.n;
      hand->getTFHandBase(names[k], f2h[k]);
      a.finger2handbase[k] = f2h[k].m;
      a.finger_status[k] = hand->_component_status[names[k]] ? 1 : 0;
    }
    const hop::Cloud& hc = hand->handCloud();
    a.hand_cloud_xyz = hc.xyz.data(), a.n_hand_cloud = hc.n;
    a.cloud_without_hand_xyz = _cloud_withouthand_raw.xyz.data(), a.n_cloud_without_hand = _cloud_withouthand_raw.n;
    hand->camToHandbase(a.cam2handbase);
    a.model_xyz = _model_xyz.data(), a.n_model = _n_model;
    for (int k = 0; k < 3; ++k) a.model_center_init[k] = _model_center_init[k];
    a.smallest_dim = _smallest_dim, a.ob_diameter = _ob_diameter;
    a.collision_thres = cfg->getf("collision_thres"), a.non_touch_dist = cfg->getf("non_touch_dist");
    a.collision_finger_dist = cfg->getf("collision_finger_dist");
    a.collision_finger_volume_ratio = cfg->getf("collision_finger_volume_ratio");
    a.voxel_size = 0.005f;
    printf("collision_dist=%f, non_touch_dist=%f\n", std::min(-_smallest_dim * a.collision_thres, -0.007f), a.non_touch_dist);
    hop::check(hop_physics_set_frame(ctx_, &a), ctx_, "hop_physics_set_frame");
    hop::check(hop_reject_by_collision(ctx_, nullptr, nullptr, nullptr), ctx_, "hop_reject_by_collision");
  }

  // selectBest (PoseEstimator.cpp:465-502)
  void selectBest(PoseHypo& best_hypo) {
    hop_lcp_opts o{cfg->getf("lcp.dist"), cfg->getf("lcp.normal_angle"), -1};  // nn_mode < 0: by size (cell lists or brute force: same bits)
    float score = 0;
    int idx = 0;
    hop::check(hop_lcp_select_best(ctx_, &o, best_hypo._pose, &score, &idx), ctx_, "hop_lcp_select_best");
    best_hypo._lcp_score = score;
    // the reference copies the winning PoseHypo: _id is the id clusterPoses(..., assign_id = true) gave it
    std::vector<int> ids((size_t)std::max(numHypos(), 1));
    int n = 0;
    hop::check(hop_hypos_download(ctx_, nullptr, nullptr, ids.data(), (int)ids.size(), &n), ctx_, "hop_hypos_download");
    best_hypo._id = (idx >= 0 && idx < n) ? ids[idx] : idx;
    best_hypo.print();
  }

  int numHypos() const { return hop_hypos_count(ctx_); }
  hop_ctx* ctx() { return ctx_; }
  ConfigParser* cfg;
  hop_s4pcs_stats last_stats{};
  float _model_center_init[3] = {0, 0, 0}, _smallest_dim = 0, _ob_diameter = 0;

 private:
  hop_ctx* ctx_ = nullptr;
  std::vector<float> _model_xyz;
  int _n_model = 0;
  hop::Cloud _cloud_withouthand_raw;
  hop::Mesh _obj_mesh;
  std::vector<uint16_t> _depth_raw;
  int _H = 0, _W = 0;
  double _depth_unit = 0.001;
  float _K[9] = {0};
  std::map<std::string, int> _mesh_ids;
};
#endif
