// PoseEstimator.h -- host-side mirror of the reference's PoseEstimator<PointT> for the hot-path members
// (src/perception/include/PoseEstimator.h:12-49), forwarding to the C-ABI of include/hop.h.  PCL cloud types are
// replaced by SoA planes (hop::Cloud).  The two rejectBy* members are "next" rows and absent.
#ifndef HOP_HOST_POSEESTIMATOR_H_
#define HOP_HOST_POSEESTIMATOR_H_
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
inline void check(int rc, hop_ctx* c, const char* where) {
  if (rc != HOP_OK) throw std::runtime_error(std::string(where) + ": " + hop_strerror(rc) + " " + (c ? hop_last_error(c) : ""));
}
}  // namespace hop

class PoseEstimator {
 public:
  // PoseEstimator(cfg, model, model001, K) (PoseEstimator.cpp:8-23)
  PoseEstimator(ConfigParser* cfg1, const hop::Cloud& model, const hop::Cloud& model001, int device = 0) : cfg(cfg1) {
    hop::check(hop_ctx_create(device, &ctx_), nullptr, "hop_ctx_create");
    hop::check(hop_set_model(ctx_, HOP_MODEL_5MM, model.xyz.data(), model.nrm.data(), model.n), ctx_, "hop_set_model(5mm)");
    hop::check(hop_set_model(ctx_, HOP_MODEL_1MM, model001.xyz.data(), model001.nrm.data(), model001.n), ctx_, "hop_set_model(1mm)");
  }
  ~PoseEstimator() { hop_ctx_destroy(ctx_); }
  PoseEstimator(const PoseEstimator&) = delete;
  PoseEstimator& operator=(const PoseEstimator&) = delete;

  // setCurScene (PoseEstimator.cpp:32-46): keeps points with confidence >= pose_estimator_high_confidence_thres
  void setCurScene(const hop::Cloud& object_segment) {
    hop::check(hop_set_scene(ctx_, object_segment.xyz.data(), object_segment.nrm.data(),
                             object_segment.conf.empty() ? nullptr : object_segment.conf.data(), object_segment.n,
                             cfg->pose_estimator_high_confidence_thres),
               ctx_, "hop_set_scene");
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
    hop_icp_opts o{10, cfg->getf("icp_angle_thres"), cfg->getf("icp_dist_thres"), 100, 3};  // nn_mode 3: NN cell lists, fused search + accumulation
    hop::check(hop_icp_refine(ctx_, &o, nullptr, nullptr), ctx_, "hop_icp_refine");
  }

  // selectBest (PoseEstimator.cpp:465-502)
  void selectBest(PoseHypo& best_hypo) {
    hop_lcp_opts o{cfg->getf("lcp.dist"), cfg->getf("lcp.normal_angle"), -1};  // nn_mode < 0: by size (cell lists or brute force: same bits)
    float score = 0;
    int idx = 0;
    hop::check(hop_lcp_select_best(ctx_, &o, best_hypo._pose, &score, &idx), ctx_, "hop_lcp_select_best");
    best_hypo._lcp_score = score;
    best_hypo._id = idx;
    best_hypo.print();
  }

  int numHypos() const { return hop_hypos_count(ctx_); }
  hop_ctx* ctx() { return ctx_; }
  ConfigParser* cfg;
  hop_s4pcs_stats last_stats{};

 private:
  hop_ctx* ctx_ = nullptr;
};
#endif
