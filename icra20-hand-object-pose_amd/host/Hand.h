// Hand.h -- host-side mirror of the reference's Hand / HandT42 for the hand-state search
// (src/perception/include/Hand.h:29-92; Hand.cpp:182-250 FingerProperty, :505-523 getTFHandBase,
// :587-600 initPSO, :603-672 matchOneComponentPSO), forwarding to the C-ABI.  The URDF/mesh loading of
// Hand::parseURDF (Hand.cpp:375-502) is replaced by addComponent() calls with link clouds already at 5 mm.
#ifndef HOP_HOST_HAND_H_
#define HOP_HOST_HAND_H_
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "PoseEstimator.h"

struct Mat4 {
  float m[16];
  static Mat4 Identity() {
    Mat4 r;
    for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.f : 0.f;
    return r;
  }
};
inline Mat4 operator*(const Mat4& a, const Mat4& b) {
  Mat4 r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += a.m[4 * i + k] * b.m[4 * k + j];
      r.m[4 * i + j] = s;
    }
  return r;
}
inline void mulPoint(const Mat4& T, const float v[4], float out[4]) {
  for (int i = 0; i < 4; ++i) out[i] = ((T.m[4 * i] * v[0] + T.m[4 * i + 1] * v[1]) + T.m[4 * i + 2] * v[2]) + T.m[4 * i + 3] * v[3];
}

// FingerProperty (Hand.h:8-21, Hand.cpp:182-250)
class FingerProperty {
 public:
  float _min_x = 0, _min_y = 0, _min_z = 0, _max_x = 0, _max_y = 0, _max_z = 0;
  float _stride_z = 0;
  std::vector<float> _hist_alongz;  // 6 x N, row-major
  int _num_division = 0;
  FingerProperty() {}
  FingerProperty(const hop::Cloud& model, int num_division) : _num_division(num_division) {
    const int n = model.n;
    const float *x = model.xyz.data(), *y = x + n, *z = y + n;
    _min_x = _min_y = _min_z = FLT_MAX;
    _max_x = _max_y = _max_z = -FLT_MAX;
    for (int i = 0; i < n; ++i) {
      _min_x = std::min(_min_x, x[i]), _min_y = std::min(_min_y, y[i]), _min_z = std::min(_min_z, z[i]);
      _max_x = std::max(_max_x, x[i]), _max_y = std::max(_max_y, y[i]), _max_z = std::max(_max_z, z[i]);
    }
    _stride_z = (_max_z - _min_z) / num_division;
    _hist_alongz.assign(6 * (size_t)num_division, 0.f);
    for (int c = 0; c < num_division; ++c)
      for (int r = 0; r < 6; ++r) H(r, c) = r < 3 ? FLT_MAX : -FLT_MAX;
    std::vector<bool> changed(num_division, false);
    for (int i = 0; i < n; ++i) {
      const int bin = getBinAlongZ(z[i]);
      H(0, bin) = std::min(H(0, bin), x[i]), H(1, bin) = std::min(H(1, bin), y[i]), H(2, bin) = std::min(H(2, bin), z[i]);
      H(3, bin) = std::max(H(3, bin), x[i]), H(4, bin) = std::max(H(4, bin), y[i]), H(5, bin) = std::max(H(5, bin), z[i]);
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
          break;
        }
  }
  float& H(int r, int c) { return _hist_alongz[(size_t)r * _num_division + c]; }
  int getBinAlongZ(float z) const {
    int bin = (int)(std::max(z - _min_z, 0.0f) / _stride_z);
    bin = std::max(bin, 0);
    return std::min(bin, _num_division - 1);
  }
};

class Hand {
 public:
  typedef Mat4 Mat;
  Hand(ConfigParser* cfg1, hop_ctx* ctx) : _handbase_in_cam(Mat4::Identity()), cfg(cfg1), ctx_(ctx) { initPSO(); }
  virtual ~Hand() {}

  // Hand::addComponent (Hand.cpp:525-534); `cloud` already down-sampled to 5 mm, in the link frame
  void addComponent(const std::string& name, const std::string& parent_name, const hop::Cloud& cloud, const Mat4& tf_in_parent) {
    _clouds[name] = cloud;
    _parent_names[name] = parent_name;
    _tf_in_parent[name] = tf_in_parent;
    _tf_self[name] = Mat4::Identity();
    _component_status[name] = false;
    if (name.find("finger") != std::string::npos) _finger_properties[name] = FingerProperty(cloud, 10);
  }

  // Hand::reset (Hand.cpp:336-361): the per-frame state goes back to its initial values, the model stays
  void reset() {
    for (auto& c : _component_status) c.second = false;
    for (auto& h : _finger_angles) h.second = 0;
    _handbase_in_cam = Mat4::Identity();
    _hand_cloud = hop::Cloud();
    _hand_clouds.clear();
    for (auto& t : _tf_self) t.second = Mat4::Identity();
  }

  // the convex mesh of a component (Hand.cpp:526-530, loaded from an OBJ file there), link frame
  void addConvexMesh(const std::string& name, const hop::Mesh& mesh) { _convex_meshes[name] = mesh; }
  // the visual mesh of a component (Hand.cpp:529, hand->_meshes: what rejectByRender draws), link frame
  void addMesh(const std::string& name, const hop::Mesh& mesh) { _meshes[name] = mesh; }

  // products of Hand::setCurScene (Hand.cpp:327-332), hand-base frame
  void setCurScene(const hop::Cloud& scene_hand_region_removed_noise, const hop::Cloud& scene_hand_region,
                   const hop::Cloud& scene_remove_swivel) {
    hop::check(hop_hand_set_scene(ctx_, scene_hand_region_removed_noise.xyz.data(), scene_hand_region_removed_noise.n,
                                  scene_hand_region.nrm.data(), scene_hand_region.n, scene_remove_swivel.xyz.data(),
                                  scene_remove_swivel.n),
               ctx_, "hop_hand_set_scene");
    _n_removed_noise = scene_hand_region_removed_noise.n, _n_remove_swivel = scene_remove_swivel.n;
  }
  // (points of scene_hand_region_removed_noise, of scene_remove_swivel) of the last setCurScene
  std::pair<int, int> sceneSizes() const { return {_n_removed_noise, _n_remove_swivel}; }

  // Hand::getTFHandBase (Hand.cpp:505-523)
  void getTFHandBase(std::string cur_name, Mat4& tf_in_handbase) {
    tf_in_handbase = Mat4::Identity();
    while (cur_name != "base_link") {
      if (_tf_self.find(cur_name) == _tf_self.end()) throw std::runtime_error("cur_name does not exist: " + cur_name);
      tf_in_handbase = _tf_in_parent[cur_name] * _tf_self[cur_name] * tf_in_handbase;
      cur_name = _parent_names[cur_name];
    }
  }

  // Hand::makeHandCloud (Hand.cpp:537-556): every component cloud in the hand-base frame at the current finger state
  // (the reference keeps one kd-tree per component in a std::map, i.e. in name order)
  void makeHandCloud() {
    _hand_clouds.clear();
    for (auto& h : _clouds) {
      Mat4 T = Mat4::Identity();
      if (h.first != "base_link") getTFHandBase(h.first, T);
      const hop::Cloud& src = h.second;
      hop::Cloud dst;
      dst.n = src.n;
      dst.xyz.resize(3 * (size_t)src.n);
      dst.nrm.resize(3 * (size_t)src.n);
      const float *x = src.xyz.data(), *y = x + src.n, *z = y + src.n;
      const float *nx = src.nrm.data(), *ny = nx + src.n, *nz = ny + src.n;
      for (int k = 0; k < 3; ++k)
        for (int i = 0; i < src.n; ++i) {
          dst.xyz[(size_t)k * src.n + i] = ((T.m[4 * k] * x[i] + T.m[4 * k + 1] * y[i]) + T.m[4 * k + 2] * z[i]) + T.m[4 * k + 3];
          dst.nrm[(size_t)k * src.n + i] = (T.m[4 * k] * nx[i] + T.m[4 * k + 1] * ny[i]) + T.m[4 * k + 2] * nz[i];
        }
      _hand_clouds[h.first] = dst;
    }
    // _hand_cloud: the components appended in map order (Hand.cpp:550)
    int total = 0;
    for (auto& h : _hand_clouds) total += h.second.n;
    _hand_cloud.n = total;
    _hand_cloud.xyz.assign(3 * (size_t)total, 0.f);
    _hand_cloud.nrm.assign(3 * (size_t)total, 0.f);
    int off = 0;
    for (auto& h : _hand_clouds) {
      for (int k = 0; k < 3; ++k) {
        std::copy(h.second.xyz.begin() + (size_t)k * h.second.n, h.second.xyz.begin() + (size_t)(k + 1) * h.second.n,
                  _hand_cloud.xyz.begin() + (size_t)k * total + off);
        std::copy(h.second.nrm.begin() + (size_t)k * h.second.n, h.second.nrm.begin() + (size_t)(k + 1) * h.second.n,
                  _hand_cloud.nrm.begin() + (size_t)k * total + off);
      }
      off += h.second.n;
    }
  }
  const hop::Cloud& handCloud() {
    if (_hand_clouds.empty()) makeHandCloud();
    return _hand_cloud;
  }
  // hand->_handbase_in_cam.inverse() (PoseEstimator.cpp:554,566): inverse of an affine matrix, adjugate in double
  void camToHandbase(float out[16]) const { affineInverse(_handbase_in_cam.m, out); }
  static void affineInverse(const float* a, float out[16]) {
    double d[16];
    affineInverseD(a, d);
    for (int i = 0; i < 16; ++i) out[i] = (float)d[i];
  }
  static void affineInverseD(const float* a, double out[16]) {
    double m[3][3], inv[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) m[i][j] = a[4 * i + j];
    const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                       m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    inv[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) / det, inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det;
    inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det, inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) / det;
    inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det, inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det;
    inv[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) / det, inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det;
    inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
    for (int i = 0; i < 16; ++i) out[i] = (i == 15) ? 1.0 : 0.0;
    for (int i = 0; i < 3; ++i) {
      double t = 0;
      for (int j = 0; j < 3; ++j) out[4 * i + j] = inv[i][j], t -= inv[i][j] * (double)a[4 * j + 3];
      out[4 * i + 3] = t;
    }
  }

  // ---- Hand::setCurScene from the clouds of the frame (Hand.cpp:279-334), camera frame in, products to the device
  // rows of a cloud selected by a byte mask
  static hop::Cloud selectRows(const std::vector<float>& xyz, const std::vector<float>& nrm, int n, const std::vector<unsigned char>& keep) {
    hop::Cloud r;
    for (int i = 0; i < n; ++i) r.n += keep[i] ? 1 : 0;
    r.xyz.resize(3 * (size_t)r.n), r.nrm.resize(3 * (size_t)r.n);
    int o = 0;
    for (int i = 0; i < n; ++i) {
      if (!keep[i]) continue;
      for (int k = 0; k < 3; ++k) r.xyz[(size_t)k * r.n + o] = xyz[(size_t)k * n + i], r.nrm[(size_t)k * r.n + o] = nrm[(size_t)k * n + i];
      ++o;
    }
    return r;
  }
  // Hand::handbaseICP (Hand.cpp:677-777): scene_organized with its normals, camera frame; corrects _handbase_in_cam
  void handbaseICP(const hop::Cloud& scene_organized, const hop::Cloud& base_link_cloud) {
    float cam_in_handbase[16];
    camToHandbase(cam_in_handbase);
    std::vector<float> sx(3 * (size_t)scene_organized.n), sn(3 * (size_t)scene_organized.n);
    int m = 0;
    hop::check(hop_voxel_downsample_normals(ctx_, scene_organized.xyz.data(), scene_organized.nrm.data(), scene_organized.n, 0.005f, sx.data(), sn.data(),
                                            scene_organized.n, &m),
               ctx_, "hop_voxel_downsample_normals");
    std::vector<float> vx(3 * (size_t)m), vn(3 * (size_t)m);
    for (int k = 0; k < 3; ++k)
      for (int i = 0; i < m; ++i) vx[(size_t)k * m + i] = sx[(size_t)k * scene_organized.n + i], vn[(size_t)k * m + i] = sn[(size_t)k * scene_organized.n + i];
    const Mat4 &t1 = _tf_in_parent["finger_1_1"], &t2 = _tf_in_parent["finger_2_1"];
    std::vector<float> hx(3 * (size_t)m), hn(3 * (size_t)m);
    std::vector<unsigned char> keep(m);
    hop::check(hop_handbase_region(ctx_, vx.data(), vn.data(), m, cam_in_handbase, t1.m[7], t1.m[11], t2.m[7], t2.m[11], hx.data(), hn.data(), keep.data()), ctx_,
               "hop_handbase_region");
    const hop::Cloud src = selectRows(hx, hn, m, keep);
    Mat4 offset = Mat4::Identity();
    if (src.n > 0) {  // Utils::runICP(scene_handbase, handbase, offset, 50, 30, 0.03, 1e-4), :734
      // a context of its own if the application gave one (setHandbaseIcpContext): the object's models and their NN lists on the
      // PoseEstimator's context then survive the frame; the hand-base cloud is uploaded there once
      hop_ctx* ic = icp_ctx_ ? icp_ctx_ : ctx_;
      hop::check(hop_set_scene(ic, src.xyz.data(), src.nrm.data(), nullptr, src.n, 0.f), ic, "hop_set_scene");
      if (!icp_ctx_ || !icp_model_set_) {
        hop::check(hop_set_model(ic, HOP_MODEL_5MM, base_link_cloud.xyz.data(), base_link_cloud.nrm.data(), base_link_cloud.n), ic, "hop_set_model");
        icp_model_set_ = icp_ctx_ != nullptr;
      }
      const Mat4 I = Mat4::Identity();
      hop::check(hop_hypos_upload(ic, I.m, nullptr, 1), ic, "hop_hypos_upload");
      hop_icp_opts o{50, 30.f, 0.03f, 0, hop::icp_nn_mode_reference()};  // nn_mode 7: Utils::runICP's own minimiser (Levenberg-Marquardt), the oracle's bits
      hop::icp_refine_reference(ic, o, "hop_icp_refine(handbaseICP)");
      Mat4 pose;
      int got = 0;
      hop::check(hop_hypos_download(ic, pose.m, nullptr, nullptr, 1, &got), ic, "hop_hypos_download");
      affineInverse(pose.m, offset.m);  // offset = refined_pose.inverse(): source -> target
    }
    const float translation = std::sqrt(offset.m[3] * offset.m[3] + offset.m[7] * offset.m[7] + offset.m[11] * offset.m[11]);
    if (translation >= 0.05) offset = Mat4::Identity();  // :740-745
    const double tr = (double)offset.m[0] + offset.m[5] + offset.m[10];
    const float rot_diff = (float)(std::acos(std::max(-1.0, std::min(1.0, (tr - 1) / 2.0))) / M_PI * 180.0);
    // R.eulerAngles(2,1,0)(1) (Eigen Geometry/EulerAngles.h:36-108)
    const float a0 = std::atan2(offset.m[4], offset.m[0]);
    const float c2 = std::sqrt(offset.m[10] * offset.m[10] + offset.m[9] * offset.m[9]);
    float pitch = a0 < 0.f ? std::atan2(-offset.m[8], -c2) : std::atan2(-offset.m[8], c2);
    pitch = std::min(std::abs(pitch), std::abs(static_cast<float>(M_PI) - pitch));
    pitch = std::min(std::abs(pitch), std::abs(static_cast<float>(M_PI) + pitch));
    if (rot_diff >= 10 || std::abs(pitch) >= 10 / 180.0 * M_PI) offset = Mat4::Identity();  // :752-756
    bool is_identity = true;
    for (int i = 0; i < 16; ++i) is_identity = is_identity && offset.m[i] == ((i % 5 == 0) ? 1.f : 0.f);
    if (!is_identity) _component_status["handbase"] = true;
    // handbase_in_cam * offset.inverse() (:771) with the inverse and the product in double, rounded once (the Python mirror does the same)
    double oi[16], prod[16];
    affineInverseD(offset.m, oi);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double acc = 0;
        for (int k = 0; k < 4; ++k) acc += (double)_handbase_in_cam.m[4 * i + k] * oi[4 * k + j];
        prod[4 * i + j] = acc;
      }
    for (int i = 0; i < 16; ++i) _handbase_in_cam.m[i] = (float)prod[i];
  }
  // HandT42::adjustHandHeight (Hand.cpp:999-1051); scene_hand_region: the 3 mm hand-region cloud, camera frame, with normals
  void adjustHandHeight(const hop::Cloud& scene_hand_region) {
    makeHandCloud();
    if (_component_status["handbase"]) return;
    float T[16];
    camToHandbase(T);
    const int n = scene_hand_region.n;
    std::vector<float> sx(3 * (size_t)n), sn(3 * (size_t)n);
    const float *x = scene_hand_region.xyz.data(), *y = x + n, *z = y + n;
    const float *nx = scene_hand_region.nrm.data(), *ny = nx + n, *nz = ny + n;
    for (int k = 0; k < 3; ++k)
      for (int i = 0; i < n; ++i) {
        sx[(size_t)k * n + i] = ((T[4 * k] * x[i] + T[4 * k + 1] * y[i]) + T[4 * k + 2] * z[i]) + T[4 * k + 3];
        sn[(size_t)k * n + i] = (T[4 * k] * nx[i] + T[4 * k + 1] * ny[i]) + T[4 * k + 2] * nz[i];
      }
    const float trial_heights[13] = {-0.03f, -0.025f, -0.02f, -0.015f, -0.01f, -0.005f, 0.f, 0.005f, 0.01f, 0.015f, 0.02f, 0.025f, 0.03f};
    int counts[13];
    hop::check(hop_hand_height_matches(ctx_, sx.data(), sn.data(), n, _hand_cloud.xyz.data(), _hand_cloud.nrm.data(), _hand_cloud.n, trial_heights, 13, counts),
               ctx_, "hop_hand_height_matches");
    int max_match = 0;
    Mat4 best_offset = Mat4::Identity();
    for (int i = 0; i < 13; ++i)
      if (counts[i] > max_match) {  // :1042-1047
        max_match = counts[i];
        best_offset = Mat4::Identity();
        best_offset.m[11] = trial_heights[i];
      }
    _handbase_in_cam = _handbase_in_cam * best_offset;  // :1050
  }
  // the filters of Hand::setCurScene (Hand.cpp:289-332) on the 3 mm hand-region cloud (camera frame, with normals)
  void setCurSceneFromRegion(const hop::Cloud& scene_hand_region) {
    float cam_in_handbase[16];
    camToHandbase(cam_in_handbase);
    const int n = scene_hand_region.n;
    std::vector<float> hx(3 * (size_t)n), hn(3 * (size_t)n);
    std::vector<unsigned char> keep(n), swivel(n);
    hop::check(hop_hand_scene_filters(ctx_, scene_hand_region.xyz.data(), scene_hand_region.nrm.data(), n, cam_in_handbase, hx.data(), hn.data(), keep.data(),
                                      swivel.data()),
               ctx_, "hop_hand_scene_filters");
    hop::Cloud in_handbase;
    in_handbase.n = n, in_handbase.xyz = hx, in_handbase.nrm = hn;
    setCurScene(selectRows(hx, hn, n, keep), in_handbase, selectRows(hx, hn, n, swivel));
  }

  // HandT42::removeSurroundingPointsAndAssignProbability (Hand.cpp:779-888); dist_thres is the SQUARED near_hand_dist,
  // as at the call site (main_realdata_auto.cpp:147-148).  Survivors in input order, camera frame, with confidence.
  hop::Cloud removeSurroundingPointsAndAssignProbability(const hop::Cloud& scene, const Mat4& handbase_in_cam, float dist_thres) {
    if (_hand_clouds.empty()) makeHandCloud();
    std::vector<hop_hand_link> links;
    for (auto& h : _hand_clouds) {
      float local = dist_thres;  // Hand.cpp:812-821
      if (h.first == "finger_2_1" || h.first == "finger_1_1") local = (float)(0.005 * 0.005);
      else if (h.first == "base" || h.first == "swivel_1" || h.first == "swivel_2") local = (float)(0.02 * 0.02);
      links.push_back(hop_hand_link{h.second.xyz.data(), h.second.n, local});
    }
    Mat4 f12, f22;
    getTFHandBase("finger_1_2", f12);
    getTFHandBase("finger_2_2", f22);
    hop::Cloud out;
    out.xyz.resize(3 * (size_t)scene.n), out.nrm.resize(3 * (size_t)scene.n), out.conf.resize(scene.n);
    int kept = 0;
    hop::check(hop_hand_remove_surrounding(ctx_, scene.xyz.data(), scene.nrm.data(), scene.n, handbase_in_cam.m, links.data(), (int)links.size(),
                                           f12.m, f22.m, _finger_properties["finger_1_2"]._min_z, out.xyz.data(), out.nrm.data(),
                                           out.conf.data(), nullptr, &kept),
               ctx_, "hop_hand_remove_surrounding");
    // compact the planes from stride scene.n to stride kept
    hop::Cloud r;
    r.n = kept;
    r.xyz.resize(3 * (size_t)kept), r.nrm.resize(3 * (size_t)kept), r.conf.assign(out.conf.begin(), out.conf.begin() + kept);
    for (int k = 0; k < 3; ++k)
      for (int i = 0; i < kept; ++i) {
        r.xyz[(size_t)k * kept + i] = out.xyz[(size_t)k * scene.n + i];
        r.nrm[(size_t)k * kept + i] = out.nrm[(size_t)k * scene.n + i];
      }
    return r;
  }

  void initPSO() {  // Hand.cpp:587-600
    hop_pso_default_settings(&_pso_settings);
    _pso_settings.n_pop = cfg->geti("hand_match.pso.n_pop");
    _pso_settings.n_gen = cfg->geti("hand_match.pso.n_gen");
    _pso_settings.check_freq = cfg->geti("hand_match.pso.check_freq");
    _pso_settings.err_tol = 1e-5;
    _pso_settings.c_cog = cfg->getf("hand_match.pso.pso_par_c_cog");
    _pso_settings.c_soc = cfg->getf("hand_match.pso.pso_par_c_soc");
    _pso_settings.initial_w = cfg->getf("hand_match.pso.pso_par_initial_w");
  }

  // Hand::matchOneComponentPSO (Hand.cpp:603-672).  use_normal / normal_angle_thres are ignored by the
  // reference body (thresholds come from the YAML, Hand.cpp:76-83); kept for signature compatibility.
  bool matchOneComponentPSO(std::string model_name, float min_angle, float max_angle, bool /*use_normal*/, float dist_thres,
                            float /*normal_angle_thres*/, float least_match) {
    _pso_settings.upper_rad = max_angle * M_PI / 180;
    _pso_settings.lower_rad = min_angle * M_PI / 180;
    std::map<std::string, std::string> pair_names{{"finger_1_1", "finger_2_1"}, {"finger_2_1", "finger_1_1"},
                                                  {"finger_1_2", "finger_2_2"}, {"finger_2_2", "finger_1_2"}};
    hop_finger_args a;
    std::memset(&a, 0, sizeof(a));
    const std::string pair_name = pair_names[model_name];
    Mat4 pair_in_base;
    auto tip = [&](const std::string& n, bool max_z, float out[4]) {
      const FingerProperty& p = _finger_properties[n];
      out[0] = p._min_x, out[1] = p._max_y, out[2] = max_z ? p._max_z : p._min_z, out[3] = 1.f;
    };
    Mat4 finger_out2parent = Mat4::Identity();
    FingerProperty finger_out_property = _finger_properties[model_name];
    float t[4];
    if (pair_name == "finger_1_1" || pair_name == "finger_2_1") {
      const std::string pair_out = pair_name == "finger_1_1" ? "finger_1_2" : "finger_2_2";
      tip(pair_out, false, t);
      getTFHandBase(pair_out, pair_in_base);
      mulPoint(pair_in_base, t, a.pair_tip1);
      getTFHandBase(pair_name, pair_in_base);
      tip(pair_name, false, t);
      mulPoint(pair_in_base, t, a.pair_tip2);
      const std::string out_name = model_name == "finger_1_1" ? "finger_1_2" : "finger_2_2";
      finger_out2parent = _tf_in_parent[out_name];
      finger_out_property = _finger_properties[out_name];
    } else {
      getTFHandBase(pair_name, pair_in_base);
      tip(pair_name, false, t);
      mulPoint(pair_in_base, t, a.pair_tip1);
      tip(pair_name, true, t);
      mulPoint(pair_in_base, t, a.pair_tip2);
    }
    const FingerProperty& fp = _finger_properties[model_name];
    a.fp_min[0] = fp._min_x, a.fp_min[1] = fp._min_y, a.fp_min[2] = fp._min_z;
    a.fp_max[0] = fp._max_x, a.fp_max[1] = fp._max_y, a.fp_max[2] = fp._max_z;
    a.fo_min[0] = finger_out_property._min_x, a.fo_min[1] = finger_out_property._min_y, a.fo_min[2] = finger_out_property._min_z;
    a.fo_max[0] = finger_out_property._max_x, a.fo_max[1] = finger_out_property._max_y, a.fo_max[2] = finger_out_property._max_z;
    a.fp_stride_z = fp._stride_z;
    a.fp_num_division = fp._num_division;
    a.fp_hist_min_y = fp._hist_alongz.data() + (size_t)1 * fp._num_division;
    Mat4 model2handbase;
    getTFHandBase(model_name, model2handbase);
    std::memcpy(a.model2handbase, model2handbase.m, sizeof(float) * 16);
    std::memcpy(a.finger_out2parent, finger_out2parent.m, sizeof(float) * 16);
    a.is_palm_side = (model_name == "finger_1_1" || model_name == "finger_2_1");
    a.is_right_side = (model_name == "finger_2_1" || model_name == "finger_2_2");
    a.gripper_min_dist = cfg->gripper_min_dist;
    a.dist_thres = dist_thres;
    const float ang = cfg->getf(a.is_palm_side ? "hand_match.finger1_normal_angle" : "hand_match.finger2_normal_angle");
    a.cos_normal_thres = (float)std::cos(ang / 180.0 * M_PI);
    a.check_normal = cfg->getb("hand_match.check_normal");
    a.max_outter_pts = cfg->geti("hand_match.max_outter_pts");
    a.outter_pt_dist = cfg->getf("hand_match.outter_pt_dist");
    a.outter_pt_dist_weight = cfg->getf("hand_match.outter_pt_dist_weight");
    const hop::Cloud& cl = _clouds[model_name];
    a.model_xyz = cl.xyz.data(), a.model_nrm = cl.nrm.data(), a.n_model = cl.n;
    hop::check(hop_hand_set_finger(ctx_, &a), ctx_, "hop_hand_set_finger");
    double angle = 0, objval = 0;
    hop::check(hop_hand_pso_search(ctx_, &_pso_settings, &angle, &objval), ctx_, "hop_hand_pso_search");
    if (-objval <= least_match) {
      std::printf("%s PSO matching failed\n", model_name.c_str());
      _tf_self[model_name] = Mat4::Identity();
      _component_status[model_name] = false;
      return false;
    }
    const float af = static_cast<float>(angle);
    Mat4 R = Mat4::Identity();
    R.m[5] = std::cos(af), R.m[6] = -std::sin(af), R.m[9] = std::sin(af), R.m[10] = std::cos(af);
    _tf_self[model_name] = R;
    _component_status[model_name] = true;
    _finger_angles[model_name] = af;
    std::printf("%s PSO final angle=%f, match_score=%f\n", model_name.c_str(), af, -objval);
    return true;
  }

  std::map<std::string, hop::Cloud> _clouds;
  std::map<std::string, std::string> _parent_names;
  std::map<std::string, Mat4> _tf_in_parent, _tf_self;
  std::map<std::string, FingerProperty> _finger_properties;
  std::map<std::string, bool> _component_status;
  std::map<std::string, float> _finger_angles;
  std::map<std::string, hop::Cloud> _hand_clouds;  // Hand::makeHandCloud products (hand-base frame)
  hop::Cloud _hand_cloud;                          // their concatenation
  std::map<std::string, hop::Mesh> _convex_meshes, _meshes;
  Mat4 _handbase_in_cam;
  ConfigParser* cfg;
  hop_pso_settings _pso_settings;

  // optional second context for handbaseICP's Utils::runICP call (see there)
  void setHandbaseIcpContext(hop_ctx* c) { icp_ctx_ = c, icp_model_set_ = false; }

 protected:
  hop_ctx* ctx_;
  hop_ctx* icp_ctx_ = nullptr;
  bool icp_model_set_ = false;
  int _n_removed_noise = 0, _n_remove_swivel = 0;
};

class HandT42 : public Hand {
 public:
  HandT42(ConfigParser* cfg1, hop_ctx* ctx) : Hand(cfg1, ctx) {}
};
#endif
