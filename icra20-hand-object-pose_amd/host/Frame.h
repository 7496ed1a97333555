// Frame.h -- one frame of the reference's drivers FROM THE DEPTH IMAGE, in C++ above the C-ABI:
// src/perception/src/app/main_realdata_auto.cpp:54-205 and the loop body of run_real_all.cpp:100-262
// (depth image -> organised cloud + integral-image normals -> 1 mm grid + hand-base crop -> Hand::setCurScene with handbaseICP ->
// finger PSO -> adjustHandHeight -> hand-point removal -> MLS normals -> generator cloud -> runSuper4pcs -> clusterPoses ->
// refineByICP -> clusterPoses -> rejectByCollisionOrNonTouching -> rejectByRender -> selectBest).  The same call order as the Python
// mirror hop_amd.run_real_all.process_frame; every cloud operation is a libhop call.
//
// Files: 16-bit PNG depth in millimetres (Utils::readDepthImage, Utils.cpp:36-55), 4 x 4 pose text files
// (Utils::parsePoseTxt, Utils.cpp:516-543) -- read here without OpenCV / libpng: read_png16 is a minimal decoder on zlib.
#ifndef HOP_HOST_FRAME_H_
#define HOP_HOST_FRAME_H_
#include <zlib.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <sstream>

#include "Hand.h"

namespace hop {

// ---- files --------------------------------------------------------------------------------------------------------------------------
// Greyscale PNG, 8 or 16 bits, non-interlaced (what the reference's depthN.png are): filters 0-4, big-endian samples.
inline void read_png16(const std::string& path, std::vector<uint16_t>& raw, int& H, int& W) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (buf.size() < 8 || std::memcmp(buf.data(), sig, 8) != 0) throw std::runtime_error("not a PNG: " + path);
  auto be32 = [&](size_t o) { return ((uint32_t)buf[o] << 24) | ((uint32_t)buf[o + 1] << 16) | ((uint32_t)buf[o + 2] << 8) | buf[o + 3]; };
  size_t pos = 8;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<unsigned char> idat;
  W = H = 0;
  while (pos + 12 <= buf.size()) {
    const uint32_t len = be32(pos);
    const std::string type(reinterpret_cast<const char*>(&buf[pos + 4]), 4);
    const size_t data = pos + 8;
    if (data + len + 4 > buf.size()) throw std::runtime_error("truncated PNG: " + path);
    if (type == "IHDR") {
      if (len != 13) throw std::runtime_error("PNG with an IHDR chunk of " + std::to_string(len) + " bytes (13 expected): " + path);
      const uint32_t w32 = be32(data), h32 = be32(data + 4);
      // (bounded before anything is sized from them: a corrupt header must not turn into a multi-gigabyte allocation)
      if (w32 == 0 || h32 == 0 || w32 > 16384u || h32 > 16384u) throw std::runtime_error("PNG with implausible dimensions " + std::to_string(w32) + " x " + std::to_string(h32) + ": " + path);
      W = (int)w32, H = (int)h32;
      depth = buf[data + 8], ctype = buf[data + 9], interlace = buf[data + 12];
    } else if (type == "IDAT")
      idat.insert(idat.end(), buf.begin() + data, buf.begin() + data + len);
    else if (type == "IEND")
      break;
    pos = data + len + 4;
  }
  if (ctype != 0 || (depth != 16 && depth != 8) || interlace != 0 || W <= 0 || H <= 0)
    throw std::runtime_error("unsupported PNG (need non-interlaced greyscale, 8 or 16 bit): " + path);
  const int bpp = depth / 8;
  const size_t stride = (size_t)W * bpp;
  std::vector<unsigned char> px((stride + 1) * (size_t)H);
  uLongf out_len = (uLongf)px.size();
  if (uncompress(px.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != px.size()) throw std::runtime_error("PNG inflate failed: " + path);
  std::vector<unsigned char> prev(stride, 0), cur(stride);
  raw.assign((size_t)H * W, 0);
  for (int r = 0; r < H; ++r) {
    const unsigned char* line = px.data() + (stride + 1) * (size_t)r;
    const int ft = line[0];
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
      int pred = 0;
      if (ft == 1) pred = a;
      else if (ft == 2) pred = b;
      else if (ft == 3) pred = (a + b) / 2;
      else if (ft == 4) {
        const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
        pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
      } else if (ft != 0)
        throw std::runtime_error("bad PNG filter: " + path);
      cur[i] = (unsigned char)(line[1 + i] + pred);
    }
    for (int c = 0; c < W; ++c) raw[(size_t)r * W + c] = bpp == 2 ? (uint16_t)((cur[2 * c] << 8) | cur[2 * c + 1]) : cur[c];
    prev.swap(cur);
  }
}

// Utils::parsePoseTxt (Utils.cpp:516-543): the first 16 blank-separated numbers, row-major
inline Mat4 parse_pose_txt(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open " + path);
  Mat4 T = Mat4::Identity();
  double v;
  int k = 0;
  while (k < 16 && (f >> v)) T.m[k++] = (float)v;
  if (k < 16) throw std::runtime_error("pose file with fewer than 16 numbers: " + path);
  return T;
}

inline Cloud read_cloud_bin(const std::string& path) {  // int32 n, int32 has_conf, xyz planes, normal planes, [conf]
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  int32_t n = 0, has_conf = 0;
  f.read(reinterpret_cast<char*>(&n), 4);
  f.read(reinterpret_cast<char*>(&has_conf), 4);
  Cloud c;
  c.n = n;
  c.xyz.resize(3 * (size_t)n);
  c.nrm.resize(3 * (size_t)n);
  f.read(reinterpret_cast<char*>(c.xyz.data()), sizeof(float) * 3 * (size_t)n);
  f.read(reinterpret_cast<char*>(c.nrm.data()), sizeof(float) * 3 * (size_t)n);
  if (has_conf) {
    c.conf.resize(n);
    f.read(reinterpret_cast<char*>(c.conf.data()), sizeof(float) * (size_t)n);
  }
  if (!f) throw std::runtime_error("short read " + path);
  return c;
}
inline Mesh read_mesh_bin(const std::string& path) {  // int32 nv, int32 nf, nv*3 float, nf*3 int32
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  int32_t nv = 0, nf = 0;
  f.read(reinterpret_cast<char*>(&nv), 4);
  f.read(reinterpret_cast<char*>(&nf), 4);
  Mesh m;
  m.V.resize(3 * (size_t)nv);
  m.F.resize(3 * (size_t)nf);
  f.read(reinterpret_cast<char*>(m.V.data()), sizeof(float) * m.V.size());
  f.read(reinterpret_cast<char*>(m.F.data()), sizeof(int32_t) * m.F.size());
  if (!f) throw std::runtime_error("short read " + path);
  return m;
}

// What the reference loads once per run (run_real_all.cpp:19-68): object clouds at 5 mm / 1 mm, object mesh, PPF key table, hand links
// with their clouds and meshes.  Directory layout = the frame directory of main_realdata_auto (model.bin, model001.bin, ppf_keys.bin,
// hand.txt, meshes.txt) -- the paper's PLY / OBJ / Boost archive / URDF files are download links.
struct Assets {
  Cloud model, model001, base_link;
  Mesh object_mesh;
  std::vector<int32_t> ppf_keys;
  struct Link {
    std::string name, parent;
    Cloud cloud;
    Mat4 tf_in_parent;
  };
  std::vector<Link> links;
  std::map<std::string, Mesh> link_meshes;

  explicit Assets(const std::string& dir0) {
    const std::string dir = dir0 + "/";
    model = read_cloud_bin(dir + "model.bin"), model001 = read_cloud_bin(dir + "model001.bin");
    {
      std::ifstream f(dir + "ppf_keys.bin", std::ios::binary);
      if (!f) throw std::runtime_error("cannot open " + dir + "ppf_keys.bin");
      int32_t n = 0;
      f.read(reinterpret_cast<char*>(&n), 4);
      ppf_keys.resize(4 * (size_t)n);
      f.read(reinterpret_cast<char*>(ppf_keys.data()), sizeof(int32_t) * 4 * (size_t)n);
    }
    std::ifstream fh(dir + "hand.txt");
    std::string line;
    while (std::getline(fh, line)) {
      std::istringstream ss(line);
      Link l;
      std::string file;
      if (!(ss >> l.name >> l.parent >> file)) continue;
      for (int i = 0; i < 16; ++i) ss >> l.tf_in_parent.m[i];
      l.cloud = read_cloud_bin(dir + file);
      links.push_back(l);
    }
    std::ifstream fb(dir + "base_link.bin", std::ios::binary);
    if (fb) base_link = read_cloud_bin(dir + "base_link.bin");
    std::ifstream fm(dir + "meshes.txt");
    std::string name, file;
    while (fm >> name >> file) {
      if (name == "object") object_mesh = read_mesh_bin(dir + file);
      else link_meshes[name] = read_mesh_bin(dir + file);
    }
  }
  void addTo(HandT42& hand) const {
    for (const auto& l : links) hand.addComponent(l.name, l.parent, l.cloud, l.tf_in_parent);
    for (const auto& m : link_meshes) {
      hand.addConvexMesh(m.first, m.second);
      hand.addMesh(m.first, m.second);
    }
  }
};

// ---- calibration (ConfigParser.cpp:44-115; run_real_all.cpp:113-114) -------------------------------------------------------------------------
inline void mat4d_mul(const double* a, const double* b, double* r) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j];
      r[4 * i + j] = s;
    }
}
inline void mat4d_affine_inverse(const double* a, double* r) {
  // general 3 x 3 inverse of the upper block (the calibration matrices are rigid up to rounding)
  const double m[9] = {a[0], a[1], a[2], a[4], a[5], a[6], a[8], a[9], a[10]};
  const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  const double inv[9] = {(m[4] * m[8] - m[5] * m[7]) / det, (m[2] * m[7] - m[1] * m[8]) / det, (m[1] * m[5] - m[2] * m[4]) / det,
                         (m[5] * m[6] - m[3] * m[8]) / det, (m[0] * m[8] - m[2] * m[6]) / det, (m[2] * m[3] - m[0] * m[5]) / det,
                         (m[3] * m[7] - m[4] * m[6]) / det, (m[1] * m[6] - m[0] * m[7]) / det, (m[0] * m[4] - m[1] * m[3]) / det};
  for (int i = 0; i < 16; ++i) r[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 3; ++i) {
    double t = 0;
    for (int j = 0; j < 3; ++j) {
      r[4 * i + j] = inv[3 * i + j];
      t -= inv[3 * i + j] * a[4 * j + 3];
    }
    r[4 * i + 3] = t;
  }
}
struct Calibration {
  float K9[9];
  double cam1_in_leftarm[16], handbase_in_palm[16];
  explicit Calibration(const ConfigParser& cfg) {
    const std::vector<float> k = cfg.getlist("cam_K");
    if (k.size() < 9) throw std::runtime_error("cam_K needs 9 numbers");
    for (int i = 0; i < 9; ++i) K9[i] = k[i];
    std::vector<float> d = cfg.has("cam1_in_leftarm") ? cfg.getlist("cam1_in_leftarm") : std::vector<float>{0, 0, 0, 0, 0, 0, 1};
    if (d.size() < 7) throw std::runtime_error("cam1_in_leftarm needs 7 numbers (x y z qx qy qz qw)");
    const double qn = std::sqrt((double)d[3] * d[3] + (double)d[4] * d[4] + (double)d[5] * d[5] + (double)d[6] * d[6]);
    const double x = d[3] / qn, y = d[4] / qn, z = d[5] / qn, w = d[6] / qn;
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    for (int i = 0; i < 16; ++i) cam1_in_leftarm[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) cam1_in_leftarm[4 * i + j] = (double)(float)R[3 * i + j];
      cam1_in_leftarm[4 * i + 3] = (double)(float)d[i];
    }
    for (int i = 0; i < 16; ++i) handbase_in_palm[i] = (i % 5 == 0) ? 1.0 : 0.0;
    if (cfg.has("handbase_in_palm")) {
      const std::vector<float> h = cfg.getlist("handbase_in_palm");
      if (h.size() < 16) throw std::runtime_error("handbase_in_palm needs 16 numbers");
      for (int i = 0; i < 16; ++i) handbase_in_palm[i] = h[i];
    }
  }
  // handbase_in_cam = cam1_in_leftarm^-1 * (leftarm_in_base^-1 * palm_in_baselink * handbase_in_palm)
  Mat4 handbaseInCam(const Mat4& leftarm_in_base, const Mat4& palm_in_baselink) const {
    double a[16], b[16], ai[16], t1[16], t2[16], ci[16], out[16];
    for (int i = 0; i < 16; ++i) a[i] = leftarm_in_base.m[i], b[i] = palm_in_baselink.m[i];
    mat4d_affine_inverse(a, ai);
    mat4d_mul(ai, b, t1);
    mat4d_mul(t1, handbase_in_palm, t2);
    mat4d_affine_inverse(cam1_in_leftarm, ci);
    mat4d_mul(ci, t2, out);
    Mat4 r;
    for (int i = 0; i < 16; ++i) r.m[i] = (float)out[i];
    return r;
  }
};

// ---- the frame -------------------------------------------------------------------------------------------------------------------------------
struct FrameInfo {
  int n_valid = 0, n_hand_region = 0, n_without_hand = 0, n_object_segment = 0, n_generated = 0, n_clusters = 0, n_after_icp = 0, n_after_physics = -1,
      n_after_render = -1;
  std::map<std::string, float> angles;
  Mat4 handbase_in_cam = Mat4::Identity();
  float score = 0;
  std::vector<std::pair<const char*, double>> stage_ms;  // host wall time per stage of process_frame, in order
};

inline Cloud compact(const std::vector<float>& x, const std::vector<float>& n, int stride, int m, const std::vector<float>* conf = nullptr) {
  Cloud c;
  c.n = m;
  c.xyz.resize(3 * (size_t)m), c.nrm.resize(3 * (size_t)m);
  for (int k = 0; k < 3; ++k)
    for (int i = 0; i < m; ++i) c.xyz[(size_t)k * m + i] = x[(size_t)k * stride + i], c.nrm[(size_t)k * m + i] = n[(size_t)k * stride + i];
  if (conf) c.conf.assign(conf->begin(), conf->begin() + m);
  return c;
}

// run_real_all.cpp:116-241 / main_realdata_auto.cpp:54-205 for one frame; returns model2scene (identity when no pose is found).
// `hand` carries the links (Assets::addTo) and, on return, the finger states; `est` owns the context.
inline Mat4 process_frame(ConfigParser& cfg, const Assets& assets, PoseEstimator& est, HandT42& hand, const std::vector<uint16_t>& depth_raw, int H, int W,
                          const float K9[9], const Mat4& handbase_in_cam_reported, double depth_unit = 0.001, bool use_physics = true, bool use_render = true,
                          FrameInfo* info_out = nullptr) {
  FrameInfo info;
  hop_ctx* ctx = est.ctx();
  const Mat4 ident = Mat4::Identity();
  auto done = [&](const Mat4& r) {
    if (info_out) *info_out = info;
    return r;
  };
  auto lap_t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* name) {
    const auto t = std::chrono::steady_clock::now();
    info.stage_ms.emplace_back(name, std::chrono::duration<double, std::milli>(t - lap_t0).count());
    lap_t0 = t;
  };
  hand.reset();
  est.reset();
  hand._handbase_in_cam = handbase_in_cam_reported;
  // :116-151 organised cloud (Utils::convert3dOrganizedRGB, Utils.cpp:79-115: a dropped pixel is (0,0,0)), integral-image normals
  const size_t npx = (size_t)H * W;
  std::vector<float> org(3 * npx, 0.f), org_n(3 * npx);
  for (int u = 0; u < H; ++u)
    for (int v = 0; v < W; ++v) {
      float d = (float)((double)(float)depth_raw[(size_t)u * W + v] * depth_unit);
      if (d > 2.0f || d < 0.1f) d = 0.f;
      if (d > 0.1f && d < 2.0f) {
        const size_t i = (size_t)u * W + v;
        org[i] = ((float)v - K9[2]) * d / K9[0];
        org[npx + i] = ((float)u - K9[5]) * d / K9[4];
        org[2 * npx + i] = d;
      }
    }
  check(hop_normals_integral_image(ctx, org.data(), H, W, 0.02f, 10.0f, 1, org_n.data()), ctx, "hop_normals_integral_image");
  lap("organised cloud + integral-image normals");
  Cloud scene_organized;  // valid z and finite normals (runICP drops NaN normals first, Utils.cpp:198-199)
  {
    std::vector<size_t> idx;
    for (size_t i = 0; i < npx; ++i) {
      const float z = org[2 * npx + i];
      if (z < 0.1f || z > 2.0f) continue;
      if (!(std::isfinite(org_n[i]) && std::isfinite(org_n[npx + i]) && std::isfinite(org_n[2 * npx + i]))) continue;
      idx.push_back(i);
    }
    scene_organized.n = (int)idx.size();
    scene_organized.xyz.resize(3 * idx.size()), scene_organized.nrm.resize(3 * idx.size());
    for (int k = 0; k < 3; ++k)
      for (size_t j = 0; j < idx.size(); ++j)
        scene_organized.xyz[(size_t)k * idx.size() + j] = org[(size_t)k * npx + idx[j]], scene_organized.nrm[(size_t)k * idx.size() + j] = org_n[(size_t)k * npx + idx[j]];
  }
  // :54-96 z pass-through, 1 mm voxel grid, hand-base crop (with the REPORTED hand-base pose), normals carried along
  float cam_in_handbase[16];
  hand.camToHandbase(cam_in_handbase);
  const float crop_min[3] = {-0.25f, -0.2f, -0.12f}, crop_max[3] = {-0.07f, 0.2f, 0.05f};
  std::vector<float> sx(3 * npx), sn(3 * npx);
  int n_rgb = 0, counts[3] = {0, 0, 0};
  check(hop_scene_from_depth_normals(ctx, depth_raw.data(), H, W, depth_unit, K9, cam_in_handbase, hand._handbase_in_cam.m, 0.001f, crop_min, crop_max, 0.02f, 10.0f,
                                     sx.data(), sn.data(), (int)npx, &n_rgb, counts),
        ctx, "hop_scene_from_depth_normals");
  lap("scene_from_depth_normals");
  info.n_valid = counts[0], info.n_hand_region = n_rgb;
  if (n_rgb == 0) return done(ident);
  const Cloud scene_rgb = compact(sx, sn, (int)npx, n_rgb);
  // :155 Hand::setCurScene (Hand.cpp:279-334): handbaseICP on the organised cloud, 3 mm hand region, outlier filters
  {
    float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
    for (int k = 0; k < 3; ++k)
      for (int i = 0; i < assets.model001.n; ++i) {
        mn[k] = std::min(mn[k], assets.model001.xyz[(size_t)k * assets.model001.n + i]);
        mx[k] = std::max(mx[k], assets.model001.xyz[(size_t)k * assets.model001.n + i]);
      }
    cfg.gripper_min_dist = 0.8f * std::min(std::min(std::abs(mn[0] - mx[0]), std::abs(mn[1] - mx[1])), std::abs(mn[2] - mx[2]));  // run_real_all.cpp:41-45
  }
  hand.handbaseICP(scene_organized, assets.base_link);
  lap("compaction + handbaseICP");
  info.handbase_in_cam = hand._handbase_in_cam;
  std::vector<float> rx(3 * (size_t)n_rgb), rn(3 * (size_t)n_rgb);
  int n_region = 0;
  check(hop_voxel_downsample_normals(ctx, scene_rgb.xyz.data(), scene_rgb.nrm.data(), n_rgb, 0.003f, rx.data(), rn.data(), n_rgb, &n_region), ctx,
        "hop_voxel_downsample_normals");
  const Cloud region = compact(rx, rn, n_rgb, n_region);
  hand.setCurSceneFromRegion(region);
  lap("3 mm region + setCurScene");
  // :158-185 finger states
  const float f1_min = cfg.getf("hand_match.finger1_min_match"), f2_min = cfg.getf("hand_match.finger2_min_match");
  const float f1_d = cfg.getf("hand_match.finger1_dist_thres"), f2_d = cfg.getf("hand_match.finger2_dist_thres");
  const float f1_a = cfg.getf("hand_match.finger1_normal_angle"), f2_a = cfg.getf("hand_match.finger2_normal_angle");
  hand.camToHandbase(cam_in_handbase);
  const char* order[2][2] = {{"finger_2_1", "finger_2_2"}, {"finger_1_1", "finger_1_2"}};
  const bool cam_right = cam_in_handbase[7] > 0;
  if (hand.sceneSizes().first > 0 && hand.sceneSizes().second > 0)
    for (int k = 0; k < 2; ++k) {
      const char* const* pr = order[cam_right ? k : 1 - k];
      if (hand.matchOneComponentPSO(pr[0], 0, 120, false, f1_d, f1_a, f1_min)) {
        info.angles[pr[0]] = hand._finger_angles[pr[0]];
        if (hand.matchOneComponentPSO(pr[1], 0, 90, true, f2_d, f2_a, f2_min)) info.angles[pr[1]] = hand._finger_angles[pr[1]];
      }
    }
  // :187-188
  lap("finger PSO");
  hand.adjustHandHeight(region);
  hand.makeHandCloud();
  // :193-199 hand points removed, confidences; :201 MLS normals; :203-225 generator cloud
  const float near = cfg.getf("near_hand_dist", 0.003f);
  Cloud finite_rgb;
  {
    std::vector<int> idx;
    for (int i = 0; i < n_rgb; ++i)
      if (std::isfinite(scene_rgb.nrm[i]) && std::isfinite(scene_rgb.nrm[(size_t)n_rgb + i]) && std::isfinite(scene_rgb.nrm[2 * (size_t)n_rgb + i])) idx.push_back(i);
    finite_rgb.n = (int)idx.size();
    finite_rgb.xyz.resize(3 * idx.size()), finite_rgb.nrm.resize(3 * idx.size());
    for (int k = 0; k < 3; ++k)
      for (size_t j = 0; j < idx.size(); ++j)
        finite_rgb.xyz[(size_t)k * idx.size() + j] = scene_rgb.xyz[(size_t)k * n_rgb + idx[j]], finite_rgb.nrm[(size_t)k * idx.size() + j] = scene_rgb.nrm[(size_t)k * n_rgb + idx[j]];
  }
  const Cloud without_hand = hand.removeSurroundingPointsAndAssignProbability(finite_rgb, hand._handbase_in_cam, near * near);
  info.n_without_hand = without_hand.n;
  lap("height + hand cloud + hand-point removal");
  if (without_hand.n < 3) return done(ident);
  const int nw = without_hand.n;
  std::vector<float> mp(3 * (size_t)nw), mnrm(3 * (size_t)nw), mcurv(nw);
  std::vector<int> mk(nw);
  int n_mls = 0;
  check(hop_normals_mls(ctx, without_hand.xyz.data(), nw, 0.003f, 2, mp.data(), mnrm.data(), mcurv.data(), mk.data(), nw, &n_mls), ctx, "hop_normals_mls");
  std::vector<float> mconf(std::max(n_mls, 1));
  for (int i = 0; i < n_mls; ++i) mconf[i] = without_hand.conf[mk[i]];
  const Cloud mls = compact(mp, mnrm, nw, n_mls, &mconf);
  std::vector<float> ox(3 * (size_t)std::max(n_mls, 1)), on(3 * (size_t)std::max(n_mls, 1)), oc(std::max(n_mls, 1));
  int n_seg = 0;
  check(hop_object_segment(ctx, mls.xyz.data(), mls.nrm.data(), mls.conf.data(), n_mls, 0.003f, ox.data(), on.data(), oc.data(), std::max(n_mls, 1), &n_seg), ctx,
        "hop_object_segment");
  info.n_object_segment = n_seg;
  lap("MLS + object segment");
  if (n_seg < 4) return done(ident);
  const Cloud object_segment = compact(ox, on, std::max(n_mls, 1), n_seg, &oc);
  if (std::getenv("HOP_APP_DEBUG")) {
    auto sum = [](const std::vector<float>& v) {
      double s = 0;
      for (float x : v) s += std::isfinite(x) ? (double)x : 1e3;
      return s;
    };
    std::printf("debug sums: scene_organized %d %.9g %.9g | scene_rgb %d %.9g %.9g | region %d %.9g | without_hand %d %.9g %.9g %.9g | mls %d %.9g %.9g | segment %d %.9g %.9g %.9g\n",
                scene_organized.n, sum(scene_organized.xyz), sum(scene_organized.nrm), scene_rgb.n, sum(scene_rgb.xyz), sum(scene_rgb.nrm), region.n, sum(region.xyz),
                without_hand.n, sum(without_hand.xyz), sum(without_hand.nrm), sum(without_hand.conf), mls.n, sum(mls.xyz), sum(mls.nrm), object_segment.n,
                sum(object_segment.xyz), sum(object_segment.nrm), sum(object_segment.conf));
  }
  if (const char* dd = std::getenv("HOP_APP_DEBUG_DIR")) {
    auto dump = [&](const char* name, const Cloud& c) {
      std::ofstream f(std::string(dd) + "/" + name, std::ios::binary);
      int32_t hdr[2] = {c.n, (int32_t)!c.conf.empty()};
      f.write(reinterpret_cast<const char*>(hdr), 8);
      f.write(reinterpret_cast<const char*>(c.xyz.data()), sizeof(float) * c.xyz.size());
      f.write(reinterpret_cast<const char*>(c.nrm.data()), sizeof(float) * c.nrm.size());
      if (!c.conf.empty()) f.write(reinterpret_cast<const char*>(c.conf.data()), sizeof(float) * c.conf.size());
    };
    dump("without_hand.bin", without_hand), dump("mls.bin", mls), dump("segment.bin", object_segment), dump("finite_rgb.bin", finite_rgb);
  }
  // :230-241
  est.setCurScene(object_segment, without_hand);
  est.setDepth(depth_raw, H, W, depth_unit, K9);
  est.registerHandMesh(&hand);
  est.registerMesh(assets.object_mesh, "object", ident.m);
  lap("estimator scene + mesh registration");
  if (!est.runSuper4pcs(assets.ppf_keys)) return done(ident);
  info.n_generated = est.numHypos();
  lap("runSuper4pcs");
  est.clusterPoses(30, 0.015, true);
  info.n_clusters = est.numHypos();
  lap("clusterPoses 1");
  est.refineByICP();
  lap("refineByICP");
  est.clusterPoses(5, 0.003, false);
  info.n_after_icp = est.numHypos();
  if (use_physics) {
    lap("clusterPoses 2");
    est.rejectByCollisionOrNonTouching(&hand);
    info.n_after_physics = est.numHypos();
    lap("rejectByCollisionOrNonTouching");
  }
  if (use_render && est.numHypos() > 0) {
    est.rejectByRender(cfg.getf("pose_estimator_wrong_ratio", 0.f), &hand);
    info.n_after_render = est.numHypos();
    lap("rejectByRender");
  }
  if (est.numHypos() == 0) return done(ident);
  PoseHypo best(-1);
  est.selectBest(best);
  info.score = best._lcp_score;
  lap("selectBest");
  Mat4 r;
  for (int i = 0; i < 16; ++i) r.m[i] = best._pose[i];
  return done(r);
}

}  // namespace hop
#endif
