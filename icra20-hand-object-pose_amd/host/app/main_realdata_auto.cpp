// main_realdata_auto.cpp -- the reference driver's hot-path call order (src/perception/src/app/
// main_realdata_auto.cpp:99-205) on top of libhop, for one frame whose clouds are given as files.
//
//   main_realdata_auto <config_autodataset.yaml> <frame_dir> [out_dir]
//   HOP_APP_REPEAT=n: the frame is processed n times (hand.reset() / est.reset() in between, as run_real_all.cpp:265-266
//   does between frames) and every pass prints "frame_ms <wall ms>": what bench.py's host_cpp leg reads.
//
// The reference loads meshes, a Boost PPF archive and a URDF, none of which ship with it; here the frame
// directory holds the already prepared clouds (what Hand::setCurScene / main :54-181 would produce):
//   model.bin model001.bin object_segment.bin   clouds (see read_cloud)
//   ppf_keys.bin                                int32 n, then n*4 int32 keys
//   hand.txt                                    one line per link: name parent cloud_file 16 floats (row-major)
//   hand_scene.bin hand_region.bin hand_swivel.bin
//   cam_side.txt                                1 if cam_in_handbase(1,3) > 0 (main :114), else 0
// and, optionally, the inputs of the physics rejection (main :185-187,201; all three or none):
//   meshes.txt                                  one line per mesh: name file ("object" and the finger links' convex meshes;
//                                               file = int32 nv, int32 nf, nv*3 float vertices, nf*3 int32 triangles)
//   cloud_withouthand.bin                       _cloud_withouthand_raw, camera frame
//   handbase_in_cam.txt                         16 floats, row-major
// and, optionally on top of those, the inputs of rejectByRender (main :202):
//   depth.bin                                   int32 H, int32 W, float64 depth unit, 9 float32 camera matrix (row-major), H*W uint16
//                                               (the link meshes of meshes.txt double as the hand's visual meshes)
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include "../Frame.h"

static hop::Cloud read_cloud(const std::string& path) {
  // int32 n, int32 has_conf, then 3n float xyz planes, 3n float normal planes, [n float conf]
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  int32_t n = 0, has_conf = 0;
  f.read(reinterpret_cast<char*>(&n), 4);
  f.read(reinterpret_cast<char*>(&has_conf), 4);
  hop::Cloud c;
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

static hop::Mesh read_mesh(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  int32_t nv = 0, nf = 0;
  f.read(reinterpret_cast<char*>(&nv), 4);
  f.read(reinterpret_cast<char*>(&nf), 4);
  hop::Mesh m;
  m.V.resize(3 * (size_t)nv);
  m.F.resize(3 * (size_t)nf);
  f.read(reinterpret_cast<char*>(m.V.data()), sizeof(float) * m.V.size());
  f.read(reinterpret_cast<char*>(m.F.data()), sizeof(int32_t) * m.F.size());
  if (!f) throw std::runtime_error("short read " + path);
  return m;
}

int main(int argc, char** argv) {
  if (argc == 3 && std::string(argv[1]) == "--dump-config") {  // key=value lines of everything the parser read
    try {
      ConfigParser cfg(argv[2]);
      for (const auto& kv : cfg.all()) std::cout << kv.first << "=" << kv.second << "\n";
      return 0;
    } catch (const std::exception& e) {
      std::fprintf(stderr, "error: %s\n", e.what());
      return 3;
    }
  }
  if (argc >= 6 && std::string(argv[2]) == "--depth") {
    // main_realdata_auto <config.yaml> --depth <assets_dir> <depth.png> <handbase_in_cam.txt> [out_dir]: the whole driver from the 16-bit
    // depth image (main_realdata_auto.cpp:54-205; the reference reads rgb / depth / palm_in_base / arm poses from the config's paths)
    try {
      ConfigParser cfg(argv[1]);
      const hop::Assets assets(argv[3]);
      const hop::Calibration cal(cfg);
      std::vector<uint16_t> depth;
      int H = 0, W = 0;
      hop::read_png16(argv[4], depth, H, W);
      const Mat4 handbase_in_cam = hop::parse_pose_txt(argv[5]);
      const std::string out_dir = argc > 6 ? argv[6] : ".";
      PoseEstimator est(&cfg, assets.model, assets.model001);
      HandT42 hand(&cfg, est.ctx());
      assets.addTo(hand);
      hop_ctx* icp_ctx = nullptr;  // handbaseICP's own context: the object's models stay on the estimator's
      hop::check(hop_ctx_create(0, &icp_ctx), nullptr, "hop_ctx_create");
      hand.setHandbaseIcpContext(icp_ctx);
      hop::FrameInfo info;
      const auto t0 = std::chrono::steady_clock::now();
      const Mat4 pose = hop::process_frame(cfg, assets, est, hand, depth, H, W, cal.K9, handbase_in_cam, 0.001, true, true, &info);
      std::printf("frame_ms %.3f\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      std::printf("valid pixels %d, hand region %d, without hand %d, object segment %d, generated %d, clusters %d, after ICP %d, after physics %d, after render %d\n",
                  info.n_valid, info.n_hand_region, info.n_without_hand, info.n_object_segment, info.n_generated, info.n_clusters, info.n_after_icp,
                  info.n_after_physics, info.n_after_render);
      std::ofstream ff(out_dir + "/model2scene.txt");
      ff.precision(9);
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) ff << pose.m[4 * r + c] << (c == 3 ? "\n" : " ");
      std::ofstream fa(out_dir + "/finger_angles.txt");
      fa.precision(9);
      for (auto& kv : info.angles) fa << kv.first << " " << kv.second << "\n";
      hop_ctx_destroy(icp_ctx);
      std::ofstream fhb(out_dir + "/handbase_in_cam.txt");
      fhb.precision(9);
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) fhb << info.handbase_in_cam.m[4 * r + c] << (c == 3 ? "\n" : " ");
      return 0;
    } catch (const std::exception& e) {
      std::fprintf(stderr, "error: %s\n", e.what());
      return 3;
    }
  }
  if (argc < 3) {
    std::cout << "usage: main_realdata_auto <config.yaml> <frame_dir> [out_dir]\n       main_realdata_auto <config.yaml> --depth <assets_dir> <depth.png> "
                 "<handbase_in_cam.txt> [out_dir]\n       main_realdata_auto --dump-config <config.yaml>\n";
    return 2;
  }
  try {
    const std::string config_dir = argv[1], frame = std::string(argv[2]) + "/", out_dir = argc > 3 ? argv[3] : argv[2];
    std::cout << "Using config file: " << config_dir << std::endl;
    ConfigParser cfg(config_dir);
    std::vector<int32_t> ppfs;
    {
      std::ifstream f(frame + "ppf_keys.bin", std::ios::binary);
      int32_t n = 0;
      f.read(reinterpret_cast<char*>(&n), 4);
      ppfs.resize(4 * (size_t)n);
      f.read(reinterpret_cast<char*>(ppfs.data()), sizeof(int32_t) * 4 * (size_t)n);
    }
    const hop::Cloud model = read_cloud(frame + "model.bin"), model001 = read_cloud(frame + "model001.bin");
    {  // main_realdata_auto.cpp:41-45
      float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
      for (int k = 0; k < 3; ++k)
        for (int i = 0; i < model001.n; ++i) {
          mn[k] = std::min(mn[k], model001.xyz[(size_t)k * model001.n + i]);
          mx[k] = std::max(mx[k], model001.xyz[(size_t)k * model001.n + i]);
        }
      cfg.gripper_min_dist = 0.8 * std::min(std::min(std::abs(mn[0] - mx[0]), std::abs(mn[1] - mx[1])), std::abs(mn[2] - mx[2]));
    }
    PoseEstimator est(&cfg, model, model001);
    HandT42 hand(&cfg, est.ctx());
    {
      std::ifstream f(frame + "hand.txt");
      std::string line;
      while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string name, parent, file;
        if (!(ss >> name >> parent >> file)) continue;
        Mat4 T;
        for (int i = 0; i < 16; ++i) ss >> T.m[i];
        hand.addComponent(name, parent, read_cloud(frame + file), T);
      }
    }
    const hop::Cloud hand_scene = read_cloud(frame + "hand_scene.bin"), hand_region = read_cloud(frame + "hand_region.bin"),
                     hand_swivel = read_cloud(frame + "hand_swivel.bin"), object_segment = read_cloud(frame + "object_segment.bin");
    const int repeat = std::getenv("HOP_APP_REPEAT") ? std::max(1, std::atoi(std::getenv("HOP_APP_REPEAT"))) : 1;
    for (int rep = 0; rep < repeat; ++rep) {
    const auto t_frame = std::chrono::steady_clock::now();
    const bool last = rep + 1 == repeat;
    hand.reset();
    est.reset();
    hand.setCurScene(hand_scene, hand_region, hand_swivel);
    int cam_right = 1;
    {
      std::ifstream f(frame + "cam_side.txt");
      if (f) f >> cam_right;
    }
    const float f1_min = cfg.getf("hand_match.finger1_min_match"), f2_min = cfg.getf("hand_match.finger2_min_match");
    const float f1_d = cfg.getf("hand_match.finger1_dist_thres"), f2_d = cfg.getf("hand_match.finger2_dist_thres");
    const float f1_a = cfg.getf("hand_match.finger1_normal_angle"), f2_a = cfg.getf("hand_match.finger2_normal_angle");
    bool match1 = false, match2 = false;
    if (cam_right) {  // main_realdata_auto.cpp:114-139
      match1 = hand.matchOneComponentPSO("finger_2_1", 0, 120, false, f1_d, f1_a, f1_min);
      if (match1) hand.matchOneComponentPSO("finger_2_2", 0, 90, true, f2_d, f2_a, f2_min);
      match2 = hand.matchOneComponentPSO("finger_1_1", 0, 120, false, f1_d, f1_a, f1_min);
      if (match2) hand.matchOneComponentPSO("finger_1_2", 0, 90, true, f2_d, f2_a, f2_min);
    } else {
      match2 = hand.matchOneComponentPSO("finger_1_1", 0, 120, false, f1_d, f1_a, f1_min);
      if (match2) hand.matchOneComponentPSO("finger_1_2", 0, 90, true, f2_d, f2_a, f2_min);
      match1 = hand.matchOneComponentPSO("finger_2_1", 0, 120, false, f1_d, f1_a, f1_min);
      if (match1) hand.matchOneComponentPSO("finger_2_2", 0, 90, true, f2_d, f2_a, f2_min);
    }
    bool physics = false, render = false;
    {
      std::ifstream fd(frame + "depth.bin", std::ios::binary);
      if (fd) {
        int32_t dh = 0, dw = 0;
        double unit = 0;
        float K9[9];
        fd.read(reinterpret_cast<char*>(&dh), 4);
        fd.read(reinterpret_cast<char*>(&dw), 4);
        fd.read(reinterpret_cast<char*>(&unit), 8);
        fd.read(reinterpret_cast<char*>(K9), sizeof(K9));
        std::vector<uint16_t> raw((size_t)dh * dw);
        fd.read(reinterpret_cast<char*>(raw.data()), sizeof(uint16_t) * raw.size());
        if (!fd) throw std::runtime_error("short read depth.bin");
        est.setDepth(raw, dh, dw, unit, K9);
        render = true;
      }
    }
    {
      std::ifstream fm(frame + "meshes.txt"), fh(frame + "handbase_in_cam.txt");
      if (fm && fh) {
        physics = true;
        for (int i = 0; i < 16; ++i) fh >> hand._handbase_in_cam.m[i];
        std::string name, file;
        hop::Mesh object_mesh;
        while (fm >> name >> file) {
          if (name == "object") object_mesh = read_mesh(frame + file);
          else {
            const hop::Mesh m = read_mesh(frame + file);
            hand.addConvexMesh(name, m);
            hand.addMesh(name, m);
          }
        }
        hand.makeHandCloud();  // main :142
        est.setCurScene(object_segment, read_cloud(frame + "cloud_withouthand.bin"));
        est.registerHandMesh(&hand);                                         // main :186
        est.registerMesh(object_mesh, "object", Mat4::Identity().m);         // main :187
      }
    }
    if (!physics) est.setCurScene(object_segment);
    const bool succeed = est.runSuper4pcs(ppfs);
    PoseHypo best(-1);
    if (!succeed) {
      std::printf("No pose found...\n");
      std::ofstream ff(out_dir + "/model2scene.txt");
      ff << "1 0 0 0\n0 1 0 0\n0 0 1 0\n0 0 0 1\n";
      return 1;
    }
    est.clusterPoses(30, 0.015, true);
    est.refineByICP();
    est.clusterPoses(5, 0.003, false);
    if (physics) {
      est.rejectByCollisionOrNonTouching(&hand);  // main :201
      std::printf("hypotheses after physics: %d\n", est.numHypos());
      if (est.numHypos() == 0) {
        std::printf("No pose left...\n");
        return 1;
      }
      if (render) est.rejectByRender(cfg.getf("lcp.dist"), &hand);  // main :202 (the threshold argument is unused there too)
    }
    est.selectBest(best);
    std::printf("frame_ms %.3f\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_frame).count());
    if (!last) continue;
    std::ofstream ff(out_dir + "/model2scene.txt");
    ff.precision(9);
    std::cout << "best tf:\n";
    for (int r = 0; r < 4; ++r) {
      for (int c = 0; c < 4; ++c) {
        ff << best._pose[4 * r + c] << (c == 3 ? "\n" : " ");
        std::cout << best._pose[4 * r + c] << (c == 3 ? "\n" : " ");
      }
    }
    std::ofstream fa(out_dir + "/finger_angles.txt");
    fa.precision(9);
    for (auto& kv : hand._finger_angles) fa << kv.first << " " << kv.second << "\n";
    }
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 3;
  }
}
