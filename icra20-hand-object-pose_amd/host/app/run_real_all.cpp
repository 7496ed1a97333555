// run_real_all.cpp -- the reference's dataset driver (src/perception/src/app/run_real_all.cpp:70-273) above libhop, on the reference's
// own directory layout:
//   <base>/<model>/<record>/{rgbN.png, depthN.png, palm_in_baseN.txt, arm_left_link_7_t_N.txt}  ->  <record>/predict/<N>/model2scene.txt
//
//   run_real_all <config_autodataset.yaml> <assets_dir> <base_dir> [model_name]
//   RANK / WORLD_SIZE in the environment shard the frames (frame index mod world, as the Python runner does; the reference is one process);
//   HOP_FORCE=1 recomputes frames whose result exists (default: resume).
// assets_dir holds what the reference loads from PLY / OBJ / Boost archive / URDF files (download links): see hop::Assets (host/Frame.h).
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <iostream>
#include <regex>

#include "../Frame.h"

static std::vector<std::string> list_dir(const std::string& d, bool dirs) {
  std::vector<std::string> out;
  DIR* dp = opendir(d.c_str());
  if (!dp) return out;
  while (dirent* e = readdir(dp)) {
    const std::string n = e->d_name;
    if (n == "." || n == "..") continue;
    struct stat st;
    if (stat((d + "/" + n).c_str(), &st) != 0) continue;
    if (dirs == (S_ISDIR(st.st_mode) != 0)) out.push_back(n);
  }
  closedir(dp);
  std::sort(out.begin(), out.end());
  return out;
}
static void mkdirs(const std::string& p) {
  std::string cur;
  std::istringstream ss(p);
  std::string part;
  if (!p.empty() && p[0] == '/') cur = "/";
  while (std::getline(ss, part, '/')) {
    if (part.empty()) continue;
    cur += part + "/";
    mkdir(cur.c_str(), 0755);
  }
}

int main(int argc, char** argv) {
  if (argc < 4) {
    std::cout << "usage: run_real_all <config.yaml> <assets_dir> <base_dir> [model_name]\n";
    return 2;
  }
  try {
    ConfigParser cfg(argv[1]);
    const hop::Assets assets(argv[2]);
    const std::string base = argv[3], model_name = argc > 4 ? argv[4] : cfg.model_name;
    const int rank = std::getenv("RANK") ? std::atoi(std::getenv("RANK")) : 0, world = std::getenv("WORLD_SIZE") ? std::max(1, std::atoi(std::getenv("WORLD_SIZE"))) : 1;
    const bool force = std::getenv("HOP_FORCE") != nullptr;
    const hop::Calibration cal(cfg);
    // run_real_all.cpp:56-68: estimator and hand are built ONCE, reset between frames (:265-266)
    PoseEstimator est(&cfg, assets.model, assets.model001);
    HandT42 hand(&cfg, est.ctx());
    assets.addTo(hand);
    hop_ctx* icp_ctx = nullptr;
    hop::check(hop_ctx_create(0, &icp_ctx), nullptr, "hop_ctx_create");
    hand.setHandbaseIcpContext(icp_ctx);
    const std::string mdir = base + "/" + model_name;
    int n_done = 0, n_skipped = 0;
    double ms_total = 0;
    const std::regex rgb_re("rgb([0-9]+)\\..*");
    for (const std::string& record : list_dir(mdir, true)) {
      const std::string rec = mdir + "/" + record;
      std::vector<int> frames;  // run_real_all.cpp:76-97: the index sits between "rgb" and the first "."
      for (const std::string& f : list_dir(rec, false)) {
        std::smatch m;
        if (std::regex_match(f, m, rgb_re)) frames.push_back(std::atoi(m[1].str().c_str()));
      }
      std::sort(frames.begin(), frames.end());
      for (int idx : frames) {
        if (idx % world != rank) continue;
        const std::string out_dir = rec + "/predict/" + std::to_string(idx), out = out_dir + "/model2scene.txt";
        {
          std::ifstream ex(out);
          if (ex && !force) {
            ++n_skipped;
            continue;
          }
        }
        const auto t0 = std::chrono::steady_clock::now();
        const Mat4 leftarm_in_base = hop::parse_pose_txt(rec + "/arm_left_link_7_t_" + std::to_string(idx) + ".txt");
        const Mat4 palm_in_baselink = hop::parse_pose_txt(rec + "/palm_in_base" + std::to_string(idx) + ".txt");
        std::vector<uint16_t> depth;
        int H = 0, W = 0;
        hop::read_png16(rec + "/depth" + std::to_string(idx) + ".png", depth, H, W);
        hop::FrameInfo info;
        const Mat4 pose = hop::process_frame(cfg, assets, est, hand, depth, H, W, cal.K9, cal.handbaseInCam(leftarm_in_base, palm_in_baselink), 0.001, true, true, &info);
        mkdirs(out_dir);
        {
          std::ofstream ff(out + ".tmp");  // a killed run never leaves a half-written result
          ff.precision(9);
          for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) ff << pose.m[4 * r + c] << (c == 3 ? "\n" : " ");
        }
        std::rename((out + ".tmp").c_str(), out.c_str());
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ms_total += ms;
        ++n_done;
        std::printf("%s/%d: %.1f ms, %d hand-region points, %d object points, %d hypotheses after ICP, score %.2f\n", record.c_str(), idx, ms, info.n_hand_region,
                    info.n_object_segment, info.n_after_icp, info.score);
      }
    }
    std::printf("rank %d of %d: %d frames written (%.1f ms per frame), %d resumed\n", rank, world, n_done, n_done ? ms_total / n_done : 0.0, n_skipped);
    hop_ctx_destroy(icp_ctx);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 3;
  }
}
