// CPU check of the C++ host's readers (host/Frame.h: read_png16, parse_pose_txt, Calibration) -- compiled and run by
// tests/test_host_io_cpu.py, which compares the printed values with the Python mirror's.  No context is created: no GPU needed.
//   host_io_check <config.yaml> <depth.png> <arm_pose.txt> <palm_pose.txt>
#include <cinttypes>
#include <iostream>

#include "../../icra20-hand-object-pose_amd/host/Frame.h"

int main(int argc, char** argv) {
  if (argc == 3 && std::string(argv[2]) == "dump") {  // every key the parser holds, "key<TAB>value"
    try {
      ConfigParser cfg(argv[1]);
      for (const auto& kv : cfg.all()) std::printf("%s\t%s\n", kv.first.c_str(), kv.second.c_str());
      return 0;
    } catch (const std::exception& e) {
      std::fprintf(stderr, "error: %s\n", e.what());
      return 3;
    }
  }
  if (argc < 5) return 2;
  if (std::string(argv[1]) == "-") {  // the PNG reader alone: "png H W sum"
    try {
      std::vector<uint16_t> d;
      int H = 0, W = 0;
      hop::read_png16(argv[2], d, H, W);
      uint64_t sum = 0;
      for (uint16_t v : d) sum += v;
      std::printf("png %d %d %" PRIu64 "\n", H, W, sum);
      return 0;
    } catch (const std::exception& e) {
      std::fprintf(stderr, "error: %s\n", e.what());
      return 3;
    }
  }
  try {
    ConfigParser cfg(argv[1]);
    const hop::Calibration cal(cfg);
    std::vector<uint16_t> d;
    int H = 0, W = 0;
    hop::read_png16(argv[2], d, H, W);
    uint64_t sum = 0, wsum = 0;
    for (size_t i = 0; i < d.size(); ++i) sum += d[i], wsum += (uint64_t)d[i] * (uint64_t)(i % 9973 + 1);
    std::printf("png %d %d %" PRIu64 " %" PRIu64 " %u %u\n", H, W, sum, wsum, (unsigned)d[0], (unsigned)d[d.size() - 1]);
    const Mat4 arm = hop::parse_pose_txt(argv[3]), palm = hop::parse_pose_txt(argv[4]);
    const Mat4 hb = cal.handbaseInCam(arm, palm);
    std::printf("K");
    for (int i = 0; i < 9; ++i) std::printf(" %.9g", cal.K9[i]);
    std::printf("\nhandbase");
    for (int i = 0; i < 16; ++i) std::printf(" %.9g", hb.m[i]);
    std::printf("\n");
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 3;
  }
}
