// ConfigParser.h -- reads config_autodataset.yaml as shipped by the reference, without yaml-cpp or ROS.
// Mirrors the key surface of the reference's ConfigParser (src/perception/include/ConfigParser.h:7-33,
// src/perception/src/ConfigParser.cpp:30-137): `yml["a"]["b"]` lookups become cfg.get("a.b"), the rosparam
// matrices (cam_K, cam1_in_leftarm, handbase_in_palm) are read from the same YAML lists.
// Supported YAML subset: nested block mappings by indentation, scalars, flow sequences `[a, b, ...]` that may
// span lines, `#` comments.  That is everything the shipped file uses.
#ifndef HOP_HOST_CONFIGPARSER_H_
#define HOP_HOST_CONFIGPARSER_H_
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

class ConfigParser {
 public:
  ConfigParser() {}
  explicit ConfigParser(const std::string& path) { parseYMLFile(path); }

  void parseYMLFile(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("ConfigParser: cannot open " + path);
    std::vector<std::pair<int, std::string>> stack;  // (indent, key)
    std::string line;
    while (std::getline(f, line)) {
      const size_t hash = find_comment(line);
      if (hash != std::string::npos) line.erase(hash);
      rtrim(line);
      if (line.find_first_not_of(" \t") == std::string::npos) continue;
      const int indent = (int)line.find_first_not_of(' ');
      const size_t colon = line.find(':', indent);
      if (colon == std::string::npos) continue;
      std::string key = line.substr(indent, colon - indent);
      std::string val = colon + 1 < line.size() ? line.substr(colon + 1) : "";
      ltrim(val);
      while (!stack.empty() && stack.back().first >= indent) stack.pop_back();
      std::string full;
      for (auto& s : stack) full += s.second + ".";
      full += key;
      if (val.empty()) {
        stack.emplace_back(indent, key);
        continue;
      }
      if (val[0] == '[') {  // flow sequence, possibly continued on following lines
        while (val.find(']') == std::string::npos && std::getline(f, line)) {
          const size_t h2 = find_comment(line);
          if (h2 != std::string::npos) line.erase(h2);
          val += " " + line;
        }
      }
      values_[full] = val;
    }
    // the reference's derived members (ConfigParser.cpp:49-115)
    model_name = get("model_name", "");
    pose_estimator_high_confidence_thres = getf("pose_estimator_high_confidence_thres", 0.8f);
    super4pcs_sample_size = geti("super4pcs_sample_size", 100);
    super4pcs_overlap = getf("super4pcs_overlap", 0.2f);
    super4pcs_delta = getf("super4pcs_delta", 0.003f);
    super4pcs_max_normal_difference = getf("super4pcs_max_normal_difference", -1.f);
    super4pcs_max_color_distance = getf("super4pcs_max_color_distance", -1.f);
    super4pcs_max_time_seconds = geti("super4pcs_max_time_seconds", 1);
  }

  const std::map<std::string, std::string>& all() const { return values_; }
  bool has(const std::string& k) const { return values_.count(k) != 0; }
  std::string get(const std::string& k, const std::string& def) const {
    auto it = values_.find(k);
    return it == values_.end() ? def : it->second;
  }
  std::string get(const std::string& k) const {
    auto it = values_.find(k);
    if (it == values_.end()) throw std::runtime_error("ConfigParser: missing key " + k);
    return it->second;
  }
  float getf(const std::string& k) const { return std::strtof(get(k).c_str(), nullptr); }
  float getf(const std::string& k, float def) const { return has(k) ? getf(k) : def; }
  int geti(const std::string& k) const { return (int)std::strtol(get(k).c_str(), nullptr, 10); }
  int geti(const std::string& k, int def) const { return has(k) ? geti(k) : def; }
  bool getb(const std::string& k) const {
    const std::string v = get(k);
    return v == "true" || v == "True" || v == "1" || v == "yes";
  }
  std::vector<float> getlist(const std::string& k) const {
    std::string v = get(k);
    for (char& ch : v)
      if (ch == '[' || ch == ']' || ch == ',') ch = ' ';
    std::istringstream ss(v);
    std::vector<float> out;
    float x;
    while (ss >> x) out.push_back(x);
    return out;
  }

  // members the hot path reads directly in the reference
  std::string model_name;
  float pose_estimator_high_confidence_thres = 0.8f;
  int super4pcs_sample_size = 100;
  float super4pcs_overlap = 0.2f, super4pcs_delta = 0.003f;
  float super4pcs_max_normal_difference = -1.f, super4pcs_max_color_distance = -1.f;
  int super4pcs_max_time_seconds = 1;
  float gripper_min_dist = 0.f;  // filled by the driver (main_realdata_auto.cpp:41-45)

 private:
  std::map<std::string, std::string> values_;
  static size_t find_comment(const std::string& s) {
    for (size_t i = 0; i < s.size(); ++i)
      if (s[i] == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return i;
    return std::string::npos;
  }
  static void rtrim(std::string& s) {
    while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r')) s.pop_back();
  }
  static void ltrim(std::string& s) {
    size_t i = 0;
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) ++i;
    s.erase(0, i);
  }
};
#endif
