// PoseHypo.h -- the reference's hypothesis record (src/perception/include/PoseHypo.h:7-26), Eigen-free.
#ifndef HOP_HOST_POSEHYPO_H_
#define HOP_HOST_POSEHYPO_H_
#include <cstdio>
struct PoseHypo {
  float _pose[16];  // row-major 4x4, model -> scene
  float _wrong_ratio = 1.f;
  float _lcp_score = 0.f;
  int _id = -1;
  PoseHypo() { setIdentity(); }
  explicit PoseHypo(int id) : _id(id) { setIdentity(); }
  PoseHypo(const float* pose16, int id, float lcp_score = 0.f) : _lcp_score(lcp_score), _id(id) {
    for (int i = 0; i < 16; ++i) _pose[i] = pose16[i];
  }
  void setIdentity() {
    for (int i = 0; i < 16; ++i) _pose[i] = (i % 5 == 0) ? 1.f : 0.f;
  }
  void print() const { std::printf("pose#%d, lcp_score=%g, wrong_ratio=%g\n", _id, _lcp_score, _wrong_ratio); }
};
#endif
