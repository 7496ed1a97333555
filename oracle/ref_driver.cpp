// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or executed from the product path.
//
// Thin C-ABI driver around the REFERENCE's own OpenGR fork, compiled from the headers where they lie
// under /root/reference (see oracle/Makefile target `ref`).  It exists only in the build container: it
// validates the CPU restatement (oracle/hop_oracle.cpp) and emits the golden vectors committed under
// tests/golden/ (script: oracle/gen_golden.py).  Nothing of the reference is copied into this repo: the
// reference classes are instantiated exactly the way the PCL wrapper does it
// (src/OpenGR_4pcs/demos/PCLWrapper/pcl/registration/super4pcs.h:122-125,
//  src/OpenGR_4pcs/demos/PCLWrapper/pcl/registration/impl/super4pcs.hpp:66-116).
//
// Build note: -O0 on purpose.  gr::KdTree::operator= has no return statement
// (src/OpenGR_4pcs/src/gr/accelerators/kdtree.h:148-156); g++ -O1+ turns the missing return into
// unreachable code and the matcher crashes in MatchBase::initKdTree (matchBase.hpp:220).  SSE2 float
// arithmetic is identical at -O0 and -O2 (no x87, no contraction without -mfma), so the vectors are
// the ones an optimised reference build would give.
#include <cstdio>
#include <cstring>
#include <vector>
#include <array>
#include <map>

#ifdef REF_O2_TIMING_BUILD
// Timing build only (oracle/_ref/libref_s4pcs_o2.so, -O2): gr::KdTree::operator= has no return statement
// (gr/accelerators/kdtree.h:148-156), which g++ -O1+ turns into unreachable code.  An explicit specialisation of that one member
// with the missing return -- the reference's files stay untouched, the other members are the reference's -- lets the optimised
// build run, so that the reference's generator can be TIMED (tools/ref_generator_timing.py); golden vectors come from the -O0 build.
#include "gr/accelerators/kdtree.h"
namespace gr {
template <>
inline bool KdTree<float, int>::operator=(const KdTree<float, int>& other) {
  mPoints = other.mPoints;
  mIndices = other.mIndices;
  mAABB = other.mAABB;
  mNodes = other.mNodes;
  _nofPointsPerCell = other._nofPointsPerCell;
  _maxDepth = other._maxDepth;
  return true;
}
}  // namespace gr
#endif
#include "gr/shared.h"
#include "gr/sampling.h"
#include "gr/utils/logger.h"
#include "gr/algorithms/match4pcsBase.h"
#include "gr/algorithms/FunctorSuper4pcs.h"
#include "gr/algorithms/PointPairFilter.h"

namespace {

struct Visitor {
  template <typename Derived>
  inline void operator()(float, float, const Eigen::MatrixBase<Derived>&) const {}
  constexpr bool needsGlobalTransformation() const { return false; }
};

using RefMatcherBase =
    gr::Match4pcsBase<gr::FunctorSuper4PCS, Visitor, gr::AdaptivePointFilter, gr::AdaptivePointFilter::Options>;

struct BaseTrace {
  std::array<int, 4> base;
  float inv1, inv2;
  std::vector<std::pair<int, int>> pairs1, pairs2;
  std::vector<std::array<int, 4>> quads;
};

// Exposes protected state and records one trace entry per successful generateCongruents().
class TracingMatcher : public RefMatcherBase {
 public:
  using Base = RefMatcherBase;
  using Base::Base;
  std::vector<BaseTrace> trace;
  bool record_pairs = true;
  bool plain = false;  // true: the reference's own generateCongruents runs untouched (no trace): see ref_use_plain_matcher

  // Same call sequence as Match4pcsBase::generateCongruents (match4pcsBase.hpp:207-281), made through the
  // reference's own member functions, so that the invariants and the two pair lists can be recorded
  // (the reference keeps them in locals).  Re-deriving the invariants afterwards is NOT possible:
  // TryQuadrilateral's permutation loop is not closed under relabelling.
  bool generateCongruents(CongruentBaseType& base, Set& quads) override {
    if (plain) return Base::generateCongruents(base, quads);  // match4pcsBase.hpp:207-281 as written by its authors
    Scalar invariant1, invariant2;
    if (!this->SelectQuadrilateral(invariant1, invariant2, base[0], base[1], base[2], base[3])) return false;
    const auto& b0 = this->base_3D_[0];
    const auto& b1 = this->base_3D_[1];
    const auto& b2 = this->base_3D_[2];
    const auto& b3 = this->base_3D_[3];
    const Scalar distance1 = (b0.pos() - b1.pos()).norm();
    const Scalar distance2 = (b2.pos() - b3.pos()).norm();
    std::vector<std::pair<int, int>> pairs1, pairs2;
    const Scalar normal_angle1 = (b0.normal() - b1.normal()).norm();
    const Scalar normal_angle2 = (b2.normal() - b3.normal()).norm();
    this->fun_.ExtractPairs(distance1, normal_angle1, Base::distance_factor * this->options_.delta, 0, 1, pairs1);
    this->fun_.ExtractPairs(distance2, normal_angle2, Base::distance_factor * this->options_.delta, 2, 3, pairs2);
    if (pairs1.size() == 0 || pairs2.size() == 0) return false;
    if (!this->fun_.FindCongruentQuadrilaterals(invariant1, invariant2, Base::distance_factor * this->options_.delta,
                                                Base::distance_factor * this->options_.delta, pairs1, pairs2, &quads))
      return false;
    BaseTrace t;
    t.base = base;
    t.inv1 = invariant1;
    t.inv2 = invariant2;
    if (record_pairs) {
      t.pairs1 = pairs1;
      t.pairs2 = pairs2;
    }
    for (const auto& q : quads) t.quads.push_back(q);
    trace.push_back(std::move(t));
    return true;
  }

  Scalar verify(const Eigen::Matrix4f& m) const { return this->Verify(m); }
  const std::vector<gr::Point3D>& sampledQ() const { return this->sampled_Q_3D_; }
  const std::vector<gr::Point3D>& sampledP() const { return this->sampled_P_3D_; }
  VectorType centroidP() const { return this->centroid_P_; }
  VectorType centroidQ() const { return this->centroid_Q_; }
  Scalar diameter() const { return this->P_diameter_; }
  int numberOfTrials() const { return this->number_of_trials_; }

  bool rigid(const Coordinates& ref, const Coordinates& cand, Eigen::Matrix4f& T, Scalar& rms) const {
    Eigen::Matrix<Scalar, 3, 1> c1 = (ref[0].pos() + ref[1].pos() + ref[2].pos()) / Scalar(3);
    Eigen::Matrix<Scalar, 3, 1> c2 = (cand[0].pos() + cand[1].pos() + cand[2].pos()) / Scalar(3.);
    return this->ComputeRigidTransformation(ref, cand, c1, c2, T, rms, false);
  }
};

struct RefCtx {
  gr::Utils::Logger logger{gr::Utils::NoLog};
  TracingMatcher* matcher = nullptr;
  std::vector<gr::Point3D> P, Q;
  ~RefCtx() { delete matcher; }
};

void fill(std::vector<gr::Point3D>& out, const float* xyz, const float* nrm, const float* prob, int n) {
  // mirrors fillPointSet, impl/super4pcs.hpp:87-103 (rgb is irrelevant: max_color_distance = -1)
  out.clear();
  out.reserve(n);
  for (int i = 0; i < n; ++i) {
    out.emplace_back(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    Eigen::Vector3f nn(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
    out[i].set_normal(nn);
    out[i].setProb(prob ? prob[i] : 1.0f);
  }
}

}  // namespace

extern "C" {

struct ref_opts {
  int sample_size;
  float overlap;
  float delta;
  float dispersion;
  int success_quadrilaterals;
  int max_time_seconds;
  float max_normal_difference;
  float max_color_distance;
};

void* ref_create(const ref_opts* o) {
  auto* c = new RefCtx;
  TracingMatcher::OptionsType opt;
  // PoseEstimator.cpp:66-73
  opt.sample_size = o->sample_size;
  opt.configureOverlap(o->overlap);
  opt.max_time_seconds = o->max_time_seconds;
  opt.delta = o->delta;
  opt.sample_dispersion = o->dispersion;
  opt.success_quadrilaterals = o->success_quadrilaterals;
  opt.max_normal_difference = o->max_normal_difference;
  opt.max_color_distance = o->max_color_distance;
  c->matcher = new TracingMatcher(opt, c->logger);
  return c;
}

void ref_destroy(void* h) { delete static_cast<RefCtx*>(h); }

void ref_set_ppf_keys(void* h, const int* keys4, int n) {
  auto* c = static_cast<RefCtx*>(h);
  c->matcher->_ppfs.clear();
  for (int i = 0; i < n; ++i) {
    std::vector<int> k(keys4 + 4 * i, keys4 + 4 * i + 4);
    c->matcher->_ppfs[k];  // membership only: the mapped lists are never read (matchBase.hpp:134,159,201)
  }
}

// The golden hypotheses normally come from "reference + the tracing override above" (the override repeats the body of
// match4pcsBase.hpp:207-281 call for call to see its locals).  With plain = 1 the override steps aside and the reference's
// own member runs: gen_golden.py emits one set that way (tests/golden/s4pcs_plain_case1.npz) and checks that it equals the
// traced run, so the override is shown not to change what the reference computes.
void ref_use_plain_matcher(void* h, int on) { static_cast<RefCtx*>(h)->matcher->plain = on != 0; }

void ref_record_pairs(void* h, int on) { static_cast<RefCtx*>(h)->matcher->record_pairs = on != 0; }

// Runs ComputeTransformation `n_calls` times on one matcher (hypotheses accumulate).
int ref_run(void* h, const float* Pxyz, const float* Pnrm, const float* Pprob, int nP, const float* Qxyz,
            const float* Qnrm, int nQ, int n_calls) {
  auto* c = static_cast<RefCtx*>(h);
  fill(c->P, Pxyz, Pnrm, Pprob, nP);
  fill(c->Q, Qxyz, Qnrm, nullptr, nQ);
  gr::UniformDistSampler sampler;
  Visitor v;
  Eigen::Matrix4f T = Eigen::Matrix4f::Identity();
  for (int k = 0; k < n_calls; ++k) c->matcher->ComputeTransformation(c->P, c->Q, T, sampler, v);
  return int(c->matcher->_pose_hypo.size());
}

int ref_num_hypos(void* h) { return int(static_cast<RefCtx*>(h)->matcher->_pose_hypo.size()); }

// row-major 4x4
void ref_get_hypos(void* h, float* pose16, float* lcp) {
  auto* m = static_cast<RefCtx*>(h)->matcher;
  for (size_t i = 0; i < m->_pose_hypo.size(); ++i) {
    for (int r = 0; r < 4; ++r)
      for (int cc = 0; cc < 4; ++cc) pose16[16 * i + 4 * r + cc] = m->_pose_hypo[i](r, cc);
    lcp[i] = m->_pose_lcp_scores[i];
  }
}

int ref_num_bases(void* h) { return int(static_cast<RefCtx*>(h)->matcher->trace.size()); }

void ref_get_base(void* h, int i, int* base4, float* inv2, int* counts3) {
  const auto& t = static_cast<RefCtx*>(h)->matcher->trace[i];
  for (int k = 0; k < 4; ++k) base4[k] = t.base[k];
  inv2[0] = t.inv1;
  inv2[1] = t.inv2;
  counts3[0] = int(t.pairs1.size());
  counts3[1] = int(t.pairs2.size());
  counts3[2] = int(t.quads.size());
}

void ref_get_base_lists(void* h, int i, int* pairs1, int* pairs2, int* quads) {
  const auto& t = static_cast<RefCtx*>(h)->matcher->trace[i];
  for (size_t k = 0; k < t.pairs1.size(); ++k) {
    pairs1[2 * k] = t.pairs1[k].first;
    pairs1[2 * k + 1] = t.pairs1[k].second;
  }
  for (size_t k = 0; k < t.pairs2.size(); ++k) {
    pairs2[2 * k] = t.pairs2[k].first;
    pairs2[2 * k + 1] = t.pairs2[k].second;
  }
  for (size_t k = 0; k < t.quads.size(); ++k)
    for (int j = 0; j < 4; ++j) quads[4 * k + j] = t.quads[k][j];
}

int ref_num_sampled_q(void* h) { return int(static_cast<RefCtx*>(h)->matcher->sampledQ().size()); }

// centred coordinates (after MatchBase::init), plus centroids and the "diameter" (matchBase.hpp:425-456)
void ref_get_state(void* h, float* Qs_xyz, float* Qs_nrm, float* cP3, float* cQ3, float* diameter,
                   int* number_of_trials) {
  auto* m = static_cast<RefCtx*>(h)->matcher;
  const auto& q = m->sampledQ();
  for (size_t i = 0; i < q.size(); ++i)
    for (int k = 0; k < 3; ++k) {
      Qs_xyz[3 * i + k] = q[i].pos()[k];
      Qs_nrm[3 * i + k] = q[i].normal()[k];
    }
  auto cp = m->centroidP();
  auto cq = m->centroidQ();
  for (int k = 0; k < 3; ++k) {
    cP3[k] = cp[k];
    cQ3[k] = cq[k];
  }
  *diameter = m->diameter();
  *number_of_trials = m->numberOfTrials();
}

// Verify (cse.hpp:346-435) of a row-major 4x4 given in the CENTRED frame, on the state left by ref_run.
float ref_verify(void* h, const float* T16) {
  auto* m = static_cast<RefCtx*>(h)->matcher;
  Eigen::Matrix4f T;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) T(r, c) = T16[4 * r + c];
  return m->verify(T);
}

// gr::computePPF (matchBase.hpp:47-68)
void ref_compute_ppf(const float* p1, const float* n1, const float* p2, const float* n2, int* key4) {
  gr::Point3D a(p1[0], p1[1], p1[2]), b(p2[0], p2[1], p2[2]);
  a.set_normal(Eigen::Vector3f(n1[0], n1[1], n1[2]));
  b.set_normal(Eigen::Vector3f(n2[0], n2[1], n2[2]));
  std::vector<int> k;
  gr::computePPF(a, b, k);
  for (int i = 0; i < 4; ++i) key4[i] = k[i];
}

// gr::pairPPFisGood (PointPairFilter.h:17-38): p,q with normals vs b0,b1 with normals. 6 floats each.
int ref_pair_ppf_is_good(const float* p, const float* q, const float* b0, const float* b1) {
  auto mk = [](const float* v) {
    gr::Point3D a(v[0], v[1], v[2]);
    a.set_normal(Eigen::Vector3f(v[3], v[4], v[5]));
    return a;
  };
  return gr::pairPPFisGood(mk(p), mk(q), mk(b0), mk(b1)) ? 1 : 0;
}

// ComputeRigidTransformation (matchBase.hpp:229-377) for 3 reference / 3 candidate points (xyz rows).
// returns 1 if ok; T16 row-major (centred frame), rms.
int ref_rigid(void* h, const float* ref9, const float* cand9, float* T16, float* rms) {
  auto* m = static_cast<RefCtx*>(h)->matcher;
  TracingMatcher::Coordinates r, c;
  for (int i = 0; i < 3; ++i) {
    r[i] = gr::Point3D(ref9[3 * i], ref9[3 * i + 1], ref9[3 * i + 2]);
    c[i] = gr::Point3D(cand9[3 * i], cand9[3 * i + 1], cand9[3 * i + 2]);
  }
  r[3] = r[0];
  c[3] = c[0];
  Eigen::Matrix4f T = Eigen::Matrix4f::Identity();
  float e = -1;
  const bool ok = m->rigid(r, c, T, e);
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) T16[4 * a + b] = T(a, b);
  *rms = e;
  return ok ? 1 : 0;
}


// ---- expression probes: the same Eigen expression shapes the reference uses, so the restatement's
// ---- operation order can be checked bit for bit (tests/test_oracle_vs_ref.py).
// (mat * pos.homogeneous()).head<3>() with mat an Eigen::Ref<const Matrix4f>, cse.hpp:390
static Eigen::Vector3f probe_tf(const Eigen::Ref<const Eigen::Matrix4f>& mat, const gr::Point3D& p) {
  return (mat * p.pos().homogeneous()).template head<3>();
}
void ref_probe_transform(const float* T16, const float* p3, float* out3) {
  Eigen::Matrix4f T;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) T(r, c) = T16[4 * r + c];
  gr::Point3D p(p3[0], p3[1], p3[2]);
  Eigen::Vector3f o = probe_tf(T, p);
  for (int k = 0; k < 3; ++k) out3[k] = o[k];
}
// out = { (a-b).squaredNorm(), a.dot(b), (a-b).norm() }, out+3 = a.normalized(), out+6 = a.cross(b)
void ref_probe_vec(const float* a3, const float* b3, float* out9) {
  Eigen::Vector3f a(a3[0], a3[1], a3[2]), b(b3[0], b3[1], b3[2]);
  out9[0] = (a - b).squaredNorm();
  out9[1] = a.dot(b);
  out9[2] = (a - b).norm();
  Eigen::Vector3f n = a.normalized();
  Eigen::Vector3f c = a.cross(b);
  for (int k = 0; k < 3; ++k) {
    out9[3 + k] = n[k];
    out9[6 + k] = c[k];
  }
}
// Quaternion::setFromTwoVectors((0,0,1), n) then q * v  (normalset.hpp:213-229); out = {w,x,y,z, rotated v}
void ref_probe_quat(const float* n3, const float* v3, float* out7) {
  Eigen::Vector3f n(n3[0], n3[1], n3[2]), v(v3[0], v3[1], v3[2]);
  Eigen::Quaternion<float> q;
  q.setFromTwoVectors(Eigen::Vector3f(0., 0., 1.), n);
  Eigen::Vector3f r = q * v;
  out7[0] = q.w();
  out7[1] = q.x();
  out7[2] = q.y();
  out7[3] = q.z();
  for (int k = 0; k < 3; ++k) out7[4 + k] = r[k];
}

}  // extern "C"
