// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or executed from the product path.
//
// The minimiser of the reference's ICP, from the reference's own vendored Eigen.
//
// Utils::runICP (src/perception/src/Utils.cpp:200-216) installs
// pcl::registration::TransformationEstimationPointToPlane, a TransformationEstimationLM: per ICP iteration PCL 1.9
// (registration/impl/transformation_estimation_lm.hpp:146-197) runs
//     Eigen::NumericalDiff<OptimizationFunctorWithIndices> num_diff(functor);
//     Eigen::LevenbergMarquardt<Eigen::NumericalDiff<OptimizationFunctorWithIndices>, float> lm(num_diff);
//     lm.minimize(x);                               // x = (tx,ty,tz, qx,qy,qz) = 0
// on the residuals (warp(x) p_src - p_tgt) . n_tgt (transformation_estimation_point_to_plane.h:77-84) with the warp of
// warp_point_rigid_6d.h:77-106.  PCL itself is absent (not vendored, not installed), but Eigen's
// unsupported/NonLinearOptimization and NumericalDiff modules -- the code that does the arithmetic -- lie under
// /root/reference/src/OpenGR_4pcs/3rdparty/Eigen and are compiled here in place (oracle/Makefile target `ref`,
// output oracle/_ref/libref_icp.so).  The functor and the warp below are the dozen lines of PCL that feed them, written
// with Eigen's own types the way PCL writes them, so that Eigen evaluates the same expressions.
//
// Also here: probes of the vendored Eigen for the functions clusterPoses calls (PoseEstimator.cpp:156,163
// eulerAngles(2,1,0); Utils.cpp:29-32 rotationGeodesicDistance) and for Matrix4f::inverse (PoseEstimator.cpp:267).
#include <cmath>
#include <cstring>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>
#include <Eigen/LU>
#include <unsupported/Eigen/NonLinearOptimization>
#include <unsupported/Eigen/NumericalDiff>

namespace {

typedef Eigen::Matrix<float, Eigen::Dynamic, 1> VectorX;
typedef Eigen::Matrix<float, 4, 1> Vector4;
typedef Eigen::Matrix<float, 4, 4> Matrix4;

// pcl::registration::WarpPointRigid6D<PointSource, PointTarget, float> (warp_point_rigid_6d.h:77-106) and
// WarpPointRigid::warpPoint (warp_point_rigid.h:86-93).
struct Warp6D {
  Matrix4 transform_matrix_;
  void setParam(const VectorX& p) {
    transform_matrix_.setZero();
    transform_matrix_(0, 3) = p[0];
    transform_matrix_(1, 3) = p[1];
    transform_matrix_(2, 3) = p[2];
    transform_matrix_(3, 3) = 1;
    Eigen::Quaternion<float> q(0, p[3], p[4], p[5]);
    q.w() = static_cast<float>(sqrt(1 - q.dot(q)));
    q.normalize();
    transform_matrix_.topLeftCorner(3, 3) = q.toRotationMatrix();
  }
  inline void warpPoint(const float* pnt_in, Vector4& pnt_out) const {
    pnt_out[0] = static_cast<float>(transform_matrix_(0, 0) * pnt_in[0] + transform_matrix_(0, 1) * pnt_in[1] +
                                    transform_matrix_(0, 2) * pnt_in[2] + transform_matrix_(0, 3));
    pnt_out[1] = static_cast<float>(transform_matrix_(1, 0) * pnt_in[0] + transform_matrix_(1, 1) * pnt_in[1] +
                                    transform_matrix_(1, 2) * pnt_in[2] + transform_matrix_(1, 3));
    pnt_out[2] = static_cast<float>(transform_matrix_(2, 0) * pnt_in[0] + transform_matrix_(2, 1) * pnt_in[1] +
                                    transform_matrix_(2, 2) * pnt_in[2] + transform_matrix_(2, 3));
    pnt_out[3] = 0.0;
  }
};

// TransformationEstimationLM::Functor + OptimizationFunctorWithIndices (transformation_estimation_lm.h:257-340,
// impl/transformation_estimation_lm.hpp:241-268) with TransformationEstimationPointToPlane::computeDistance.
struct Functor {
  typedef float Scalar;
  enum { InputsAtCompileTime = Eigen::Dynamic, ValuesAtCompileTime = Eigen::Dynamic };
  typedef Eigen::Matrix<float, Eigen::Dynamic, 1> InputType;
  typedef Eigen::Matrix<float, Eigen::Dynamic, 1> ValueType;
  typedef Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic> JacobianType;

  int m_data_points_;
  const float *src, *tgt, *nrm;  // m x 3 each (source point, matched target point, its normal)
  mutable Warp6D warp;
  Functor(int m, const float* s, const float* t, const float* n) : m_data_points_(m), src(s), tgt(t), nrm(n) {}
  int values() const { return m_data_points_; }

  int operator()(const VectorX& x, VectorX& fvec) const {
    warp.setParam(x);
    for (int i = 0; i < values(); ++i) {
      Vector4 p_src_warped;
      warp.warpPoint(src + 3 * i, p_src_warped);
      Vector4 t(tgt[3 * i], tgt[3 * i + 1], tgt[3 * i + 2], 0);
      Vector4 n(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2], 0);
      fvec[i] = (p_src_warped - t).dot(n);
    }
    return 0;
  }
};

}  // namespace

extern "C" {

// One call of TransformationEstimationLM::estimateRigidTransformation on m correspondences.
// stats: [0] LevenbergMarquardtSpace::Status, [1] nfev, [2] iter; fnorm_out: lm.fnorm.
// Returns 0; -1 when m < 4 (PCL prints an error and leaves the caller's matrix untouched,
// transformation_estimation_lm.hpp:158-164) -- T16_out is not written then.
int ref_lm_point_to_plane(int m, const float* src_xyz, const float* tgt_xyz, const float* tgt_nrm, float* T16_out,
                          float* x6_out, int* stats, float* fnorm_out) {
  if (m < 4) return -1;
  VectorX x(6);
  x.setConstant(6, 0);
  Functor functor(m, src_xyz, tgt_xyz, tgt_nrm);
  Eigen::NumericalDiff<Functor> num_diff(functor);
  Eigen::LevenbergMarquardt<Eigen::NumericalDiff<Functor>, float> lm(num_diff);
  int info = lm.minimize(x);
  functor.warp.setParam(x);
  const Matrix4 T = functor.warp.transform_matrix_;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) T16_out[4 * i + j] = T(i, j);
  if (x6_out)
    for (int j = 0; j < 6; ++j) x6_out[j] = x[j];
  if (stats) stats[0] = info, stats[1] = (int)lm.nfev, stats[2] = (int)lm.iter;
  if (fnorm_out) *fnorm_out = lm.fnorm;
  return 0;
}

// warp matrix of a parameter vector (row-major 4x4)
void ref_probe_warp6(const float* x6, float* T16) {
  VectorX x(6);
  for (int j = 0; j < 6; ++j) x[j] = x6[j];
  Warp6D w;
  w.setParam(x);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) T16[4 * i + j] = w.transform_matrix_(i, j);
}

// residual vector of a parameter vector
void ref_probe_residuals(int m, const float* src_xyz, const float* tgt_xyz, const float* tgt_nrm, const float* x6, float* fvec_out) {
  VectorX x(6), f(m);
  for (int j = 0; j < 6; ++j) x[j] = x6[j];
  Functor functor(m, src_xyz, tgt_xyz, tgt_nrm);
  functor(x, f);
  for (int i = 0; i < m; ++i) fvec_out[i] = f[i];
}

// forward-difference Jacobian Eigen::NumericalDiff hands the minimiser (column-major m x 6 -> row-major out)
void ref_probe_jacobian(int m, const float* src_xyz, const float* tgt_xyz, const float* tgt_nrm, const float* x6, float* jac_out) {
  VectorX x(6);
  for (int j = 0; j < 6; ++j) x[j] = x6[j];
  Functor functor(m, src_xyz, tgt_xyz, tgt_nrm);
  Eigen::NumericalDiff<Functor> num_diff(functor);
  Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic> J(m, 6);
  num_diff.df(x, J);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < 6; ++j) jac_out[6 * i + j] = J(i, j);
}

// Eigen::Matrix3f::eulerAngles(2,1,0) as clusterPoses calls it (PoseEstimator.cpp:156,163); R row-major.
void ref_probe_euler_zyx(const float* R9, float* out3) {
  Eigen::Matrix3f R;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R(i, j) = R9[3 * i + j];
  Eigen::Vector3f rpy = R.eulerAngles(2, 1, 0);
  out3[0] = rpy(0), out3[1] = rpy(1), out3[2] = rpy(2);
}

// rotationGeodesicDistance, Utils.cpp:29-32 (the expression as written there, on Eigen types)
float ref_probe_geodesic(const float* R1_9, const float* R2_9) {
  Eigen::Matrix3f R1, R2;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R1(i, j) = R1_9[3 * i + j], R2(i, j) = R2_9[3 * i + j];
  return std::acos(((R1 * R2).trace() - 1) / 2.0);
}

// (t0-t1).norm() of clusterPoses (PoseEstimator.cpp:148-153)
float ref_probe_tdiff_norm(const float* t0, const float* t1) {
  Eigen::Vector3f a(t0[0], t0[1], t0[2]), b(t1[0], t1[1], t1[2]);
  return (a - b).norm();
}

// transformation.inverse() * model2scene, PoseEstimator.cpp:267 (Matrix4f, row-major in and out)
void ref_probe_inverse_times(const float* Ticp16, const float* pose16, float* out16) {
  Matrix4 A, B;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) A(i, j) = Ticp16[4 * i + j], B(i, j) = pose16[4 * i + j];
  Matrix4 C = A.inverse() * B;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) out16[4 * i + j] = C(i, j);
}

// final_transformation_ = transformation_ * final_transformation_ (icp.hpp), Matrix4f product
void ref_probe_mul4(const float* A16, const float* B16, float* out16) {
  Matrix4 A, B;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) A(i, j) = A16[4 * i + j], B(i, j) = B16[4 * i + j];
  Matrix4 C = A * B;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) out16[4 * i + j] = C(i, j);
}

}  // extern "C"
