// ref_sdf_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A C wrapper around the reference's own vendored libigl (header-only mode), compiled in place from
// /root/reference/src/perception/include/igl with the Eigen the reference vendors under
// src/OpenGR_4pcs/3rdparty/Eigen (see oracle/Makefile, target `ref`).  Nothing is copied from the reference; the
// output (oracle/_ref/libref_sdf.so) is git-ignored.  It exposes exactly the call SDFchecker makes
// (SDFchecker.cpp:115-134: igl::signed_distance(pts, V, F, SIGNED_DISTANCE_TYPE_PSEUDONORMAL, lower, upper, S, I, C, N)
// on float matrices) so that oracle/sdf_oracle.cpp can be pinned against it and tests/golden/sdf_*.npz generated.
#include <Eigen/Core>
#include "igl/signed_distance.h"

extern "C" int ref_igl_signed_distance(const float* P, int np, const float* V, int nv, const int* F, int nf, float lower,
                                       float upper, float* S_out, int* I_out, float* C_out) {
  Eigen::MatrixXf Pm(np, 3), Vm(nv, 3);
  Eigen::MatrixXi Fm(nf, 3);
  for (int i = 0; i < np; i++)
    for (int j = 0; j < 3; j++) Pm(i, j) = P[3 * i + j];
  for (int i = 0; i < nv; i++)
    for (int j = 0; j < 3; j++) Vm(i, j) = V[3 * i + j];
  for (int i = 0; i < nf; i++)
    for (int j = 0; j < 3; j++) Fm(i, j) = F[3 * i + j];
  Eigen::VectorXf S, I;
  Eigen::MatrixXf C, N;
  igl::signed_distance(Pm, Vm, Fm, igl::SIGNED_DISTANCE_TYPE_PSEUDONORMAL, lower, upper, S, I, C, N);
  for (int i = 0; i < np; i++) {
    S_out[i] = S(i);
    if (I_out) I_out[i] = (int)I(i);
    if (C_out)
      for (int j = 0; j < 3; j++) C_out[3 * i + j] = C(i, j);
  }
  return 0;
}
