/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the reference's hot path -- wenbowen123/icra20-hand-object-pose:
 *   generator  : src/OpenGR_4pcs/src/gr/algorithms/{matchBase.hpp,match4pcsBase.hpp,
 *                congruentSetExplorationBase.hpp,FunctorSuper4pcs.h,pairCreationFunctor.h,PointPairFilter.h}
 *                + src/OpenGR_4pcs/src/gr/accelerators/{normalset.h,normalset.hpp,utils.h,kdtree.h}
 *                + src/OpenGR_4pcs/src/gr/sampling.h
 *   scoring    : src/perception/src/Utils.cpp:372-444 (computeLCP), :188-229 (runICP call site),
 *                src/perception/src/PoseEstimator.cpp:106-233 (clusterPoses), :235-275, :465-502
 *   hand search: src/perception/src/Hand.cpp:10-178 (objFuncPSO), :182-250 (FingerProperty),
 *                src/perception/include/unconstrained/pso.hpp:146-351 (pso_int)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / reported baseline.  The product (icra20-hand-object-pose_amd/csrc) never links it.
 *
 * Pinning status
 *   - generator + Verify + PPF key + 3-point fit: PINNED against the reference's own code compiled in
 *     place (oracle/ref_driver.cpp -> oracle/_ref/libref_s4pcs.so) and against the golden vectors it
 *     emitted (tests/golden/s4pcs_*.npz, script oracle/gen_golden.py).
 *   - computeLCP, ICP, clusterPoses, objFuncPSO, PSO loop: PARITY UNPINNED.  Their arithmetic lives in
 *     PCL 1.9 / FLANN / Armadillo, none of which is vendored or installed; they are restated from the
 *     call sites above and from the published behaviour of those libraries.
 *
 * Conventions: clouds are SoA planes, `xyz` = [x0..x(n-1) | y0.. | z0..] (3*n floats), same for
 * normals.  Poses are row-major 4x4 float.  All arithmetic is IEEE float without contraction
 * (-ffp-contract=off) in the operation order Eigen 3.3.90 uses for the corresponding expressions.
 */
#ifndef HOP_ORACLE_H_
#define HOP_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- pure functions (generator) */
/* gr::computePPF, matchBase.hpp:47-68.  Normals are raw (normalised here as Point3D::set_normal does). */
void orc_compute_ppf(const float* p1, const float* n1, const float* p2, const float* n2, int* key4);
/* gr::pairPPFisGood, PointPairFilter.h:17-38; each argument = {x,y,z,nx,ny,nz}, normals raw. */
int orc_pair_ppf_is_good(const float* p, const float* q, const float* b0, const float* b1);
/* MatchBase::ComputeRigidTransformation, matchBase.hpp:229-377 (computeScale=false, max_angle<0).
 * ref9/cand9 = three xyz rows each; T16 row-major in the frame of the inputs. returns 1 on success. */
int orc_rigid(const float* ref9, const float* cand9, float* T16, float* rms);
/* probes of the Eigen-order helpers (see ref_driver.cpp ref_probe_*) */
void orc_probe_transform(const float* T16, const float* p3, float* out3);
void orc_probe_vec(const float* a3, const float* b3, float* out9);
void orc_probe_quat(const float* n3, const float* v3, float* out7);

/* ---------------------------------------------------------------- generator */
typedef struct {
  int sample_size;            /* super4pcs_sample_size */
  float overlap;              /* super4pcs_overlap (only feeds the dead number_of_trials formula) */
  float delta;                /* super4pcs_delta */
  float dispersion;           /* super4pcs_dispersion */
  int success_quadrilaterals; /* super4pcs_success_quadrilaterals */
  int n_trials;               /* <=0: the reference's effective value, 30 (cse.hpp:78,90-100) */
  unsigned int random_seed;   /* std::mt19937::default_seed = 5489 in the reference */
} orc_s4pcs_opts;

void* orc_s4pcs_create(const orc_s4pcs_opts* o);
void orc_s4pcs_destroy(void* h);
void orc_s4pcs_set_keys(void* h, const int* keys4, int n);
/* runs ComputeTransformation n_calls times on one matcher; returns the number of hypotheses held */
int orc_s4pcs_run(void* h, const float* Pxyz, const float* Pnrm, const float* Pprob, int nP,
                  const float* Qxyz, const float* Qnrm, int nQ, int n_calls);
int orc_s4pcs_num_hypos(void* h);
void orc_s4pcs_get_hypos(void* h, float* pose16, float* lcp);
int orc_s4pcs_num_bases(void* h);
void orc_s4pcs_get_base(void* h, int i, int* base4, float* inv2, int* counts3);
void orc_s4pcs_get_base_lists(void* h, int i, int* pairs1, int* pairs2, int* quads);
int orc_s4pcs_num_sampled_q(void* h);
/* centred sampled Q (SoA planes), centroids, "diameter" and the count of c<-1 quaternion fallbacks */
void orc_s4pcs_get_state(void* h, float* Qs_xyz, float* Qs_nrm, float* cP3, float* cQ3, float* diameter,
                         int* n_quat_fallback);
/* Verify on the state of the last run; T16 in the centred frame */
float orc_s4pcs_verify(void* h, const float* T16);

/* ---------------------------------------------------------------- batched scoring */
/* CongruentSetExplorationBase::Verify, cse.hpp:346-435: inlier count of T*Qs against P within delta.
 * use_tree: 0 brute force, 1 kd-tree (identical results). */
void orc_verify_batch(const float* Pxyz, int nP, const float* Qxyz, int nQ, const float* T16, int H,
                      float delta, int use_tree, int* count_out);
/* Utils::computeLCP (Utils.cpp:372-444) with use_normal=use_dot_score=use_reciprocal=true, weights 1,
 * applied to model transformed by each pose (PoseEstimator.cpp:482-489). */
void orc_compute_lcp_batch(const float* Sxyz, const float* Snrm, int nS, const float* Mxyz,
                           const float* Mnrm, int nM, const float* pose16, int H, float dist_thres,
                           float angle_deg, int use_tree, float* score_out);
/* refineByICP body (PoseEstimator.cpp:258-270) for each pose: point-to-plane ICP, source = scene,
 * target = model transformed by the pose; pose <- T_icp^-1 * pose.  iters_out/converged_out optional. */
void orc_icp_refine_batch(const float* Sxyz, const float* Snrm, int nS, const float* Mxyz,
                          const float* Mnrm, int nM, float* pose16_inout, int H, int max_iter,
                          float angle_deg, float max_corr_dist, int use_tree, int* iters_out,
                          int* converged_out);
/* PoseEstimator::clusterPoses (PoseEstimator.cpp:106-233). sym_deg = object_symmetry x,y,z (degrees).
 * Sorts by (lcp desc, id asc), greedy clustering; writes indices (into the input arrays) of the kept
 * cluster heads in order; returns their number. */
int orc_cluster_poses(const float* pose16, const float* lcp, const int* ids, int H, float angle_deg,
                      float dist, const float* sym_deg3, int* keep_out);

/* ---------------------------------------------------------------- hand-state search */
typedef struct {
  /* FingerProperty (Hand.cpp:182-250) of the finger being matched and of its outer link */
  float fp_min[3], fp_max[3];
  float fp_stride_z;
  int fp_num_division;
  const float* fp_hist_min_y; /* _hist_alongz row 1: per-z-bin min y, fp_num_division floats */
  float fo_min[3], fo_max[3]; /* finger_out_property extremes */
  float model2handbase[16];   /* row-major */
  float finger_out2parent[16];
  float pair_tip1[4], pair_tip2[4];
  int is_palm_side;  /* name == finger_1_1 || finger_2_1 */
  int is_right_side; /* name == finger_2_1 || finger_2_2 */
  float gripper_min_dist;
  float dist_thres;
  float cos_normal_thres; /* cos(finger{1,2}_normal_angle) */
  int check_normal;
  int max_outter_pts;
  float outter_pt_dist, outter_pt_dist_weight;
  /* clouds, SoA planes */
  const float *model_xyz, *model_nrm;
  int n_model;
  const float *scene_xyz;            /* kd-tree cloud: scene_hand_region_removed_noise */
  int n_scene;
  const float *scene_nrm_lookup;     /* normals of scene_hand_region (Hand.cpp:91 quirk), n_lookup pts */
  int n_lookup;
  const float *swivel_xyz;           /* scene_remove_swivel */
  int n_swivel;
} orc_finger_args;

double orc_pso_objective(const orc_finger_args* a, double angle);
void orc_pso_objective_batch(const orc_finger_args* a, const double* angles, int n, double* cost_out);
typedef struct {
  int n_pop, n_gen, check_freq;
  double c_cog, c_soc, initial_w, w_min, w_max, err_tol;
  double lower_rad, upper_rad;
  uint64_t seed;
} orc_pso_settings;
/* pso_int (pso.hpp:146-351).  RNG: std::mt19937_64(seed) + uniform_real_distribution<double>(0,1). */
int orc_pso_search(const orc_finger_args* a, const orc_pso_settings* s, double* best_angle,
                   double* objval);
/* FingerProperty constructor: fills min/max/stride and hist (6 x num_division, row-major) */
void orc_finger_property(const float* xyz, int n, int num_division, float* min3, float* max3,
                         float* stride_z, float* hist6xN);

/* ---- SURVEY.md 8(f) row N1 (sdf_oracle.cpp) ---- */
/* igl::signed_distance(P, V, F, SIGNED_DISTANCE_TYPE_PSEUDONORMAL, lower, upper, S, I, C, N) on float matrices
 * (SDFchecker.cpp:115-134).  P: np x 3 row-major, V: nv x 3 row-major, F: nf x 3; pose16 (row-major, may be NULL) is
 * applied to V first as SDFchecker::registerMesh / transformMesh do. */
int orc_sdf_signed_distance(const float* P, int np, const float* V, int nv, const int* F, int nf, const float* pose16,
                            float lower, float upper, float* S_out, int* I_out);
/* pcl::VoxelGrid (Utils::downsamplePointCloud, Utils.cpp:334-340), xyz only; SoA planes in (stride n) and out (stride cap) */
int orc_voxel_downsample(const float* xyz_planes, int n, float leaf, float* out_planes, int cap, int* n_out);
/* main_realdata_auto.cpp:156-177: 3 mm voxel grid over xyz + normals, normals towards the camera, nearest-point confidence */
int orc_object_segment(const float* xyz_planes, const float* nrm_planes, const float* conf, int n, float leaf, float* out_xyz, float* out_nrm,
                       float* out_conf, int cap, int* n_out);
/* Hand::setCurScene (Hand.cpp:289-321): hand-base transform, two radius outlier filters, statistical outlier filter, x pass-through */
int orc_hand_scene_filters(const float* xyz_planes, const float* nrm_planes, int n, const float* cam_in_handbase16, float* hb_xyz, float* hb_nrm,
                           unsigned char* keep_noise, unsigned char* keep_swivel);
/* pcl::VoxelGrid over xyz + normals; Hand::handbaseICP's source cloud (Hand.cpp:685-729) */
int orc_voxel_downsample_normals(const float* xyz_planes, const float* nrm_planes, int n, float leaf, float* out_xyz, float* out_nrm, int cap, int* n_out);
int orc_handbase_region(const float* xyz_planes, const float* nrm_planes, int n, const float* cam_in_handbase16, float y1, float z1, float y2, float z2,
                        float* hb_xyz, float* hb_nrm, unsigned char* keep);
/* HandT42::adjustHandHeight matching loop (Hand.cpp:1010-1049) */
int orc_hand_height_matches(const float* scene_xyz, const float* scene_nrm, int n_scene, const float* hand_xyz, const float* hand_nrm, int n_hand,
                            const float* heights, int n_heights, int* counts);
/* scene front end of main_realdata_auto.cpp:54-96 (depth -> cloud, z pass-through, voxel grid, hand-base crop) */
int orc_scene_from_depth(const unsigned short* depth_raw, int H, int W, double depth_unit, const float* K9, const float* cam_in_handbase16,
                         const float* handbase_in_cam16, float leaf, const float* crop_min3, const float* crop_max3, float* out_planes,
                         int cap, int* n_out, int* counts4);
typedef struct {
  /* meshes as registered with SDFchecker: "object" at identity, the four finger links at getTFHandBase(link)
   * (PoseEstimator.cpp:503-520); order finger_1_1, finger_1_2, finger_2_1, finger_2_2 */
  const float* object_V; int object_nv; const int* object_F; int object_nf;
  const float* finger_V[4]; int finger_nv[4]; const int* finger_F[4]; int finger_nf[4];
  const float* finger_mesh_pose[4];  /* row-major 4x4 or NULL */
  /* hand->_clouds[finger] (link frame, SoA planes) and getTFHandBase(finger); finger_status = _component_status */
  const float* finger_xyz[4]; int finger_n[4]; const float* finger2handbase[4]; int finger_status[4];
  const float* hand_cloud_xyz; int n_hand_cloud;                 /* hand->_hand_cloud, hand-base frame */
  const float* cloud_without_hand_xyz; int n_cloud_without_hand; /* _cloud_withouthand_raw, camera frame */
  float cam2handbase[16];
  const float* model_xyz; int n_model;                           /* _model */
  float model_center_init[3], smallest_dim, ob_diameter;
  float collision_thres, non_touch_dist, collision_finger_dist, collision_finger_volume_ratio, voxel_size;
} orc_physics_args;
int orc_reject_by_collision(const orc_physics_args* a, const float* poses16, int H, unsigned char* keep, float* diag8);

/* glibc-compatible float acos used by the PPF tests (restated from fdlibm e_acosf.c) */
float orc_acosf(float x);

#ifdef __cplusplus
}
#endif
#endif
