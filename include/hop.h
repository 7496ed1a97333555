/* hop.h -- C-ABI of the MI355X-native hot path of wenbowen123/icra20-hand-object-pose.
 *
 * One shared library (libhop.so, HIP for gfx950) behind plain-C entry points: no C++ types, no torch
 * types, caller-owned host buffers in and out.  Every entry point names the reference interface it
 * replaces (paths relative to the reference root; the reference has no FFI layer of its own -- the
 * boundary is the set of C++ member functions main_realdata_auto.cpp:99-205 calls).
 * INTEGRATION.md shows the reference-side shim a maintainer would add.
 *
 * Conventions
 *   - clouds are SoA planes: xyz = [x0..x(n-1) | y0.. | z0..] (3*n floats); normals likewise.
 *   - poses are row-major 4x4 float (model -> scene).
 *   - every function returns HOP_OK (0) or a negative hop_status; nothing aborts or throws.
 *   - one ctx per (host thread, GPU); calls on one ctx are serialised by the caller.
 *   - the library has no CPU fallback: without a HIP device hop_ctx_create fails with HOP_E_NO_DEVICE.
 */
#ifndef HOP_H_
#define HOP_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HOP_ABI_VERSION 2 /* 2: hop_frames_allgather, hop_topk_pack_device, hop_topk_allgather_device, hop_comm_info, hop_cluster_pose_terms (round 3); icp nn_mode 7 */

typedef enum {
  HOP_OK = 0,
  HOP_E_INVALID = -1,     /* bad argument */
  HOP_E_NO_DEVICE = -2,   /* no usable HIP device */
  HOP_E_HIP = -3,         /* a HIP runtime call failed (see hop_last_error) */
  HOP_E_CAPACITY = -4,    /* caller buffer or internal capacity too small */
  HOP_E_STATE = -5,       /* call order (e.g. generate before set_scene) */
  HOP_E_NO_HYPOTHESIS = -6, /* generator produced nothing: runSuper4pcs would return false */
  HOP_E_ALLOC = -7,
  HOP_E_COMM = -8         /* RCCL unavailable or a collective failed (see hop_comm_last_error) */
} hop_status;

typedef struct hop_ctx hop_ctx;

int hop_abi_version(void);
const char* hop_strerror(int status);
/* text of the last HIP error seen by this ctx ("" if none) */
const char* hop_last_error(const hop_ctx* ctx);

int hop_ctx_create(int device, hop_ctx** out);
void hop_ctx_destroy(hop_ctx* ctx);
/* blocks until everything queued on the ctx streams is done */
int hop_synchronize(hop_ctx* ctx);

/* ------------------------------------------------------------------------------------------------
 * Clouds
 * ---------------------------------------------------------------------------------------------- */
/* PoseEstimator::setCurScene (src/perception/src/PoseEstimator.cpp:32-46): `xyz/nrm/conf` is the object
 * segment; points with conf < high_confidence_thres are dropped, the rest is _scene_high_confidence
 * (P of the generator, source of ICP, scene of computeLCP).  Pass thres <= 0 to keep everything. */
int hop_set_scene(hop_ctx* ctx, const float* xyz, const float* nrm, const float* conf, int n,
                  float high_confidence_thres);
int hop_scene_size(const hop_ctx* ctx);

#define HOP_MODEL_5MM 0 /* `_model`    : Q of the generator, ICP target  (main_realdata_auto.cpp:38)  */
#define HOP_MODEL_1MM 1 /* `_model001` : computeLCP model                (main_realdata_auto.cpp:37)  */
/* PoseEstimator ctor (PoseEstimator.cpp:8-23) */
int hop_set_model(hop_ctx* ctx, int level, const float* xyz, const float* nrm, int n);

/* pcl::Super4PCS::setPPFHash (demos/PCLWrapper/pcl/registration/super4pcs.h:149-152).  Only key
 * membership is ever used by the reference (matchBase.hpp:134,159,201), so the table is the key set:
 * nkeys rows of 4 ints (dist mm bin, three angle-degree bins). */
int hop_set_ppf_keys(hop_ctx* ctx, const int32_t* keys4, int nkeys);

/* ------------------------------------------------------------------------------------------------
 * Generator: PoseEstimator::runSuper4pcs (PoseEstimator.cpp:62-100) ->
 * gr::Match4pcsBase::ComputeTransformation (src/OpenGR_4pcs/src/gr/algorithms/
 * congruentSetExplorationBase.hpp:66-125).  Hypotheses stay resident on the device; they are also
 * copied out when poses16_out / lcp_out are non-NULL (cap = capacity in hypotheses).
 * Output order is canonical: base trial, then pair lists in (i desc-major, j) order -- see DESIGN.md.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int sample_size;            /* super4pcs_sample_size            (config_autodataset.yaml:133) */
  float overlap;              /* super4pcs_overlap                (:134)  */
  float delta;                /* super4pcs_delta                  (:135)  */
  float dispersion;           /* super4pcs_dispersion             (:136)  */
  int success_quadrilaterals; /* super4pcs_success_quadrilaterals (:137)  */
  int max_time_seconds;       /* super4pcs_max_time_seconds       (:140); <=0 disables the wall-clock cut */
  int n_trials;               /* base trials; <=0 = the reference's effective 30 (cse.hpp:78,90-100) */
  unsigned int random_seed;   /* gr MatchBase Options::randomSeed, std::mt19937::default_seed = 5489 */
  float max_normal_difference; /* must be < 0 (as shipped); other filters are not implemented */
  float max_color_distance;    /* must be < 0 */
  int verify_mode;            /* 0 = brute force LDS-tiled, 1 = voxel grid, 2 = EXIST-mode NN cell lists (all: same counts) */
} hop_s4pcs_opts;

typedef struct {
  int n_trials_run, n_bases, n_hypotheses;
  long long n_pairs, n_quads, n_candidates; /* summed over bases */
  int n_sampled_q;
  float centroid_p[3], centroid_q[3], diameter;
  double ms_select, ms_device; /* host base selection / device time (events) */
} hop_s4pcs_stats;

void hop_s4pcs_default_opts(hop_s4pcs_opts* o);
int hop_s4pcs_generate(hop_ctx* ctx, const hop_s4pcs_opts* opts, float* poses16_out, float* lcp_out,
                       int cap, int* n_out, hop_s4pcs_stats* stats_out);
/* read-back of the per-base trace of the last generate (tests): base ids, invariants, list sizes */
int hop_s4pcs_num_bases(const hop_ctx* ctx);
int hop_s4pcs_get_base(const hop_ctx* ctx, int i, int* base4, float* inv2, int* counts3);
/* sampled, centred Q of the last generate (SoA planes, n = stats.n_sampled_q) */
int hop_s4pcs_get_sampled_q(const hop_ctx* ctx, float* xyz, float* nrm);

/* CongruentSetExplorationBase::Verify (cse.hpp:346-435) for H transforms given in the centred frame
 * of the last generate (or of hop_verify_set_clouds).  count_out[h] = inliers among the n_q samples. */
int hop_verify_set_clouds(hop_ctx* ctx, const float* p_xyz, int n_p, const float* q_xyz, int n_q);
int hop_verify_batch(hop_ctx* ctx, const float* T16, int H, float delta, int mode, int* count_out);

/* ------------------------------------------------------------------------------------------------
 * Resident hypothesis set (PoseEstimator::_pose_hypos).  The scoring calls below work on it.
 * ---------------------------------------------------------------------------------------------- */
/* replace the set by caller poses (ids = 0..H-1, scores as given or 0) */
int hop_hypos_upload(hop_ctx* ctx, const float* poses16, const float* scores, int H);
int hop_hypos_count(const hop_ctx* ctx);
int hop_hypos_download(hop_ctx* ctx, float* poses16_out, float* scores_out, int* ids_out, int cap,
                       int* n_out);
/* keep the k best by (score desc, emission order asc) = HypoCompare (PoseEstimator.cpp:110-123);
 * the survivors are renumbered 0..k-1 in that order. */
int hop_hypos_keep_topk(hop_ctx* ctx, int k);

/* PoseEstimator::refineByICP (PoseEstimator.cpp:235-275) + Utils::runICP (Utils.cpp:188-229) on the
 * resident set: source = scene, target = HOP_MODEL_5MM transformed by each pose; pose <- T_icp^-1 * pose.
 * The reference keeps only the first 100 hypotheses (PoseEstimator.cpp:241): pass max_hypotheses=100
 * for that, <=0 for "all". */
typedef struct {
  int max_iter;        /* 10                       (PoseEstimator.cpp:266) */
  float angle_deg;     /* icp_angle_thres = 45     (config_autodataset.yaml:127) */
  float max_corr_dist; /* icp_dist_thres  = 0.01   (:126) */
  int max_hypotheses;
  int nn_mode;         /* 0 brute force, 1 ring-expanding voxel grid, 2 NN cell lists, 3 NN cell lists with search and
                      * accumulation in one kernel (same correspondences in all modes), 4 as 3 with the accumulated
                      * transform applied once; modes 0-4 take ONE Gauss-Newton step of the linearised point-to-plane
                      * problem per ICP iteration.
                      * 5: the reference's own minimiser -- PCL's TransformationEstimationPointToPlane is
                      * Eigen::LevenbergMarquardt<NumericalDiff<...>, float> on (t, quaternion xyz) run to ITS stopping rule per
                      * ICP iteration (Utils.cpp:200-216; Eigen's NonLinearOptimization as vendored under
                      * src/OpenGR_4pcs/3rdparty/Eigen/unsupported), with PCL's gates (strict normal test against the double
                      * threshold, double distance gate) and stop rules (absolute MSE only, identity increment).  Several
                      * passes over the correspondences per ICP iteration (one per function evaluation, every residual and
                      * Jacobian entry the float the reference evaluates): ~14x the time of mode 4.
                      * 6: the same minimiser evaluated from the 13 x 13 moment matrix of the correspondences (the residual is
                      * linear in [R | t]): ONE pass per ICP iteration, the whole Levenberg-Marquardt run per hypothesis in the
                      * solve kernel.  Exact arithmetic where PCL rounds each residual to float; closer to the reference's
                      * default build than its -march=native build is (profiles/r03_icp_lm_deltas.json).  Per-lane float sums,
                      * reciprocal estimates in the solve: results reproducible on the GPU, not on a CPU.
                      * 7: the moment form made REPRODUCIBLE ANYWHERE -- the 13 components of u are put on a power-of-two grid
                      * (12 bits, nearest integer of the exact product: 30 um on positions, 2e-4 on normals, 4 um on the residual -- two orders below the noise of the
                      * reference's float forward differences), the moment matrix is their exact integer sum (any order of the
                      * lanes / wavefronts / workgroups gives the same 64-bit integers), and the solve uses IEEE + - * / sqrt fma in
                      * a fixed order: the CPU statement of the algorithm (oracle minimiser 7) returns the same bits, so the chain
                      * above it returns the same pose.  As close to the reference's run as mode 6.  The sums run on the matrix cores
                      * (v_mfma_i32_16x16x64_i8 on a byte split of the 13-bit operands: k_icp_fusedq_momm) once the device has passed a check
                      * of the operand layout, else -- or with HOP_ICP_MFMA=0 -- on the vector units (k_icp_fusedq_momi): the same integers.
                      * Modes 6 and 7 walk the packed (8-byte) cell lists, which exist for a HOP_MODEL_5MM of < 65535 points whose
                      * lists fit the 16-bit cell frame.  Without them mode 6 runs as mode 5 (same minimiser, float per-pass form);
                      * mode 7 returns HOP_E_STATE (hop_last_error says why) rather than bits other than the ones it promises. */
} hop_icp_opts;
int hop_icp_refine(hop_ctx* ctx, const hop_icp_opts* opts, int* iterations_out /*H or NULL*/,
                   int* converged_out /*H or NULL*/);

/* PoseEstimator::selectBest (PoseEstimator.cpp:465-502) + Utils::computeLCP (Utils.cpp:372-444):
 * scores every resident pose against HOP_MODEL_1MM, stores the score as its _lcp_score and returns the
 * arg-max (first maximum in set order, strict '>' as the reference). */
typedef struct {
  float dist;      /* lcp.dist = 0.001          (config_autodataset.yaml:116) */
  float angle_deg; /* lcp.normal_angle = 10     (:117) */
  int nn_mode;     /* 0 brute force, 1 voxel grids, 2 NN cell lists (same scores, bit for bit); < 0: chosen by problem size */
} hop_lcp_opts;
int hop_lcp_select_best(hop_ctx* ctx, const hop_lcp_opts* opts, float* best_pose16_out,
                        float* best_score_out, int* best_index_out);

/* PoseEstimator::clusterPoses(angle_deg, dist, assign_id) (PoseEstimator.cpp:106-233) on the resident
 * set (host-side greedy pass over the sorted set, as in the reference). sym_deg = object_symmetry x,y,z. */
int hop_cluster_poses(hop_ctx* ctx, float angle_deg, float dist, const float* sym_deg3, int assign_id);
/* the same as a pure function on caller arrays; keep_out receives indices of the cluster heads */
int hop_cluster_poses_host(const float* poses16, const float* scores, const int* ids, int H,
                           float angle_deg, float dist, const float* sym_deg3, int* keep_out,
                           int* n_keep_out);

/* the three quantities clusterPoses compares for a pair of poses, as pure host functions (known-answer tests against the
 * reference's vendored Eigen): out5 = { eulerAngles(2,1,0) of pose_a's rotation [3] (PoseEstimator.cpp:156),
 * rotationGeodesicDistance(R_a, R_b) (Utils.cpp:29-32), (t_a - t_b).norm() (PoseEstimator.cpp:148-153) } */
int hop_cluster_pose_terms(const float* pose_a16, const float* pose_b16, float* out5);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU exchange: each rank packs its k best rows, ranks all-gather the tables (RCCL through
 * torch.distributed, see bench.py) and merge them identically.  Row = {score f32, id i32, pose 16 f32}.
 * ---------------------------------------------------------------------------------------------- */
#define HOP_TOPK_ROW_FLOATS 18
int hop_topk_pack(hop_ctx* ctx, int k, int id_offset, float* rows_out /* k*18 floats */, int* n_rows_out);
/* pure function: merge n_tables tables of k rows each (rows with id < 0 are padding) into the k best */
int hop_topk_merge(const float* tables, int n_tables, int k, float* rows_out, int* n_rows_out);

/* The exchange itself, inside the library (the reference has no multi-GPU path): one RCCL communicator per
 * process / GPU, one ncclAllGather of the k x 72-byte table per frame from device buffers on the communicator's own
 * stream, then hop_topk_merge -- the same k rows on every rank.  RCCL is loaded with dlopen on first use.
 *   hop_comm_unique_id   rank 0: ncclGetUniqueId; the launcher hands the 128 bytes to the other ranks (any channel)
 *   hop_comm_create      every rank: ncclCommInitRank on `device`
 *   hop_topk_allgather   collective: every rank calls it once per frame, in the same order */
#define HOP_COMM_ID_BYTES 128
typedef struct hop_comm hop_comm;
int hop_comm_unique_id(unsigned char id_out[HOP_COMM_ID_BYTES]);
int hop_comm_create(int device, const unsigned char id[HOP_COMM_ID_BYTES], int rank, int world, hop_comm** out);
void hop_comm_destroy(hop_comm* comm);
const char* hop_comm_last_error(const hop_comm* comm /* may be NULL: why RCCL could not be loaded */);
int hop_topk_allgather(hop_comm* comm, const float* rows_in, int k, float* merged_out, int* n_rows_out);
/* the resident set's top-k table written on the device (rows_dev: device memory, k rows) -- no download, no host sort */
int hop_topk_pack_device(hop_ctx* ctx, int k, int id_offset, float* rows_dev, int* n_rows_out);
/* the exchange with the table on the device: rows_dev (device memory, written by hop_topk_pack_device) -> ncclAllGather -> merge
 * kernel; only the k merged rows return to the host.  Same result as hop_topk_pack + hop_topk_allgather. */
int hop_topk_allgather_device(hop_comm* comm, const float* rows_dev, int k, float* merged_out, int* n_rows_out);
/* ncclCommCount of the communicator, mean wall time [us] and number of the top-k exchanges so far */
int hop_comm_info(hop_comm* comm, int* rccl_ranks_out, double* mean_exchange_us_out, long* exchanges_out);
/* BASELINE configs[3] (frames sharded over the GPUs, "RCCL gather of per-frame best pose"; the reference writes one 4 x 4 matrix per
 * frame, run_real_all.cpp:256-262): rows of HOP_FRAME_ROW_FLOATS floats = { frame index, pose[16] row-major }, n_local <= rows_per_rank
 * rows from this rank, ONE ncclAllGather; all_out: world * rows_per_rank rows in rank order, padding rows have frame index -1. */
#define HOP_FRAME_ROW_FLOATS 17
int hop_frames_allgather(hop_comm* comm, const float* rows_local, int n_local, int rows_per_rank, float* all_out);

/* ------------------------------------------------------------------------------------------------
 * Hand-state search: Hand::matchOneComponentPSO (src/perception/src/Hand.cpp:603-672) ->
 * optim::pso (include/unconstrained/pso.hpp:146-351) -> objFuncPSO (Hand.cpp:10-178).
 * ---------------------------------------------------------------------------------------------- */
/* Hand::setCurScene products (Hand.cpp:327-332): kd-tree cloud (scene_hand_region_removed_noise),
 * normals of scene_hand_region (indexed with the kd-tree's indices, Hand.cpp:91) and scene_remove_swivel;
 * all in the hand-base frame. */
int hop_hand_set_scene(hop_ctx* ctx, const float* scene_xyz, int n_scene, const float* lookup_nrm,
                       int n_lookup, const float* swivel_xyz, int n_swivel);

typedef struct {
  float fp_min[3], fp_max[3]; /* FingerProperty of the finger (Hand.cpp:182-236) */
  float fp_stride_z;
  int fp_num_division;
  const float* fp_hist_min_y; /* _hist_alongz row 1 */
  float fo_min[3], fo_max[3]; /* finger_out_property extremes */
  float model2handbase[16];   /* getTFHandBase(name) (Hand.cpp:505-523) */
  float finger_out2parent[16];
  float pair_tip1[4], pair_tip2[4]; /* Hand.cpp:611-643 */
  int is_palm_side;                 /* finger_1_1 / finger_2_1 */
  int is_right_side;                /* finger_2_1 / finger_2_2 */
  float gripper_min_dist;           /* cfg.gripper_min_dist (main_realdata_auto.cpp:41-45) */
  float dist_thres;                 /* finger{1,2}_dist_thres */
  float cos_normal_thres;           /* cos(finger{1,2}_normal_angle) */
  int check_normal;
  int max_outter_pts;
  float outter_pt_dist, outter_pt_dist_weight;
  const float *model_xyz, *model_nrm; /* finger cloud @5 mm, link frame, SoA */
  int n_model;
} hop_finger_args;
int hop_hand_set_finger(hop_ctx* ctx, const hop_finger_args* args);
/* objFuncPSO for n angles (radians) at once */
int hop_hand_pso_eval_batch(hop_ctx* ctx, const double* angles, int n, double* cost_out);
/* How the outer-side penalty of objFuncPSO (Hand.cpp:141-152, one float accumulated over the no-swivel scene) is summed:
 * 0 (default) in scene order, bit-equal to the reference's loop; 1 by a fixed-order tree reduction on the GPU (equal to
 * ~1e-6 relative; the order of a float sum is not part of the reference's interface).  Applies to the calls that follow. */
int hop_hand_set_sum_mode(hop_ctx* ctx, int mode);

typedef struct {
  int n_pop, n_gen, check_freq;     /* hand_match.pso.* (config_autodataset.yaml:108-114) */
  double c_cog, c_soc, initial_w;
  double w_min, w_max, err_tol;     /* OptimLib defaults 0.10 / 0.99, err_tol 1e-5 (Hand.cpp:594) */
  double lower_rad, upper_rad;      /* bounds (Hand.cpp:606-609) */
  uint64_t seed;                    /* arma_rng::set_seed(0) (pso.hpp:152) */
} hop_pso_settings;
void hop_pso_default_settings(hop_pso_settings* s);
/* optim::pso: returns the best angle and the objective value (ArgPasser::objval) */
int hop_hand_pso_search(hop_ctx* ctx, const hop_pso_settings* s, double* best_angle_out,
                        double* objval_out);

/* ------------------------------------------------------------------------------------------------
 * "Next" row N3a: HandT42::removeSurroundingPointsAndAssignProbability (Hand.cpp:779-888), the step
 * between the hand search and the generator (main_realdata_auto.cpp:142-148): scene points near any hand link are
 * removed, the others get the confidence 1 - exp(-lambda d) of their distance to the hand, points on the outer side of
 * the distal fingers are removed, and the survivors return in the camera frame.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float* xyz;     /* link cloud in the hand-base frame (Hand::makeHandCloud, Hand.cpp:537-556), SoA planes x[n] y[n] z[n] */
  int n;
  float sq_dist_thres;  /* local_dist_thres of this link (Hand.cpp:812-821), squared metres */
} hop_hand_link;
/* scene_xyz/scene_nrm: camera frame, SoA planes of n points.  links: in the reference's std::map (name) order.
 * Outputs: SoA planes of capacity n (plane stride n), the first *n_out entries valid, in input order (the reference's
 * order depends on OpenMP scheduling); keep_index[k] = input index of survivor k (may be NULL). */
int hop_hand_remove_surrounding(hop_ctx* ctx, const float* scene_xyz, const float* scene_nrm, int n,
                                const float handbase_in_cam[16], const hop_hand_link* links, int n_links,
                                const float finger12_in_handbase[16], const float finger22_in_handbase[16],
                                float finger12_min_z, float* out_xyz, float* out_nrm, float* out_conf,
                                int* keep_index, int* n_out);

/* ------------------------------------------------------------------------------------------------
 * "Next" row N4, the PPF key table of a model: the pair loop of the offline tool
 * src/perception/src/app/computePPF.cpp:17-38,88-100 -- computePPF(points[i], points[j]) for every i < j on the 5 mm
 * model cloud with its normals (normalised once, as there) -- so that the table runSuper4pcs consumes
 * (hop_set_ppf_keys) can be regenerated for any object.  The tool's voxel / MLS-normal preprocessing is not part of it.
 * keys4_out: up to `cap` rows of 4 int32 (dist mm, three angles in degrees), sorted lexicographically, unique;
 * *n_keys is the number of distinct valid keys (HOP_E_CAPACITY if it exceeds cap).
 * ---------------------------------------------------------------------------------------------- */
int hop_model_ppf_keys(hop_ctx* ctx, const float* xyz, const float* nrm, int n, int32_t* keys4_out, int cap, int* n_keys);

/* ------------------------------------------------------------------------------------------------
 * "Next" row N1: physics rejection.
 *   SDFchecker::registerMesh            src/perception/src/SDFchecker.cpp:36-78    -> hop_sdf_register_mesh
 *   SDFchecker::getSignedDistanceMinMaxWithRegistered   SDFchecker.cpp:115-134      -> hop_sdf_signed_distance
 *     (igl::signed_distance, SIGNED_DISTANCE_TYPE_PSEUDONORMAL, float, bounds -FLT_MAX / FLT_MAX as every call site
 *      passes them: a point at distance exactly 0 yields NaN, include/igl/signed_distance.cpp:117-156)
 *   Utils::downsamplePointCloud (pcl::VoxelGrid, xyz)   src/perception/src/Utils.cpp:334-340 -> hop_voxel_downsample
 *   PoseEstimator::registerHandMesh / rejectByCollisionOrNonTouching   PoseEstimator.cpp:503-520, 524-735
 *                                                                       -> hop_physics_set_frame + hop_reject_by_collision
 * Meshes: V is nv x 3 row-major, F is nf x 3 vertex indices (triangles); pose16 (row-major, may be NULL) is applied to
 * the vertices at registration as the reference does.  Registration prepares libigl's face / edge / vertex
 * pseudonormals and a box tree on the host (the counterpart of loading the OBJ file); every query runs on the GPU.
 * Clouds are SoA planes (x plane, y plane, z plane of n floats) like everywhere in this header.
 * ---------------------------------------------------------------------------------------------- */
#define HOP_SDF_MAX_MESHES 16
int hop_sdf_register_mesh(hop_ctx* ctx, int mesh_id, const float* V, int nv, const int32_t* F, int nf, const float* pose16);
/* SDFchecker::transformMesh (SDFchecker.cpp:81-87) as an absolute pose relative to the registered vertices: the mesh is
 * not rebuilt, queries are moved by the inverse pose instead (distances agree with re-registering under the pose to the
 * rounding of the motion; where libigl's sign hinges on a tie between faces, the tie can break differently).  pose16 NULL:
 * back to the registered vertices. */
int hop_sdf_set_mesh_pose(hop_ctx* ctx, int mesh_id, const float* pose16);
/* dists[n] (NaN for points on the surface); faces[n] (may be NULL): closest face, nf + 1 where dists is NaN;
 * *min_dist / *max_dist: S.minCoeff() / S.maxCoeff() over the non-NaN entries (FLT_MAX / -FLT_MAX when n == 0). */
int hop_sdf_signed_distance(hop_ctx* ctx, int mesh_id, const float* pts_xyz, int n, float* dists, int32_t* faces, float* min_dist,
                            float* max_dist);
/* out_xyz: SoA planes with plane stride cap; centroids in ascending voxel-index order (x fastest), as pcl::VoxelGrid
 * emits them.  HOP_E_CAPACITY when *n_out > cap or when the grid would overflow an int (PCL warns and gives up there). */
int hop_voxel_downsample(hop_ctx* ctx, const float* xyz, int n, float leaf, float* out_xyz, int cap, int* n_out);
typedef struct {
  int object_mesh;    /* registered at identity: est.registerMesh(object_mesh_path, "object", I), main_realdata_auto.cpp:187 */
  int finger_mesh[4]; /* registered at getTFHandBase(link); order finger_1_1, finger_1_2, finger_2_1, finger_2_2 */
  /* hand->_clouds[finger] (link frame), getTFHandBase(finger), hand->_component_status[finger] */
  const float* finger_xyz[4];
  int finger_n[4];
  const float* finger2handbase[4]; /* row-major 4x4 */
  int finger_status[4];
  const float* hand_cloud_xyz; /* hand->_hand_cloud, hand-base frame */
  int n_hand_cloud;
  const float* cloud_without_hand_xyz; /* _cloud_withouthand_raw, camera frame */
  int n_cloud_without_hand;
  float cam2handbase[16]; /* hand->_handbase_in_cam.inverse(), row-major */
  const float* model_xyz; /* _model */
  int n_model;
  float model_center_init[3], smallest_dim, ob_diameter; /* PoseEstimator.cpp:12-20 */
  float collision_thres, non_touch_dist, collision_finger_dist, collision_finger_volume_ratio; /* config_autodataset.yaml:128-131 */
  float voxel_size; /* 0.005, PoseEstimator.cpp:556 */
} hop_physics_args;
/* per-frame inputs of rejectByCollisionOrNonTouching: uploads the clouds, moves the finger clouds and the scene into
 * the hand-base frame, applies the voxel grid */
int hop_physics_set_frame(hop_ctx* ctx, const hop_physics_args* args);
/* filters the resident hypothesis set (hop_hypos_*) in place; survivors keep their relative order.  keep_out[H] and
 * diag8_out[H x 8] (both may be NULL) describe the incoming set of *n_in hypotheses: diag = {deciding check (0 kept,
 * 1 scene point inside the object, 2 hand point colliding, 3 finger cloud colliding, 4 one side not touching,
 * 5 object inside a finger), the two single-point distances, the minimum distance of each finger cloud, the smallest
 * object-to-finger-mesh minimum}; entries of checks that did not run are NaN. */
int hop_reject_by_collision(hop_ctx* ctx, unsigned char* keep_out, float* diag8_out, int* n_in);
/* device time (HIP events on the ctx stream) of the last hop_physics_set_frame / hop_reject_by_collision, in ms */
int hop_physics_timing(hop_ctx* ctx, double* ms_set_frame, double* ms_reject);

/* ------------------------------------------------------------------------------------------------
 * "Next" row N2: PoseEstimator::rejectByRender (src/perception/src/PoseEstimator.cpp:345-463) without OpenGL:
 * Renderer::addObject / doRender (src/perception/src/Renderer.cpp:41-81), the projection of
 * src/depth_sim/src/range_likelihood.cpp:391-449 and the depth read-back of src/depth_sim/src/simulation_io.cpp:411-460
 * restated as a z-buffer rasteriser (hop_render.hip; parity unpinned: no OpenGL exists here to pin coverage against).
 *   hop_render_set_frame    the frame's depth image (Utils::readDepthImage: raw * depth_unit, out of [0.1, 2.0] -> 0), the camera
 *                           intrinsics (Renderer(H, W, fx, fy, cx, cy)) and the hand: the triangles of every hand mesh whose
 *                           component was matched, ALREADY moved by handbase_in_cam * getTFHandBase(name) (:362-383), one soup
 *   hop_render_set_object   the object mesh in the model frame (_obj_mesh); every hypothesis moves it by its pose (:388)
 *   hop_render_depth        Renderer::doRender for one object pose (NULL: hand only): depth in metres (millimetre-rounded,
 *                           clamped to [0.1, 2.0], background 2.0) and owner per pixel (0 nothing, 1 hand, 2 object = the
 *                           blue pixels of the colour image)
 *   hop_reject_by_render    scores every resident hypothesis (_wrong_ratio = roi_weight * roi_diff / roi_cnt + bg_diff / bg_cnt,
 *                           :398-444) and keeps the max(int(keep_ratio * n), 10) smallest, ascending (:447-461; ties by position).
 *                           sum_mode 0: the per-pixel terms are added into one float in row order, as the reference does (the
 *                           result depends on that order); 1: reduced in double.  wrong_ratio_out: n values in the order of
 *                           the set before the call; keep_index_out: positions (before the call) of the survivors.
 * ---------------------------------------------------------------------------------------------- */
int hop_render_set_frame(hop_ctx* ctx, const uint16_t* depth_raw, int H, int W, double depth_unit, const float K9[9], const float* hand_V,
                         int hand_nv, const int32_t* hand_F, int hand_nf);
int hop_render_set_object(hop_ctx* ctx, const float* V, int nv, const int32_t* F, int nf);
int hop_render_depth(hop_ctx* ctx, const float* pose16, float* depth_m_out, unsigned char* owner_out);
int hop_reject_by_render(hop_ctx* ctx, float roi_weight, float keep_ratio, int sum_mode, float* wrong_ratio_out, int* keep_index_out,
                         int* n_keep_out);

/* ------------------------------------------------------------------------------------------------
 * "Next" row N3b: the scene front end of src/perception/src/app/main_realdata_auto.cpp:54-96 without
 * colours and normals -- Utils::readDepthImage (Utils.cpp:36-55, depth_unit = SR300_DEPTH_UNIT), Utils::convert3dOrganizedRGB
 * (Utils.cpp:79-115, K9 = cam_intrinsic row-major), the z pass-through [0.1, 2.0], the voxel grid at `leaf` (0.001),
 * the move into the hand-base frame (cam_in_handbase = handbase_in_cam.inverse()), the three pass-through filters on
 * z, x, y (crop_min / crop_max as x, y, z; main :79-94: x [-0.25,-0.07], y [-0.2,0.2], z [-0.12,0.05]) and the move back.
 * out_xyz: SoA planes with plane stride cap, voxel order; counts3 (may be NULL): valid pixels, points after the voxel
 * grid, points after the crop.
 * ---------------------------------------------------------------------------------------------- */
int hop_scene_from_depth(hop_ctx* ctx, const uint16_t* depth_raw, int H, int W, double depth_unit, const float K9[9],
                         const float cam_in_handbase[16], const float handbase_in_cam[16], float leaf, const float crop_min[3],
                         const float crop_max[3], float* out_xyz, int cap, int* n_out, int* counts3);

/* "Next" row N3g: the two normal estimators of the driver (PCL 1.9, restated from its published sources: parity unpinned).
 *   hop_normals_integral_image   Utils::calNormalIntegralImage (src/perception/src/Utils.cpp:293-329; main_realdata_auto.cpp:61 passes
 *                                method -1 = SIMPLE_3D_GRADIENT, max_depth_change_factor 0.02, smoothing 10, depth-dependent):
 *                                pcl::IntegralImageNormalEstimation on the ORGANISED cloud.  xyz: SoA planes of H*W floats, row-major
 *                                pixels; a dropped pixel is (0,0,0) in the reference (Utils.cpp:92) -- non-finite values are read as 0.
 *                                nrm_out: planes of H*W, NaN where PCL leaves the normal undefined (border of `smoothing` pixels,
 *                                depth edges, zero gradient); normals flipped towards the origin.  H <= 1025.
 *   hop_normals_mls              Utils::calNormalMLS (Utils.cpp:268-289; main :153 passes radius 0.003): pcl::MovingLeastSquares,
 *                                polynomial order 2 (0 / 1: plane only), normals, SIMPLE projection, no upsampling.  Points with fewer than
 *                                3 neighbours in the radius are dropped (mls.process + getCorrespondingIndices + ExtractIndices);
 *                                the survivors come back PROJECTED onto the fitted surface with its normal and PCL's curvature,
 *                                in input order (PCL's OpenMP version emits thread chunks in arrival order); keep_index[k] = input
 *                                index of output k.  Outputs: planes with stride cap.
 *   hop_scene_from_depth_normals hop_scene_from_depth with the integral-image normals computed on the organised cloud and carried
 *                                through the pass-through, the voxel grid (pcl::VoxelGrid averages every field: normals summed and
 *                                normalised per voxel, NaN if any member is NaN) and the crop: the cloud Hand::setCurScene receives
 *                                (main_realdata_auto.cpp:54-96,100). */
int hop_normals_integral_image(hop_ctx* ctx, const float* xyz, int H, int W, float max_depth_change_factor, float normal_smoothing_size,
                               int depth_dependent_smoothing, float* nrm_out);
int hop_normals_mls(hop_ctx* ctx, const float* xyz, int n, float search_radius, int polynomial_order, float* out_xyz, float* out_nrm,
                    float* out_curvature, int* keep_index, int cap, int* n_out);
int hop_scene_from_depth_normals(hop_ctx* ctx, const uint16_t* depth_raw, int H, int W, double depth_unit, const float K9[9],
                                 const float cam_in_handbase[16], const float handbase_in_cam[16], float leaf, const float crop_min[3],
                                 const float crop_max[3], float max_depth_change_factor, float normal_smoothing_size, float* out_xyz,
                                 float* out_nrm, int cap, int* n_out, int* counts3);

/* "Next" row N3c: the generator's input cloud from the dense hand-free cloud (main_realdata_auto.cpp:156-177): voxel grid at
 * `leaf` (0.003) over xyz and normals (pcl::VoxelGrid / CentroidPoint: xyz averaged, normals summed and normalised),
 * removeAllNaNFromPointCloud, pcl::flipNormalTowardsViewpoint(0,0,0), confidence of the nearest dense point (1-NN).
 * Inputs: the dense cloud after Utils::calNormalMLS (SoA planes) with its confidences; outputs with plane stride cap. */
int hop_object_segment(hop_ctx* ctx, const float* xyz, const float* nrm, const float* conf, int n, float leaf, float* out_xyz,
                       float* out_nrm, float* out_conf, int cap, int* n_out);

/* "Next" row N3d: what Hand::setCurScene derives from the 3 mm hand-region cloud after handbaseICP (src/perception/src/Hand.cpp:289-321):
 * the cloud moved into the hand-base frame (pcl::transformPointCloudWithNormals by cam_in_handbase), two
 * pcl::RadiusOutlierRemoval passes (0.02 m / 30 neighbours, 0.04 m / 100), pcl::StatisticalOutlierRemoval (mean_k 20,
 * 2 sigma) and the x pass-through [-0.25, -0.1].  hb_xyz / hb_nrm: `scene_in_handbase` (SoA planes, stride n);
 * keep_noise[i] / keep_swivel[i]: whether input point i is in scene_hand_region_removed_noise / scene_remove_swivel
 * (the three clouds hop_hand_set_scene takes: removed_noise xyz, the normals of scene_in_handbase, remove_swivel xyz). */
int hop_hand_scene_filters(hop_ctx* ctx, const float* xyz, const float* nrm, int n, const float cam_in_handbase[16], float* hb_xyz,
                           float* hb_nrm, unsigned char* keep_noise, unsigned char* keep_swivel);

/* "Next" row N3e: the pieces of Hand::handbaseICP (src/perception/src/Hand.cpp:677-777) that touch clouds.
 *   hop_voxel_downsample_normals  Utils::downsamplePointCloud on a cloud with normals (:680, leaf 0.005): centroids and the
 *                                 normalised normal sums (pcl CentroidPoint), ascending voxel order
 *   hop_handbase_region           the ICP source (:685-729): hand-base transform, pass-through x [-0.07, 0.03] and
 *                                 z [-0.18, 0.01], finger connections removed (y1,z1 / y2,z2 = translation of finger_1_1 /
 *                                 finger_2_1 in their parent); hb_xyz / hb_nrm: every input point in the hand-base frame
 *                                 (planes, stride n), keep[i]: whether it belongs to the source cloud
 * The ICP itself is Utils::runICP (Utils.cpp:188-229), the function refineByICP calls: hop_set_scene (source) /
 * hop_set_model (target: the base_link cloud) / hop_hypos_upload (identity) / hop_icp_refine {50, 30, 0.03}; the offset
 * is the inverse of the refined pose.  The acceptance rules of :740-772 are host logic (host/Hand.h, api.py). */
int hop_voxel_downsample_normals(hop_ctx* ctx, const float* xyz, const float* nrm, int n, float leaf, float* out_xyz, float* out_nrm,
                                 int cap, int* n_out);
int hop_handbase_region(hop_ctx* ctx, const float* xyz, const float* nrm, int n, const float cam_in_handbase[16], float y1, float z1,
                        float y2, float z2, float* hb_xyz, float* hb_nrm, unsigned char* keep);

/* "Next" row N3f: the matching loop of HandT42::adjustHandHeight (src/perception/src/Hand.cpp:1010-1049).  scene: the 3 mm
 * hand-region cloud in the hand-base frame with normals (hb_xyz / hb_nrm of hop_hand_scene_filters); hand: Hand::_hand_cloud
 * with normals (hand-base frame).  counts[t]: hand points whose nearest scene point, after the hand is shifted by heights[t]
 * along z, is within 5 mm with normals within 45 degrees.  The choice of the best height (first maximum, :1042-1047) and
 * the update of _handbase_in_cam (:1050) are host logic. */
int hop_hand_height_matches(hop_ctx* ctx, const float* scene_xyz, const float* scene_nrm, int n_scene, const float* hand_xyz,
                            const float* hand_nrm, int n_hand, const float* heights, int n_heights, int* counts);

/* ------------------------------------------------------------------------------------------------
 * Measurement helpers (bench.py): device time in ms of the kernels launched by the last call of the
 * named stage, measured with HIP events on the ctx stream; and launch counts.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  double ms_verify, ms_gen_other, ms_icp_nn, ms_icp_solve, ms_lcp_fwd, ms_lcp_rev, ms_pso, ms_ppf_matrix;
  long long n_verify_launches, n_icp_nn_launches, n_lcp_launches, n_pso_launches;
  long long pairs_verify, pairs_icp, pairs_lcp, pairs_pso; /* algorithmic point-pair evaluations */
  /* per-kernel spans of the cell-list paths: ms_icp_nn is the correspondence kernel alone when nn_mode >= 2
   * (ms_icp_accum the normal-equation kernel), ms_lcp_fwd the fused computeLCP kernel, ms_lcp_sum the ordered sum,
   * ms_quads the congruent-quadrilateral kernel (also included in ms_gen_other), ms_build the per-frame
   * acceleration structures (grids, NN cell lists). */
  double ms_icp_accum, ms_lcp_sum, ms_quads, ms_build;
  long long n_quads_launches, n_build_launches;
} hop_timing;
int hop_timing_reset(hop_ctx* ctx);
int hop_timing_get(hop_ctx* ctx, hop_timing* out);
int hop_timing_enable(hop_ctx* ctx, int on);

#ifdef __cplusplus
}
#endif
#endif /* HOP_H_ */
