// hop_device.h -- argument blocks and record layouts shared by the kernels (hop_kernels.hip) and the
// host side of libhop (hop_ctx.hip).  Plain structs, passed to kernels by value.
#ifndef HOP_DEVICE_H_
#define HOP_DEVICE_H_

#include <hip/hip_runtime.h>
#include "hop_math.h"

namespace hop {

constexpr int NN_TILE = 2048;  // targets per LDS tile (float4 each -> 32 KiB)
constexpr int NN_CH = 16;      // targets per min-chunk of the scan
constexpr int PPF_ROWS = 64;   // rows of the PPF matrix handled per block
constexpr int ICP_NACC = 32;   // 21 (lower triangle of J^T J) + 6 (J^T r) + mse + count + 3 (sum of matched source points)
constexpr int FIT_QUEUES = 64;    // sub-queues between the quadrilateral test and fit stages (power of two)
constexpr int ICP_NMOM = 74;         // nn_mode 6: 73 distinct entries of the 13 x 13 moment matrix + the sum of squared correspondence distances
constexpr int ICP_NMOM_STRIDE = 76;  // + the accepted count, padded
constexpr int ICP_NMOMI = 92;         // nn_mode 7: the 91 entries of the lower triangle of the integer moment matrix + the gridded squared distances
constexpr int ICP_NMOMI_STRIDE = 94;  // (64-bit words per block and hypothesis) + the accepted count, padded
constexpr int ICP_MOM_BITS = 12;      // nn_mode 7: |gridded component| <= 2^12, products <= 2^24: a lane may add 127 of them in 32 bits (it adds <= 2 ICP_ACCUM_R)
constexpr int ICP_ACCUM_R = 32;  // points per thread of the accumulation kernel of the cell-list path
constexpr float GRID_MARGIN = 1.0e-5f;  // metres; bounds | ||T^-1 s - m|| - ||s - T m|| | for rigid float poses
constexpr int MAX_RING = 64;   // samples on the normal cone (normalset.hpp:208-210; <= 2*ceil(2*pi*atan(pi)*3.5) = 56)

// one base (4 points of P) of the generator, prepared by the host for a batch
struct BaseDev {
  float bpos[4][3];  // centred P coordinates of base_3D_[0..3]
  float dist1, dist2;  // |b0-b1|, |b2-b3| (match4pcsBase.hpp:250-251)
  float inv1, inv2;    // quadrilateral invariants (TryQuadrilateral)
  EdgeFeat e1, e2;     // pairPPFisGood features of the two base edges
  int nb_sample;       // cone samples (normalset.hpp:208-210)
  float ring[MAX_RING][3];
};

struct PpfMatrixArgs {
  const float *x, *y, *z, *nx, *ny, *nz;  // centred P, normals already "PPF-normalised"
  int n, words;                            // words = ceil(n/64)
  const unsigned* bitmap;                  // key set as a direct-address bitmap
  int dist_bins;
  unsigned long long* out;  // n x words
  const float* angle_thr;   // 32 cosine thresholds of the angle bins (ppf_angle_bin_thr), or null: literal acosf path
};

struct PairArgs {
  const BaseDev* bases;
  const float *qx, *qy, *qz, *qnx, *qny, *qnz;  // sampled, centred Q
  int nq;
  float eps;  // distance_factor * delta
  unsigned *pairs1, *pairs2;  // [nbases][cap] packed (a<<16|b)
  int *cnt1, *cnt2;
  int cap;
  int* overflow;
};

struct QuadElem {  // per first pair
  short cx, cy, cz, nid;
  float px, py, pz;
};
struct QuadQuery {  // per second pair
  short cx, cy, cz, pad;
  float px, py, pz;
  unsigned mask[11];  // 343 normal bins
};

struct QuadPrepArgs {
  const BaseDev* bases;
  const float *qx, *qy, *qz;  // sampled centred Q (world)
  const float *ux, *uy, *uz;  // same points in the unit cube (pairCreationFunctor.h:104-108,155-157)
  const unsigned *pairs1, *pairs2;
  const int *cnt1, *cnt2;
  int cap;
  NsetGeom geom;
  QuadElem* elems;
  QuadQuery* queries;
};

struct Candidate {
  float T[12];  // centred-frame transform (rows 0..2)
  float c1[3], c2[3];
  unsigned long long key;
};

struct QuadArgs {
  const BaseDev* bases;
  const float *qx, *qy, *qz;
  const unsigned *pairs1, *pairs2;
  const int *cnt1, *cnt2;
  int cap;
  NsetGeom geom;
  const QuadElem* elems;
  const QuadQuery* queries;
  float dist_thr2;  // distance_threshold2 (unsquared delta)
  float delta;
  int base_index0;
  Candidate* cands;
  int* cand_counts;
  int* cand_count;
  int cand_cap;
  int* nquads;  // per base: congruent quadrilaterals found (before the rigid-fit gate)
  int* overflow;
  int4* fit_queue;  // [FIT_QUEUES][fit_cap]: (base, first pair, second pair) of the quadrilaterals that passed the cheap tests
  int* fit_count;   // [FIT_QUEUES]
  int fit_cap;
};

struct VerifyArgs {
  const float *px, *py, *pz;  // centred P
  int np;
  const float *qx, *qy, *qz;  // sampled centred Q
  int nq;
  const float* T;  // transforms, t_stride floats apart, first 12 floats = rows 0..2
  int t_stride;
  const int* n_cand_ptr;  // device counter (or null -> n_cand)
  int n_cand, cand_cap;
  float sq_eps;
  int* counts;
};

struct GridDev {
  float ox, oy, oz, inv_cell;
  float cell;
  int dx, dy, dz;
  const int* cell_start;  // dx*dy*dz + 1
  const float4* pts;      // points sorted by cell; .w carries the original index (int bits)
};

// "NN cell lists": for every voxel of the target's (padded) bounding box, the points that can be the nearest
// neighbour of SOME query inside that voxel (and lie within the gating distance).  One lookup + a short
// contiguous list replaces the ring walk.
struct CellListDev {
  float ox, oy, oz, inv_cell, cell;
  int dx, dy, dz;
  const int* start;   // dx*dy*dz + 1
  const int2* range;  // dx*dy*dz: (start[c], start[c+1]) side by side -- what the lookups read
  const float4* pts;  // concatenated lists; .w = original index (int bits)
  const float4* nrm;  // normals of the same entries (or null)
  // head[cell] (computeLCP's reduced-sum kernel; or null) = the list's FIRST entry inline: x, y, z as float bits and
  // (list position | min(length, 255) << 24): a list of one entry -- most of the 1 mm lists -- costs one 16-byte access instead of the
  // range record plus the entry (cell_list_heads, k_cell_heads)
  const uint4* head;
  float gox, goy, goz;     // -origin * inv_cell: grid coordinate = fma(q, inv_cell, go)
  const float4* nrm_idx;   // normals of the cloud by ORIGINAL index (AoS copy; 16 B per point, cache resident)
  const float4* pts_idx;   // points of the cloud by original index (AoS copy)
  // PACKED lists (ICP model lists, cells_nnq).  Entries are 8 bytes: the three coordinates quantised to 16 bits in a
  // per-cell frame (origin = cell corner - q_R, step = (cell + 2 q_R) / 65535) and the original index in 16 bits.  Lists
  // are stored as 16-byte chunks of two entries, an even number of chunks per list (unused slots are empty:
  // 0xFFFFFFFF 0xFFFFFFFF, coordinates farther than any real entry); rec[cell] = (first chunk, chunks).
  const uint2* rec;
  const uint4* qlist;
  float q_cs, q_rs;        // local query coordinate in whole steps = (int)fma(fraction of the grid coordinate, q_cs, q_rs)
  float q_step2;           // step^2: squared step-unit distances -> m^2
  float q_eq;              // bound of the position error of a dequantised entry + that of the rounded query (metres)
  float q_sa, q_sb;        // S = f2 q_sa + q_sb >= sqrt(f2): the root-free bound in cells_nnq's tolerance (q_sa = 1 / (2 c), q_sb = c / 2)
  float q_tk, q_t0;        // 4.2 q_eq and 2.1 q_eq^2 + 2 step^2 of that tolerance
};
constexpr unsigned Q_EMPTY_HI = 0xFFFF0000u;  // high word >= this: empty slot

struct CellListBuildArgs {
  const float *x, *y, *z;
  int n;
  float ox, oy, oz, cell;
  int dx, dy, dz;
  float max_dist, margin;
  float* u2;      // per cell: min over points of the squared farthest-corner distance
  int* count;     // per cell
  const int* start;
  float4* pts;
  const float *nx, *ny, *nz;  // optional normals of the cloud ...
  float4* nrm;                // ... copied next to pts (null: none)
  float dom_eps;              // domination margin on squared distances
  int pair_max;               // per-frame lists: the pairwise pruning pass runs for at most this many survivors (longer: kept as they are)
};

struct EmitArgs {
  const Candidate* cands;
  const int* cand_counts;
  const int* cand_count;
  int cand_cap;
  float cp[3], cq[3];
  int nq;
  float* pose;
  float* score;
  unsigned long long* key;
  unsigned* inv_count;
  int* hyp_count;
  int hyp_cap;
  int* cand_total;  // running sum of candidates over batches
  int* overflow;
};

struct LcpArgs {
  const float *sx, *sy, *sz, *snx, *sny, *snz;
  int ns;
  const float *mx, *my, *mz, *mnx, *mny, *mnz;
  int nm;
  const float* pose;
  int h0;
  float dist, cos_thres;
  int* rev_idx;
  float* rev_d2;
  float* terms;
  float* score;
  GridDev model_grid;  // nn_mode 1: model in its rest frame, cell >= dist + margin
  GridDev scene_grid;  //            scene, cell >= dist + margin
  CellListDev model_cells;  // nn_mode 2: NN cell lists of the model (rest frame), max_dist = dist
  CellListDev scene_cells;  //            and of the scene
  // grid path walks the scene in a spatially sorted order (neighbouring lanes touch neighbouring cells);
  // `perm` maps the sorted position back to the caller's index, where the terms are stored.
  const float *qx, *qy, *qz, *qnx, *qny, *qnz;
  const int* perm;
  const float* pose_inv;  // [H][12] inverse poses (cell-list path)
  const int* inv_perm;  // caller index -> sorted position (cell-list path: the term table is indexed by sorted position)
  int tiles_per_wave;   // nn_mode 3: 64-point tiles a wavefront adds before it reduces (lcp_tiles_per_wave), ONE value for every batch of a call
};

struct IcpState {
  float T_inc[12];
  float final_tf[16];
  double mse_prev;
  int iterations, active, converged, pad;
};

// nn_mode 5: state of Eigen::LevenbergMarquardt::minimize for one hypothesis (csrc/hop_icp_lm.hip)
constexpr float LM_SQRT_EPS_F = 3.4526698300124393e-04f;  // sqrt(FLT_EPSILON) in float: ftol, xtol, NumericalDiff's eps
struct LmDev {
  float x[6], xc[6], p[6], h[6];  // accepted parameters, candidate to evaluate, last step, forward-difference steps at xc
  float W[7][12];                 // warp matrices at xc and at xc + h_j e_j
  double A[21], g[6], ff;         // J^T J (packed lower triangle), J^T f, |f|^2 at x
  double diag[6], delta, par, xnorm, fnorm, gnorm, pnorm;
  double mse_sum;                 // sum of squared correspondence distances of this ICP iteration
  int iter, nfev, status, phase, cnt, waiting;
  static constexpr bool fast_lmpar = false;  // nn_mode 5 keeps the oracle's operation sequence
  static constexpr bool regs_lmpar = false;  // ... and lmpar2 in its general, pivoted form (the oracle's lm_par)
};

struct IcpArgs {
  const float *sx, *sy, *sz, *snx, *sny, *snz;
  int ns;
  const float *mx, *my, *mz, *mnx, *mny, *mnz;
  int nm;
  float* pose;
  int h0;
  int iter, max_iter;
  float max_d2, cos_thr;
  float* moved;     // [hb][6][ns]
  double* partial;  // [hb][blocks][ICP_NACC]
  IcpState* state;
  GridDev model_grid;  // nn_mode 1: model in its rest frame
  int max_ring;        // rings needed to cover max_corr_dist
  CellListDev cells;   // nn_mode 2: NN cell lists of the model in its rest frame
  int* corr_idx;       // nn_mode 2: [hb][ns] list position of the accepted correspondence (or -1)
  float* hist;         // nn_mode 2: [hb][max_iter][12] increments solved so far
  const float* pose_inv;  // nn_mode 2: [H][12] inverse of the input poses
  const float4 *s_pts4, *s_nrm4;  // nn_mode 3/4: the Morton-ordered source as AoS float4 (two 16-byte loads per point)
  LmDev* lm;                      // nn_mode 5: [hb]
  // nn_mode 7: the grid of the moment form -- powers of two (as floats, and their exponents) that scale n_a p'_b, n_a, r0 and the squared
  // correspondence distance to integers, and the clamps (oracle: mom_spec)
  float mom_s_np, mom_s_n, mom_s_r, mom_s_d, mom_lim, mom_lim_d;
  int mom_k_np, mom_k_n, mom_k_r, mom_k_d;
};

struct PsoParticle {
  float T[12];     // cur_model2handbase
  float Tinv[12];  // its inverse
  int skip, pad[3];
};

struct PsoArgs {
  const PsoParticle* particles;
  const float *mx, *my, *mz, *mnx, *mny, *mnz;
  int nm;
  const float *sx, *sy, *sz;
  int ns;
  const float *lnx, *lny, *lnz;
  int n_lookup;
  const float *wx, *wy, *wz;
  int n_swivel;
  float dist_thres, cos_normal_thres;
  int check_normal;
  float fp_min_z, fp_stride_z;
  int fp_num_division;
  const float* hist_min_y;
  GridDev scene_grid;  // hand scene on a voxel grid (cell = dist_thres/2): exact NN within dist_thres by ring search
  int max_ring;
  int use_grid;             // 0 brute force, 1 ring search on scene_grid, 2 NN cell lists
  CellListDev scene_cells;  // hand scene NN cell lists (max_dist = dist_thres)
  int n_particles;
  int sum_mode;  // 0: outer-side terms added in scene order (the reference's float sum), 1: block reduction
  int* match_count;
  float* outer_terms;  // [n_swivel][n_particles]
  float* outer_sum;
  int* outer_cnt;
};

struct SurroundArgs {
  const float *sx, *sy, *sz, *snx, *sny, *snz;  // scene, camera frame
  int n;
  float cam2hb[12], hb2cam[12], f1inv[12], f2inv[12];
  const float4* link_pts;   // all links concatenated, hand-base frame; .w unused
  const int* link_start;    // n_links + 1
  const float* link_thres;  // squared distance threshold per link
  int n_links;
  float min_z;
  float* hbp;               // [6][n] scratch: point and normal in the hand-base frame
  float* conf;              // [n]
  int* keep;                // [n] 0/1
};
struct SurroundOutArgs {
  const float* hbp;
  const float* conf;
  const int *keep, *pos;    // flag and its exclusive scan
  int n;
  float hb2cam[12];
  float *ox, *oy, *oz, *onx, *ony, *onz, *oconf;
  int* oindex;
};
void launch_model_ppf_keys(const float* x, const float* y, const float* z, const float* nx, const float* ny, const float* nz, int n,
                           int dist_bins, unsigned* bitmap, int* overflow, hipStream_t s);
void launch_hand_surround(const SurroundArgs& a, hipStream_t s);
void launch_hand_surround_out(const SurroundOutArgs& a, hipStream_t s);

void icp_counters_read(unsigned long long* out8, bool reset);
void lm_counters_read(unsigned long long* out8, bool reset);
// launchers (hop_kernels.hip)
void launch_ppf_matrix(const PpfMatrixArgs& a, hipStream_t s);
void launch_pairs(const PairArgs& a, int nbases, hipStream_t s);
void launch_quad_prep(const QuadPrepArgs& a, int nbases, int max_items, hipStream_t s);
void launch_quads(const QuadArgs& a, int nbases, int blocks_per_base, hipStream_t s);
void launch_verify(const VerifyArgs& a, int mode, const GridDev* gd, int blocks, hipStream_t s);
void launch_emit(const EmitArgs& a, int blocks, hipStream_t s);
void launch_gather_hypos(const unsigned* perm, int n, const float* pose_in, const float* score_in, float* pose_out,
                         float* score_out, int* id_out, hipStream_t s);
void launch_iota(unsigned* p, int n, hipStream_t s);
void launch_topk_pack(const unsigned* order, int H, int k, int id_offset, const float* pose, const float* score, const int* ids, float* rows, hipStream_t s);
void launch_score_keys(const float* score, const int* ids, int n, unsigned long long* key, hipStream_t s);
void launch_lcp_reverse(const LcpArgs& a, int hb, hipStream_t s);
void launch_lcp_forward(const LcpArgs& a, int hb, hipStream_t s);
void launch_lcp_sum(const LcpArgs& a, int hb, hipStream_t s);
void launch_lcp_grid(const LcpArgs& a, int hb, hipStream_t s);
void launch_lcp_cells(const LcpArgs& a, int hb, hipStream_t s);
void launch_lcp_sum_t(const LcpArgs& a, int hb, hipStream_t s);
int lcp_cells_row_stride(int hb);
int lcp_tiles_per_wave(int ns, int H);
void launch_lcp_cells_fast(const LcpArgs& a, int hb, hipStream_t s);
void launch_lcp_sum_partial(const LcpArgs& a, int hb, hipStream_t s);
void launch_cell_list_local_flag(const CellListBuildArgs& a, const GridDev& g, int* flag, hipStream_t s);
void launch_cell_list_local_work(const int* flag, const int* flag_scan, int ncell, int* work, hipStream_t s);
void launch_cell_list_local(const CellListBuildArgs& a, const GridDev& g, bool write, int exist_mode, const int* work, int nwork, int* keep_buf,
                            int lanes, hipStream_t s);
int cell_list_local_keep();
void launch_pose_inverse(const float* pose, int n, float* inv12, hipStream_t s);
void launch_unit_normals(const float* nx, const float* ny, const float* nz, int n, float* ux, float* uy, float* uz, hipStream_t s);
void launch_cell_ranges(const int* start, int ncell, int2* range, hipStream_t s);
void launch_verify_cells(const VerifyArgs& a, const CellListDev& cl, int blocks, hipStream_t s);
void launch_lcp_cells(const LcpArgs& a, int hb, hipStream_t s);
void launch_lcp_sum_t(const LcpArgs& a, int hb, hipStream_t s);
int lcp_cells_row_stride(int hb);
int icp_blocks_per_hyp(int ns, bool cells);
void launch_icp_fusedq_momi(const IcpArgs& a, int hb, hipStream_t s);
void launch_icp_fusedq_momm(const IcpArgs& a, int hb, hipStream_t s);  // the same sums on the matrix cores
void launch_cell_heads(const int2* range, const float4* pts, int ncell, uint4* head, hipStream_t s);
void lcp_counters_read(unsigned long long* out4, bool reset);
void launch_dev_selftest_scalar(int n, const float* x, const float* y, const int* ia, const int* ib, const int* ic, unsigned* out, hipStream_t s);
void launch_dev_selftest_mfma(int tiles, const int* a, const int* b, const int* c, int* d, hipStream_t s);
void launch_dev_selftest_momm(int batches, const int* U, const unsigned long long* mask, int* out, hipStream_t s);
void launch_icp_lm7_solve(const IcpArgs& a, int hb, int nblocks, hipStream_t s);
void launch_icp_fusedq_mom(const IcpArgs& a, int hb, hipStream_t s);
void launch_icp_lm6_solve(const IcpArgs& a, int hb, int nblocks, hipStream_t s);
void launch_icp_lm_begin(const IcpArgs& a, int hb, hipStream_t s);
void launch_icp_lm_pass(const IcpArgs& a, int hb, bool first, hipStream_t s);
void launch_icp_lm_solve(const IcpArgs& a, int hb, int nblocks, bool first, unsigned* n_waiting, hipStream_t s);
void launch_icp_init(IcpState* st, int hb, hipStream_t s);
void launch_icp_nn(const IcpArgs& a, int hb, hipStream_t s);
void launch_icp_nn_grid(const IcpArgs& a, int hb, hipStream_t s);
void launch_icp_corr_cells(const IcpArgs& a, int hb, hipStream_t s);
void launch_icp_accum(const IcpArgs& a, int hb, hipStream_t s);
void launch_icp_fused(const IcpArgs& a, int hb, hipStream_t s);
void launch_icp_fusedq(const IcpArgs& a, int hb, bool composed, hipStream_t s);
void launch_soa_to_aos4(const float* x, const float* y, const float* z, int n, float4* out, hipStream_t s);
void launch_cell_list_bounds(const CellListBuildArgs& a, hipStream_t s);
void launch_cell_list_count(const CellListBuildArgs& a, hipStream_t s);
void launch_cell_list_fill(const CellListBuildArgs& a, hipStream_t s);
void launch_icp_solve(const IcpArgs& a, int hb, int nblocks, hipStream_t s);
void launch_icp_finish(const IcpArgs& a, int hb, int* iters, int* conv, hipStream_t s);
void launch_pso(const PsoArgs& a, int n_particles, hipStream_t s);
void launch_grid_cell_ids(const float* x, const float* y, const float* z, int n, const GridDev& gd, int* cell_of, int* cell_count,
                          hipStream_t s);

}  // namespace hop
#endif
