"""TEST INFRASTRUCTURE ONLY: ctypes bindings of the CPU oracle (liboracle.so) and, when it has been
built in the build container, of the reference's own OpenGR fork (oracle/_ref/libref_s4pcs.so).

Importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  Never the product.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from contextlib import contextmanager

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libref_s4pcs.so")
REF_SDF_SO = os.path.join(HERE, "_ref", "libref_sdf.so")
REF_ICP_SO = os.path.join(HERE, "_ref", "libref_icp.so")
REF_ICP_NATIVE_SO = os.path.join(HERE, "_ref", "libref_icp_native.so")  # same source, -march=native: build-to-build spread of the reference

fp = C.POINTER(C.c_float)
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


def F(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(fp)


def I(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ip)


def D(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(dp)


def soa(a):
    """(n,3) -> contiguous (3,n) float32 planes."""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).T)


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < max(
            os.path.getmtime(os.path.join(HERE, f)) for f in ("hop_oracle.cpp", "sdf_oracle.cpp", "normals_oracle.cpp", "render_oracle.cpp", "hop_oracle.h")):
        subprocess.check_call(["make", "-C", HERE, "_build/liboracle.so"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


@contextmanager
def quiet_stdout():
    """The reference headers print to stdout from C++ (match4pcsBase.hpp:264-265 etc.)."""
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    try:
        yield
    finally:
        try:
            C.CDLL(None).fflush(None)   # what the C++ side left in stdio's buffer goes to /dev/null too, not to the caller's stdout at exit
        except (OSError, AttributeError):
            pass
        os.dup2(saved, 1)
        os.close(devnull)
        os.close(saved)


class S4Opts(C.Structure):
    _fields_ = [("sample_size", C.c_int), ("overlap", C.c_float), ("delta", C.c_float), ("dispersion", C.c_float),
                ("success_quadrilaterals", C.c_int), ("n_trials", C.c_int), ("random_seed", C.c_uint)]


class RefOpts(C.Structure):
    _fields_ = [("sample_size", C.c_int), ("overlap", C.c_float), ("delta", C.c_float), ("dispersion", C.c_float),
                ("success_quadrilaterals", C.c_int), ("max_time_seconds", C.c_int),
                ("max_normal_difference", C.c_float), ("max_color_distance", C.c_float)]


class FingerArgs(C.Structure):
    _fields_ = [
        ("fp_min", C.c_float * 3), ("fp_max", C.c_float * 3), ("fp_stride_z", C.c_float), ("fp_num_division", C.c_int),
        ("fp_hist_min_y", fp), ("fo_min", C.c_float * 3), ("fo_max", C.c_float * 3),
        ("model2handbase", C.c_float * 16), ("finger_out2parent", C.c_float * 16),
        ("pair_tip1", C.c_float * 4), ("pair_tip2", C.c_float * 4), ("is_palm_side", C.c_int),
        ("is_right_side", C.c_int), ("gripper_min_dist", C.c_float), ("dist_thres", C.c_float),
        ("cos_normal_thres", C.c_float), ("check_normal", C.c_int), ("max_outter_pts", C.c_int),
        ("outter_pt_dist", C.c_float), ("outter_pt_dist_weight", C.c_float),
        ("model_xyz", fp), ("model_nrm", fp), ("n_model", C.c_int),
        ("scene_xyz", fp), ("n_scene", C.c_int), ("scene_nrm_lookup", fp), ("n_lookup", C.c_int),
        ("swivel_xyz", fp), ("n_swivel", C.c_int),
    ]


class PsoSettings(C.Structure):
    _fields_ = [("n_pop", C.c_int), ("n_gen", C.c_int), ("check_freq", C.c_int), ("c_cog", C.c_double),
                ("c_soc", C.c_double), ("initial_w", C.c_double), ("w_min", C.c_double), ("w_max", C.c_double),
                ("err_tol", C.c_double), ("lower_rad", C.c_double), ("upper_rad", C.c_double), ("seed", C.c_uint64)]


class PhysicsArgs(C.Structure):
    _fields_ = [("object_V", fp), ("object_nv", C.c_int), ("object_F", ip), ("object_nf", C.c_int),
                ("finger_V", fp * 4), ("finger_nv", C.c_int * 4), ("finger_F", ip * 4), ("finger_nf", C.c_int * 4),
                ("finger_mesh_pose", fp * 4),
                ("finger_xyz", fp * 4), ("finger_n", C.c_int * 4), ("finger2handbase", fp * 4), ("finger_status", C.c_int * 4),
                ("hand_cloud_xyz", fp), ("n_hand_cloud", C.c_int),
                ("cloud_without_hand_xyz", fp), ("n_cloud_without_hand", C.c_int),
                ("cam2handbase", C.c_float * 16),
                ("model_xyz", fp), ("n_model", C.c_int),
                ("model_center_init", C.c_float * 3), ("smallest_dim", C.c_float), ("ob_diameter", C.c_float),
                ("collision_thres", C.c_float), ("non_touch_dist", C.c_float), ("collision_finger_dist", C.c_float),
                ("collision_finger_volume_ratio", C.c_float), ("voxel_size", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        L.orc_acosf.restype = C.c_float
        L.orc_acosf.argtypes = [C.c_float]
        L.orc_s4pcs_create.restype = C.c_void_p
        L.orc_s4pcs_create.argtypes = [C.POINTER(S4Opts)]
        L.orc_s4pcs_destroy.argtypes = [C.c_void_p]
        L.orc_s4pcs_set_keys.argtypes = [C.c_void_p, ip, C.c_int]
        L.orc_s4pcs_run.argtypes = [C.c_void_p, fp, fp, fp, C.c_int, fp, fp, C.c_int, C.c_int]
        L.orc_s4pcs_num_hypos.argtypes = [C.c_void_p]
        L.orc_s4pcs_get_hypos.argtypes = [C.c_void_p, fp, fp]
        L.orc_s4pcs_num_bases.argtypes = [C.c_void_p]
        L.orc_s4pcs_get_base.argtypes = [C.c_void_p, C.c_int, ip, fp, ip]
        L.orc_s4pcs_get_base_lists.argtypes = [C.c_void_p, C.c_int, ip, ip, ip]
        L.orc_s4pcs_num_sampled_q.argtypes = [C.c_void_p]
        L.orc_s4pcs_get_state.argtypes = [C.c_void_p, fp, fp, fp, fp, fp, ip]
        L.orc_s4pcs_verify.restype = C.c_float
        L.orc_s4pcs_verify.argtypes = [C.c_void_p, fp]
        L.orc_verify_batch.argtypes = [fp, C.c_int, fp, C.c_int, fp, C.c_int, C.c_float, C.c_int, ip]
        L.orc_compute_lcp_batch.argtypes = [fp, fp, C.c_int, fp, fp, C.c_int, fp, C.c_int, C.c_float, C.c_float,
                                            C.c_int, fp]
        L.orc_icp_refine_batch.argtypes = [fp, fp, C.c_int, fp, fp, C.c_int, fp, C.c_int, C.c_int, C.c_float,
                                           C.c_float, C.c_int, ip, ip]
        L.orc_cluster_poses.restype = C.c_int
        L.orc_cluster_poses.argtypes = [fp, fp, ip, C.c_int, C.c_float, C.c_float, fp, ip]
        L.orc_pso_objective.restype = C.c_double
        L.orc_pso_objective.argtypes = [C.POINTER(FingerArgs), C.c_double]
        L.orc_pso_objective_batch.argtypes = [C.POINTER(FingerArgs), dp, C.c_int, dp]
        L.orc_pso_search.argtypes = [C.POINTER(FingerArgs), C.POINTER(PsoSettings), dp, dp]
        L.orc_finger_property.argtypes = [fp, C.c_int, C.c_int, fp, fp, fp, fp]
        L.orc_model_ppf_keys.restype = C.c_int
        L.orc_model_ppf_keys.argtypes = [fp, fp, C.c_int, ip, C.c_int]
        L.orc_hand_remove_surrounding.restype = C.c_int
        L.orc_hand_remove_surrounding.argtypes = [fp, fp, C.c_int, fp, C.POINTER(fp), ip, fp, C.c_int, fp, fp, C.c_float, fp, fp, fp, ip]
        L.orc_sdf_signed_distance.argtypes = [fp, C.c_int, fp, C.c_int, ip, C.c_int, fp, C.c_float, C.c_float, fp, ip]
        L.orc_voxel_downsample.argtypes = [fp, C.c_int, C.c_float, fp, C.c_int, ip]
        L.orc_scene_from_depth.argtypes = [C.POINTER(C.c_ushort), C.c_int, C.c_int, C.c_double, fp, fp, fp, C.c_float, fp, fp, fp, C.c_int, ip, ip]
        L.orc_object_segment.argtypes = [fp, fp, fp, C.c_int, C.c_float, fp, fp, fp, C.c_int, ip]
        L.orc_hand_scene_filters.argtypes = [fp, fp, C.c_int, fp, fp, fp, C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte)]
        L.orc_voxel_downsample_normals.argtypes = [fp, fp, C.c_int, C.c_float, fp, fp, C.c_int, ip]
        L.orc_handbase_region.argtypes = [fp, fp, C.c_int, fp, C.c_float, C.c_float, C.c_float, C.c_float, fp, fp, C.POINTER(C.c_ubyte)]
        L.orc_hand_height_matches.argtypes = [fp, fp, C.c_int, fp, fp, C.c_int, fp, C.c_int, ip]
        L.orc_reject_by_collision.argtypes = [C.POINTER(PhysicsArgs), fp, C.c_int, C.POINTER(C.c_ubyte), fp]
        L.orc_compute_ppf.argtypes = [fp, fp, fp, fp, ip]
        L.orc_pair_ppf_is_good.argtypes = [fp, fp, fp, fp]
        L.orc_rigid.argtypes = [fp, fp, fp, fp]
        L.orc_probe_transform.argtypes = [fp, fp, fp]
        L.orc_probe_vec.argtypes = [fp, fp, fp]
        L.orc_probe_quat.argtypes = [fp, fp, fp]
        L.orc_set_lm_estimator.argtypes = [C.c_void_p]
        L.orc_lm_point_to_plane.restype = C.c_int
        L.orc_lm_point_to_plane.argtypes = [C.c_int, fp, fp, fp, fp, fp, ip]
        L.orc_lm_warp6.argtypes = [fp, fp]
        llp = C.POINTER(C.c_longlong)
        L.orc_set_mom_bits.argtypes = [C.c_int]
        L.orc_lm_point_to_plane_moments.argtypes = [dp, dp, fp, fp, ip]
        L.orc_mom_accumulate.argtypes = [C.c_int, fp, fp, fp, fp, fp, C.c_double, C.c_float, C.c_int, llp, llp, dp, ip]
        L.orc_lm_residuals_jacobian.argtypes = [C.c_int, fp, fp, fp, fp, fp, fp]
        L.orc_euler_zyx.argtypes = [fp, fp]
        L.orc_geodesic.restype = C.c_float
        L.orc_geodesic.argtypes = [fp, fp]
        L.orc_tdiff_norm.restype = C.c_float
        L.orc_tdiff_norm.argtypes = [fp, fp]
        L.orc_inverse_times.argtypes = [fp, fp, fp]
        L.orc_mul4.argtypes = [fp, fp, fp]
        _lib = L
    return _lib


_ref = None


def ref_available():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [C.POINTER(RefOpts)]
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_set_ppf_keys.argtypes = [C.c_void_p, ip, C.c_int]
        L.ref_record_pairs.argtypes = [C.c_void_p, C.c_int]
        L.ref_run.argtypes = [C.c_void_p, fp, fp, fp, C.c_int, fp, fp, C.c_int, C.c_int]
        L.ref_num_hypos.argtypes = [C.c_void_p]
        L.ref_get_hypos.argtypes = [C.c_void_p, fp, fp]
        L.ref_num_bases.argtypes = [C.c_void_p]
        L.ref_get_base.argtypes = [C.c_void_p, C.c_int, ip, fp, ip]
        L.ref_get_base_lists.argtypes = [C.c_void_p, C.c_int, ip, ip, ip]
        L.ref_num_sampled_q.argtypes = [C.c_void_p]
        L.ref_get_state.argtypes = [C.c_void_p, fp, fp, fp, fp, fp, ip]
        L.ref_verify.restype = C.c_float
        L.ref_verify.argtypes = [C.c_void_p, fp]
        L.ref_compute_ppf.argtypes = [fp, fp, fp, fp, ip]
        L.ref_pair_ppf_is_good.argtypes = [fp, fp, fp, fp]
        L.ref_rigid.argtypes = [C.c_void_p, fp, fp, fp, fp]
        L.ref_probe_transform.argtypes = [fp, fp, fp]
        L.ref_probe_vec.argtypes = [fp, fp, fp]
        L.ref_probe_quat.argtypes = [fp, fp, fp]
        _ref = L
    return _ref


# ------------------------------------------------------------------------------ generator wrappers
class _GenBase:
    """Shared read-out logic; `self.L` / `self.h` / prefix differ."""

    def hypos(self):
        n = self._call("num_hypos")
        pose = np.zeros((n, 16), np.float32)
        lcp = np.zeros(n, np.float32)
        if n:
            self._call("get_hypos", F(pose), F(lcp))
        return pose.reshape(n, 4, 4), lcp

    def bases(self):
        out = []
        for i in range(self._call("num_bases")):
            b4 = np.zeros(4, np.int32)
            inv = np.zeros(2, np.float32)
            c3 = np.zeros(3, np.int32)
            self._call("get_base", i, I(b4), F(inv), I(c3))
            p1 = np.zeros((max(c3[0], 1), 2), np.int32)
            p2 = np.zeros((max(c3[1], 1), 2), np.int32)
            q = np.zeros((max(c3[2], 1), 4), np.int32)
            self._call("get_base_lists", i, I(p1), I(p2), I(q))
            out.append(dict(base=b4, inv=inv, pairs1=p1[:c3[0]], pairs2=p2[:c3[1]], quads=q[:c3[2]]))
        return out


class OracleS4PCS(_GenBase):
    def __init__(self, sample_size=100, overlap=0.2, delta=0.003, dispersion=0.5, success_quadrilaterals=10,
                 n_trials=0, random_seed=5489):
        self.L = lib()
        o = S4Opts(sample_size, overlap, delta, dispersion, success_quadrilaterals, n_trials, random_seed)
        self.h = C.c_void_p(self.L.orc_s4pcs_create(C.byref(o)))

    def _call(self, name, *a):
        return getattr(self.L, "orc_s4pcs_" + name)(self.h, *a)

    def __del__(self):
        try:
            self.L.orc_s4pcs_destroy(self.h)
        except Exception:
            pass

    def set_keys(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        self._call("set_keys", I(keys), len(keys))

    def run(self, Pxyz, Pnrm, Pprob, Qxyz, Qnrm, n_calls=1):
        P, Pn, Q, Qn = soa(Pxyz), soa(Pnrm), soa(Qxyz), soa(Qnrm)
        pr = np.ascontiguousarray(Pprob, dtype=np.float32)
        return self._call("run", F(P), F(Pn), F(pr), P.shape[1], F(Q), F(Qn), Q.shape[1], n_calls)

    def state(self):
        n = self._call("num_sampled_q")
        q = np.zeros((3, n), np.float32)
        qn = np.zeros((3, n), np.float32)
        cp = np.zeros(3, np.float32)
        cq = np.zeros(3, np.float32)
        d = np.zeros(1, np.float32)
        nf = np.zeros(1, np.int32)
        self._call("get_state", F(q), F(qn), F(cp), F(cq), F(d), I(nf))
        return dict(Qs=q.T.copy(), Qs_nrm=qn.T.copy(), cP=cp, cQ=cq, diameter=float(d[0]), n_quat_fallback=int(nf[0]))

    def verify(self, T):
        T = np.ascontiguousarray(T, dtype=np.float32).reshape(16)
        return float(self._call("verify", F(T)))


class RefS4PCS(_GenBase):
    """The reference's own matcher (build container only)."""

    def __init__(self, sample_size=100, overlap=0.2, delta=0.003, dispersion=0.5, success_quadrilaterals=10,
                 max_time_seconds=10 ** 9, record_pairs=True, plain=False):
        self.L = ref()
        o = RefOpts(sample_size, overlap, delta, dispersion, success_quadrilaterals, max_time_seconds, -1.0, -1.0)
        self.h = C.c_void_p(self.L.ref_create(C.byref(o)))
        self.L.ref_record_pairs(self.h, 1 if record_pairs else 0)
        self.L.ref_use_plain_matcher.argtypes = [C.c_void_p, C.c_int]
        self.L.ref_use_plain_matcher(self.h, 1 if plain else 0)   # plain: the reference's own generateCongruents, no tracing override

    def _call(self, name, *a):
        return getattr(self.L, "ref_" + name)(self.h, *a)

    def __del__(self):
        try:
            self.L.ref_destroy(self.h)
        except Exception:
            pass

    def set_keys(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        self.L.ref_set_ppf_keys(self.h, I(keys), len(keys))

    def run(self, Pxyz, Pnrm, Pprob, Qxyz, Qnrm, n_calls=1):
        P = np.ascontiguousarray(Pxyz, np.float32)
        Pn = np.ascontiguousarray(Pnrm, np.float32)
        Q = np.ascontiguousarray(Qxyz, np.float32)
        Qn = np.ascontiguousarray(Qnrm, np.float32)
        pr = np.ascontiguousarray(Pprob, dtype=np.float32)
        with quiet_stdout():
            return self._call("run", F(P), F(Pn), F(pr), len(P), F(Q), F(Qn), len(Q), n_calls)

    def state(self):
        n = self._call("num_sampled_q")
        q = np.zeros((n, 3), np.float32)
        qn = np.zeros((n, 3), np.float32)
        cp = np.zeros(3, np.float32)
        cq = np.zeros(3, np.float32)
        d = np.zeros(1, np.float32)
        nt = np.zeros(1, np.int32)
        self._call("get_state", F(q), F(qn), F(cp), F(cq), F(d), I(nt))
        return dict(Qs=q, Qs_nrm=qn, cP=cp, cQ=cq, diameter=float(d[0]), number_of_trials=int(nt[0]))

    def verify(self, T):
        T = np.ascontiguousarray(T, dtype=np.float32).reshape(16)
        return float(self._call("verify", F(T)))


# ------------------------------------------------------------------------------ scoring wrappers
def verify_batch(P, Qs, T, delta, use_tree=True):
    Pp, Qp = soa(P), soa(Qs)
    T = np.ascontiguousarray(T, np.float32).reshape(-1, 16)
    out = np.zeros(len(T), np.int32)
    lib().orc_verify_batch(F(Pp), Pp.shape[1], F(Qp), Qp.shape[1], F(T), len(T), delta, int(use_tree), I(out))
    return out


def compute_lcp_batch(S, Sn, M, Mn, poses, dist=0.001, angle_deg=10.0, use_tree=True):
    Sp, Snp, Mp, Mnp = soa(S), soa(Sn), soa(M), soa(Mn)
    T = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    out = np.zeros(len(T), np.float32)
    lib().orc_compute_lcp_batch(F(Sp), F(Snp), Sp.shape[1], F(Mp), F(Mnp), Mp.shape[1], F(T), len(T), dist,
                                angle_deg, int(use_tree), F(out))
    return out


def icp_refine_batch(S, Sn, M, Mn, poses, max_iter=10, angle_deg=45.0, max_corr_dist=0.01, use_tree=True):
    Sp, Snp, Mp, Mnp = soa(S), soa(Sn), soa(M), soa(Mn)
    T = np.ascontiguousarray(poses, np.float32).reshape(-1, 16).copy()
    it = np.zeros(len(T), np.int32)
    cv = np.zeros(len(T), np.int32)
    lib().orc_icp_refine_batch(F(Sp), F(Snp), Sp.shape[1], F(Mp), F(Mnp), Mp.shape[1], F(T), len(T), max_iter,
                               angle_deg, max_corr_dist, int(use_tree), I(it), I(cv))
    return T.reshape(-1, 4, 4), it, cv


def icp_refine_batch_variant(S, Sn, M, Mn, poses, max_iter=10, angle_deg=45.0, max_corr_dist=0.01, minimiser=0, strict_normal=False, relative_stop=True):
    """run_icp with the deviations of the restatement switched (see IcpVariant in hop_oracle.cpp); kd-tree NN."""
    Sp, Snp, Mp, Mnp = soa(S), soa(Sn), soa(M), soa(Mn)
    T = np.ascontiguousarray(poses, np.float32).reshape(-1, 16).copy()
    it = np.zeros(len(T), np.int32)
    cv = np.zeros(len(T), np.int32)
    lib().orc_icp_refine_batch_variant(F(Sp), F(Snp), Sp.shape[1], F(Mp), F(Mnp), Mp.shape[1], F(T), len(T), int(max_iter), C.c_float(angle_deg),
                                       C.c_float(max_corr_dist), int(minimiser), int(bool(strict_normal)), int(bool(relative_stop)), I(it), I(cv))
    return T.reshape(-1, 4, 4), it, cv


# ---- the reference's ICP minimiser (Eigen::LevenbergMarquardt from the reference's vendored Eigen) --------------------------
MIN_GN, MIN_LM_RESTATED, MIN_LM_REF, MIN_LM_EXACT, MIN_LM_MOMENT = 0, 4, 5, 6, 7
_ref_icp = None
_ref_icp_native = None


def ref_icp_available():
    return os.path.exists(REF_ICP_SO)


def ref_icp_use(native=False):
    """Which build of the reference's minimiser liboracle's run_icp (minimiser 5) calls: the default g++ build (SSE2, no
    contraction: the one the goldens come from) or the -march=native build of the same source."""
    global _ref_icp_native
    L = ref_icp()
    if native:
        if _ref_icp_native is None:
            _ref_icp_native = C.CDLL(REF_ICP_NATIVE_SO)
        L = _ref_icp_native
    lib().orc_set_lm_estimator(C.cast(L.ref_lm_point_to_plane, C.c_void_p))


def ref_icp():
    """oracle/_ref/libref_icp.so (oracle/ref_icp_driver.cpp compiled against /root/reference/src/OpenGR_4pcs/3rdparty/Eigen).
    Loading it also hands its minimiser to liboracle's run_icp (minimiser 5)."""
    global _ref_icp
    if _ref_icp is None:
        L = C.CDLL(REF_ICP_SO)
        L.ref_lm_point_to_plane.restype = C.c_int
        L.ref_lm_point_to_plane.argtypes = [C.c_int, fp, fp, fp, fp, fp, ip, fp]
        L.ref_probe_warp6.argtypes = [fp, fp]
        L.ref_probe_residuals.argtypes = [C.c_int, fp, fp, fp, fp, fp]
        L.ref_probe_jacobian.argtypes = [C.c_int, fp, fp, fp, fp, fp]
        L.ref_probe_euler_zyx.argtypes = [fp, fp]
        L.ref_probe_geodesic.restype = C.c_float
        L.ref_probe_geodesic.argtypes = [fp, fp]
        L.ref_probe_tdiff_norm.restype = C.c_float
        L.ref_probe_tdiff_norm.argtypes = [fp, fp]
        L.ref_probe_inverse_times.argtypes = [fp, fp, fp]
        L.ref_probe_mul4.argtypes = [fp, fp, fp]
        lib().orc_set_lm_estimator(C.cast(L.ref_lm_point_to_plane, C.c_void_p))
        _ref_icp = L
    return _ref_icp


def _aos3(a):
    return np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1, 3))


def lm_point_to_plane(src, tgt, nrm, ref=False):
    """One TransformationEstimationLM::estimateRigidTransformation on given correspondences.
    ref=True: Eigen's own LevenbergMarquardt (oracle/_ref); False: the restatement in liboracle.
    Returns (T 4x4, x6, (status, nfev, iter))."""
    a, b, c = _aos3(src), _aos3(tgt), _aos3(nrm)
    T = np.zeros(16, np.float32)
    x = np.zeros(6, np.float32)
    st = np.zeros(3, np.int32)
    if ref:
        fn = np.zeros(1, np.float32)
        rc = ref_icp().ref_lm_point_to_plane(len(a), F(a), F(b), F(c), F(T), F(x), I(st), F(fn))
    else:
        rc = lib().orc_lm_point_to_plane(len(a), F(a), F(b), F(c), F(T), F(x), I(st))
    if rc != 0:
        return None, None, None
    return T.reshape(4, 4), x, tuple(int(v) for v in st)


def lm_warp6(x6, ref=False):
    x = np.ascontiguousarray(x6, np.float32)
    T = np.zeros(16, np.float32)
    (ref_icp().ref_probe_warp6 if ref else lib().orc_lm_warp6)(F(x), F(T))
    return T.reshape(4, 4)


def lm_residuals_jacobian(src, tgt, nrm, x6, ref=False):
    a, b, c = _aos3(src), _aos3(tgt), _aos3(nrm)
    x = np.ascontiguousarray(x6, np.float32)
    f = np.zeros(len(a), np.float32)
    J = np.zeros((len(a), 6), np.float32)
    if ref:
        ref_icp().ref_probe_residuals(len(a), F(a), F(b), F(c), F(x), F(f))
        ref_icp().ref_probe_jacobian(len(a), F(a), F(b), F(c), F(x), F(J))
    else:
        lib().orc_lm_residuals_jacobian(len(a), F(a), F(b), F(c), F(x), F(f), F(J))
    return f, J


def set_mom_bits(bits):
    """grid of the moment form (minimiser 7): |integer| <= 2^bits; 12 (csrc/hop_device.h ICP_MOM_BITS, the oracle's default) is what the GPU's nn_mode 7 uses -- other values: precision experiments"""
    lib().orc_set_mom_bits(int(bits))


def lm_point_to_plane_moments(M, c):
    """Eigen's minimiser from a 13 x 13 moment matrix (the moment form, minimiser 7) -> (T, x, (status, nfev, iter))"""
    Md = np.ascontiguousarray(M, np.float64).reshape(169)
    cd = np.ascontiguousarray(c, np.float64).reshape(3)
    T = np.zeros(16, np.float32)
    x = np.zeros(6, np.float32)
    st = np.zeros(3, np.int32)
    lib().orc_lm_point_to_plane_moments(Md.ctypes.data_as(dp), cd.ctypes.data_as(dp), F(T), F(x), I(st))
    return T.reshape(4, 4), x, tuple(int(v) for v in st)


MOM_BITS = 12  # = csrc/hop_device.h ICP_MOM_BITS = the oracle's g_mom_bits (tests/test_abi_cpu.py keeps the three in step)


def mom_accumulate(src, tgt, nrm, d2, ctr, model_radius, max_corr_dist=0.01, bits=MOM_BITS):
    """integer moment sums of the given correspondences -> (M int64 13x13, d2q, M in metres float64 13x13, (k_np, k_n, k_r, k_d))"""
    a, b, c = (np.ascontiguousarray(v, np.float32).reshape(-1, 3) for v in (src, tgt, nrm))
    d = np.ascontiguousarray(d2, np.float32).reshape(-1)
    ct = np.ascontiguousarray(ctr, np.float32).reshape(3)
    Mi = np.zeros(169, np.int64)
    Md = np.zeros(169, np.float64)
    dq = C.c_longlong(0)
    sc = np.zeros(4, np.int32)
    lib().orc_mom_accumulate(len(a), F(a), F(b), F(c), F(d), F(ct), C.c_double(model_radius), C.c_float(max_corr_dist), int(bits),
                             Mi.ctypes.data_as(C.POINTER(C.c_longlong)), C.byref(dq), Md.ctypes.data_as(C.POINTER(C.c_double)), I(sc))
    return Mi.reshape(13, 13), int(dq.value), Md.reshape(13, 13), tuple(int(v) for v in sc)


def icp_refine_batch_lm(S, Sn, M, Mn, poses, max_iter=10, angle_deg=45.0, max_corr_dist=0.01, ref=False, exact=False, moment=False):
    """refineByICP with the reference's minimiser: PCL's gates and stopping rules (strict normal test, absolute MSE only)
    around Levenberg-Marquardt -- ref=True: Eigen's own code from the reference tree (needs oracle/_ref/libref_icp.so),
    False: the restatement (what the GPU's nn_mode 5 is compared with); exact=True: the same with every residual in double (nn_mode 6);
    moment=True: the moment form with integer-exact sums (nn_mode 7: the GPU returns these bits)."""
    if ref:
        ref_icp()
    return icp_refine_batch_variant(S, Sn, M, Mn, poses, max_iter, angle_deg, max_corr_dist,
                                    minimiser=MIN_LM_REF if ref else (MIN_LM_MOMENT if moment else (MIN_LM_EXACT if exact else MIN_LM_RESTATED)),
                                    strict_normal=True, relative_stop=False)


def euler_zyx(R, ref=False):
    r = np.ascontiguousarray(np.asarray(R, np.float32).reshape(9))
    out = np.zeros(3, np.float32)
    (ref_icp().ref_probe_euler_zyx if ref else lib().orc_euler_zyx)(F(r), F(out))
    return out


def geodesic(R1, R2, ref=False):
    a = np.ascontiguousarray(np.asarray(R1, np.float32).reshape(9))
    b = np.ascontiguousarray(np.asarray(R2, np.float32).reshape(9))
    return float((ref_icp().ref_probe_geodesic if ref else lib().orc_geodesic)(F(a), F(b)))


def tdiff_norm(t0, t1, ref=False):
    a = np.ascontiguousarray(t0, np.float32)
    b = np.ascontiguousarray(t1, np.float32)
    return float((ref_icp().ref_probe_tdiff_norm if ref else lib().orc_tdiff_norm)(F(a), F(b)))


def inverse_times(Ticp, pose, ref=False):
    a = np.ascontiguousarray(np.asarray(Ticp, np.float32).reshape(16))
    b = np.ascontiguousarray(np.asarray(pose, np.float32).reshape(16))
    out = np.zeros(16, np.float32)
    (ref_icp().ref_probe_inverse_times if ref else lib().orc_inverse_times)(F(a), F(b), F(out))
    return out.reshape(4, 4)


def mul4(A, B, ref=False):
    a = np.ascontiguousarray(np.asarray(A, np.float32).reshape(16))
    b = np.ascontiguousarray(np.asarray(B, np.float32).reshape(16))
    out = np.zeros(16, np.float32)
    (ref_icp().ref_probe_mul4 if ref else lib().orc_mul4)(F(a), F(b), F(out))
    return out.reshape(4, 4)


def cluster_poses(poses, lcp, ids, angle_deg, dist, sym_deg):
    T = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    lcp = np.ascontiguousarray(lcp, np.float32)
    ids = np.ascontiguousarray(ids, np.int32)
    sym = np.ascontiguousarray(sym_deg, np.float32)
    keep = np.zeros(len(T), np.int32)
    n = lib().orc_cluster_poses(F(T), F(lcp), I(ids), len(T), angle_deg, dist, F(sym), I(keep))
    return keep[:n].copy()


def hand_remove_surrounding(scene_xyz, scene_nrm, handbase_in_cam, links, finger12_in_handbase, finger22_in_handbase, min_z):
    """links: list of (xyz (n,3) in the hand-base frame, squared distance threshold), in name order.
    Returns (xyz, nrm, conf, keep_index) of the survivors."""
    n = len(scene_xyz)
    X, Nn = soa(scene_xyz), soa(scene_nrm)
    clouds = [soa(l[0]) for l in links]
    arr = (C.POINTER(C.c_float) * len(links))(*[F(c) for c in clouds])
    ln = np.array([len(l[0]) for l in links], np.int32)
    th = np.array([l[1] for l in links], np.float32)
    ox, on = np.zeros((3, max(n, 1)), np.float32), np.zeros((3, max(n, 1)), np.float32)
    oc, ki = np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.int32)
    T = np.ascontiguousarray(handbase_in_cam, np.float32).reshape(16)
    f1 = np.ascontiguousarray(finger12_in_handbase, np.float32).reshape(16)
    f2 = np.ascontiguousarray(finger22_in_handbase, np.float32).reshape(16)
    k = lib().orc_hand_remove_surrounding(F(X), F(Nn), n, F(T), arr, I(ln), F(th), len(links), F(f1), F(f2), float(min_z), F(ox), F(on), F(oc), I(ki))
    return ox[:, :k].T.copy(), on[:, :k].T.copy(), oc[:k].copy(), ki[:k].copy()


def model_ppf_keys(xyz, nrm, cap=1 << 20):
    X, Nn = soa(xyz), soa(nrm)
    out = np.zeros((cap, 4), np.int32)
    k = lib().orc_model_ppf_keys(F(X), F(Nn), X.shape[1], I(out), cap)
    return out[:k].copy()


# ---------------------------------------------------------------------------------------------- row N1
FLT_MAX = float(np.finfo(np.float32).max)
_ref_sdf = None


def ref_sdf_available():
    return os.path.exists(REF_SDF_SO)


def ref_signed_distance(P, V, Fi, lower=-FLT_MAX, upper=FLT_MAX):
    """The reference's own libigl (oracle/_ref/libref_sdf.so): (S, I, C)."""
    global _ref_sdf
    if _ref_sdf is None:
        _ref_sdf = C.CDLL(REF_SDF_SO)
        _ref_sdf.ref_igl_signed_distance.argtypes = [fp, C.c_int, fp, C.c_int, ip, C.c_int, C.c_float, C.c_float, fp, ip, fp]
    P = np.ascontiguousarray(P, np.float32)
    V = np.ascontiguousarray(V, np.float32)
    Fi = np.ascontiguousarray(Fi, np.int32)
    S, Ii, Cc = np.zeros(len(P), np.float32), np.zeros(len(P), np.int32), np.zeros((len(P), 3), np.float32)
    _ref_sdf.ref_igl_signed_distance(F(P), len(P), F(V), len(V), I(Fi), len(Fi), lower, upper, F(S), I(Ii), F(Cc))
    return S, Ii, Cc


def sdf_signed_distance(P, V, Fi, pose=None, lower=-FLT_MAX, upper=FLT_MAX):
    P = np.ascontiguousarray(P, np.float32)
    V = np.ascontiguousarray(V, np.float32)
    Fi = np.ascontiguousarray(Fi, np.int32)
    S, Ii = np.zeros(len(P), np.float32), np.zeros(len(P), np.int32)
    T = None if pose is None else np.ascontiguousarray(pose, np.float32).reshape(16)
    lib().orc_sdf_signed_distance(F(P), len(P), F(V), len(V), I(Fi), len(Fi), None if T is None else F(T), lower, upper, F(S), I(Ii))
    return S, Ii


def voxel_downsample(xyz, leaf):
    X = soa(xyz)
    n = X.shape[1]
    out = np.zeros((3, max(n, 1)), np.float32)
    k = C.c_int(0)
    lib().orc_voxel_downsample(F(X), n, leaf, F(out), max(n, 1), C.byref(k))
    return out[:, :k.value].T.copy()


def physics_args(p):
    """p: dict with the fields of hop_amd.api.PhysicsInputs (numpy arrays); returns (PhysicsArgs, keepalive)."""
    a, keep = PhysicsArgs(), []

    def f32(x):
        x = np.ascontiguousarray(x, np.float32)
        keep.append(x)
        return x

    def i32(x):
        x = np.ascontiguousarray(x, np.int32)
        keep.append(x)
        return x

    V, Fi = f32(p["object_V"]), i32(p["object_F"])
    a.object_V, a.object_nv, a.object_F, a.object_nf = F(V), len(V), I(Fi), len(Fi)
    for k in range(4):
        V, Fi = f32(p["finger_V"][k]), i32(p["finger_F"][k])
        a.finger_V[k], a.finger_nv[k], a.finger_F[k], a.finger_nf[k] = F(V), len(V), I(Fi), len(Fi)
        a.finger_mesh_pose[k] = F(f32(np.asarray(p["finger_mesh_pose"][k]).reshape(16)))
        X = f32(soa(p["finger_xyz"][k]))
        a.finger_xyz[k], a.finger_n[k] = F(X), X.shape[1]
        a.finger2handbase[k] = F(f32(np.asarray(p["finger2handbase"][k]).reshape(16)))
        a.finger_status[k] = int(p["finger_status"][k])
    X = f32(soa(p["hand_cloud"]))
    a.hand_cloud_xyz, a.n_hand_cloud = F(X), X.shape[1]
    X = f32(soa(p["cloud_without_hand"]))
    a.cloud_without_hand_xyz, a.n_cloud_without_hand = F(X), X.shape[1]
    a.cam2handbase = (C.c_float * 16)(*np.asarray(p["cam2handbase"], np.float32).reshape(16))
    X = f32(soa(p["model"]))
    a.model_xyz, a.n_model = F(X), X.shape[1]
    a.model_center_init = (C.c_float * 3)(*np.asarray(p["model_center_init"], np.float32))
    for k in ("smallest_dim", "ob_diameter", "collision_thres", "non_touch_dist", "collision_finger_dist",
              "collision_finger_volume_ratio", "voxel_size"):
        setattr(a, k, float(p[k]))
    return a, keep


def reject_by_collision(p, poses):
    a, keep = physics_args(p)
    T = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    out = np.zeros(len(T), np.uint8)
    diag = np.zeros((len(T), 8), np.float32)
    lib().orc_reject_by_collision(C.byref(a), F(T), len(T), out.ctypes.data_as(C.POINTER(C.c_ubyte)), F(diag))
    return out.astype(bool), diag


def scene_from_depth(depth_raw, depth_unit, K, cam_in_handbase, handbase_in_cam, leaf, crop_min, crop_max):
    """Returns (xyz (n,3), counts[valid pixels, after voxel grid, after crop])."""
    d = np.ascontiguousarray(depth_raw, np.uint16)
    H, W = d.shape
    K9 = np.ascontiguousarray(K, np.float32).reshape(9)
    A = np.ascontiguousarray(cam_in_handbase, np.float32).reshape(16)
    B = np.ascontiguousarray(handbase_in_cam, np.float32).reshape(16)
    lo, hi = np.ascontiguousarray(crop_min, np.float32), np.ascontiguousarray(crop_max, np.float32)
    cap = H * W
    out = np.zeros((3, cap), np.float32)
    n = C.c_int(0)
    counts = np.zeros(4, np.int32)
    lib().orc_scene_from_depth(d.ctypes.data_as(C.POINTER(C.c_ushort)), H, W, depth_unit, F(K9), F(A), F(B), leaf, F(lo), F(hi), F(out), cap, C.byref(n), I(counts))
    return out[:, :n.value].T.copy(), counts[:3].copy()


def object_segment(xyz, nrm, conf, leaf=0.003):
    X, Nn = soa(xyz), soa(nrm)
    cf = np.ascontiguousarray(conf, np.float32)
    n = X.shape[1]
    cap = max(n, 1)
    ox, on, oc = np.zeros((3, cap), np.float32), np.zeros((3, cap), np.float32), np.zeros(cap, np.float32)
    k = C.c_int(0)
    lib().orc_object_segment(F(X), F(Nn), F(cf), n, leaf, F(ox), F(on), F(oc), cap, C.byref(k))
    return ox[:, :k.value].T.copy(), on[:, :k.value].T.copy(), oc[:k.value].copy()


def hand_scene_filters(xyz, nrm, cam_in_handbase):
    """Hand::setCurScene (Hand.cpp:289-321): (xyz, nrm in the hand-base frame, keep flags of removed_noise, of remove_swivel)."""
    X, Nn = soa(xyz), soa(nrm)
    n = X.shape[1]
    T = np.ascontiguousarray(cam_in_handbase, np.float32).reshape(16)
    hx, hn = np.zeros((3, max(n, 1)), np.float32), np.zeros((3, max(n, 1)), np.float32)
    k1, k2 = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
    lib().orc_hand_scene_filters(F(X), F(Nn), n, F(T), F(hx), F(hn), k1.ctypes.data_as(C.POINTER(C.c_ubyte)), k2.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return hx[:, :n].T.copy(), hn[:, :n].T.copy(), k1[:n].astype(bool), k2[:n].astype(bool)


def voxel_downsample_normals(xyz, nrm, leaf):
    X, Nn = soa(xyz), soa(nrm)
    n = X.shape[1]
    cap = max(n, 1)
    ox, on = np.zeros((3, cap), np.float32), np.zeros((3, cap), np.float32)
    k = C.c_int(0)
    lib().orc_voxel_downsample_normals(F(X), F(Nn), n, leaf, F(ox), F(on), cap, C.byref(k))
    return ox[:, :k.value].T.copy(), on[:, :k.value].T.copy()


def handbase_region(xyz, nrm, cam_in_handbase, y1, z1, y2, z2):
    X, Nn = soa(xyz), soa(nrm)
    n = X.shape[1]
    T = np.ascontiguousarray(cam_in_handbase, np.float32).reshape(16)
    hx, hn = np.zeros((3, max(n, 1)), np.float32), np.zeros((3, max(n, 1)), np.float32)
    k = np.zeros(max(n, 1), np.uint8)
    lib().orc_handbase_region(F(X), F(Nn), n, F(T), y1, z1, y2, z2, F(hx), F(hn), k.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return hx[:, :n].T.copy(), hn[:, :n].T.copy(), k[:n].astype(bool)


def hand_height_matches(scene_xyz, scene_nrm, hand_xyz, hand_nrm, heights):
    S, Sn, Hx, Hn = soa(scene_xyz), soa(scene_nrm), soa(hand_xyz), soa(hand_nrm)
    h = np.ascontiguousarray(heights, np.float32)
    out = np.zeros(len(h), np.int32)
    lib().orc_hand_height_matches(F(S), F(Sn), S.shape[1], F(Hx), F(Hn), Hx.shape[1], F(h), len(h), I(out))
    return out


def normals_integral_image(xyz_organized, max_depth_change_factor=0.02, smoothing=10.0, depth_dependent=True):
    """pcl::IntegralImageNormalEstimation as Utils::calNormalIntegralImage configures it (Utils.cpp:293-329, method -1).
    xyz_organized: (H, W, 3); returns (H, W, 3) normals, NaN where undefined."""
    a = np.asarray(xyz_organized, np.float32)
    H, W = a.shape[:2]
    planes = np.ascontiguousarray(a.reshape(H * W, 3).T)
    out = np.zeros((3, H * W), np.float32)
    lib().orc_normals_integral_image(F(planes), H, W, C.c_float(max_depth_change_factor), C.c_float(smoothing), int(bool(depth_dependent)), F(out))
    return out.T.reshape(H, W, 3).copy()


def normals_mls(xyz, radius=0.003, order=2):
    """pcl::MovingLeastSquares as Utils::calNormalMLS configures it (Utils.cpp:268-289): (projected xyz, normals, curvature,
    input index) of the points that have >= 3 neighbours."""
    X = soa(xyz)
    n = X.shape[1]
    cap = max(n, 1)
    ox, on = np.zeros((3, cap), np.float32), np.zeros((3, cap), np.float32)
    oc, ki = np.zeros(cap, np.float32), np.zeros(cap, np.int32)
    k = C.c_int(0)
    lib().orc_normals_mls(F(X), n, C.c_float(radius), int(order), F(ox), F(on), F(oc), I(ki), cap, C.byref(k))
    m = k.value
    return ox[:, :m].T.copy(), on[:, :m].T.copy(), oc[:m].copy(), ki[:m].copy()


def scene_from_depth_normals(depth_raw, depth_unit, K, cam_in_handbase, handbase_in_cam, leaf, crop_min, crop_max, factor=0.02, smoothing=10.0):
    d = np.ascontiguousarray(depth_raw, np.uint16)
    H, W = d.shape
    K9 = np.ascontiguousarray(K, np.float32).reshape(9)
    A = np.ascontiguousarray(cam_in_handbase, np.float32).reshape(16)
    B = np.ascontiguousarray(handbase_in_cam, np.float32).reshape(16)
    lo, hi = np.ascontiguousarray(crop_min, np.float32), np.ascontiguousarray(crop_max, np.float32)
    cap = H * W
    ox, on = np.zeros((3, cap), np.float32), np.zeros((3, cap), np.float32)
    n = C.c_int(0)
    lib().orc_scene_from_depth_normals(d.ctypes.data_as(C.POINTER(C.c_ushort)), H, W, C.c_double(depth_unit), F(K9), F(A), F(B), C.c_float(leaf), F(lo), F(hi),
                                       C.c_float(factor), C.c_float(smoothing), F(ox), F(on), cap, C.byref(n))
    return ox[:, :n.value].T.copy(), on[:, :n.value].T.copy()


def _mesh(V, Fi):
    V = np.ascontiguousarray(np.asarray(V, np.float32).reshape(-1, 3))
    Fi = np.ascontiguousarray(np.asarray(Fi, np.int32).reshape(-1, 3))
    return V, Fi


def render(hand_V, hand_F, obj_V, obj_F, K, H, W):
    """Renderer::doRender of hand + object meshes given in the camera frame: (depth metres (H, W), owner (H, W))."""
    hv, hf = _mesh(hand_V, hand_F)
    ov, of = _mesh(obj_V, obj_F)
    K9 = np.ascontiguousarray(K, np.float32).reshape(9)
    d = np.zeros((H, W), np.float32)
    o = np.zeros((H, W), np.uint8)
    lib().orc_render(F(hv), I(hf), len(hf), F(ov), I(of), len(of), F(K9), H, W, F(d), o.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return d, o


def read_depth_image(depth_raw, unit=0.001):
    """Utils::readDepthImage (Utils.cpp:36-55)."""
    d = (np.asarray(depth_raw).astype(np.float32).astype(np.float64) * unit).astype(np.float32)
    d[(d > 2.0) | (d < 0.1)] = 0
    return d


def reject_by_render(depth_raw, unit, K, hand_V, hand_F, obj_V, obj_F, poses, roi_weight, keep_ratio):
    """PoseEstimator::rejectByRender: (wrong_ratio of every hypothesis, kept indices in pop order)."""
    dm = np.ascontiguousarray(read_depth_image(depth_raw, unit))
    H, W = dm.shape
    hv, hf = _mesh(hand_V, hand_F)
    ov, of = _mesh(obj_V, obj_F)
    K9 = np.ascontiguousarray(K, np.float32).reshape(9)
    T = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    wr = np.zeros(max(len(T), 1), np.float32)
    keep = np.zeros(max(len(T), 1), np.int32)
    nk = C.c_int(0)
    lib().orc_reject_by_render(F(dm), H, W, F(K9), F(hv), I(hf), len(hf), F(ov), len(ov), I(of), len(of), F(T), len(T), C.c_float(roi_weight), C.c_float(keep_ratio),
                               F(wr), I(keep), C.byref(nk))
    return wr[:len(T)].copy(), keep[:nk.value].copy()
