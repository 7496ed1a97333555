"""ctypes binding of the C-ABI in include/hop.h (libhop.so) and Python mirrors of the reference's host classes.

There is no CPU fallback here: importing works anywhere (so CPU-only tests can check symbols), but every
compute entry point needs a HIP device and raises ``HopError`` otherwise.

Mirrors (same names / argument meaning as the reference):
  ``PoseHypo``       include/PoseHypo.h:7-26
  ``PoseEstimator``  include/PoseEstimator.h:12-49  (setCurScene, runSuper4pcs, clusterPoses, refineByICP, selectBest)
  ``HandT42``        include/Hand.h:29-92           (getTFHandBase, matchOneComponentPSO)
"""
from __future__ import annotations

import ctypes as C
import math
import os
import sys
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libhop.so")
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "hop.h")

fp = C.POINTER(C.c_float)
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)

HOP_MODEL_5MM = 0
HOP_MODEL_1MM = 1
TOPK_ROW_FLOATS = 18
FRAME_ROW_FLOATS = 17  # hop_frames_allgather: frame index, pose[16]


class HopError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__(f"{where}: status {status} ({_strerror(status)}) {detail}")


class S4pcsOpts(C.Structure):
    _fields_ = [("sample_size", C.c_int), ("overlap", C.c_float), ("delta", C.c_float), ("dispersion", C.c_float),
                ("success_quadrilaterals", C.c_int), ("max_time_seconds", C.c_int), ("n_trials", C.c_int),
                ("random_seed", C.c_uint), ("max_normal_difference", C.c_float), ("max_color_distance", C.c_float),
                ("verify_mode", C.c_int)]


class S4pcsStats(C.Structure):
    _fields_ = [("n_trials_run", C.c_int), ("n_bases", C.c_int), ("n_hypotheses", C.c_int), ("n_pairs", C.c_longlong),
                ("n_quads", C.c_longlong), ("n_candidates", C.c_longlong), ("n_sampled_q", C.c_int),
                ("centroid_p", C.c_float * 3), ("centroid_q", C.c_float * 3), ("diameter", C.c_float),
                ("ms_select", C.c_double), ("ms_device", C.c_double)]


HOP_ABI_VERSION = 2
# hop_icp_opts.nn_mode of the mirrors: 7 = the reference's Levenberg-Marquardt on (t, quaternion) to its stopping rule per ICP iteration, from
# integer-exact moment sums with an IEEE-only solve: the poses the CPU oracle (minimiser 7) returns, bit for bit.  HOP_ICP_NN_MODE overrides it
# without a rebuild (host/PoseEstimator.h icp_nn_mode_reference: e.g. 6, the float-sum form round 3 measured on hardware).
def _icp_nn_mode_reference():
    """HOP_ICP_NN_MODE as host/PoseEstimator.h icp_nn_mode_reference reads it: an integer in 0..7, anything else (empty, text, out of range) is
    the default 7 with a line on stderr -- never an exception at import, never an out-of-range mode passed through."""
    e = os.environ.get("HOP_ICP_NN_MODE")
    if e is None or e == "":
        return 7
    try:
        m = int(e.strip())
    except ValueError:
        m = -1
    if not 0 <= m <= 7:
        sys.stderr.write(f"hop: HOP_ICP_NN_MODE={e!r} is not an integer in 0..7; using 7\n")
        return 7
    return m


ICP_NN_MODE_REFERENCE = _icp_nn_mode_reference()
ICP_NN_MODE_GN = 4         # one Gauss-Newton step per ICP iteration (faster; not what PCL computes)


class IcpOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("angle_deg", C.c_float), ("max_corr_dist", C.c_float),
                ("max_hypotheses", C.c_int), ("nn_mode", C.c_int)]


class LcpOpts(C.Structure):
    _fields_ = [("dist", C.c_float), ("angle_deg", C.c_float), ("nn_mode", C.c_int)]


class FingerArgs(C.Structure):
    _fields_ = [("fp_min", C.c_float * 3), ("fp_max", C.c_float * 3), ("fp_stride_z", C.c_float),
                ("fp_num_division", C.c_int), ("fp_hist_min_y", fp), ("fo_min", C.c_float * 3), ("fo_max", C.c_float * 3),
                ("model2handbase", C.c_float * 16), ("finger_out2parent", C.c_float * 16), ("pair_tip1", C.c_float * 4),
                ("pair_tip2", C.c_float * 4), ("is_palm_side", C.c_int), ("is_right_side", C.c_int),
                ("gripper_min_dist", C.c_float), ("dist_thres", C.c_float), ("cos_normal_thres", C.c_float),
                ("check_normal", C.c_int), ("max_outter_pts", C.c_int), ("outter_pt_dist", C.c_float),
                ("outter_pt_dist_weight", C.c_float), ("model_xyz", fp), ("model_nrm", fp), ("n_model", C.c_int)]


class PsoSettings(C.Structure):
    _fields_ = [("n_pop", C.c_int), ("n_gen", C.c_int), ("check_freq", C.c_int), ("c_cog", C.c_double),
                ("c_soc", C.c_double), ("initial_w", C.c_double), ("w_min", C.c_double), ("w_max", C.c_double),
                ("err_tol", C.c_double), ("lower_rad", C.c_double), ("upper_rad", C.c_double), ("seed", C.c_uint64)]


class HandLink(C.Structure):
    _fields_ = [("xyz", C.POINTER(C.c_float)), ("n", C.c_int), ("sq_dist_thres", C.c_float)]


class PhysicsArgs(C.Structure):
    _fields_ = [("object_mesh", C.c_int), ("finger_mesh", C.c_int * 4),
                ("finger_xyz", fp * 4), ("finger_n", C.c_int * 4), ("finger2handbase", fp * 4), ("finger_status", C.c_int * 4),
                ("hand_cloud_xyz", fp), ("n_hand_cloud", C.c_int),
                ("cloud_without_hand_xyz", fp), ("n_cloud_without_hand", C.c_int),
                ("cam2handbase", C.c_float * 16),
                ("model_xyz", fp), ("n_model", C.c_int),
                ("model_center_init", C.c_float * 3), ("smallest_dim", C.c_float), ("ob_diameter", C.c_float),
                ("collision_thres", C.c_float), ("non_touch_dist", C.c_float), ("collision_finger_dist", C.c_float),
                ("collision_finger_volume_ratio", C.c_float), ("voxel_size", C.c_float)]


class Timing(C.Structure):
    _fields_ = [("ms_verify", C.c_double), ("ms_gen_other", C.c_double), ("ms_icp_nn", C.c_double),
                ("ms_icp_solve", C.c_double), ("ms_lcp_fwd", C.c_double), ("ms_lcp_rev", C.c_double),
                ("ms_pso", C.c_double), ("ms_ppf_matrix", C.c_double), ("n_verify_launches", C.c_longlong),
                ("n_icp_nn_launches", C.c_longlong), ("n_lcp_launches", C.c_longlong), ("n_pso_launches", C.c_longlong),
                ("pairs_verify", C.c_longlong), ("pairs_icp", C.c_longlong), ("pairs_lcp", C.c_longlong),
                ("pairs_pso", C.c_longlong), ("ms_icp_accum", C.c_double), ("ms_lcp_sum", C.c_double), ("ms_quads", C.c_double),
                ("ms_build", C.c_double), ("n_quads_launches", C.c_longlong), ("n_build_launches", C.c_longlong)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/hop.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
SIGNATURES = {
    "hop_abi_version": (C.c_int, []),
    "hop_strerror": (C.c_char_p, [C.c_int]),
    "hop_last_error": (C.c_char_p, [_vp]),
    "hop_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "hop_ctx_destroy": (None, [_vp]),
    "hop_synchronize": (C.c_int, [_vp]),
    "hop_set_scene": (C.c_int, [_vp, fp, fp, fp, C.c_int, C.c_float]),
    "hop_scene_size": (C.c_int, [_vp]),
    "hop_set_model": (C.c_int, [_vp, C.c_int, fp, fp, C.c_int]),
    "hop_set_ppf_keys": (C.c_int, [_vp, ip, C.c_int]),
    "hop_s4pcs_default_opts": (None, [C.POINTER(S4pcsOpts)]),
    "hop_s4pcs_generate": (C.c_int, [_vp, C.POINTER(S4pcsOpts), fp, fp, C.c_int, ip, C.POINTER(S4pcsStats)]),
    "hop_s4pcs_num_bases": (C.c_int, [_vp]),
    "hop_s4pcs_get_base": (C.c_int, [_vp, C.c_int, ip, fp, ip]),
    "hop_s4pcs_get_sampled_q": (C.c_int, [_vp, fp, fp]),
    "hop_verify_set_clouds": (C.c_int, [_vp, fp, C.c_int, fp, C.c_int]),
    "hop_verify_batch": (C.c_int, [_vp, fp, C.c_int, C.c_float, C.c_int, ip]),
    "hop_hypos_upload": (C.c_int, [_vp, fp, fp, C.c_int]),
    "hop_hypos_count": (C.c_int, [_vp]),
    "hop_hypos_download": (C.c_int, [_vp, fp, fp, ip, C.c_int, ip]),
    "hop_hypos_keep_topk": (C.c_int, [_vp, C.c_int]),
    "hop_icp_refine": (C.c_int, [_vp, C.POINTER(IcpOpts), ip, ip]),
    "hop_lcp_select_best": (C.c_int, [_vp, C.POINTER(LcpOpts), fp, fp, ip]),
    "hop_cluster_poses": (C.c_int, [_vp, C.c_float, C.c_float, fp, C.c_int]),
    "hop_cluster_poses_host": (C.c_int, [fp, fp, ip, C.c_int, C.c_float, C.c_float, fp, ip, ip]),
    "hop_cluster_pose_terms": (C.c_int, [fp, fp, fp]),
    "hop_topk_pack": (C.c_int, [_vp, C.c_int, C.c_int, fp, ip]),
    "hop_topk_merge": (C.c_int, [fp, C.c_int, C.c_int, fp, ip]),
    "hop_hand_set_scene": (C.c_int, [_vp, fp, C.c_int, fp, C.c_int, fp, C.c_int]),
    "hop_model_ppf_keys": (C.c_int, [_vp, fp, fp, C.c_int, ip, C.c_int, ip]),
    "hop_hand_set_finger": (C.c_int, [_vp, C.POINTER(FingerArgs)]),
    "hop_hand_remove_surrounding": (C.c_int, [_vp, fp, fp, C.c_int, fp, C.POINTER(HandLink), C.c_int, fp, fp, C.c_float, fp, fp, fp, ip, ip]),
    "hop_comm_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "hop_comm_create": (C.c_int, [C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.POINTER(_vp)]),
    "hop_comm_destroy": (None, [_vp]),
    "hop_comm_last_error": (C.c_char_p, [_vp]),
    "hop_topk_allgather": (C.c_int, [_vp, fp, C.c_int, fp, ip]),
    "hop_topk_pack_device": (C.c_int, [_vp, C.c_int, C.c_int, C.c_void_p, ip]),
    "hop_topk_allgather_device": (C.c_int, [_vp, C.c_void_p, C.c_int, fp, ip]),
    "hop_comm_info": (C.c_int, [_vp, ip, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "hop_frames_allgather": (C.c_int, [_vp, fp, C.c_int, C.c_int, fp]),
    "hop_hand_set_sum_mode": (C.c_int, [_vp, C.c_int]),
    "hop_hand_pso_eval_batch": (C.c_int, [_vp, dp, C.c_int, dp]),
    "hop_pso_default_settings": (None, [C.POINTER(PsoSettings)]),
    "hop_hand_pso_search": (C.c_int, [_vp, C.POINTER(PsoSettings), dp, dp]),
    "hop_sdf_register_mesh": (C.c_int, [_vp, C.c_int, fp, C.c_int, ip, C.c_int, fp]),
    "hop_sdf_set_mesh_pose": (C.c_int, [_vp, C.c_int, fp]),
    "hop_sdf_signed_distance": (C.c_int, [_vp, C.c_int, fp, C.c_int, fp, ip, fp, fp]),
    "hop_voxel_downsample": (C.c_int, [_vp, fp, C.c_int, C.c_float, fp, C.c_int, ip]),
    "hop_scene_from_depth": (C.c_int, [_vp, C.POINTER(C.c_ushort), C.c_int, C.c_int, C.c_double, fp, fp, fp, C.c_float, fp, fp, fp, C.c_int, ip, ip]),
    "hop_object_segment": (C.c_int, [_vp, fp, fp, fp, C.c_int, C.c_float, fp, fp, fp, C.c_int, ip]),
    "hop_render_set_frame": (C.c_int, [_vp, C.POINTER(C.c_ushort), C.c_int, C.c_int, C.c_double, fp, fp, C.c_int, ip, C.c_int]),
    "hop_render_set_object": (C.c_int, [_vp, fp, C.c_int, ip, C.c_int]),
    "hop_render_depth": (C.c_int, [_vp, fp, fp, C.POINTER(C.c_ubyte)]),
    "hop_reject_by_render": (C.c_int, [_vp, C.c_float, C.c_float, C.c_int, fp, ip, ip]),
    "hop_normals_integral_image": (C.c_int, [_vp, fp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, fp]),
    "hop_normals_mls": (C.c_int, [_vp, fp, C.c_int, C.c_float, C.c_int, fp, fp, fp, ip, C.c_int, ip]),
    "hop_scene_from_depth_normals": (C.c_int, [_vp, C.POINTER(C.c_ushort), C.c_int, C.c_int, C.c_double, fp, fp, fp, C.c_float, fp, fp, C.c_float, C.c_float,
                                              fp, fp, C.c_int, ip, ip]),
    "hop_hand_scene_filters": (C.c_int, [_vp, fp, fp, C.c_int, fp, fp, fp, C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte)]),
    "hop_voxel_downsample_normals": (C.c_int, [_vp, fp, fp, C.c_int, C.c_float, fp, fp, C.c_int, ip]),
    "hop_handbase_region": (C.c_int, [_vp, fp, fp, C.c_int, fp, C.c_float, C.c_float, C.c_float, C.c_float, fp, fp, C.POINTER(C.c_ubyte)]),
    "hop_hand_height_matches": (C.c_int, [_vp, fp, fp, C.c_int, fp, fp, C.c_int, fp, C.c_int, ip]),
    "hop_physics_set_frame": (C.c_int, [_vp, C.POINTER(PhysicsArgs)]),
    "hop_reject_by_collision": (C.c_int, [_vp, C.POINTER(C.c_ubyte), fp, ip]),
    "hop_physics_timing": (C.c_int, [_vp, dp, dp]),
    "hop_timing_reset": (C.c_int, [_vp]),
    "hop_timing_get": (C.c_int, [_vp, C.POINTER(Timing)]),
    "hop_timing_enable": (C.c_int, [_vp, C.c_int]),
}

_lib = None


def build_library(force=False):
    """hipcc cross-compiles gfx950 without a GPU; the .so stays in-tree (icra20-hand-object-pose_amd/lib)."""
    src = os.path.join(HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", src], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HopError(-2, "load", f"{LIB_PATH} is missing: run __graft_entry__.build() (no CPU fallback exists)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.hop_abi_version() != HOP_ABI_VERSION:  # the struct layouts and entry points this binding was written for (include/hop.h)
            raise HopError(-1, "load", f"{LIB_PATH} has ABI version {L.hop_abi_version()}, this binding expects {HOP_ABI_VERSION}: rebuild")
        _lib = L
    return _lib


def _strerror(status):
    try:
        return lib().hop_strerror(status).decode()
    except Exception:
        return "?"


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
    """(n,3) array -> contiguous (3,n) float32 planes (the ABI's cloud layout)."""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).T)


class Context:
    """One hop_ctx: a HIP device, its stream and the resident clouds / hypothesis set."""

    def __init__(self, device=0):
        self.L = lib()
        h = _vp()
        rc = self.L.hop_ctx_create(device, C.byref(h))
        if rc:
            raise HopError(rc, "hop_ctx_create")
        self.h = h
        self.device = device
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.L.hop_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, where, ok=()):
        if rc and rc not in ok:
            raise HopError(rc, where, self.L.hop_last_error(self.h).decode())
        return rc

    # ---- clouds
    def set_scene(self, xyz, nrm, conf=None, high_confidence_thres=0.0):
        X, Nn = soa(xyz), soa(nrm)
        cf = None if conf is None else np.ascontiguousarray(conf, dtype=np.float32)
        self._chk(self.L.hop_set_scene(self.h, F(X), F(Nn), F(cf) if cf is not None else None, X.shape[1],
                                       high_confidence_thres), "hop_set_scene")
        return self.L.hop_scene_size(self.h)

    def set_model(self, level, xyz, nrm):
        X, Nn = soa(xyz), soa(nrm)
        self._chk(self.L.hop_set_model(self.h, level, F(X), F(Nn), X.shape[1]), "hop_set_model")

    def set_ppf_keys(self, keys):
        k = np.ascontiguousarray(keys, dtype=np.int32).reshape(-1, 4)
        self._chk(self.L.hop_set_ppf_keys(self.h, I(k), len(k)), "hop_set_ppf_keys")

    # ---- generator
    def default_s4pcs_opts(self, **kw):
        o = S4pcsOpts()
        self.L.hop_s4pcs_default_opts(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def s4pcs_generate(self, opts, download=True, cap=None):
        st = S4pcsStats()
        n = C.c_int(0)
        if not download:
            rc = self.L.hop_s4pcs_generate(self.h, C.byref(opts), None, None, 0, C.byref(n), C.byref(st))
            self._chk(rc, "hop_s4pcs_generate", ok=(-6,))
            return None, None, st
        cap = cap or (1 << 16)
        while True:
            pose = np.zeros((cap, 16), np.float32)
            lcp = np.zeros(cap, np.float32)
            rc = self.L.hop_s4pcs_generate(self.h, C.byref(opts), F(pose), F(lcp), cap, C.byref(n), C.byref(st))
            if rc == -4 and n.value > cap:
                # the set is resident; fetch it with a big enough buffer instead of re-running
                pose = np.zeros((n.value, 16), np.float32)
                lcp = np.zeros(n.value, np.float32)
                ids = np.zeros(n.value, np.int32)
                m = C.c_int(0)
                self._chk(self.L.hop_hypos_download(self.h, F(pose), F(lcp), I(ids), n.value, C.byref(m)), "hop_hypos_download")
                return pose.reshape(-1, 4, 4), lcp, st
            self._chk(rc, "hop_s4pcs_generate", ok=(-6,))
            return pose[:n.value].reshape(-1, 4, 4).copy(), lcp[:n.value].copy(), st

    def s4pcs_bases(self, successful_only=True):
        """Per-base trace of the last generate.  A base is "successful" when generateCongruents would return
        true (both pair lists and the quadrilateral set non-empty, match4pcsBase.hpp:269-280); the reference
        only counts / keeps those."""
        out = []
        for i in range(self.L.hop_s4pcs_num_bases(self.h)):
            b4 = np.zeros(4, np.int32)
            inv = np.zeros(2, np.float32)
            c3 = np.zeros(3, np.int32)
            self._chk(self.L.hop_s4pcs_get_base(self.h, i, I(b4), F(inv), I(c3)), "hop_s4pcs_get_base")
            ok = c3[0] > 0 and c3[1] > 0 and c3[2] > 0
            if ok or not successful_only:
                out.append(dict(base=b4, inv=inv, n_pairs1=int(c3[0]), n_pairs2=int(c3[1]), n_quads=int(c3[2]), success=bool(ok)))
        return out

    def s4pcs_sampled_q(self, n):
        x = np.zeros((3, n), np.float32)
        nn = np.zeros((3, n), np.float32)
        self._chk(self.L.hop_s4pcs_get_sampled_q(self.h, F(x), F(nn)), "hop_s4pcs_get_sampled_q")
        return x.T.copy(), nn.T.copy()

    def verify_set_clouds(self, P, Q):
        Pp, Qp = soa(P), soa(Q)
        self._chk(self.L.hop_verify_set_clouds(self.h, F(Pp), Pp.shape[1], F(Qp), Qp.shape[1]), "hop_verify_set_clouds")

    def verify_batch(self, T, delta, mode=0):
        T = np.ascontiguousarray(T, dtype=np.float32).reshape(-1, 16)
        out = np.zeros(len(T), np.int32)
        self._chk(self.L.hop_verify_batch(self.h, F(T), len(T), delta, mode, I(out)), "hop_verify_batch")
        return out

    # ---- physics row (N1)
    def sdf_register_mesh(self, mesh_id, V, Fi, pose=None):
        V = np.ascontiguousarray(V, np.float32).reshape(-1, 3)
        Fi = np.ascontiguousarray(Fi, np.int32).reshape(-1, 3)
        T = None if pose is None else np.ascontiguousarray(pose, np.float32).reshape(16)
        self._chk(self.L.hop_sdf_register_mesh(self.h, mesh_id, F(V), len(V), I(Fi), len(Fi), F(T) if T is not None else None),
                  "hop_sdf_register_mesh")

    def sdf_set_mesh_pose(self, mesh_id, pose=None):
        T = None if pose is None else np.ascontiguousarray(pose, np.float32).reshape(16)
        self._chk(self.L.hop_sdf_set_mesh_pose(self.h, mesh_id, F(T) if T is not None else None), "hop_sdf_set_mesh_pose")

    def sdf_signed_distance(self, mesh_id, pts):
        """SDFchecker::getSignedDistanceMinMaxWithRegistered: (dists, faces, min_dist, max_dist)."""
        X = soa(pts)
        n = X.shape[1]
        d, f = np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.int32)
        mn, mx = C.c_float(0), C.c_float(0)
        self._chk(self.L.hop_sdf_signed_distance(self.h, mesh_id, F(X), n, F(d), I(f), C.byref(mn), C.byref(mx)), "hop_sdf_signed_distance")
        return d[:n], f[:n], mn.value, mx.value

    def voxel_downsample(self, xyz, leaf):
        X = soa(xyz)
        n = X.shape[1]
        out = np.zeros((3, max(n, 1)), np.float32)
        k = C.c_int(0)
        self._chk(self.L.hop_voxel_downsample(self.h, F(X), n, leaf, F(out), max(n, 1), C.byref(k)), "hop_voxel_downsample")
        return out[:, :k.value].T.copy()

    def scene_from_depth(self, depth_raw, depth_unit, K, cam_in_handbase, handbase_in_cam, leaf=0.001,
                         crop_min=(-0.25, -0.2, -0.12), crop_max=(-0.07, 0.2, 0.05)):
        """main_realdata_auto.cpp:54-96 (no colours, no normals): (xyz (n,3) camera frame, counts[valid px, voxels, cropped])."""
        d = np.ascontiguousarray(depth_raw, np.uint16)
        H, W = d.shape
        K9 = np.ascontiguousarray(K, np.float32).reshape(9)
        A = np.ascontiguousarray(cam_in_handbase, np.float32).reshape(16)
        B = np.ascontiguousarray(handbase_in_cam, np.float32).reshape(16)
        lo, hi = np.ascontiguousarray(crop_min, np.float32), np.ascontiguousarray(crop_max, np.float32)
        cap = H * W
        out = np.zeros((3, cap), np.float32)
        n = C.c_int(0)
        counts = np.zeros(3, np.int32)
        self._chk(self.L.hop_scene_from_depth(self.h, d.ctypes.data_as(C.POINTER(C.c_ushort)), H, W, depth_unit, F(K9), F(A), F(B), leaf, F(lo), F(hi),
                                              F(out), cap, C.byref(n), I(counts)), "hop_scene_from_depth")
        return out[:, :n.value].T.copy(), counts

    def scene_from_depth_normals(self, depth_raw, depth_unit, K, cam_in_handbase, handbase_in_cam, leaf=0.001,
                                 crop_min=(-0.25, -0.2, -0.12), crop_max=(-0.07, 0.2, 0.05), max_depth_change_factor=0.02, smoothing=10.0):
        """main_realdata_auto.cpp:54-96 with the integral-image normals of :61: (xyz, nrm (n,3) camera frame, counts)."""
        d = np.ascontiguousarray(depth_raw, np.uint16)
        H, W = d.shape
        K9 = np.ascontiguousarray(K, np.float32).reshape(9)
        A = np.ascontiguousarray(cam_in_handbase, np.float32).reshape(16)
        B = np.ascontiguousarray(handbase_in_cam, np.float32).reshape(16)
        lo, hi = np.ascontiguousarray(crop_min, np.float32), np.ascontiguousarray(crop_max, np.float32)
        cap = H * W
        out, on = np.zeros((3, cap), np.float32), np.zeros((3, cap), np.float32)
        n = C.c_int(0)
        counts = np.zeros(3, np.int32)
        self._chk(self.L.hop_scene_from_depth_normals(self.h, d.ctypes.data_as(C.POINTER(C.c_ushort)), H, W, depth_unit, F(K9), F(A), F(B), leaf, F(lo), F(hi),
                                                      max_depth_change_factor, smoothing, F(out), F(on), cap, C.byref(n), I(counts)),
                  "hop_scene_from_depth_normals")
        return out[:, :n.value].T.copy(), on[:, :n.value].T.copy(), counts

    # ---- rejectByRender (N2)
    def render_set_frame(self, depth_raw, depth_unit, K, hand_V, hand_F):
        d = np.ascontiguousarray(depth_raw, np.uint16)
        H, W = d.shape
        K9 = np.ascontiguousarray(K, np.float32).reshape(9)
        V = np.ascontiguousarray(np.asarray(hand_V, np.float32).reshape(-1, 3))
        Fi = np.ascontiguousarray(np.asarray(hand_F, np.int32).reshape(-1, 3))
        self._render_shape = (H, W)
        self._chk(self.L.hop_render_set_frame(self.h, d.ctypes.data_as(C.POINTER(C.c_ushort)), H, W, depth_unit, F(K9), F(V) if len(V) else None, len(V),
                                              I(Fi) if len(Fi) else None, len(Fi)), "hop_render_set_frame")

    def render_set_object(self, V, Fi):
        V = np.ascontiguousarray(np.asarray(V, np.float32).reshape(-1, 3))
        Fi = np.ascontiguousarray(np.asarray(Fi, np.int32).reshape(-1, 3))
        self._chk(self.L.hop_render_set_object(self.h, F(V), len(V), I(Fi), len(Fi)), "hop_render_set_object")

    def render_depth(self, pose=None):
        """Renderer::doRender: (depth metres (H, W), owner (H, W): 0 nothing, 1 hand, 2 object)."""
        H, W = self._render_shape
        d = np.zeros((H, W), np.float32)
        o = np.zeros((H, W), np.uint8)
        T = None if pose is None else np.ascontiguousarray(pose, np.float32).reshape(16)
        self._chk(self.L.hop_render_depth(self.h, F(T) if T is not None else None, F(d), o.ctypes.data_as(C.POINTER(C.c_ubyte))), "hop_render_depth")
        return d, o

    def reject_by_render(self, roi_weight, keep_ratio, sum_mode=0):
        """PoseEstimator::rejectByRender on the resident set: (wrong_ratio of every hypothesis before the call, kept positions)."""
        n = max(self.hypos_count(), 1)
        wr = np.zeros(n, np.float32)
        keep = np.zeros(n, np.int32)
        nk = C.c_int(0)
        self._chk(self.L.hop_reject_by_render(self.h, roi_weight, keep_ratio, int(sum_mode), F(wr), I(keep), C.byref(nk)), "hop_reject_by_render")
        return wr, keep[:nk.value].copy()

    def normals_integral_image(self, xyz_organized, max_depth_change_factor=0.02, smoothing=10.0, depth_dependent=True):
        """Utils::calNormalIntegralImage (Utils.cpp:293-329, method -1) on an organised cloud (H, W, 3): (H, W, 3) normals."""
        a = np.asarray(xyz_organized, np.float32)
        H, W = a.shape[:2]
        planes = np.ascontiguousarray(a.reshape(H * W, 3).T)
        out = np.zeros((3, H * W), np.float32)
        self._chk(self.L.hop_normals_integral_image(self.h, F(planes), H, W, max_depth_change_factor, smoothing, int(bool(depth_dependent)), F(out)),
                  "hop_normals_integral_image")
        return out.T.reshape(H, W, 3).copy()

    def normals_mls(self, xyz, radius=0.003, order=2):
        """Utils::calNormalMLS (Utils.cpp:268-289): (projected xyz, normals, curvature, input index of every output)."""
        X = soa(xyz)
        n = X.shape[1]
        cap = max(n, 1)
        ox, on = np.zeros((3, cap), np.float32), np.zeros((3, cap), np.float32)
        oc, ki = np.zeros(cap, np.float32), np.zeros(cap, np.int32)
        k = C.c_int(0)
        self._chk(self.L.hop_normals_mls(self.h, F(X), n, radius, int(order), F(ox), F(on), F(oc), I(ki), cap, C.byref(k)), "hop_normals_mls")
        m = k.value
        return ox[:, :m].T.copy(), on[:, :m].T.copy(), oc[:m].copy(), ki[:m].copy()

    def object_segment(self, xyz, nrm, conf, leaf=0.003):
        """main_realdata_auto.cpp:156-177: (xyz, nrm, conf) of the generator's input cloud."""
        X, Nn = soa(xyz), soa(nrm)
        cf = np.ascontiguousarray(conf, np.float32)
        n = X.shape[1]
        cap = max(n, 1)
        ox, on, oc = np.zeros((3, cap), np.float32), np.zeros((3, cap), np.float32), np.zeros(cap, np.float32)
        k = C.c_int(0)
        self._chk(self.L.hop_object_segment(self.h, F(X), F(Nn), F(cf), n, leaf, F(ox), F(on), F(oc), cap, C.byref(k)), "hop_object_segment")
        return ox[:, :k.value].T.copy(), on[:, :k.value].T.copy(), oc[:k.value].copy()

    def hand_scene_filters(self, xyz, nrm, cam_in_handbase):
        """Hand::setCurScene (Hand.cpp:289-321): (xyz, nrm in the hand-base frame, keep flags of removed_noise, of remove_swivel)."""
        X, Nn = soa(xyz), soa(nrm)
        n = X.shape[1]
        T = np.ascontiguousarray(cam_in_handbase, np.float32).reshape(16)
        hx, hn = np.zeros((3, max(n, 1)), np.float32), np.zeros((3, max(n, 1)), np.float32)
        k1, k2 = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
        self._chk(self.L.hop_hand_scene_filters(self.h, F(X), F(Nn), n, F(T), F(hx), F(hn), k1.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                                k2.ctypes.data_as(C.POINTER(C.c_ubyte))), "hop_hand_scene_filters")
        return hx[:, :n].T.copy(), hn[:, :n].T.copy(), k1[:n].astype(bool), k2[:n].astype(bool)

    def voxel_downsample_normals(self, xyz, nrm, leaf):
        X, Nn = soa(xyz), soa(nrm)
        n = X.shape[1]
        cap = max(n, 1)
        ox, on = np.zeros((3, cap), np.float32), np.zeros((3, cap), np.float32)
        k = C.c_int(0)
        self._chk(self.L.hop_voxel_downsample_normals(self.h, F(X), F(Nn), n, leaf, F(ox), F(on), cap, C.byref(k)), "hop_voxel_downsample_normals")
        return ox[:, :k.value].T.copy(), on[:, :k.value].T.copy()

    def handbase_region(self, xyz, nrm, cam_in_handbase, y1, z1, y2, z2):
        X, Nn = soa(xyz), soa(nrm)
        n = X.shape[1]
        T = np.ascontiguousarray(cam_in_handbase, np.float32).reshape(16)
        hx, hn = np.zeros((3, max(n, 1)), np.float32), np.zeros((3, max(n, 1)), np.float32)
        k = np.zeros(max(n, 1), np.uint8)
        self._chk(self.L.hop_handbase_region(self.h, F(X), F(Nn), n, F(T), y1, z1, y2, z2, F(hx), F(hn), k.ctypes.data_as(C.POINTER(C.c_ubyte))),
                  "hop_handbase_region")
        return hx[:, :n].T.copy(), hn[:, :n].T.copy(), k[:n].astype(bool)

    def hand_height_matches(self, scene_xyz, scene_nrm, hand_xyz, hand_nrm, heights):
        S, Sn, Hx, Hn = soa(scene_xyz), soa(scene_nrm), soa(hand_xyz), soa(hand_nrm)
        h = np.ascontiguousarray(heights, np.float32)
        out = np.zeros(len(h), np.int32)
        self._chk(self.L.hop_hand_height_matches(self.h, F(S), F(Sn), S.shape[1], F(Hx), F(Hn), Hx.shape[1], F(h), len(h), I(out)), "hop_hand_height_matches")
        return out

    def physics_set_frame(self, p):
        """p: dict -- object_mesh, finger_mesh[4] (registered ids), finger_xyz[4] ((n,3), link frame), finger2handbase[4],
        finger_status[4], hand_cloud, cloud_without_hand, cam2handbase, model, model_center_init and the scalars of
        hop_physics_args."""
        a, keep = PhysicsArgs(), []

        def planes(x):
            x = soa(x)
            keep.append(x)
            return x

        a.object_mesh = int(p["object_mesh"])
        for k in range(4):
            a.finger_mesh[k] = int(p["finger_mesh"][k])
            X = planes(p["finger_xyz"][k])
            a.finger_xyz[k], a.finger_n[k] = F(X), X.shape[1]
            T = np.ascontiguousarray(p["finger2handbase"][k], np.float32).reshape(16)
            keep.append(T)
            a.finger2handbase[k] = F(T)
            a.finger_status[k] = int(p["finger_status"][k])
        X = planes(p["hand_cloud"])
        a.hand_cloud_xyz, a.n_hand_cloud = F(X), X.shape[1]
        X = planes(p["cloud_without_hand"])
        a.cloud_without_hand_xyz, a.n_cloud_without_hand = F(X), X.shape[1]
        a.cam2handbase = (C.c_float * 16)(*np.asarray(p["cam2handbase"], np.float32).reshape(16))
        X = planes(p["model"])
        a.model_xyz, a.n_model = F(X), X.shape[1]
        a.model_center_init = (C.c_float * 3)(*np.asarray(p["model_center_init"], np.float32))
        for k in ("smallest_dim", "ob_diameter", "collision_thres", "non_touch_dist", "collision_finger_dist",
                  "collision_finger_volume_ratio", "voxel_size"):
            setattr(a, k, float(p[k]))
        self._chk(self.L.hop_physics_set_frame(self.h, C.byref(a)), "hop_physics_set_frame")

    def reject_by_collision(self):
        """Filters the resident set; returns (keep mask, diag (H,8)) of the incoming set."""
        H = self.hypos_count()
        keep = np.zeros(max(H, 1), np.uint8)
        diag = np.zeros((max(H, 1), 8), np.float32)
        n = C.c_int(0)
        self._chk(self.L.hop_reject_by_collision(self.h, keep.ctypes.data_as(C.POINTER(C.c_ubyte)), F(diag), C.byref(n)), "hop_reject_by_collision")
        return keep[:H].astype(bool), diag[:H]

    def physics_timing(self):
        a, b = C.c_double(0), C.c_double(0)
        self.L.hop_physics_timing(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    # ---- resident set
    def hypos_upload(self, poses, scores=None):
        T = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 16)
        s = None if scores is None else np.ascontiguousarray(scores, dtype=np.float32)
        self._chk(self.L.hop_hypos_upload(self.h, F(T), F(s) if s is not None else None, len(T)), "hop_hypos_upload")

    def hypos_count(self):
        return self.L.hop_hypos_count(self.h)

    def hypos_download(self):
        n = self.hypos_count()
        pose = np.zeros((max(n, 1), 16), np.float32)
        sc = np.zeros(max(n, 1), np.float32)
        ids = np.zeros(max(n, 1), np.int32)
        m = C.c_int(0)
        self._chk(self.L.hop_hypos_download(self.h, F(pose), F(sc), I(ids), max(n, 1), C.byref(m)), "hop_hypos_download")
        return pose[:n].reshape(-1, 4, 4), sc[:n], ids[:n]

    def hypos_keep_topk(self, k):
        self._chk(self.L.hop_hypos_keep_topk(self.h, k), "hop_hypos_keep_topk")

    def icp_refine(self, max_iter=10, angle_deg=45.0, max_corr_dist=0.01, max_hypotheses=0, nn_mode=0, want_stats=False):
        o = IcpOpts(max_iter, angle_deg, max_corr_dist, max_hypotheses, nn_mode)
        if want_stats:
            n = min(self.hypos_count(), max_hypotheses) if max_hypotheses > 0 else self.hypos_count()
            it = np.zeros(max(n, 1), np.int32)
            cv = np.zeros(max(n, 1), np.int32)
            self._chk(self.L.hop_icp_refine(self.h, C.byref(o), I(it), I(cv)), "hop_icp_refine")
            return it[:n], cv[:n]
        self._chk(self.L.hop_icp_refine(self.h, C.byref(o), None, None), "hop_icp_refine")

    def selfcheck(self, force=True):
        """What the library's first-use checks of the gfx950-specific instructions said on this device (hop_debug_selfcheck; a development
        aid outside include/hop.h).  Returns {"qrank": True | False | None, "mfma": ..., "icp_engine": "momm" | "momi" | None}: None = not
        checked yet.  False means the library runs a substitute kernel on this device (and has said so on stderr)."""
        L = self.L
        fn = L.hop_debug_selfcheck
        fn.restype, fn.argtypes = C.c_int, [_vp, C.c_int]
        bits = fn(self.h, 1 if force else 0)
        eng_fn = L.hop_debug_icp_engine
        eng_fn.restype, eng_fn.argtypes = C.c_int, [_vp]
        eng = eng_fn(self.h)
        return {"qrank": (bool(bits & 2) if bits & 1 else None), "mfma": (bool(bits & 8) if bits & 4 else None),
                "icp_engine": {1: "momm", 0: "momi"}.get(eng)}

    def icp_refine_reference(self, max_iter, angle_deg, max_corr_dist, max_hypotheses=0, want_stats=False):
        """The mirrors' refinement (host/PoseEstimator.h icp_refine_reference): ICP_NN_MODE_REFERENCE, and where nn_mode 7 refuses for want of
        packed model lists (HOP_E_STATE: a 5 mm model of >= 65535 points, a list outside the 16-bit cell frame) one retry with nn_mode 5 -- the
        same minimiser in its per-evaluation float form -- announced on stderr."""
        try:
            return self.icp_refine(max_iter, angle_deg, max_corr_dist, max_hypotheses, ICP_NN_MODE_REFERENCE, want_stats)
        except HopError as e:
            if e.status != -5 or ICP_NN_MODE_REFERENCE != 7:
                raise
            sys.stderr.write(f"hop: {e} -- retrying with nn_mode 5 (per-evaluation float form of the same minimiser)\n")
            return self.icp_refine(max_iter, angle_deg, max_corr_dist, max_hypotheses, 5, want_stats)

    def lcp_select_best(self, dist=0.001, angle_deg=10.0, nn_mode=0):
        o = LcpOpts(dist, angle_deg, nn_mode)
        pose = np.zeros(16, np.float32)
        sc = C.c_float(0)
        idx = C.c_int(0)
        self._chk(self.L.hop_lcp_select_best(self.h, C.byref(o), F(pose), C.byref(sc), C.byref(idx)), "hop_lcp_select_best")
        return pose.reshape(4, 4), float(sc.value), int(idx.value)

    def cluster_poses(self, angle_deg, dist, sym_deg, assign_id):
        s = np.ascontiguousarray(sym_deg, dtype=np.float32)
        self._chk(self.L.hop_cluster_poses(self.h, angle_deg, dist, F(s), int(bool(assign_id))), "hop_cluster_poses")

    def topk_pack(self, k, id_offset=0):
        rows = np.zeros((k, TOPK_ROW_FLOATS), np.float32)
        n = C.c_int(0)
        self._chk(self.L.hop_topk_pack(self.h, k, id_offset, F(rows), C.byref(n)), "hop_topk_pack")
        return rows, n.value

    def topk_pack_device(self, k, id_offset, rows_dev_ptr):
        """the same table written into DEVICE memory (k * 18 floats at rows_dev_ptr): no download of the set, no host sort"""
        n = C.c_int(0)
        self._chk(self.L.hop_topk_pack_device(self.h, int(k), int(id_offset), C.c_void_p(int(rows_dev_ptr)), C.byref(n)), "hop_topk_pack_device")
        return n.value

    # ---- hand
    def hand_set_scene(self, scene_xyz, lookup_nrm, swivel_xyz):
        S, Ln, W = soa(scene_xyz), soa(lookup_nrm), soa(swivel_xyz)
        self._chk(self.L.hop_hand_set_scene(self.h, F(S), S.shape[1], F(Ln), Ln.shape[1], F(W), W.shape[1]), "hop_hand_set_scene")

    def model_ppf_keys(self, xyz, nrm, cap=1 << 20):
        """Key table of a model cloud (computePPF.cpp:88-100): (n_keys, 4) int32, sorted."""
        X, Nn = soa(xyz), soa(nrm)
        out = np.zeros((cap, 4), np.int32)
        k = C.c_int(0)
        self._chk(self.L.hop_model_ppf_keys(self.h, F(X), F(Nn), X.shape[1], I(out), cap, C.byref(k)), "hop_model_ppf_keys")
        return out[:k.value].copy()

    def hand_remove_surrounding(self, scene_xyz, scene_nrm, handbase_in_cam, links, finger12_in_handbase, finger22_in_handbase, min_z):
        """links: list of (xyz (n,3) in the hand-base frame, squared distance threshold), in name order.
        Returns (xyz, nrm, conf, keep_index) of the survivors, in input order, camera frame."""
        X, Nn = soa(scene_xyz), soa(scene_nrm)
        n = X.shape[1]
        clouds = [soa(l[0]) for l in links]
        arr = (HandLink * max(len(links), 1))()
        for k, (cl, l) in enumerate(zip(clouds, links)):
            arr[k].xyz, arr[k].n, arr[k].sq_dist_thres = F(cl), cl.shape[1], float(np.float32(l[1]))
        ox, on = np.zeros((3, max(n, 1)), np.float32), np.zeros((3, max(n, 1)), np.float32)
        oc, ki = np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.int32)
        T = np.ascontiguousarray(handbase_in_cam, np.float32).reshape(16)
        f1 = np.ascontiguousarray(finger12_in_handbase, np.float32).reshape(16)
        f2 = np.ascontiguousarray(finger22_in_handbase, np.float32).reshape(16)
        k = C.c_int(0)
        self._chk(self.L.hop_hand_remove_surrounding(self.h, F(X), F(Nn), n, F(T), arr, len(links), F(f1), F(f2), float(min_z), F(ox), F(on),
                                                     F(oc), I(ki), C.byref(k)), "hop_hand_remove_surrounding")
        k = k.value
        return ox[:, :k].T.copy(), on[:, :k].T.copy(), oc[:k].copy(), ki[:k].copy()

    def hand_set_finger(self, args: "FingerArgs"):
        self._chk(self.L.hop_hand_set_finger(self.h, C.byref(args)), "hop_hand_set_finger")

    def hand_set_sum_mode(self, mode):
        self._chk(self.L.hop_hand_set_sum_mode(self.h, int(mode)), "hop_hand_set_sum_mode")

    def hand_pso_eval_batch(self, angles):
        a = np.ascontiguousarray(angles, dtype=np.float64)
        out = np.zeros(len(a), np.float64)
        self._chk(self.L.hop_hand_pso_eval_batch(self.h, D(a), len(a), D(out)), "hop_hand_pso_eval_batch")
        return out

    def hand_pso_search(self, settings: "PsoSettings"):
        ang = C.c_double(0)
        val = C.c_double(0)
        self._chk(self.L.hop_hand_pso_search(self.h, C.byref(settings), C.byref(ang), C.byref(val)), "hop_hand_pso_search")
        return float(ang.value), float(val.value)

    # ---- timing
    def timing_enable(self, on=True):
        self.L.hop_timing_enable(self.h, int(on))

    def timing_reset(self):
        self.L.hop_timing_reset(self.h)

    def timing_get(self):
        t = Timing()
        self._chk(self.L.hop_timing_get(self.h, C.byref(t)), "hop_timing_get")
        return t.as_dict()

    def synchronize(self):
        self._chk(self.L.hop_synchronize(self.h), "hop_synchronize")


def organized_cloud(depth_raw, K, depth_unit=0.001):
    """Utils::readDepthImage + Utils::convert3dOrganizedRGB (Utils.cpp:36-55,79-115) without colours: (H, W, 3) float32, a
    dropped pixel is (0, 0, 0) (bad_point)."""
    K = np.asarray(K, np.float32).reshape(3, 3)
    d = (np.asarray(depth_raw).astype(np.float32).astype(np.float64) * depth_unit).astype(np.float32)
    d[(d > 2.0) | (d < 0.1)] = 0
    H, W = d.shape
    v, u = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    ok = (d > 0.1) & (d < 2.0)
    x = ((v - K[0, 2]) * d / K[0, 0]).astype(np.float32)
    y = ((u - K[1, 2]) * d / K[1, 1]).astype(np.float32)
    return np.where(ok[..., None], np.stack([x, y, d], axis=-1), 0).astype(np.float32)


class Comm:
    """One RCCL communicator per process / GPU (hop_comm_*): the per-frame all-gather + merge of the top-k tables."""

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * 128)()
        rc = lib().hop_comm_unique_id(buf)
        if rc:
            raise HopError(rc, "hop_comm_unique_id", (lib().hop_comm_last_error(None) or b"").decode())
        return bytes(buf)

    def __init__(self, device, unique_id, rank, world):
        self.L = lib()
        self.h = _vp()
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        rc = self.L.hop_comm_create(int(device), buf, int(rank), int(world), C.byref(self.h))
        if rc:
            raise HopError(rc, "hop_comm_create", (self.L.hop_comm_last_error(None) or b"").decode())

    def topk_allgather(self, rows, k):
        t = np.ascontiguousarray(rows, dtype=np.float32).reshape(k, TOPK_ROW_FLOATS)
        out = np.zeros((k, TOPK_ROW_FLOATS), np.float32)
        n = C.c_int(0)
        rc = self.L.hop_topk_allgather(self.h, F(t), k, F(out), C.byref(n))
        if rc:
            raise HopError(rc, "hop_topk_allgather", (self.L.hop_comm_last_error(self.h) or b"").decode())
        return out, n.value

    def topk_allgather_device(self, rows_dev_ptr, k):
        """rows on the device (Context.topk_pack_device) -> ncclAllGather -> merge kernel: only the k merged rows come back to the host"""
        out = np.zeros((k, TOPK_ROW_FLOATS), np.float32)
        n = C.c_int(0)
        rc = self.L.hop_topk_allgather_device(self.h, C.c_void_p(int(rows_dev_ptr)), int(k), F(out), C.byref(n))
        if rc:
            raise HopError(rc, "hop_topk_allgather_device", (self.L.hop_comm_last_error(self.h) or b"").decode())
        return out, n.value

    def info(self):
        """(ncclCommCount, mean exchange time in microseconds, number of exchanges)"""
        n, us, cnt = C.c_int(0), C.c_double(0), C.c_long(0)
        self.L.hop_comm_info(self.h, C.byref(n), C.byref(us), C.byref(cnt))
        return n.value, us.value, cnt.value

    def frames_allgather(self, rows, rows_per_rank, world):
        """C4: rows (n_local, 17) = frame index + 4 x 4 pose; returns (world * rows_per_rank, 17), padding rows have index -1"""
        r = np.ascontiguousarray(rows, np.float32).reshape(-1, FRAME_ROW_FLOATS)
        out = np.zeros((world * rows_per_rank, FRAME_ROW_FLOATS), np.float32)
        rc = self.L.hop_frames_allgather(self.h, F(r) if len(r) else None, len(r), int(rows_per_rank), F(out))
        if rc:
            raise HopError(rc, "hop_frames_allgather", (self.L.hop_comm_last_error(self.h) or b"").decode())
        return out

    def close(self):
        if self.h:
            self.L.hop_comm_destroy(self.h)
            self.h = _vp()


def topk_merge(tables, k):
    t = np.ascontiguousarray(tables, dtype=np.float32).reshape(-1, k, TOPK_ROW_FLOATS)
    out = np.zeros((k, TOPK_ROW_FLOATS), np.float32)
    n = C.c_int(0)
    rc = lib().hop_topk_merge(F(t), t.shape[0], k, F(out), C.byref(n))
    if rc:
        raise HopError(rc, "hop_topk_merge")
    return out, n.value


def cluster_poses_host(poses, scores, ids, angle_deg, dist, sym_deg):
    T = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 16)
    s = np.ascontiguousarray(scores, dtype=np.float32)
    i = np.ascontiguousarray(ids, dtype=np.int32)
    sym = np.ascontiguousarray(sym_deg, dtype=np.float32)
    keep = np.zeros(max(len(T), 1), np.int32)
    n = C.c_int(0)
    rc = lib().hop_cluster_poses_host(F(T), F(s), I(i), len(T), angle_deg, dist, F(sym), I(keep), C.byref(n))
    if rc:
        raise HopError(rc, "hop_cluster_poses_host")
    return keep[:n.value].copy()


def compute_ppf(ctx, cloud_xyz, downsample_size=0.001, normal_radius=0.003, ppf_density=0.005):
    """The offline tool src/perception/src/app/computePPF.cpp:56-107 chained from the library's calls: 1 mm voxel grid, centring on the
    bounding-box mid point, Utils::calNormalMLS (radius 3 mm, points replaced by their projections), normals flipped outward
    (flipNormalTowardsViewpoint towards the origin, then negated), the cloud the tool saves as model.ply, the 5 mm voxel grid of it and
    the PPF keys of all its ordered pairs.  Returns dict(model001=(xyz, nrm) [centred, as model.ply], model=(xyz, nrm) [5 mm, moved back
    as the tool does before it writes the table], keys (n, 4) int32, mid (3,))."""
    x1 = ctx.voxel_downsample(np.ascontiguousarray(cloud_xyz, np.float32), downsample_size)
    mn, mx = x1.min(axis=0), x1.max(axis=0)
    mid = ((mn + mx).astype(np.float32).astype(np.float64) / 2.0).astype(np.float32)     # (minPt.x + maxPt.x) / 2.0 assigned to a float
    xc = (x1 - mid).astype(np.float32)
    px, pn, _, _ = ctx.normals_mls(xc, normal_radius, 2)
    # pcl::flipNormalTowardsViewpoint(pt, 0, 0, 0, n): flip if the normal points away from the viewpoint; then the tool negates it
    vp = -px.astype(np.float32)
    flip = np.einsum("ij,ij->i", vp, pn) < 0
    pn = np.where(flip[:, None], -pn, pn)
    pn = (-pn).astype(np.float32)
    m5x, m5n = ctx.voxel_downsample_normals(px, pn, ppf_density)
    keys = ctx.model_ppf_keys(m5x, m5n)
    return dict(model001=(px, pn), model=((m5x + mid).astype(np.float32), m5n), keys=keys, mid=mid)


def cluster_pose_terms(pose_a, pose_b):
    """(eulerAngles(2,1,0) of pose_a [3], rotationGeodesicDistance(R_a, R_b), |t_a - t_b|) as clusterPoses evaluates them."""
    a = np.ascontiguousarray(pose_a, dtype=np.float32).reshape(16)
    b = np.ascontiguousarray(pose_b, dtype=np.float32).reshape(16)
    out = np.zeros(5, np.float32)
    rc = lib().hop_cluster_pose_terms(F(a), F(b), F(out))
    if rc:
        raise HopError(rc, "hop_cluster_pose_terms")
    return out


def rows_to_hypos(rows):
    rows = np.asarray(rows, dtype=np.float32).reshape(-1, TOPK_ROW_FLOATS)
    score = rows[:, 0].copy()
    ids = rows[:, 1].copy().view(np.int32)
    pose = rows[:, 2:].reshape(-1, 4, 4).copy()
    return pose, score, ids


# =============================================================================================== mirrors
class PoseHypo:
    """include/PoseHypo.h:7-26"""

    def __init__(self, pose=None, id=-1, lcp_score=0.0):
        self._pose = np.eye(4, dtype=np.float32) if pose is None else np.asarray(pose, np.float32).reshape(4, 4)
        self._wrong_ratio = 1.0
        self._lcp_score = float(lcp_score)
        self._id = int(id)


class PoseEstimator:
    """Mirror of PoseEstimator<PointT> for the hot-path members (PoseEstimator.h:15-22).

    ``cfg`` is the parsed config_autodataset.yaml (a dict); ``model``/``model001`` are (xyz, normals).
    """

    def __init__(self, cfg, model, model001, ctx=None, device=0):
        self.cfg = cfg
        self.ctx = ctx or Context(device)
        self.ctx.set_model(HOP_MODEL_5MM, *model)
        self.ctx.set_model(HOP_MODEL_1MM, *model001)
        self.ctx.model_owner = self
        self._models = ((np.asarray(model[0], np.float32), np.asarray(model[1], np.float32)),
                        (np.asarray(model001[0], np.float32), np.asarray(model001[1], np.float32)))
        self._pose_hypos = []
        self.last_stats = None
        self._model = np.asarray(model[0], np.float32)
        # PoseEstimator.cpp:12-20: centroid (pcl::computeCentroid sums in float, in order), bounding box of the 1 mm model
        m1 = np.asarray(model001[0], np.float32)
        self._model_center_init = (np.cumsum(m1, axis=0, dtype=np.float32)[-1] / np.float32(len(m1))).astype(np.float32)
        ext = (m1.max(axis=0) - m1.min(axis=0)).astype(np.float32)
        self._smallest_dim = np.float32(ext.min())
        self._ob_diameter = np.float32(np.sqrt(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]))
        self._cloud_withouthand_raw = None
        self._mesh_ids = {}
        self._slots = dict(self.MESH_SLOTS)   # per instance: new names do not leak into other estimators
        self._obj_mesh = None
        self._depth_raw = self._depth_unit = self._K = None

    def setCurScene(self, object_segment_xyz, object_segment_nrm, confidence, cloud_withouthand_raw=None, depth_raw=None, depth_unit=0.001, K=None):
        """PoseEstimator::setCurScene (PoseEstimator.cpp:32-46); depth_raw / K: the frame's depth image and intrinsics
        (_depth_meters, the constructor's K), needed by rejectByRender only."""
        thres = float(self.cfg.get("pose_estimator_high_confidence_thres", 0.8))
        if getattr(self.ctx, "model_owner", self) is not self:   # HandT42.handbaseICP borrowed the context's model slot
            self.ctx.set_model(HOP_MODEL_5MM, *self._models[0])
            self.ctx.set_model(HOP_MODEL_1MM, *self._models[1])
            self.ctx.model_owner = self
        if cloud_withouthand_raw is not None:
            self._cloud_withouthand_raw = np.asarray(cloud_withouthand_raw, np.float32)
        if depth_raw is not None:
            self._depth_raw, self._depth_unit, self._K = np.asarray(depth_raw, np.uint16), float(depth_unit), np.asarray(K, np.float32)
        return self.ctx.set_scene(object_segment_xyz, object_segment_nrm, confidence, thres)

    # ---- physics row (N1)
    MESH_SLOTS = {"object": 0, "finger_1_1": 1, "finger_1_2": 2, "finger_2_1": 3, "finger_2_2": 4}

    def registerMesh(self, V, Fi, name, pose=None):
        """PoseEstimator::registerMesh -> SDFchecker::registerMesh (PoseEstimator.cpp:505-508); the OBJ file of the
        reference is passed as arrays."""
        mid = self._slots.setdefault(name, len(self._slots))
        self.ctx.sdf_register_mesh(mid, V, Fi, pose)
        self._mesh_ids[name] = mid
        if name == "object":
            self._obj_mesh = (np.asarray(V, np.float32), np.asarray(Fi, np.int32))   # _obj_mesh, what rejectByRender draws

    def rejectByRender(self, projection_thres, hand, handbase_in_cam, sum_mode=0):
        """PoseEstimator::rejectByRender (PoseEstimator.cpp:345-463; projection_thres is unused there as well): the meshes
        of the matched hand components at handbase_in_cam * getTFHandBase(name), the object mesh under every hypothesis,
        _wrong_ratio from the depth difference to the frame's depth image, the best render_keep_hypo share kept.
        Returns (wrong ratios of the incoming set, kept positions)."""
        if self._depth_raw is None or self._obj_mesh is None:
            raise HopError(-5, "rejectByRender", "setCurScene(depth_raw=..., K=...) and registerMesh(..., 'object') come first")
        Vs, Fs, off = [], [], 0
        for name, (V, Fi) in hand.hand.meshes.items():          # hand->_meshes, map order; unmatched components are skipped (:364-365)
            if not hand._component_status.get(name, False):
                continue
            T = np.asarray(handbase_in_cam, np.float32) @ hand.getTFHandBase(name)
            V = np.asarray(V, np.float32)
            Vs.append((V @ T[:3, :3].T + T[:3, 3]).astype(np.float32))
            Fs.append(np.asarray(Fi, np.int32) + off)
            off += len(V)
        hV = np.concatenate(Vs) if Vs else np.zeros((0, 3), np.float32)
        hF = np.concatenate(Fs) if Fs else np.zeros((0, 3), np.int32)
        self.ctx.render_set_frame(self._depth_raw, self._depth_unit, self._K, hV, hF)
        self.ctx.render_set_object(*self._obj_mesh)
        return self.ctx.reject_by_render(float(self.cfg.get("render_roi_weight", 2.0)), float(self.cfg.get("render_keep_hypo", 0.3)), sum_mode)

    def registerHandMesh(self, hand):
        """PoseEstimator.cpp:510-520: the four finger links' convex meshes at getTFHandBase(link)."""
        for name in ("finger_1_1", "finger_1_2", "finger_2_1", "finger_2_2"):
            if name in hand.hand.meshes:
                V, Fi = hand.hand.meshes[name]
                self.registerMesh(V, Fi, name, hand.getTFHandBase(name))

    def rejectByCollisionOrNonTouching(self, hand, handbase_in_cam):
        """PoseEstimator.cpp:524-735.  Returns (keep mask, diagnostics) of the incoming hypothesis set, or None when
        pose_estimator_use_physics is off."""
        if not bool(self.cfg.get("pose_estimator_use_physics", True)):
            return None
        names = ("finger_1_1", "finger_1_2", "finger_2_1", "finger_2_2")
        clouds = getattr(hand, "_hand_clouds", None) or hand.makeHandCloud()
        p = dict(
            object_mesh=self._mesh_ids["object"], finger_mesh=[self._mesh_ids[n] for n in names],
            finger_xyz=[np.asarray(hand.hand.clouds[n][0], np.float32) for n in names],
            finger2handbase=[hand.getTFHandBase(n) for n in names],
            finger_status=[int(bool(hand._component_status.get(n, False))) for n in names],
            hand_cloud=np.concatenate([clouds[n] for n in sorted(clouds)]).astype(np.float32),  # (*_hand_cloud) += ..., map order
            cloud_without_hand=self._cloud_withouthand_raw,
            cam2handbase=np.linalg.inv(np.asarray(handbase_in_cam, np.float64)).astype(np.float32),
            model=self._model, model_center_init=self._model_center_init,
            smallest_dim=self._smallest_dim, ob_diameter=self._ob_diameter,
            collision_thres=float(self.cfg["collision_thres"]), non_touch_dist=float(self.cfg["non_touch_dist"]),
            collision_finger_dist=float(self.cfg["collision_finger_dist"]),
            collision_finger_volume_ratio=float(self.cfg["collision_finger_volume_ratio"]), voxel_size=0.005)
        self.ctx.physics_set_frame(p)
        return self.ctx.reject_by_collision()

    def runSuper4pcs(self, ppfs, n_trials=0, verify_mode=2):
        """PoseEstimator.cpp:62-100.  ``ppfs``: (n,4) int key table (the reference passes a std::map whose
        keys are these rows).  Returns False when no hypothesis is found, as the reference does."""
        self.ctx.set_ppf_keys(ppfs)
        c = self.cfg
        o = self.ctx.default_s4pcs_opts(
            sample_size=int(c["super4pcs_sample_size"]), overlap=float(c["super4pcs_overlap"]),
            max_time_seconds=int(c["super4pcs_max_time_seconds"]), delta=float(c["super4pcs_delta"]),
            dispersion=float(c["super4pcs_dispersion"]), success_quadrilaterals=int(c["super4pcs_success_quadrilaterals"]),
            max_normal_difference=float(c["super4pcs_max_normal_difference"]),
            max_color_distance=float(c["super4pcs_max_color_distance"]), n_trials=n_trials, verify_mode=verify_mode)
        _, _, st = self.ctx.s4pcs_generate(o, download=False)
        self.last_stats = st
        return st.n_hypotheses > 0

    def clusterPoses(self, angle_diff, dist_diff, assign_id):
        name = self.cfg["model_name"]
        s = self.cfg["object_symmetry"][name]
        self.ctx.cluster_poses(angle_diff, dist_diff, [s["x"], s["y"], s["z"]], assign_id)

    def refineByICP(self):
        # nn_mode 7: the reference's minimiser (PCL's point-to-plane estimator = Eigen's Levenberg-Marquardt, Utils.cpp:200-216)
        self.ctx.icp_refine_reference(10, float(self.cfg["icp_angle_thres"]), float(self.cfg["icp_dist_thres"]), max_hypotheses=100)

    def selectBest(self):
        pose, score, idx = self.ctx.lcp_select_best(float(self.cfg["lcp"]["dist"]), float(self.cfg["lcp"]["normal_angle"]), -1)
        ids = self.ctx.hypos_download()[2]     # the reference copies the winning PoseHypo: _id is the id clusterPoses assigned
        return PoseHypo(pose, int(ids[idx]) if 0 <= idx < len(ids) else idx, score)

    def hypos(self):
        pose, sc, ids = self.ctx.hypos_download()
        return [PoseHypo(pose[i], ids[i], sc[i]) for i in range(len(sc))]


def finger_property(xyz, num_division=10):
    """FingerProperty::FingerProperty (Hand.cpp:182-236): extremes, z stride and per-z-bin min/max (6 x N)."""
    xyz = np.asarray(xyz, np.float32)
    mn, mx = xyz.min(axis=0), xyz.max(axis=0)
    stride = np.float32((mx[2] - mn[2]) / np.float32(num_division))
    hist = np.empty((6, num_division), np.float32)
    hist[:3] = np.finfo(np.float32).max
    hist[3:] = -np.finfo(np.float32).max
    bins = (np.maximum(xyz[:, 2] - mn[2], np.float32(0)) / stride).astype(np.int32)
    bins = np.clip(bins, 0, num_division - 1)
    changed = np.zeros(num_division, bool)
    for b in range(num_division):
        sel = xyz[bins == b]
        if len(sel):
            hist[:3, b] = sel.min(axis=0)
            hist[3:, b] = sel.max(axis=0)
            changed[b] = True
    for i in range(num_division):
        if changed[i]:
            continue
        for j in range(i + 1, num_division):
            if changed[j]:
                hist[:, i] = hist[:, j]
                changed[i] = True
                break
    if not changed[-1]:
        for i in range(num_division - 2, -1, -1):
            if changed[i]:
                hist[:, -1] = hist[:, i]
                break
    return dict(min=mn, max=mx, stride_z=stride, hist=hist, num_division=num_division)


def euler_zyx_pitch(R):
    """Second angle of Eigen's R.eulerAngles(2,1,0) (Geometry/EulerAngles.h:36-108)."""
    R = np.asarray(R, np.float64)
    a0 = math.atan2(R[1, 0], R[0, 0])
    c2 = math.hypot(R[2, 2], R[2, 1])
    if a0 < 0:
        return math.atan2(-R[2, 0], -c2)
    return math.atan2(-R[2, 0], c2)


class HandT42:
    """Mirror of Hand/HandT42 for the hand-state search (Hand.h:29-92).

    ``hand`` is a kinematic description (hop_amd.synth.HandModel or the equivalent parsed from a URDF):
    link clouds at 5 mm, parents and link->parent transforms.
    """

    PAIR = {"finger_1_1": "finger_2_1", "finger_2_1": "finger_1_1", "finger_1_2": "finger_2_2", "finger_2_2": "finger_1_2"}

    def __init__(self, cfg, hand, ctx=None, device=0):
        self.cfg = cfg
        self.hand = hand
        self._ctx = ctx
        self._device = device
        self._tf_self = {n: np.eye(4, dtype=np.float32) for n in hand.parents}
        self._component_status = {n: False for n in hand.parents}
        self._finger_properties = {n: finger_property(hand.clouds[n][0], 10) for n in hand.parents if "finger" in n}
        self.gripper_min_dist = 0.0
        self._icp_ctx = None
        hm = cfg["hand_match"]
        self._pso = dict(n_pop=int(hm["pso"]["n_pop"]), n_gen=int(hm["pso"]["n_gen"]), check_freq=int(hm["pso"]["check_freq"]),
                         c_cog=float(hm["pso"]["pso_par_c_cog"]), c_soc=float(hm["pso"]["pso_par_c_soc"]),
                         initial_w=float(hm["pso"]["pso_par_initial_w"]))

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = Context(self._device)
        return self._ctx

    def setHandbaseIcpContext(self, ctx):
        """handbaseICP runs Utils::runICP on its own scene / model: give it a context of its own (host/Hand.h
        setHandbaseIcpContext does the same) and the PoseEstimator's models, with the lists built for them, stay on the main one."""
        self._icp_ctx = ctx

    def setCurScene(self, scene_removed_noise_xyz, scene_hand_region_nrm, scene_remove_swivel_xyz):
        """Products of Hand::setCurScene (Hand.cpp:327-332), all in the hand-base frame."""
        self.ctx.hand_set_scene(scene_removed_noise_xyz, scene_hand_region_nrm, scene_remove_swivel_xyz)

    def handbaseICP(self, scene_xyz_cam, scene_nrm_cam, handbase_in_cam):
        """Hand::handbaseICP (Hand.cpp:677-777): corrects handbase_in_cam by a point-to-plane ICP of the scene around the
        palm against the base_link cloud.  Returns (new handbase_in_cam, cam2handbase_offset).  Uses the context's scene,
        5 mm model and hypothesis slots for the ICP (Utils::runICP is the function refineByICP calls): run it before the
        PoseEstimator of the frame is set up, as main_realdata_auto.cpp does (:99 vs :183) -- or, with setHandbaseIcpContext, a
        context of its own for the ICP (the voxel grid and the crop before it use no slot)."""
        c = self.ctx
        ci = self._icp_ctx or c
        handbase_in_cam = np.asarray(handbase_in_cam, np.float32)
        cam_in_handbase = np.linalg.inv(handbase_in_cam.astype(np.float64)).astype(np.float32)
        sx, sn = c.voxel_downsample_normals(scene_xyz_cam, scene_nrm_cam, 0.005)
        t1, t2 = self.hand.tf_in_parent["finger_1_1"], self.hand.tf_in_parent["finger_2_1"]
        hx, hn, keep = c.handbase_region(sx, sn, cam_in_handbase, float(t1[1, 3]), float(t1[2, 3]), float(t2[1, 3]), float(t2[2, 3]))
        offset = np.eye(4, dtype=np.float32)
        if keep.sum() > 0:
            bx, bn = self.hand.clouds["base_link"]
            ci.set_scene(hx[keep], hn[keep], None, 0.0)         # Utils::runICP source (pclSegment)
            if getattr(ci, "model_owner", None) is not self.hand:  # target (pclModel): base_link, uploaded once per context
                ci.set_model(HOP_MODEL_5MM, bx, bn)
                ci.model_owner = self.hand                       # (a PoseEstimator on this context uploads its models again)
            ci.hypos_upload(np.eye(4, dtype=np.float32)[None])
            ci.icp_refine_reference(50, 30.0, 0.03)
            pose, _, _ = ci.hypos_download()
            offset = np.linalg.inv(pose[0].astype(np.float64)).astype(np.float32)  # source -> target
        translation = float(np.linalg.norm(offset[:3, 3]))
        if translation >= 0.05:                                   # :740-745
            offset = np.eye(4, dtype=np.float32)
        R = offset[:3, :3].astype(np.float64)
        rot_diff = math.degrees(math.acos(max(-1.0, min(1.0, (np.trace(R) - 1) / 2.0))))
        pitch = euler_zyx_pitch(offset[:3, :3])
        pitch = min(abs(pitch), abs(math.pi - pitch))
        pitch = min(abs(pitch), abs(math.pi + pitch))
        if rot_diff >= 10 or abs(pitch) >= 10 / 180.0 * math.pi:  # :752-756
            offset = np.eye(4, dtype=np.float32)
        if not np.array_equal(offset, np.eye(4, dtype=np.float32)):
            self._component_status["handbase"] = True
        new = (handbase_in_cam.astype(np.float64) @ np.linalg.inv(offset.astype(np.float64))).astype(np.float32)
        return new, offset

    TRIAL_HEIGHTS = (-0.03, -0.025, -0.02, -0.015, -0.01, -0.005, 0, 0.005, 0.01, 0.015, 0.02, 0.025, 0.03)

    def adjustHandHeight(self, region_xyz_cam, region_nrm_cam, handbase_in_cam):
        """HandT42::adjustHandHeight (Hand.cpp:999-1051): when handbaseICP did not fix the hand base, try 13 offsets along
        the hand-base z axis and keep the one under which most hand points find a scene point within 5 mm / 45 degrees.
        ``region``: the 3 mm hand-region cloud (camera frame).  Returns (new handbase_in_cam, best height, counts)."""
        clouds = self.makeHandCloud()
        handbase_in_cam = np.asarray(handbase_in_cam, np.float32)
        if self._component_status.get("handbase", False):
            return handbase_in_cam, 0.0, None
        cam_in_handbase = np.linalg.inv(handbase_in_cam.astype(np.float64)).astype(np.float32)
        x = np.asarray(region_xyz_cam, np.float32)
        n = np.asarray(region_nrm_cam, np.float32)
        T = cam_in_handbase
        sx, sn = np.empty_like(x), np.empty_like(n)
        for k in range(3):   # pcl::transformPointCloudWithNormals in float
            sx[:, k] = ((T[k, 0] * x[:, 0] + T[k, 1] * x[:, 1]) + T[k, 2] * x[:, 2]) + T[k, 3]
            sn[:, k] = (T[k, 0] * n[:, 0] + T[k, 1] * n[:, 1]) + T[k, 2] * n[:, 2]
        names = sorted(clouds)
        hx = np.concatenate([clouds[k] for k in names]).astype(np.float32)
        hn = np.concatenate([self._hand_cloud_normals[k] for k in names]).astype(np.float32)
        counts = self.ctx.hand_height_matches(sx, sn, hx, hn, self.TRIAL_HEIGHTS)
        best_height, max_match = 0.0, 0
        for h, cnt in zip(self.TRIAL_HEIGHTS, counts):   # first strict maximum (:1042-1047)
            if cnt > max_match:
                max_match, best_height = int(cnt), float(h)
        offset = np.eye(4, dtype=np.float32)
        offset[2, 3] = best_height
        new = handbase_in_cam.copy()      # handbase_in_cam * offset (:1050) in float, term by term (no fused multiply-add of a BLAS kernel)
        new[:3, 3] = handbase_in_cam[:3, 2] * np.float32(best_height) + handbase_in_cam[:3, 3]
        return new, best_height, counts

    def setCurSceneFromRegion(self, region_xyz_cam, region_nrm_cam, handbase_in_cam):
        """Hand::setCurScene from the 3 mm hand-region cloud in the camera frame (Hand.cpp:289-332, after handbaseICP):
        hand-base transform, the two radius outlier filters, the statistical outlier filter and the x pass-through run on
        the GPU (hop_hand_scene_filters); the three products go to hop_hand_set_scene.  Returns their sizes."""
        cam_in_handbase = np.linalg.inv(np.asarray(handbase_in_cam, np.float64)).astype(np.float32)
        hx, hn, keep, swivel = self.ctx.hand_scene_filters(region_xyz_cam, region_nrm_cam, cam_in_handbase)
        self.setCurScene(hx[keep], hn, hx[swivel])
        return int(keep.sum()), len(hx), int(swivel.sum())

    def makeHandCloud(self):
        """Hand::makeHandCloud (Hand.cpp:537-556): every component cloud in the hand-base frame at the current finger
        state, keyed by name (the reference keeps a kd-tree per component; std::map order = sorted names)."""
        out, nrm = {}, {}
        for name in sorted(self.hand.clouds):
            T = self.getTFHandBase(name) if name != "base_link" else np.eye(4, dtype=np.float32)
            x = np.asarray(self.hand.clouds[name][0], np.float32)
            p = np.empty_like(x)
            for k in range(3):   # pcl::transformPointCloud: ((m0 x + m1 y) + m2 z) + m3 in float
                p[:, k] = ((T[k, 0] * x[:, 0] + T[k, 1] * x[:, 1]) + T[k, 2] * x[:, 2]) + T[k, 3]
            out[name] = p
            m = np.asarray(self.hand.clouds[name][1], np.float32)
            q = np.empty_like(m)
            for k in range(3):   # normals: rotation part only
                q[:, k] = (T[k, 0] * m[:, 0] + T[k, 1] * m[:, 1]) + T[k, 2] * m[:, 2]
            nrm[name] = q
        self._hand_clouds = out
        self._hand_cloud_normals = nrm
        return out

    @staticmethod
    def local_dist_thres(name, dist_thres):
        """Hand.cpp:812-821 (squared metres, stored as float)."""
        if name in ("finger_2_1", "finger_1_1"):
            return np.float32(0.005 * 0.005)
        if name in ("base", "swivel_1", "swivel_2"):
            return np.float32(0.02 * 0.02)
        return np.float32(dist_thres)

    def removeSurroundingPointsAndAssignProbability(self, scene_xyz, scene_nrm, handbase_in_cam, dist_thres):
        """HandT42::removeSurroundingPointsAndAssignProbability (Hand.cpp:779-888); ``dist_thres`` is the SQUARED
        near_hand_dist as at the call site (main_realdata_auto.cpp:147-148).  Returns (xyz, nrm, confidence, index)."""
        clouds = getattr(self, "_hand_clouds", None) or self.makeHandCloud()
        links = [(clouds[name], self.local_dist_thres(name, dist_thres)) for name in sorted(clouds)]
        return self.ctx.hand_remove_surrounding(scene_xyz, scene_nrm, handbase_in_cam, links, self.getTFHandBase("finger_1_2"),
                                                self.getTFHandBase("finger_2_2"), float(self._finger_properties["finger_1_2"]["min"][2]))

    def getTFHandBase(self, name):
        """Hand.cpp:505-523 (float matrix products, link -> hand base)."""
        T = np.eye(4, dtype=np.float32)
        cur = name
        while cur != "base_link":
            T = (self.hand.tf_in_parent[cur].astype(np.float32) @ self._tf_self[cur]).astype(np.float32) @ T
            T = T.astype(np.float32)
            cur = self.hand.parents[cur]
        return T

    def _tip(self, name, use_max_z):
        p = self._finger_properties[name]
        return np.array([p["min"][0], p["max"][1], p["max"][2] if use_max_z else p["min"][2], 1.0], np.float32)

    def finger_args(self, model_name, dist_thres):
        """Fills the ArgPasser fields of Hand::matchOneComponentPSO (Hand.cpp:611-653)."""
        hm = self.cfg["hand_match"]
        pair_name = self.PAIR[model_name]
        a = FingerArgs()
        if pair_name in ("finger_1_1", "finger_2_1"):
            pair_out = "finger_1_2" if pair_name == "finger_1_1" else "finger_2_2"
            tip1 = self.getTFHandBase(pair_out) @ self._tip(pair_out, False)
            tip2 = self.getTFHandBase(pair_name) @ self._tip(pair_name, False)
            out_name = "finger_1_2" if model_name == "finger_1_1" else "finger_2_2"
            fo2p = self.hand.tf_in_parent[out_name].astype(np.float32)
            fo = self._finger_properties[out_name]
        else:
            Tb = self.getTFHandBase(pair_name)
            tip1 = Tb @ self._tip(pair_name, False)
            tip2 = Tb @ self._tip(pair_name, True)
            fo2p = np.eye(4, dtype=np.float32)
            fo = self._finger_properties[model_name]
        fpp = self._finger_properties[model_name]
        for k in range(3):
            a.fp_min[k], a.fp_max[k] = float(fpp["min"][k]), float(fpp["max"][k])
            a.fo_min[k], a.fo_max[k] = float(fo["min"][k]), float(fo["max"][k])
        a.fp_stride_z = float(fpp["stride_z"])
        a.fp_num_division = fpp["num_division"]
        hist = np.ascontiguousarray(fpp["hist"][1], np.float32)
        m2h = np.ascontiguousarray(self.getTFHandBase(model_name), np.float32).reshape(16)
        xyz, nrm = self.hand.clouds[model_name]
        X, Nn = soa(xyz), soa(nrm)
        self._keep = [hist, X, Nn]
        a.fp_hist_min_y = F(hist)
        for k in range(16):
            a.model2handbase[k] = float(m2h[k])
            a.finger_out2parent[k] = float(fo2p.reshape(16)[k])
        for k in range(4):
            a.pair_tip1[k], a.pair_tip2[k] = float(tip1[k]), float(tip2[k])
        a.is_palm_side = int(model_name in ("finger_1_1", "finger_2_1"))
        a.is_right_side = int(model_name in ("finger_2_1", "finger_2_2"))
        a.gripper_min_dist = float(self.gripper_min_dist)
        a.dist_thres = float(dist_thres)
        ang = hm["finger1_normal_angle"] if a.is_palm_side else hm["finger2_normal_angle"]
        a.cos_normal_thres = float(np.float32(math.cos(np.float32(ang) / 180.0 * math.pi)))
        a.check_normal = int(bool(hm["check_normal"]))
        a.max_outter_pts = int(hm["max_outter_pts"])
        a.outter_pt_dist = float(hm["outter_pt_dist"])
        a.outter_pt_dist_weight = float(hm["outter_pt_dist_weight"])
        a.model_xyz, a.model_nrm, a.n_model = F(X), F(Nn), X.shape[1]
        return a

    def pso_settings(self, min_angle, max_angle, seed=0):
        s = PsoSettings()
        lib().hop_pso_default_settings(C.byref(s))
        for k, v in self._pso.items():
            setattr(s, k, v)
        s.lower_rad = min_angle * math.pi / 180
        s.upper_rad = max_angle * math.pi / 180
        s.seed = seed
        return s

    def matchOneComponentPSO(self, model_name, min_angle, max_angle, use_normal, dist_thres, normal_angle_thres, least_match):
        """Hand.cpp:603-672.  ``use_normal`` and ``normal_angle_thres`` are ignored, as in the reference body
        (thresholds are re-read from the YAML, Hand.cpp:76-83)."""
        args = self.finger_args(model_name, dist_thres)
        self.ctx.hand_set_finger(args)
        angle, objval = self.ctx.hand_pso_search(self.pso_settings(min_angle, max_angle))
        self.last_objval = objval
        self._hand_clouds = None   # the link moves (or resets): a cached makeHandCloud() result would be stale
        if -objval <= least_match:
            self._tf_self[model_name] = np.eye(4, dtype=np.float32)
            self._component_status[model_name] = False
            return False
        a = np.float32(angle)
        T = np.eye(4, dtype=np.float32)
        T[1, 1], T[1, 2], T[2, 1], T[2, 2] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
        self._tf_self[model_name] = T
        self._component_status[model_name] = True
        self.last_angle = float(a)
        return True
