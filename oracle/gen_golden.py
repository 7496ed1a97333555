"""TEST INFRASTRUCTURE ONLY -- golden-vector generator (runs in the build container, where
/root/reference exists).

Runs the REFERENCE's own OpenGR fork (oracle/_ref/libref_s4pcs.so, compiled in place from
/root/reference by `make -C oracle ref`) on seeded synthetic inputs and stores inputs + outputs as
small .npz fixtures under tests/golden/.  The fixtures are data only (no reference source text).

    python oracle/gen_golden.py            # regenerates tests/golden/s4pcs_case*.npz, kat_*.npz
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import orc  # noqa: E402
from orc import F, I  # noqa: E402

spec = importlib.util.spec_from_file_location("hop_synth", os.path.join(ROOT, "icra20-hand-object-pose_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec)
sys.modules["hop_synth"] = synth
spec.loader.exec_module(synth)

OUT = os.path.join(ROOT, "tests", "golden")

# (name, scene points, scene seed, n_calls, sample_size, success_quadrilaterals)
CASES = [
    ("case1", 300, 7, 1, 100, 10),   # as-shipped option values (config_autodataset.yaml:133-140)
    ("case2", 500, 3, 1, 100, 10),
    ("case3", 250, 21, 2, 60, 4),    # repeated calls on one matcher (hypotheses accumulate)
    # BASELINE configs[0] ("C1-synthetic-model"): the hand-region cloud of the reference's example/depth7.png
    # (tests/golden/depth7_hand_region.npz, made by tools/make_depth7_fixture.py) against the synthetic ellipse
    ("depth7", -1, 0, 1, 100, 10),
]


def sort_rows(a):
    a = np.asarray(a)
    if len(a) == 0:
        return a
    return a[np.lexsort(a.T[::-1])]


def canonical_hypos(pose, lcp):
    """Sort by (lcp desc, then the 9 rotation entries ascending): a total order on the multiset."""
    flat = pose.reshape(len(pose), 16)
    rot = flat[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]]
    order = np.lexsort(tuple(rot[:, ::-1].T) + (-lcp,))
    return pose[order], lcp[order]


def gen_case(name, n_scene, seed, n_calls, sample_size, succ):
    mx, mn = synth.ellipsoid_model_spacing(0.005)
    keys = synth.ppf_key_table()
    if n_scene < 0:
        g = np.load(os.path.join(OUT, "depth7_hand_region.npz"))

        class Real:
            xyz, nrm = g["xyz"], g["nrm"]
            conf = np.ones(len(g["xyz"]), np.float32)
            gt_pose = np.eye(4, dtype=np.float32)
        sc = Real
    else:
        sc = synth.make_scene(n_scene, seed=seed)
    r = orc.RefS4PCS(sample_size=sample_size, success_quadrilaterals=succ)
    r.set_keys(keys)
    n = r.run(sc.xyz, sc.nrm, sc.conf, mx, mn, n_calls)
    st = r.state()
    bases = r.bases()
    pose, lcp = r.hypos()
    pose, lcp = canonical_hypos(pose, lcp)
    # Verify KATs on the final state: perturbed hypotheses in the centred frame
    rng = np.random.default_rng(1234 + seed)
    Ts, vs = [], []
    for k in range(200):
        Tg = pose[rng.integers(len(pose))].astype(np.float64)
        Tc = Tg.copy()
        Tc[:3, 3] = Tg[:3, :3] @ st["cQ"].astype(np.float64) + Tg[:3, 3] - st["cP"]
        d = synth.se3(synth.rot_from_axis_angle(rng.normal(size=3), rng.uniform(0, 0.3) ** 2),
                      rng.normal(size=3) * 0.002)
        Tc = (Tc @ d).astype(np.float32)
        Ts.append(Tc)
        vs.append(r.verify(Tc))
    data = dict(
        P_xyz=sc.xyz, P_nrm=sc.nrm, P_conf=sc.conf, Q_xyz=mx, Q_nrm=mn, keys=keys, gt_pose=sc.gt_pose,
        opts=np.array([sample_size, succ, n_calls], np.int32), opts_f=np.array([0.2, 0.003, 0.5], np.float32),
        Qs=st["Qs"], Qs_nrm=st["Qs_nrm"], cP=st["cP"], cQ=st["cQ"], diameter=np.float32(st["diameter"]),
        number_of_trials=np.int32(st["number_of_trials"]),
        n_bases=np.int32(len(bases)),
        base_ids=np.array([b["base"] for b in bases], np.int32),
        base_inv=np.array([b["inv"] for b in bases], np.float32),
        hyp_pose=pose, hyp_lcp=lcp,
        verify_T=np.array(Ts, np.float32), verify_lcp=np.array(vs, np.float32),
    )
    for i, b in enumerate(bases):
        data[f"pairs1_{i}"] = sort_rows(b["pairs1"]).astype(np.int16)
        data[f"pairs2_{i}"] = sort_rows(b["pairs2"]).astype(np.int16)
        data[f"quads_{i}"] = sort_rows(b["quads"]).astype(np.int16)
    path = os.path.join(OUT, f"s4pcs_{name}.npz")
    np.savez_compressed(path, **data)
    print(name, "hypotheses", n, "bases", len(bases), "->", path, os.path.getsize(path) // 1024, "KiB")


def gen_plain(name, n_scene, seed, n_calls, sample_size, succ):
    """Hypotheses of the UNTOUCHED reference matcher (no tracing override of generateCongruents, oracle/ref_driver.cpp) on
    the inputs of case `name`; must equal the traced run that produced s4pcs_<name>.npz."""
    mx, mn = synth.ellipsoid_model_spacing(0.005)
    keys = synth.ppf_key_table()
    sc = synth.make_scene(n_scene, seed=seed)
    out = {}
    for plain in (True, False):
        r = orc.RefS4PCS(sample_size=sample_size, success_quadrilaterals=succ, plain=plain)
        r.set_keys(keys)
        r.run(sc.xyz, sc.nrm, sc.conf, mx, mn, n_calls)
        out[plain] = canonical_hypos(*r.hypos())
    assert np.array_equal(out[True][0], out[False][0]) and np.array_equal(out[True][1], out[False][1]), "the tracing override changed the result"
    g = np.load(os.path.join(OUT, f"s4pcs_{name}.npz"))
    assert np.array_equal(out[True][0], g["hyp_pose"]) and np.array_equal(out[True][1], g["hyp_lcp"]), "differs from the committed traced golden"
    path = os.path.join(OUT, f"s4pcs_plain_{name}.npz")
    np.savez_compressed(path, hyp_pose=out[True][0], hyp_lcp=out[True][1], opts=np.array([sample_size, succ, n_calls, n_scene, seed], np.int32))
    print("plain", name, "hypotheses", len(out[True][1]), "== traced run ->", path, os.path.getsize(path) // 1024, "KiB")


def gen_kats():
    """Known-answer tests of the pure functions, straight from the reference build."""
    R = orc.ref()
    rng = np.random.default_rng(99)
    n = 4000
    p1 = (rng.normal(size=(n, 3)) * 0.03).astype(np.float32)
    p2 = (rng.normal(size=(n, 3)) * 0.03).astype(np.float32)
    n1 = rng.normal(size=(n, 3)).astype(np.float32)
    n2 = rng.normal(size=(n, 3)).astype(np.float32)
    n2[::10] = n1[::10]          # acos(>1) -> NaN -> INT_MIN path
    n2[5::17] = -n1[5::17]
    key = np.zeros((n, 4), np.int32)
    for i in range(n):
        R.ref_compute_ppf(F(p1[i]), F(n1[i]), F(p2[i]), F(n2[i]), I(key[i]))
    # pair filter
    m = 6000
    pts = np.concatenate([rng.normal(size=(m, 4, 3)) * 0.03, rng.normal(size=(m, 4, 3))], axis=2).astype(np.float32)
    for k in range(0, m, 2):  # make segment lengths comparable so the later tests are reached
        d = np.linalg.norm(pts[k, 0, :3] - pts[k, 1, :3])
        u = rng.normal(size=3)
        u /= np.linalg.norm(u)
        pts[k, 3, :3] = pts[k, 2, :3] + (u * (d + rng.normal() * 0.003)).astype(np.float32)
    good = np.zeros(m, np.int8)
    for k in range(m):
        a = np.ascontiguousarray(pts[k])
        good[k] = R.ref_pair_ppf_is_good(F(a[0]), F(a[1]), F(a[2]), F(a[3]))
    # 3-point rigid fit
    h = orc.RefS4PCS()
    q = 2000
    ref9 = (rng.normal(size=(q, 3, 3)) * 0.03).astype(np.float32)
    cand9 = np.zeros_like(ref9)
    T = np.zeros((q, 16), np.float32)
    rms = np.zeros(q, np.float32)
    ok = np.zeros(q, np.int8)
    for k in range(q):
        Rm = synth.random_rotation(rng)
        t = rng.normal(size=3) * 0.05
        c = (ref9[k].astype(np.float64) - t) @ Rm  # cand = R^T (ref - t)  => ref = R cand + t
        c += rng.normal(size=(3, 3)) * (0.0005 if k % 3 else 0.004)
        cand9[k] = c.astype(np.float32)
        e = np.zeros(1, np.float32)
        ok[k] = R.ref_rigid(h.h, F(np.ascontiguousarray(ref9[k].reshape(9))),
                            F(np.ascontiguousarray(cand9[k].reshape(9))), F(T[k]), F(e))
        rms[k] = e[0]
    path = os.path.join(OUT, "kat_pure.npz")
    np.savez_compressed(path, ppf_p1=p1, ppf_n1=n1, ppf_p2=p2, ppf_n2=n2, ppf_key=key, pf_pts=pts, pf_good=good,
                        rigid_ref=ref9, rigid_cand=cand9, rigid_T=T, rigid_rms=rms, rigid_ok=ok)
    print("kat_pure ->", path, os.path.getsize(path) // 1024, "KiB")


def sdf_query_points(rng, V, Fi, n_box=1200, n_far=300, n_vert=250, n_edge=250):
    """Query points around a mesh: in and around the bounding box, far away, next to vertices and to edge midpoints."""
    lo, hi = V.min(0), V.max(0)
    c, e = (lo + hi) / 2, hi - lo
    return np.concatenate([
        c + (rng.random((n_box, 3)) - 0.5) * e * 1.6,
        c + (rng.random((n_far, 3)) - 0.5) * e * 6,
        V[rng.integers(0, len(V), n_vert)] + rng.normal(scale=1e-4, size=(n_vert, 3)),
        (V[Fi[:, 0]] + V[Fi[:, 1]])[rng.integers(0, len(Fi), n_edge)] / 2 + rng.normal(scale=1e-3, size=(n_edge, 3)),
    ]).astype(np.float32)


def sdf_special_meshes():
    """Edge cases: one triangle, an open bumpy sheet, a closed box with a zero-area face, a duplicated vertex and an
    unreferenced vertex, and two disconnected components."""
    rng = np.random.default_rng(77)
    single = (np.array([[0, 0, 0], [0.03, 0, 0], [0, 0.02, 0.01]], np.float32), np.array([[0, 1, 2]], np.int32))
    g = np.linspace(-0.03, 0.03, 7)
    X, Y = np.meshgrid(g, g, indexing="ij")
    Z = 0.004 * np.sin(40 * X) * np.cos(30 * Y)
    Vs = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1).astype(np.float32)
    Fs = []
    for i in range(6):
        for j in range(6):
            a, b, c, d = i * 7 + j, (i + 1) * 7 + j, (i + 1) * 7 + j + 1, i * 7 + j + 1
            Fs += [(a, b, c), (a, c, d)]
    sheet = (Vs, np.asarray(Fs, np.int32))
    Vb, Fb = synth.box_mesh((-0.01, -0.02, -0.015), (0.01, 0.02, 0.015), (1, 2, 1))
    Vd = np.concatenate([Vb, Vb[:1], [[0.5, 0.5, 0.5]]]).astype(np.float32)  # duplicate of vertex 0, an unused vertex
    Fd = np.concatenate([Fb, [[0, 1, 1]], [[len(Vb), 2, 3]]]).astype(np.int32)  # zero-area face, a face on the duplicate
    degenerate = (Vd, Fd)
    V1, F1 = synth.ellipsoid_mesh((0.01, 0.01, 0.01), subdiv=1)
    two = (np.concatenate([V1, V1 + np.float32([0.05, 0, 0])]).astype(np.float32), np.concatenate([F1, F1 + len(V1)]).astype(np.int32))
    return {"single": single, "sheet": sheet, "degenerate": degenerate, "two_parts": two}


def gen_sdf():
    """Row N1: signed distances of the reference's own libigl (oracle/_ref/libref_sdf.so) on the synthetic meshes."""
    rng = np.random.default_rng(2024)
    meshes = {"ellipsoid": synth.ellipsoid_mesh(subdiv=2), "box": synth.box_mesh((0, 0, 0), (0.02, 0.012, 0.06)),
              "torus": synth.torus_mesh(), "lshape": synth.lshape_mesh()}
    meshes.update(sdf_special_meshes())
    data = {}
    for name, (V, Fi) in meshes.items():
        P = sdf_query_points(rng, V, Fi)
        S, Ii, Cc = orc.ref_signed_distance(P, V, Fi)
        data[f"{name}_V"], data[f"{name}_F"], data[f"{name}_P"] = V, Fi, P
        data[f"{name}_S"], data[f"{name}_I"], data[f"{name}_C"] = S, Ii, Cc
        print("sdf", name, "faces", len(Fi), "points", len(P), "inside", int((S < 0).sum()))
    path = os.path.join(OUT, "sdf_igl.npz")
    np.savez_compressed(path, **data)
    print("sdf ->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    if not orc.ref_available():
        orc.build()
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    if not only:
        gen_kats()
    if not only or "sdf" in only:
        gen_sdf()
    for c in CASES:
        if not only or c[0] in only:
            gen_case(*c)
    if not only or "plain" in only:
        gen_plain(*[c for c in CASES if c[0] == "case1"][0])
