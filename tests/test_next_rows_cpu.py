"""CPU checks of the oracle restatements of the "next" rows against hand-computed cases."""
import math

import numpy as np


def test_remove_surrounding_hand_computed(orc):
    """HandT42::removeSurroundingPointsAndAssignProbability (Hand.cpp:779-888) on a case small enough to do by hand:
    identity frames, one link 'finger_1_1' (threshold (5 mm)^2) with a single point at the origin and one link with the
    caller's threshold (1 cm)^2 at x = 0.1."""
    eye = np.eye(4, dtype=np.float32)
    far = eye.copy()
    far[:3, 3] = [10, 10, 10]           # distal finger frames far away: the outer-side test never fires (y' < 0 but z' < min_z)
    links = [(np.float32([[0, 0, 0]]), np.float32(0.005 * 0.005)), (np.float32([[0.1, 0, 0]]), np.float32(0.01 * 0.01))]
    scene = np.float32([
        [0.004, 0, 0],       # within 5 mm of link 0                       -> removed
        [0.003, 0, 0.0049],  # d = 5.7 mm > 5 mm, planar 3 mm, |dz| <= 5 mm -> removed by the planar rule
        [0.003, 0, 0.0051],  # same but |dz| > 5 mm; link 1 is 9.7 cm away -> kept, min_dist = |p - link0|
        [0.093, 0, 0],       # 7 mm from link 1 (threshold 1 cm)           -> removed
        [0.05, 0.03, 0],     # far from both                                -> kept
    ])
    nrm = np.tile(np.float32([0, 0, 1]), (len(scene), 1))
    x, n, c, idx = orc.hand_remove_surrounding(scene, nrm, eye, links, far, far, 0.0)
    assert idx.tolist() == [2, 4]
    assert np.array_equal(x, scene[[2, 4]]) and np.array_equal(n, nrm[[2, 4]])
    lam = np.float32(231.04906018664843)
    d2 = np.float32(math.sqrt(float(np.float32(0.003) ** 2 + np.float32(0.0051) ** 2)))
    d4 = min(np.float32(np.linalg.norm(scene[4])), np.float32(np.linalg.norm(scene[4] - np.float32([0.1, 0, 0]))))
    exp = [1 - math.exp(-float(lam * d2)), 1 - math.exp(-float(lam * d4))]
    assert np.allclose(c, exp, rtol=2e-6, atol=0)
    # outer side of a distal finger: identity finger frame, min_z = 0: points with y < 0 and z >= 0 go
    x2, _, _, idx2 = orc.hand_remove_surrounding(np.float32([[0.05, -0.03, 0.01], [0.05, -0.03, -0.01], [0.05, 0.03, 0.01]]),
                                                 np.tile(np.float32([0, 0, 1]), (3, 1)), eye, links, eye, far, 0.0)
    assert idx2.tolist() == [1, 2]
    # a rigid camera pose changes nothing but the output frame
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = np.float32([[0, -1, 0], [1, 0, 0], [0, 0, 1]])
    T[:3, 3] = [0.2, -0.1, 0.5]
    cam = (scene @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    x3, _, c3, idx3 = orc.hand_remove_surrounding(cam, nrm, T, links, far, far, 0.0)
    assert idx3.tolist() == [2, 4] and np.abs(x3 - cam[[2, 4]]).max() < 1e-6 and np.allclose(c3, c, rtol=1e-4)


def test_model_ppf_keys_restatement(orc, hop):
    """Pair loop of the offline computePPF tool (computePPF.cpp:17-38,88-100): a hand-computable pair, and agreement
    with the numpy construction the synthetic key table uses."""
    # two points 10 mm apart along x with normals +z and +x: dist 10 -> bin 10; angles 90, 0, 90 -> bins 90, 0, 90
    k = orc.model_ppf_keys(np.float32([[0, 0, 0], [0.01, 0, 0]]), np.float32([[0, 0, 2], [3, 0, 0]]))
    assert k.tolist() == [[10, 90, 0, 90]]
    synth = hop.synth
    mx, mn = synth.ellipsoid_model_spacing(0.01)
    a = orc.model_ppf_keys(mx, mn)
    b = synth.ppf_keys_numpy(mx, mn)
    sa, sb = set(map(tuple, a.tolist())), set(map(tuple, b.tolist()))
    assert len(sa) > 100 and len(sa & sb) > 0.98 * len(sa | sb)   # numpy's arccos/float paths differ on a few boundary pairs
    assert a.tolist() == sorted(a.tolist())
