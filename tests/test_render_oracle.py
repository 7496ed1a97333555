"""CPU tests of the oracle's restatement of PoseEstimator::rejectByRender (OpenGL is absent: "parity
unpinned") on triangles whose image can be computed by hand."""
import numpy as np

K = np.array([[600.0, 0, 320.0], [0, 600.0, 230.0], [0, 0, 1]], np.float32)   # cy off-centre: the vertical mirror shows
H, W = 480, 640


def _quad(x0, x1, y0, y1, z):
    V = np.array([[x0, y0, z], [x1, y0, z], [x1, y1, z], [x0, y1, z]], np.float32)
    F = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    return V, F


def test_fronto_parallel_square_pixels_and_millimetres(orc):
    # a square at z = 0.5004 m spanning x in [-0.05, 0.05], y in [-0.02, 0.03] (camera frame, y down)
    V, F = _quad(-0.05, 0.05, -0.02, 0.03, 0.5004)
    none = (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32))
    d, o = orc.render(none[0], none[1], V, F, K, H, W)
    # window x = fx X / Z + cx; window y = fy Y / Z + (H - cy); a pixel is covered when its centre (+0.5) is inside
    xw0, xw1 = 600 * -0.05 / 0.5004 + 320, 600 * 0.05 / 0.5004 + 320
    yw0, yw1 = 600 * -0.02 / 0.5004 + (480 - 230), 600 * 0.03 / 0.5004 + (480 - 230)
    cols = [w for w in range(W) if xw0 <= w + 0.5 <= xw1]
    rows = [h for h in range(H) if yw0 <= h + 0.5 <= yw1]
    exp = np.zeros((H, W), bool)
    exp[np.ix_(rows, cols)] = True
    assert np.array_equal(o == 2, exp) and exp.sum() > 5000
    assert np.all(d[exp] == np.float32(0.5)) and np.all(d[~exp] == np.float32(2.0))      # 500.4 mm -> 500; background 2.0
    V2, F2 = _quad(-0.05, 0.05, -0.02, 0.03, 0.5006)
    d2, _ = orc.render(none[0], none[1], V2, F2, K, H, W)
    assert np.all(d2[d2 < 1.9] == np.float32(0.501))


def test_hand_in_front_wins_and_clipping(orc):
    hand = _quad(-0.02, 0.02, -0.02, 0.02, 0.40)
    obj = _quad(-0.05, 0.05, -0.05, 0.05, 0.60)
    d, o = orc.render(hand[0], hand[1], obj[0], obj[1], K, H, W)
    assert (o == 1).sum() > 0 and (o == 2).sum() > (o == 1).sum()
    assert np.all(d[o == 1] == np.float32(0.4)) and np.all(d[o == 2] == np.float32(0.6))
    # same depth: GL_LESS keeps the first drawn (the hand)
    obj2 = _quad(-0.05, 0.05, -0.05, 0.05, 0.40)
    d, o2 = orc.render(hand[0], hand[1], obj2[0], obj2[1], K, H, W)
    assert np.array_equal(o2 == 1, o == 1)
    # beyond the far plane / before the near plane: nothing is drawn
    far = _quad(-0.05, 0.05, -0.05, 0.05, 2.5)
    near = _quad(-0.005, 0.005, -0.005, 0.005, 0.05)
    d, o = orc.render(far[0], far[1], near[0], near[1], K, H, W)
    assert not o.any() and np.all(d == np.float32(2.0))


def test_wrong_ratio_by_hand(orc):
    """real image = the render of the true pose: the true hypothesis scores the background term only; a shifted one pays
    2.0 per uncovered / newly covered object pixel"""
    V, F = _quad(-0.05, 0.05, -0.05, 0.05, 0.5)
    none = (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32))
    d, o = orc.render(none[0], none[1], V, F, K, H, W)
    raw = np.where(o == 2, 500, 0).astype(np.uint16)        # the sensor sees the square and nothing else
    I4 = np.eye(4, dtype=np.float32)
    sh = I4.copy()
    sh[0, 3] = 0.01                                         # 12 pixels to the right
    back = I4.copy()
    back[2, 3] = 0.004                                      # 4 mm farther: same pixels (nearly), |sim - real| = 0.004
    wr, keep = orc.reject_by_render(raw, 0.001, K, none[0], none[1], V, F, np.stack([I4, sh, back]), 2.0, 0.3)
    n_obj = int((o == 2).sum())
    n_bg = H * W - n_obj
    # true pose: roi term 0; every background pixel has real = 0 -> diff 2.0 -> bg_diff / bg_cnt = 2.0 exactly
    assert wr[0] == np.float32(2.0)
    # farther pose: roi pixels (a few less: the square shrinks) differ by 4 mm; uncovered true pixels have sim = 2.0 -> diff 2.0 in bg
    assert 2.0 + 2.0 * 0.0039 < wr[2] < 2.0 + 2.0 * 0.0041 + 1e-3
    # shifted pose: 12 of ~120 columns leave the roi (real invalid there -> 2.0 each) -> roi_diff / roi_cnt = 2.0 * 12 / 120
    assert abs(wr[1] - (2.0 + 2.0 * 2.0 * 12 / 120)) < 0.02
    assert keep.tolist() == [0, 2, 1]                       # n < 10: everything is kept, ascending wrong ratio


def test_triangles_crossing_the_near_plane_are_clipped_not_dropped(orc):
    """a ground-like quad in the plane Y = 0.05 m from Z = -1 (behind the camera) to Z = 1.5: OpenGL clips it at the near plane and
    draws the rest; row h shows Z = fy * 0.05 / (h + 0.5 - (H - cy))"""
    V = np.array([[-0.4, 0.05, -1.0], [0.4, 0.05, -1.0], [0.4, 0.05, 1.5], [-0.4, 0.05, 1.5]], np.float32)
    F = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    none = (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32))
    d, o = orc.render(none[0], none[1], V, F, K, H, W)
    rows = np.arange(H) + 0.5 - (H - 230.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        Z = np.where(rows > 0, 600 * np.float64(np.float32(0.05)) / rows, np.inf)
    seen = (Z >= 0.1) & (Z <= 1.5)
    assert seen.sum() > 150
    mid = W // 2
    assert np.array_equal(o[:, mid] == 2, seen)
    exp = np.round(Z[seen] * 1000) / 1000
    assert np.abs(d[seen, mid] - exp).max() < 1.01e-3      # (the read-back's float expression rounds to the millimetre)
    # the part of the image the quad covers widens towards the camera: at the bottom row it spans the whole width
    assert (o[H - 1] == 2).all() and not (o[0] == 2).any()
