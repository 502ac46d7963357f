"""Hand-derived known answers for the soft-silhouette rasteriser (reference smal_fitter/p3d_renderer.py:26-39,65-68:
pytorch3d 0.2.5 RasterizationSettings(blur_radius = log(1/1e-4 - 1) * 1e-4, faces_per_pixel = 100) + SoftSilhouetteShader
(sigma 1e-4) + transform_points_screen).

pytorch3d itself is not installable here, so the oracle's rasteriser cannot be pinned against
its output.  These cases pin it -- and the HIP kernels, under `-m gpu` -- against closed forms instead: every expected
value below is computed from the construction of the scene with two-line formulas (point-segment distance, 2-D
barycentrics, sigmoid), never by running rasteriser code.

Scenes are built in NDC and mapped back to world space through the camera of SURVEY App. A.2:
    x_ndc = -s x / (2.7 - z),  y_ndc = s y / (2.7 - z),  z_view = 2.7 - z,  s = 1 / tan(30 deg)
    pixel (row r, col c) has its centre at  x_p = 1 - (2c + 1)/S,  y_p = 1 - (2r + 1)/S.
Each case returns (verts (V,3) float64 world, faces (F,3) int, S, [(row, col, expected_sil), ...]).
"""
import math

import numpy as np

SIGMA = 1e-4
BLUR = math.log(1.0 / 1e-4 - 1.0) * 1e-4          # compared with the SQUARED distance
S_CAM = 1.0 / math.tan(math.radians(30.0))
CAM_DIST = 2.7
K = 100


def world_from_ndc(x_ndc, y_ndc, z_view):
    return np.array([-x_ndc * z_view / S_CAM, y_ndc * z_view / S_CAM, CAM_DIST - z_view], np.float64)


def pixel_centre(row, col, S):
    return 1.0 - (2.0 * col + 1.0) / S, 1.0 - (2.0 * row + 1.0) / S


def sigmoid(x):
    return 1.0 / (1.0 + math.exp(-x))


def seg_dist2(px, py, ax, ay, bx, by):
    ex, ey = bx - ax, by - ay
    t = min(1.0, max(0.0, ((px - ax) * ex + (py - ay) * ey) / (ex * ex + ey * ey)))
    qx, qy = ax + t * ex - px, ay + t * ey - py
    return qx * qx + qy * qy


def tri_dist2(p, tri):
    return min(seg_dist2(p[0], p[1], *tri[i], *tri[j]) for i, j in ((0, 1), (0, 2), (1, 2)))


def barycentric(p, tri):
    (ax, ay), (bx, by), (cx, cy) = tri
    den = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)
    w1 = ((p[0] - ax) * (cy - ay) - (p[1] - ay) * (cx - ax)) / den
    w2 = ((bx - ax) * (p[1] - ay) - (by - ay) * (p[0] - ax)) / den
    return 1.0 - w1 - w2, w1, w2


def single_face_sil(p, tri):
    """one face: sil = sigmoid(-d/sigma), d = -dist^2 inside, +dist^2 outside; nothing beyond the blur radius"""
    w = barycentric(p, tri)
    inside = all(x > 0 for x in w)
    d2 = tri_dist2(p, tri)
    if not inside and d2 >= BLUR:
        return 0.0
    return sigmoid((d2 if inside else -d2) / SIGMA)


def case_single_triangle(x_edge_px_offset=0.25, S=256):
    """One large triangle with a vertical edge x = x_e between pixel columns; pixels of the middle row at
    0.25 / 0.75 / 1.25 / ... pixels either side of it.  Far from the other two edges, sil = sigmoid(-+(x_p - x_e)^2 / sigma),
    and exactly 0 beyond sqrt(blur) = 3.885 pixels (at 256^2)."""
    px = 2.0 / S
    x_e = 1.0 - (2.0 * (S // 2 - 1) + 1.0) / S - x_edge_px_offset * px      # `offset` pixels towards -x of column S/2 - 1
    tri = [(x_e, -0.5), (x_e, 0.5), (-0.6, 0.0)]                            # interior: x < x_e
    verts = np.stack([world_from_ndc(x, y, 2.0) for x, y in tri])
    faces = np.array([[0, 1, 2]])
    row = S // 2
    checks = []
    for col in range(S // 2 - 8, S // 2 + 8):
        p = pixel_centre(row, col, S)
        inside = p[0] < x_e
        d2 = (p[0] - x_e) ** 2
        exp = sigmoid(d2 / SIGMA) if inside else (sigmoid(-d2 / SIGMA) if d2 < BLUR else 0.0)
        checks.append((row, col, exp))
    return verts, faces, S, checks, dict(tri=tri, x_e=x_e)


def case_blur_cutoff(S=256):
    """the blur radius is compared with the squared distance: a pixel 3.88 px from the edge is a candidate
    (p = sigmoid(-d^2/sigma) ~ 1.0e-4), one 3.89 px away is not (sqrt(blur) = 3.8846 px at 256^2)"""
    out = []
    for dist_px in (3.88, 3.89):
        col = S // 2 - 5
        px = 2.0 / S
        x_p = 1.0 - (2.0 * col + 1.0) / S
        x_e = x_p - dist_px * px                                              # pixel lies outside (x_p > x_e)
        tri = [(x_e, -0.5), (x_e, 0.5), (-0.6, 0.0)]
        verts = np.stack([world_from_ndc(x, y, 2.0) for x, y in tri])
        d2 = (dist_px * px) ** 2
        exp = sigmoid(-d2 / SIGMA) if d2 < BLUR else 0.0
        out.append((verts, np.array([[0, 1, 2]]), S, [(S // 2, col, exp)]))
    return out


def case_k_nearest(S=64):
    """150 faces stacked in depth over the same pixels.  Faces 0..49 are the FARTHEST (geometry A), faces 50..149 the 100
    NEAREST (geometry B): with faces_per_pixel = 100 a pixel sees geometry B only -- not the first 100 in index order, not all
    150.  sil = 1 - (1 - p_B)^100."""
    px = 2.0 / S
    row, col = S // 2, S // 2 - 2
    x_p, _ = pixel_centre(row, col, S)
    d_a, d_b = 0.8 * px, 0.9 * px
    verts, faces = [], []
    for i in range(150):
        far = i < 50
        x_e = x_p - (d_a if far else d_b)
        z_view = (3.0 if far else 2.0) + 1e-3 * i
        for x, y in ((x_e, -0.5), (x_e, 0.5), (-0.6, 0.0)):
            verts.append(world_from_ndc(x, y, z_view))
        faces.append([3 * i, 3 * i + 1, 3 * i + 2])
    p_a, p_b = sigmoid(-d_a ** 2 / SIGMA), sigmoid(-d_b ** 2 / SIGMA)
    exp = 1.0 - (1.0 - p_b) ** K
    wrong_first100 = 1.0 - (1.0 - p_a) ** 50 * (1.0 - p_b) ** 50
    wrong_all = 1.0 - (1.0 - p_a) ** 50 * (1.0 - p_b) ** 100
    return np.stack(verts), np.array(faces), S, [(row, col, exp)], dict(wrong_first100=wrong_first100, wrong_all=wrong_all)


def case_degenerate(S=64):
    """a sliver with |signed area| = 4e-9 <= kEpsilon (1e-8) is culled as a whole; the same sliver with area 4e-8 is
    rasterised like any face (the pixel half a pixel away from it gets sigmoid(-d^2/sigma))"""
    px = 2.0 / S
    row, col = S // 2, S // 2
    x_p, y_p = pixel_centre(row, col, S)
    out = []
    for area in (4e-9, 4e-8):
        base = 0.02
        h = area / base                                  # |(c - a) x (b - a)| = base * h
        y0 = y_p - 0.5 * px
        tri = [(x_p - base / 2, y0), (x_p + base / 2, y0), (x_p, y0 - h)]
        verts = np.stack([world_from_ndc(x, y, 2.0) for x, y in tri])
        exp = 0.0 if area <= 1e-8 else single_face_sil((x_p, y_p), tri)
        out.append((verts, np.array([[0, 1, 2]]), S, [(row, col, exp)]))
    return out


def case_behind_camera(S=64):
    """a face with one vertex behind the camera plane (z_view = -0.5): max z >= 0, so the face is kept; a pixel is a candidate
    only where the depth interpolated with the screen-space barycentrics is >= 0 (1 - 1.5 w2 >= 0 here)"""
    world = np.array([[-0.2, -0.2, 1.7], [0.2, -0.2, 1.7], [0.0, 0.3, 3.2]], np.float64)
    zv = CAM_DIST - world[:, 2]
    tri = [(-S_CAM * world[i, 0] / zv[i], S_CAM * world[i, 1] / zv[i]) for i in range(3)]
    checks = []
    for y_target in (-0.5, -0.9):
        row = int(round(((1.0 - y_target) * S - 1.0) / 2.0))
        col = S // 2
        p = pixel_centre(row, col, S)
        w = barycentric(p, tri)
        pz = sum(wi * zi for wi, zi in zip(w, zv))
        assert all(x > 0 for x in w), "test pixel must lie inside the 2-D triangle"
        exp = single_face_sil(p, tri) if pz >= 0 else 0.0
        checks.append((row, col, exp))
    assert checks[0][2] > 0.5 and checks[1][2] == 0.0
    return world, np.array([[0, 1, 2]]), S, checks


def keypoint_known_answers(S=256):
    """transform_points_screen + the reference's (row, col) reordering, SURVEY App. A.2:
    col = (S-1)/2 (1 + s x / (2.7 - z)),  row = (S-1)/2 (1 - s y / (2.7 - z))"""
    pts = np.array([[0.0, 0.0, 0.0], [0.5, 0.25, 0.2], [-0.3, 0.1, -0.4]], np.float64)
    half = (S - 1) / 2.0
    exp = np.stack([half * (1.0 - S_CAM * pts[:, 1] / (CAM_DIST - pts[:, 2])),
                    half * (1.0 + S_CAM * pts[:, 0] / (CAM_DIST - pts[:, 2]))], 1)
    assert abs(exp[0, 0] - 127.5) < 1e-12 and abs(exp[0, 1] - 127.5) < 1e-12
    return pts, exp


def edge_shift_gradient(S=256):
    """d sil / d x_e for the single-triangle case at the pixel 0.25 px outside the edge: sil = sigmoid(-(x_p - x_e)^2 / sigma)
    => d sil / d x_e = sil (1 - sil) * 2 (x_p - x_e) / sigma.  Moving the edge = moving its two end points a and b, so the
    sum of the two vertices' NDC-x gradients equals it; one world unit in x is -s / z_view NDC units."""
    verts, faces, S, checks, info = case_single_triangle(0.25, S)
    row, col = S // 2, S // 2 - 1
    x_p, _ = pixel_centre(row, col, S)
    x_e = info["x_e"]
    sil = sigmoid(-(x_p - x_e) ** 2 / SIGMA)
    dsil_dxe = sil * (1.0 - sil) * 2.0 * (x_p - x_e) / SIGMA
    dsil_dworldx_sum = dsil_dxe * (-S_CAM / 2.0)           # z_view = 2.0 for every vertex of the case
    return verts, faces, S, (row, col), dsil_dworldx_sum


def vertex_nearest_gradient(S=256):
    """SURVEY App. B, last row: a pixel whose nearest feature of the triangle is a VERTEX -- the one place where the exact adjoint of
    the point-segment distance and the one with the edge parameter t left unclamped (some pytorch3d 0.2.x sources,
    PointLineDistanceBackward) differ.  Triangle a, b, c with the pixel outside beyond a, 1 px above the line through a and b and
    1.5 px "before" a along it: t = (p - a).(b - a) / |b - a|^2 < 0 is clamped to 0, dist = |p - a|^2 (edges a-b and a-c tie at a;
    both restatements and the kernels take the first minimum, a-b, like pytorch3d's e01 <= e02 test).  With
    sil = sigmoid(-dist / sigma), g = d sil / d dist = -sil (1 - sil) / sigma:
      exact:      d sil / d a = g * 2 (a - p),                    d sil / d b = 0
      unclamped:  q = (a + t (b - a)) - p (perpendicular foot),   d sil / d a = g (1 - t) 2 q,   d sil / d b = g t 2 q
    Returns verts (world), faces, S, (row, col), {"exact": (V,2) d sil / d world xy, "unclamped": ...}."""
    px = 2.0 / S
    row, col = S // 2, S // 2
    x_p, y_p = pixel_centre(row, col, S)
    a = (x_p - 1.5 * px, y_p - 1.0 * px)                 # edge a -> b runs along -x (towards larger columns... NDC +x is left)
    b = (a[0] - 0.4, a[1])
    c = (a[0] - 0.2, a[1] - 0.3)
    tri = [a, b, c]
    verts = np.stack([world_from_ndc(x, y, 2.0) for x, y in tri])
    ex, ey = b[0] - a[0], b[1] - a[1]
    t = ((x_p - a[0]) * ex + (y_p - a[1]) * ey) / (ex * ex + ey * ey)
    assert t < 0.0
    dist = (x_p - a[0]) ** 2 + (y_p - a[1]) ** 2
    assert dist < BLUR and all(w <= 0 for w in barycentric((x_p, y_p), tri)[1:2])    # a candidate, outside
    assert abs(dist - tri_dist2((x_p, y_p), tri)) < 1e-18
    sil = sigmoid(-dist / SIGMA)
    g = -sil * (1.0 - sil) / SIGMA
    exact = np.zeros((3, 2))
    exact[0] = [g * 2.0 * (a[0] - x_p), g * 2.0 * (a[1] - y_p)]
    qx, qy = a[0] + t * ex - x_p, a[1] + t * ey - y_p
    unclamped = np.zeros((3, 2))
    unclamped[0] = [g * (1.0 - t) * 2.0 * qx, g * (1.0 - t) * 2.0 * qy]
    unclamped[1] = [g * t * 2.0 * qx, g * t * 2.0 * qy]
    # NDC -> world at z_view = 2: x_ndc = -s x / z, y_ndc = s y / z
    to_world = np.array([-S_CAM / 2.0, S_CAM / 2.0])
    assert np.abs(exact - unclamped).max() > 0.2 * np.abs(exact).max()     # the case discriminates
    return verts, np.array([[0, 1, 2]]), S, (row, col), {"exact": exact * to_world, "unclamped": unclamped * to_world, "sil": sil}
