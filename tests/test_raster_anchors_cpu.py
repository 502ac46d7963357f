"""The rasteriser oracle against hand-derived closed forms (tests/raster_anchors.py): both restatements of pytorch3d
0.2.5 -- oracle/smal_oracle.py (torch float64, pair list) and oracle/raster_naive.c (literal float32) -- must reproduce
them.  This is the pin the oracle's renderer has in the absence of pytorch3d: single-face
silhouette values at known pixel distances, the blur cut-off on the squared distance, the K = 100 nearest-in-depth
truncation, the degenerate-area cull, the per-pixel depth test of a face crossing the camera plane, and the keypoint
projection.  The same cases run on the HIP kernels in tests/test_gpu_anchors.py."""
import numpy as np
import pytest
import torch

from oracle import raster_naive, smal_oracle as so
from tests import raster_anchors as ra


def _oracle_sil(verts, faces, S, dtype=torch.float64):
    v = torch.from_numpy(np.asarray(verts, np.float64)).to(dtype)[None]
    return so.soft_silhouette(v, torch.from_numpy(np.asarray(faces, np.int64)), S)[0].double().numpy()


def _naive_sil(verts, faces, S):
    v = torch.from_numpy(np.asarray(verts, np.float64))
    xn, yn, zv = so.world_to_ndc(v)
    v_ndc = np.stack([xn.numpy(), yn.numpy(), zv.numpy()], 1).astype(np.float32)
    return raster_naive.forward(v_ndc, np.asarray(faces, np.int32), S, want_fragments=False)[0].astype(np.float64)


def _check(sil, checks, tol):
    for row, col, exp in checks:
        got = sil[row, col]
        assert abs(got - exp) < tol * (1.0 + abs(exp)), (row, col, got, exp)
        if exp == 0.0:
            assert got == 0.0, (row, col, got)


@pytest.mark.parametrize("offset", [0.25, 0.5])
def test_single_triangle_closed_form(offset):
    verts, faces, S, checks, _ = ra.case_single_triangle(offset)
    assert any(e == 0.0 for _, _, e in checks) and any(0.4 < e < 0.6 for _, _, e in checks)
    _check(_oracle_sil(verts, faces, S), checks, 1e-9)
    _check(_naive_sil(verts, faces, S), checks, 2e-4)          # float32 squared distances / sigma = 1e-4


def test_blur_is_compared_with_the_squared_distance():
    (inside_case, outside_case) = ra.case_blur_cutoff()
    for verts, faces, S, checks in (inside_case, outside_case):
        _check(_oracle_sil(verts, faces, S), checks, 1e-9)
        _check(_naive_sil(verts, faces, S), checks, 2e-6)
    assert 0.9e-4 < inside_case[3][0][2] < 1.1e-4 and outside_case[3][0][2] == 0.0


def test_only_the_100_nearest_in_depth_count():
    verts, faces, S, checks, wrong = ra.case_k_nearest()
    (row, col, exp), = checks
    assert abs(exp - wrong["wrong_first100"]) > 0.05 and abs(exp - wrong["wrong_all"]) > 0.05   # the case discriminates
    for sil, tol in ((_oracle_sil(verts, faces, S), 1e-9), (_naive_sil(verts, faces, S), 5e-4)):
        assert abs(sil[row, col] - exp) < tol, (sil[row, col], exp, wrong)


def test_degenerate_faces_are_culled():
    culled, kept = ra.case_degenerate()
    for verts, faces, S, checks in (culled, kept):
        _check(_oracle_sil(verts, faces, S), checks, 1e-7)
    assert _oracle_sil(*culled[:3]).max() == 0.0
    assert kept[3][0][2] > 0.05
    # float32 restatement: the area of the culled sliver (4e-9) is well below kEpsilon in float32 too
    assert _naive_sil(*culled[:3]).max() == 0.0
    _check(_naive_sil(*kept[:3]), kept[3], 2e-3)


def test_face_crossing_the_camera_plane():
    verts, faces, S, checks = ra.case_behind_camera()
    _check(_oracle_sil(verts, faces, S), checks, 1e-9)
    _check(_naive_sil(verts, faces, S), checks, 1e-4)


def test_keypoint_projection_known_answers():
    pts, exp = ra.keypoint_known_answers()
    got = so.project_points(torch.from_numpy(pts)[None], 256)[0].numpy()
    assert np.abs(got - exp).max() < 1e-10


def test_edge_shift_gradient_closed_form():
    verts, faces, S, (row, col), dsum = ra.edge_shift_gradient()
    v = torch.from_numpy(verts)[None].requires_grad_(True)
    sil = so.soft_silhouette(v, torch.from_numpy(faces.astype(np.int64)), S)
    sil[0, row, col].backward()
    got = float(v.grad[0, 0, 0] + v.grad[0, 1, 0])             # the edge's two end points, world x
    assert abs(got - dsum) < 1e-6 * abs(dsum), (got, dsum)
    assert abs(float(v.grad[0, 2, 0])) < 1e-12                 # the far vertex does not move that edge


@pytest.mark.parametrize("unclamped", [False, True])
def test_vertex_nearest_gradient_both_adjoint_conventions(unclamped):
    """SURVEY App. B's switch: where the nearest feature is a vertex, the exact adjoint and the unclamped-t one differ by known
    closed forms; both oracle restatements reproduce the one their flag selects (the forward is the same either way)"""
    verts, faces, S, (row, col), exp = ra.vertex_nearest_gradient()
    want = exp["unclamped" if unclamped else "exact"]
    so.EDGE_T_UNCLAMPED = unclamped
    try:
        v = torch.from_numpy(verts)[None].requires_grad_(True)
        sil = so.soft_silhouette(v, torch.from_numpy(faces.astype(np.int64)), S)
        assert abs(float(sil[0, row, col].detach()) - exp["sil"]) < 1e-9
        sil[0, row, col].backward()
    finally:
        so.EDGE_T_UNCLAMPED = False
    got = v.grad[0, :, :2].numpy()
    assert np.abs(got - want).max() < 1e-7 * np.abs(want).max(), (got, want)
    # the literal float32 restatement (gradient with respect to the NDC positions; z_view = 2 for every vertex)
    xn, yn, zv = so.world_to_ndc(torch.from_numpy(verts))
    v_ndc = np.stack([xn.numpy(), yn.numpy(), zv.numpy()], 1).astype(np.float32)
    sil_n, p2f, _, dists, _ = raster_naive.forward(v_ndc, faces.astype(np.int32), S)
    gs = np.zeros((S, S), np.float32)
    gs[row, col] = 1.0
    gv = raster_naive.backward(v_ndc, faces.astype(np.int32), S, p2f, dists, gs, unclamped_t=unclamped)
    to_world = np.array([-ra.S_CAM / 2.0, ra.S_CAM / 2.0])
    assert np.abs(gv * to_world - want).max() < 2e-3 * np.abs(want).max(), (gv * to_world, want)
