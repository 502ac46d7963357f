#!/usr/bin/env python3
"""Generate the committed synthetic SMAL-topology mesh `smalify_amd/data/synth_mesh.npz`.

The real SMAL model files are not redistributable / not present, so tests and the
benchmark run on a procedural stand-in with *exactly* the SMAL dimensions: V = 3889 vertices,
F = 7774 faces, closed genus-0, mirror-symmetric about y = 0 with 135 vertices on the symmetry
plane at the vertex ids the reference's template alignment hard-codes
(reference smal_model/smal_basics.py:9, consumed at :11-29).

Construction (nothing is read from the reference's template OBJ):
  * half surface = Delaunay triangulation of a unit disk: 135 points on the rim (the symmetry
    plane) + 1877 interior points on a sunflower spiral  ->  2*1877 + 135 - 2 = 3887 triangles
  * Lambert equal-area lift of the disk to a hemisphere (y >= 0), mirrored to y <= 0
  * radial "quadruped blob" deformation (body ellipsoid + 4 leg lobes + head + tail lobes)
  * vertex ids: rim vertices take the 135 centre ids; mirror pairs take consecutive free ids
    (left = even slot, right = odd slot) so that "k-th left vertex <-> k-th right vertex in index
    order", which is what reference smal_basics.py:27 relies on.

The output is committed so that golden fixtures do not depend on qhull reproducing the same
triangulation on another machine.
"""
import os
import sys

import numpy as np
from scipy.spatial import Delaunay

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from smalify_amd.smal_topology import CENTER_VERTEX_IDS, NUM_VERTS, NUM_FACES  # noqa: E402


def blob_radius(d):
    """Radial scale of the quadruped blob for unit directions d (n,3); x = nose, z = up."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    # base ellipsoid: semi-axes (length, half-width, half-height)
    a, b, c = 0.52, 0.15, 0.19
    r = 1.0 / np.sqrt((x / a) ** 2 + (y / b) ** 2 + (z / c) ** 2)

    def lobe(cx, cy, cz, amp, sharp):
        cvec = np.array([cx, cy, cz], dtype=np.float64)
        cvec /= np.linalg.norm(cvec)
        cosang = d @ cvec
        return amp * np.exp(sharp * (cosang - 1.0))

    bump = np.zeros_like(r)
    for sx in (+0.55, -0.55):          # front / back legs
        for sy in (+0.28, -0.28):      # left / right
            bump += lobe(sx, sy, -0.78, 1.55, 38.0)
    bump += lobe(1.0, 0.0, 0.28, 0.55, 16.0)    # head / neck
    bump += lobe(-1.0, 0.0, 0.22, 0.50, 60.0)   # tail
    for sy in (+0.45, -0.45):                   # ears
        bump += lobe(0.80, sy, 0.62, 0.35, 160.0)
    return r * (1.0 + bump)


def main():
    n_rim = len(CENTER_VERTEX_IDS)            # 135
    n_int = (NUM_VERTS - n_rim) // 2          # 1877
    assert n_rim + 2 * n_int == NUM_VERTS

    # --- disk points -----------------------------------------------------------------------
    ang = 2.0 * np.pi * (np.arange(n_rim) + 0.5) / n_rim
    rim = np.stack([np.cos(ang), np.sin(ang)], 1)
    k = np.arange(n_int) + 0.5
    golden = np.pi * (3.0 - np.sqrt(5.0))
    rad = np.sqrt(k / (n_int + 0.5 * n_rim))          # keeps the outer ring inside the rim
    interior = np.stack([rad * np.cos(golden * k), rad * np.sin(golden * k)], 1)
    pts = np.concatenate([rim, interior], 0)          # (2012, 2)

    tri = Delaunay(pts).simplices.astype(np.int64)
    assert tri.shape[0] == 2 * n_int + n_rim - 2, tri.shape
    # consistent CCW orientation in the disk
    e1 = pts[tri[:, 1]] - pts[tri[:, 0]]
    e2 = pts[tri[:, 2]] - pts[tri[:, 0]]
    flip = (e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]) < 0
    tri[flip] = tri[flip][:, [0, 2, 1]]

    # --- Lambert equal-area lift: disk radius rho in [0,1] -> polar angle from +y -----------
    rho = np.linalg.norm(pts, axis=1)
    rho[:n_rim] = 1.0
    phi = np.arctan2(pts[:, 1], pts[:, 0])
    cos_t = 1.0 - rho ** 2                      # y component (1 at the pole, 0 on the rim)
    sin_t = np.sqrt(np.clip(1.0 - cos_t ** 2, 0.0, None))
    half = np.stack([sin_t * np.cos(phi), cos_t, sin_t * np.sin(phi)], 1)   # y >= 0
    half[:n_rim, 1] = 0.0

    # --- vertex ids ------------------------------------------------------------------------
    centre = np.asarray(CENTER_VERTEX_IDS, dtype=np.int64)
    free = np.setdiff1d(np.arange(NUM_VERTS), centre)           # ascending
    left_ids, right_ids = free[0::2], free[1::2]
    assert len(left_ids) == n_int and len(right_ids) == n_int

    dirs = np.zeros((NUM_VERTS, 3))
    dirs[centre] = half[:n_rim]
    # reference convention: left = y < 0, right = y > 0 (smal_basics.py:24-25)
    dirs[right_ids] = half[n_rim:]
    dirs[left_ids] = half[n_rim:] * np.array([1.0, -1.0, 1.0])

    id_pos = np.concatenate([centre, right_ids])      # half-mesh index -> global id (y >= 0 side)
    id_neg = np.concatenate([centre, left_ids])
    f_pos = id_pos[tri]
    f_neg = id_neg[tri][:, [0, 2, 1]]                 # mirrored side: flip winding
    faces = np.concatenate([f_pos, f_neg], 0)
    assert faces.shape == (NUM_FACES, 3)

    verts = dirs * blob_radius(dirs)[:, None]

    # outward orientation check (signed volume > 0), else flip everything
    v0, v1, v2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    vol = np.einsum("ij,ij->i", v0, np.cross(v1, v2)).sum() / 6.0
    if vol < 0:
        faces = faces[:, [0, 2, 1]]

    sym = np.arange(NUM_VERTS)
    sym[left_ids] = right_ids
    sym[right_ids] = left_ids

    # sanity: closed manifold, Euler characteristic 2
    edges = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0), 1)
    uniq, cnt = np.unique(edges, axis=0, return_counts=True)
    assert (cnt == 2).all()
    assert NUM_VERTS - len(uniq) + NUM_FACES == 2

    out = os.path.join(os.path.dirname(__file__), "..", "smalify_amd", "data", "synth_mesh.npz")
    np.savez_compressed(out, verts=verts.astype(np.float32), faces=faces.astype(np.int16),
                        sym_idx=sym.astype(np.int16))
    print("wrote", os.path.abspath(out), "V", verts.shape, "F", faces.shape,
          "bbox", verts.min(0), verts.max(0))


if __name__ == "__main__":
    main()
