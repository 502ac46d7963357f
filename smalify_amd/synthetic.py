"""Synthetic SMAL-shaped model and synthetic fitting problems.

The real SMAL pickles and the BADJA / StanfordExtra datasets are un-vendored submodules of the
reference and there is no network, so tests and bench.py use a procedural stand-in
with the exact SMAL dimensions and pickle layout (`synthetic_smal_dicts`), pushed through the same
one-time preparation as a real model (`model_io.prepare_model`).

`synthetic_smal_dicts` returns dictionaries with *the keys and dtypes the reference's loader reads*
(reference smal_model/smal_torch.py:36-96, smal_fitter/smal_fitter.py:40-43), so the very same
objects can be pickled and fed to the imported reference when golden vectors are generated
(tests/golden/make_golden.py).
"""
from __future__ import annotations

import os

import numpy as np

from . import smal_topology as topo
from . import model_io

_MESH_PATH = os.path.join(os.path.dirname(__file__), "data", "synth_mesh.npz")

# rest-pose joint centres of the stand-in quadruped (x = nose, y = left(-)/right(+), z = up)
_JOINT_POS = np.array(
    [[-0.30, 0, 0.00], [-0.25, 0, 0.00], [-0.15, 0, 0.01], [-0.05, 0, 0.01], [0.05, 0, 0.01],
     [0.15, 0, 0.01], [0.25, 0, 0.01],
     [0.28, -0.08, -0.08], [0.29, -0.09, -0.18], [0.29, -0.09, -0.28], [0.30, -0.09, -0.38],
     [0.28, 0.08, -0.08], [0.29, 0.09, -0.18], [0.29, 0.09, -0.28], [0.30, 0.09, -0.38],
     [0.38, 0, 0.05], [0.52, 0, 0.09],
     [-0.28, -0.08, -0.08], [-0.29, -0.09, -0.18], [-0.29, -0.09, -0.28], [-0.30, -0.09, -0.38],
     [-0.28, 0.08, -0.08], [-0.29, 0.09, -0.18], [-0.29, 0.09, -0.28], [-0.30, 0.09, -0.38],
     [-0.42, 0, 0.04], [-0.46, 0, 0.05], [-0.50, 0, 0.06], [-0.54, 0, 0.07], [-0.58, 0, 0.08],
     [-0.62, 0, 0.085], [-0.66, 0, 0.09],
     [0.62, 0, 0.05], [0.50, -0.07, 0.16], [0.50, 0.07, 0.16]], dtype=np.float64)
assert _JOINT_POS.shape == (topo.NUM_JOINTS, 3)


def load_synth_mesh():
    m = np.load(_MESH_PATH)
    return m["verts"].astype(np.float64), m["faces"].astype(np.int64), m["sym_idx"].astype(np.int64)


def synthetic_smal_dicts(seed=0, dense_weights=False, symmetric_basis=False):
    """-> (dd, data, sym_idx) in the reference's pickle layout.

    dd   : f (F,3) uint32, v_template (V,3) f64, shapedirs (V,3,41) f64, posedirs (V,3,306) f64,
           J_regressor scipy csc (35,V), weights (V,35) f64, kintree_table (2,35) uint32
    data : cluster_means (5,41), cluster_cov list of 5 (41,41) SPD
    dense_weights: every vertex gets a non-zero weight for every joint (stress variant); the default
           follows rigged-mesh practice: <= 4 influences per vertex, sparse joint regressor.
    symmetric_basis: the shape basis is made mirror-symmetric about the y = 0 plane (like a real SMAL model's): every family mean
           then leaves the template left/right balanced, so ALL five shape families load (with the default basis families 2
           and 3 stop where the reference stops, smal_basics.py:32-35).  The default stays as it was: the golden fixtures
           belong to it.
    """
    import scipy.sparse as sp

    rs = np.random.RandomState(seed)
    verts, faces, sym_idx = load_synth_mesh()
    nv = verts.shape[0]

    # --- smooth shape basis: quadratic fields of position, decaying amplitude --------------------
    x, y, z = verts[:, 0], verts[:, 1], verts[:, 2]
    basis = np.stack([np.ones(nv), x, y, z, x * x, y * y, z * z, x * y, x * z, y * z], 1)   # (V,10)
    shapedirs = np.zeros((nv, 3, 41))
    for b in range(41):
        coef = rs.randn(10, 3) * np.array([0.3, 1, 1, 1, 2, 2, 2, 2, 2, 2])[:, None]
        shapedirs[:, :, b] = (basis @ coef) * (0.035 / (1.0 + 0.15 * b))

    if symmetric_basis:
        mirror = np.array([1.0, -1.0, 1.0])[None, :, None]
        shapedirs = 0.5 * (shapedirs + mirror * shapedirs[sym_idx])      # centre vertices (sym_idx[v] = v): no y displacement

    # --- pose-corrective basis: dense, small ------------------------------------------------------
    posedirs = rs.randn(nv, 3, topo.NUM_POSE_FEATURES) * 0.004

    # --- skinning weights / joint regressor from distances to the joint centres ------------------
    d2 = ((verts[:, None, :] - _JOINT_POS[None, :, :]) ** 2).sum(-1)          # (V,35)
    if dense_weights:
        w = np.exp(-d2 / (2 * 0.08 ** 2)) + 1e-3
    else:
        w = np.zeros_like(d2)
        near = np.argsort(d2, axis=1)[:, :4]
        rows = np.arange(nv)[:, None]
        w[rows, near] = np.exp(-d2[rows, near] / (2 * 0.05 ** 2)) + 1e-6
    weights = w / w.sum(1, keepdims=True)

    jr = np.zeros((topo.NUM_JOINTS, nv))
    for j in range(topo.NUM_JOINTS):
        near = np.argsort(d2[:, j])[:24]
        ww = np.exp(-d2[near, j] / (2 * 0.06 ** 2)) + 1e-3
        jr[j, near] = ww / ww.sum()
    J_regressor = sp.csc_matrix(jr)

    kintree = np.zeros((2, topo.NUM_JOINTS), dtype=np.uint32)
    kintree[0] = topo.SYNTH_PARENTS.astype(np.int64).astype(np.uint32)       # root -> 4294967295
    kintree[1] = np.arange(topo.NUM_JOINTS)

    dd = dict(f=faces.astype(np.uint32), v_template=verts.copy(), shapedirs=shapedirs,
              posedirs=posedirs, J_regressor=J_regressor, weights=weights, kintree_table=kintree)

    means = rs.randn(5, 41) * 0.25
    covs = []
    for _ in range(5):
        a = rs.randn(41, 41) * 0.2
        covs.append(a @ a.T + 0.05 * np.eye(41))
    data = dict(cluster_means=means, cluster_cov=covs)
    return dd, data, sym_idx


def synthetic_model(seed=0, shape_family_id=1, dense_weights=False):
    """SMALModelData of the stand-in model, prepared exactly like a real one."""
    dd, data, sym = synthetic_smal_dicts(seed, dense_weights)
    return model_io.prepare_model(dd, data, sym, shape_family_id)


def synthetic_pose_prior(seed=7):
    """Stand-in for the walking pose prior when the reference's data/priors is not reachable
    (GPU box): lower-triangular precision factor (105,105), mean (105,), mask (first 3 = 0)."""
    rs = np.random.RandomState(seed)
    a = rs.randn(105, 105) * 0.3
    cov = a @ a.T / 105.0 + 0.02 * np.eye(105)
    prec = np.linalg.cholesky(np.linalg.inv(cov))
    mean = rs.randn(105) * 0.05
    mean[:3] = 0.0
    mask = np.ones(105, dtype=np.float32)
    mask[:3] = 0.0
    return prec.astype(np.float32), mean.astype(np.float32), mask


def synthetic_shape_prior(seed=11, dim=26):
    rs = np.random.RandomState(seed)
    a = rs.randn(dim, 13) * 0.4                 # rank-deficient like unity_betas (13 samples)
    cov = a @ a.T / 13.0
    mean = (rs.randn(dim) * 0.2).astype(np.float32)
    return model_io.shape_prior_from_cov(cov, mean)


def ground_truth_params(num_frames, seed=1234, mean_betas=None, mean_logscale=None):
    """Smooth random ground-truth fit parameters for a synthetic sequence.

    joint_rotations ~ 0.15 N(0,1), global_rotation = init + 0.1 N(0,1),
    trans = (0.05,-0.03,0.1) + 0.02 N(0,1); 5-tap box filter over frames.
    """
    rs = np.random.RandomState(seed)
    init = model_io.initial_global_rotation()

    def smooth(a):
        if a.shape[0] < 2:
            return a
        pad = np.concatenate([a[:1]] * 2 + [a] + [a[-1:]] * 2, 0)
        return sum(pad[i:i + a.shape[0]] for i in range(5)) / 5.0

    jr = smooth(0.15 * rs.randn(num_frames, topo.NUM_POSE, 3))
    gr = smooth(init[None, :] + 0.1 * rs.randn(num_frames, 3))
    tr = smooth(np.array([0.05, -0.03, 0.1])[None, :] + 0.02 * rs.randn(num_frames, 3))
    betas = np.zeros(topo.NUM_BETAS) if mean_betas is None else np.asarray(mean_betas, dtype=np.float64)
    ls = np.zeros(topo.NUM_LOGSCALES) if mean_logscale is None else np.asarray(mean_logscale, dtype=np.float64)
    return dict(global_rotation=gr.astype(np.float32), joint_rotations=jr.astype(np.float32),
                trans=tr.astype(np.float32), betas=betas.astype(np.float32),
                log_beta_scales=ls.astype(np.float32))


def keypoint_noise_and_visibility(num_frames, seed=4321, sigma_px=1.0, p_visible=0.85):
    rs = np.random.RandomState(seed)
    noise = (sigma_px * rs.randn(num_frames, topo.NUM_KEYPOINTS, 2)).astype(np.float32)
    vis = (rs.rand(num_frames, topo.NUM_KEYPOINTS) < p_visible).astype(np.float32)
    return noise, vis
