"""Problems and parity cases of the 3D mesh-fitting objective, shared by tests/test_mesh3d_cpu.py (host shim vs oracle),
tests/test_gpu_fit3d.py (HIP through the C-ABI vs oracle) and __graft_entry__/tools.  Nothing here asserts.
Oracle = oracle/mesh3d_oracle.py (test infrastructure)."""
from __future__ import annotations

import numpy as np
import torch

from oracle import mesh3d_oracle as mo
from oracle import smal_oracle as so
from smalify_amd import synthetic

WEIGHT_KEYS = ("w_chamfer", "w_edge", "w_normal", "w_laplacian")


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def objective_problem(N, S, seed, deform=True):
    """N noisy copies of the synthetic SMAL template against S target points from a perturbed, rescaled surface"""
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    rs = np.random.RandomState(seed)
    V = md.num_verts
    base = np.asarray(md.v_template, np.float32)
    lbs = (base[None] + 0.01 * rs.randn(N, V, 3)).astype(np.float32)
    trans = (0.05 * rs.randn(N, 3)).astype(np.float32)
    dfm = (0.003 * rs.randn(N, V, 3)).astype(np.float32) if deform else None
    idx = rs.randint(0, V, size=(N, S))
    pts = (1.1 * base[idx] + 0.02 * rs.randn(N, S, 3) + 0.03).astype(np.float32)
    return md, lbs, trans, dfm, pts


def oracle_objective(verts32, pts32, faces, weights):
    """oracle value and gradient at the float32 vertices -> (total, {term: value}, d total / d verts)"""
    edges, pairs = mo.unique_edges(faces), mo.face_pairs(faces)
    tv = torch.from_numpy(np.asarray(verts32)).double().requires_grad_(True)
    total, terms = mo.objective(tv, torch.from_numpy(np.asarray(pts32)).double(), edges, pairs, dict(zip(WEIGHT_KEYS, weights)))
    total.backward()
    return float(total.detach()), {k: float(v.detach()) for k, v in terms.items()}, tv.grad.numpy()


def target_meshes_from_smal(md, N, seed):
    """N posed + reshaped SMAL meshes (oracle LBS), centred and scaled like fitter_3d/utils.py:237-241 -> verts list, faces"""
    rs = np.random.RandomState(seed)
    om = so.OracleModel(md)
    beta = torch.from_numpy(0.5 * rs.randn(N, 20))
    theta = torch.from_numpy(0.12 * rs.randn(N, 35, 3))
    ls = torch.from_numpy(0.05 * rs.randn(N, 6))
    verts = so.smal_forward(om, beta, theta, ls)[0].numpy()
    out = []
    for v in verts:
        v = v - v.mean(0)
        out.append((v / np.abs(v).max()).astype(np.float32))
    return out, np.asarray(md.faces, np.int64)


def synthetic_smal_data(seed=0):
    return synthetic.synthetic_smal_dicts(seed=seed)[1]
