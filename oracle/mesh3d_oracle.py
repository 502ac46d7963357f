"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch float64 + autograd) of the objective of the reference's 3D mesh
fitter, fitter_3d/trainer.py:205-227 (Stage.forward): chamfer distance to points sampled from the target meshes, plus
edge-length, normal-consistency and uniform-Laplacian regularisers of the deformed SMAL mesh, and the torch.optim.Adam
step of trainer.py:194 (default betas).  Only tests/ may import this file.

PARITY UNPINNED: the four loss terms and the sampler live in PyTorch3D (pinned by the reference's README to v0.2.5,
pytorch3d/loss/{chamfer,mesh_edge_loss,mesh_normal_consistency,mesh_laplacian_smoothing}.py and
pytorch3d/ops/sample_points_from_meshes.py), which is absent from /root/reference and from this image.  What follows
restates the published v0.2.5 algorithms; tests/test_mesh3d_cpu.py checks them against hand-derived closed forms
on small meshes.  PINNED against the imported reference (tests/golden/make_golden_fit3d.py ->
tests/golden/reference_golden_fit3d.npz, checked in tests/test_mesh3d_cpu.py): fitter_verts (SMAL3DFitter.forward,
trainer.py:94-108: the LBS of smal_oracle.py + trans + deform_verts) and Adam over the parameter groups of a Stage
(SMALParamGroup + torch.optim.Adam, trainer.py:111-153,194) incl. the frozen log_beta_scales.

v0.2.5 definitions restated here (N meshes in the batch, all reductions at their defaults):
  chamfer_distance(x, y)        cham_x[n] = mean_i min_j |x_ni - y_nj|^2, cham_y likewise; (sum_n cham_x + cham_y) / N
  mesh_edge_loss(m, 0)          per mesh: mean over unique edges of |v0 - v1|^2; mean over meshes
  mesh_normal_consistency(m)    per mesh: mean over pairs of faces sharing an edge (v0,v1), third vertices a and b, of
                                1 - cos(n0, n1), n0 = (v1-v0) x (a-v0), n1 = -(v1-v0) x (b-v0); mean over meshes.
                                cos = <n0,n1> * rsqrt(max(|n0|^2 |n1|^2, eps^2)), eps = 1e-8 (torch 1.6 cosine_similarity)
  mesh_laplacian_smoothing(m, "uniform")   per mesh: mean over vertices of | mean_{u in N(i)} v_u - v_i |; mean over meshes
  sample_points_from_meshes(m, S)  face ~ multinomial(area), (u, v) ~ U(0,1)^2, p = (1-sqrt u) a + sqrt u (1-v) b + sqrt u v c
"""
import numpy as np
import torch

DEFAULT_WEIGHTS = dict(w_chamfer=1.0, w_edge=1.0, w_normal=0.01, w_laplacian=0.1)   # trainer.py:31
COS_EPS = 1e-8


# ---------------------------------------------------------------------------------------------------------------
# topology of one triangle mesh (what Meshes.edges_packed / laplacian_packed / the face-pair tables hold)
# ---------------------------------------------------------------------------------------------------------------
def unique_edges(faces):
    """sorted unique (lo, hi) vertex pairs, (E,2) int64"""
    f = np.asarray(faces, dtype=np.int64)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
    e.sort(axis=1)
    return np.unique(e, axis=0)


def face_pairs(faces):
    """every pair of faces sharing an edge -> (P,4) int64 rows (v0, v1, a, b): the shared edge (v0 < v1) and the two
    opposite vertices; edges with k faces contribute k (k-1) / 2 rows (mesh_normal_consistency.py v0.2.5)"""
    f = np.asarray(faces, dtype=np.int64)
    by_edge = {}
    for fi, (i, j, k) in enumerate(f):
        for (p, q, r) in ((i, j, k), (j, k, i), (k, i, j)):
            key = (min(p, q), max(p, q))
            by_edge.setdefault(key, []).append(r)
    rows = []
    for (v0, v1) in sorted(by_edge):
        opp = by_edge[(v0, v1)]
        for x in range(len(opp)):
            for y in range(x + 1, len(opp)):
                rows.append((v0, v1, opp[x], opp[y]))
    return np.asarray(rows, dtype=np.int64).reshape(-1, 4)


# ---------------------------------------------------------------------------------------------------------------
# loss terms (verts: (N,V,3) torch tensor, any float dtype)
# ---------------------------------------------------------------------------------------------------------------
def chamfer(points, verts):
    """pytorch3d.loss.chamfer_distance(points, verts)[0], v0.2.5 defaults"""
    total = verts.sum() * 0.0
    for n in range(verts.shape[0]):                                               # one mesh at a time: (S,V,3) temporaries
        d2 = ((points[n, :, None, :] - verts[n, None, :, :]) ** 2).sum(-1)        # (S,V)
        total = total + d2.min(dim=1).values.mean() + d2.min(dim=0).values.mean()
    return total / verts.shape[0]


def edge_loss(verts, edges):
    e = torch.as_tensor(edges)
    d = verts[:, e[:, 0]] - verts[:, e[:, 1]]
    return ((d.norm(dim=2)) ** 2.0).mean(dim=1).sum() / verts.shape[0]


def normal_consistency(verts, pairs):
    p = torch.as_tensor(pairs)
    if p.numel() == 0:
        return verts.sum() * 0.0
    v0, v1, a, b = (verts[:, p[:, i]] for i in range(4))
    n0 = torch.cross(v1 - v0, a - v0, dim=2)
    n1 = -torch.cross(v1 - v0, b - v0, dim=2)
    w12 = (n0 * n1).sum(-1)
    w1 = (n0 * n0).sum(-1)
    w2 = (n1 * n1).sum(-1)
    cos = w12 * torch.rsqrt(torch.clamp(w1 * w2, min=COS_EPS * COS_EPS))
    return (1.0 - cos).mean(dim=1).sum() / verts.shape[0]


def laplacian_uniform(verts, edges):
    e = torch.as_tensor(edges)
    N, V = verts.shape[0], verts.shape[1]
    deg = torch.zeros(V, dtype=verts.dtype)
    deg.index_add_(0, e[:, 0], torch.ones(e.shape[0], dtype=verts.dtype))
    deg.index_add_(0, e[:, 1], torch.ones(e.shape[0], dtype=verts.dtype))
    acc = torch.zeros_like(verts)
    acc = acc.index_add(1, e[:, 0], verts[:, e[:, 1]])
    acc = acc.index_add(1, e[:, 1], verts[:, e[:, 0]])
    inv = torch.where(deg > 0, 1.0 / deg.clamp(min=1.0), torch.zeros_like(deg))
    lv = acc * inv[None, :, None] - verts * (deg > 0).to(verts.dtype)[None, :, None]
    # torch's norm backward is 0 at the origin; sqrt(x + 0) would give nan there, so mask explicitly
    sq = (lv * lv).sum(-1)
    nrm = torch.where(sq > 0, torch.sqrt(torch.where(sq > 0, sq, torch.ones_like(sq))), torch.zeros_like(sq))
    return nrm.mean(dim=1).sum() / N


def objective(verts, points, edges, pairs, weights=None):
    """Stage.forward (trainer.py:205-227) -> (total, dict of the unweighted terms); a term is skipped when its weight
    is <= 0 (trainer.py:203)"""
    w = dict(DEFAULT_WEIGHTS)
    if weights:
        w.update(weights)
    terms = {}
    total = verts.sum() * 0.0
    if w["w_chamfer"] > 0:
        terms["chamfer"] = chamfer(points, verts)
        total = total + w["w_chamfer"] * terms["chamfer"]
    if w["w_edge"] > 0:
        terms["edge"] = edge_loss(verts, edges)
        total = total + w["w_edge"] * terms["edge"]
    if w["w_normal"] > 0:
        terms["normal"] = normal_consistency(verts, pairs)
        total = total + w["w_normal"] * terms["normal"]
    if w["w_laplacian"] > 0:
        terms["laplacian"] = laplacian_uniform(verts, edges)
        total = total + w["w_laplacian"] * terms["laplacian"]
    return total, terms


# ---------------------------------------------------------------------------------------------------------------
# sampler (distribution only: the reference draws from torch's global generator, no stream can be matched)
# ---------------------------------------------------------------------------------------------------------------
def face_areas(verts, faces):
    v = np.asarray(verts, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int64)
    return 0.5 * np.linalg.norm(np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]), axis=1)


def barycentric_point(verts, faces, face_idx, u, v):
    """the map of _rand_barycentric_coords (sample_points_from_meshes.py v0.2.5)"""
    vv = np.asarray(verts, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int64)[face_idx]
    su = np.sqrt(u)
    w0, w1, w2 = 1.0 - su, su * (1.0 - v), su * v
    return w0[:, None] * vv[f[:, 0]] + w1[:, None] * vv[f[:, 1]] + w2[:, None] * vv[f[:, 2]]


# ---------------------------------------------------------------------------------------------------------------
# SMAL3DFitter.forward + Stage.step on top of the LBS oracle
# ---------------------------------------------------------------------------------------------------------------
def fitter_verts(oracle_model, params):
    """SMAL3DFitter.forward (trainer.py:94-108): LBS verts + trans + deform_verts"""
    from oracle import smal_oracle as so
    theta = torch.cat([params["global_rot"][:, None, :], params["joint_rot"]], dim=1)
    verts = so.smal_forward(oracle_model, params["betas"], theta, params["log_beta_scales"])[0]
    verts = verts + params["trans"][:, None, :]
    if params.get("deform_verts") is not None:
        verts = verts + params["deform_verts"]
    return verts


class Adam:
    """torch.optim.Adam with per-parameter learning rates (trainer.py:121-153,194): betas (0.9, 0.999), eps 1e-8"""

    def __init__(self, lrs, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lrs, self.b1, self.b2, self.eps = dict(lrs), beta1, beta2, eps
        self.state = {}

    def step(self, params, grads):
        for k, lr in self.lrs.items():
            g = grads.get(k)
            if g is None:
                continue
            st = self.state.setdefault(k, dict(t=0, m=torch.zeros_like(g), v=torch.zeros_like(g)))
            st["t"] += 1
            st["m"] = self.b1 * st["m"] + (1 - self.b1) * g
            st["v"] = self.b2 * st["v"] + (1 - self.b2) * g * g
            bc1 = 1 - self.b1 ** st["t"]
            bc2 = 1 - self.b2 ** st["t"]
            denom = st["v"].sqrt() / np.sqrt(bc2) + self.eps
            params[k] = params[k] - (lr / bc1) * st["m"] / denom
