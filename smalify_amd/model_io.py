"""Host-side, one-time preparation of the SMAL model constants (numpy, float64 -> float32).

Mirrors what the reference does once at construction time:
  * SMAL.__init__                          reference smal_model/smal_torch.py:24-96
  * align_smal_template_to_symmetry_axis   reference smal_model/smal_basics.py:7-37
  * shape-family statistics                reference smal_fitter/smal_fitter.py:40-72
The result is a plain `SMALModelData` of contiguous float32 / int32 arrays that the C-ABI
(`smalfit_model_create`, include/smalfit.h) uploads to HBM.  Nothing here runs per iteration.

The reference unpickles py2 pickles that embed chumpy objects; chumpy is not a dependency here —
`_LenientUnpickler` maps every chumpy class onto a tiny stand-in that keeps the `x` ndarray
(SURVEY.md Appendix C.2).
"""
from __future__ import annotations

import dataclasses
import pickle

import numpy as np

from . import smal_topology as topo


class _ChStandIn:
    """Stand-in for chumpy.ch.Ch leaves: the pickled state carries the value under 'x'."""

    def __setstate__(self, state):
        self.__dict__.update(state)

    @property
    def r(self):
        return self.x


class _LenientUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] == "chumpy":
            return _ChStandIn
        return super().find_class(module, name)


def load_pickle(path):
    """latin1 unpickle tolerant of chumpy leaves (reference uses encoding='latin1', smal_torch.py:31-34)."""
    with open(path, "rb") as f:
        return _LenientUnpickler(f, encoding="latin1").load()


def as_ndarray(x):
    """undo_chumpy (smal_torch.py:20-21) without chumpy; also densifies scipy sparse matrices."""
    if isinstance(x, np.ndarray):
        return x
    if hasattr(x, "todense"):
        return np.asarray(x.todense())
    if hasattr(x, "r"):
        return np.asarray(x.r)
    return np.asarray(x)


def align_template_to_symmetry_axis(v, sym_idx):
    """Restatement of reference smal_basics.py:7-37 (same arithmetic, same quirks).

    * subtracts the *scalar* mean over all 3V numbers (smal_basics.py:11)
    * shifts y by the mean y of the 135 centre vertices, then zeroes their y
    * overwrites the right side by the mirrored left side, pairing the k-th right vertex with the
      k-th left vertex in index order (boolean-mask assignment at smal_basics.py:27)
    Returns (v_sym, left_inds, right_inds, center_inds).  Raises if |left| != |right|
    (the reference drops into pdb there, smal_basics.py:32-35).
    """
    v = np.array(v, dtype=np.float64, copy=True)
    centre = np.asarray(topo.CENTER_VERTEX_IDS)
    v = v - np.mean(v)
    v[:, 1] = v[:, 1] - np.mean(v[centre, 1])
    v[centre, 1] = 0.0
    sym_idx = np.asarray(sym_idx).astype(np.int64).reshape(-1)
    left = v[:, 1] < 0
    right = v[:, 1] > 0
    center = v[:, 1] == 0
    v[left[sym_idx]] = np.array([1.0, -1.0, 1.0]) * v[left]
    left_inds, right_inds, center_inds = (np.where(m)[0] for m in (left, right, center))
    if len(left_inds) != len(right_inds):
        raise ValueError("SMAL template is not left/right balanced: %d vs %d"
                         % (len(left_inds), len(right_inds)))
    return v, left_inds, right_inds, center_inds


@dataclasses.dataclass
class SMALModelData:
    """Device-ready constants. Row-major, contiguous; column index of the blend bases = 3*v + axis."""
    v_template: np.ndarray     # (V, 3)  f32, family mean added + symmetrised
    shapedirs: np.ndarray      # (NB_all, 3V) f32  (first 20 rows are used by the fitter)
    posedirs: np.ndarray       # (306, 3V) f32
    J_regressor: np.ndarray    # (V, 35) f32 dense
    weights: np.ndarray        # (V, 35) f32
    parents: np.ndarray        # (35,) int32, parents[0] = -1
    faces: np.ndarray          # (F, 3) int32
    left_inds: np.ndarray
    right_inds: np.ndarray
    center_inds: np.ndarray

    @property
    def num_verts(self):
        return self.v_template.shape[0]

    @property
    def num_faces(self):
        return self.faces.shape[0]


def prepare_model(dd, data=None, sym_idx=None, shape_family_id=-1):
    """dict with the SMAL pickle's keys -> SMALModelData  (reference smal_torch.py:36-96).

    dd keys: f, v_template, shapedirs (V,3,NB), J_regressor (35,V sparse or dense), posedirs (V,3,306),
             kintree_table (2,35), weights (V,35)
    data   : dict with cluster_means (needed when shape_family_id != -1, smal_torch.py:58-67)
    sym_idx: (V,) mirror permutation (symIdx.pkl)
    """
    faces = as_ndarray(dd["f"]).astype(np.int32)
    v_template = as_ndarray(dd["v_template"]).astype(np.float64)
    nverts = v_template.shape[0]
    sd = as_ndarray(dd["shapedirs"])
    num_betas = sd.shape[-1]
    shapedir = np.reshape(sd, [-1, num_betas]).T.copy()                   # (NB, 3V)  smal_torch.py:53-54
    if shape_family_id != -1:
        betas = np.asarray(data["cluster_means"])[shape_family_id]
        v_template = v_template + np.matmul(betas[None, :], shapedir).reshape(-1, nverts, 3)[0]
    if sym_idx is None:
        raise ValueError("sym_idx is required (reference reads config.SMAL_SYM_FILE)")
    v_sym, left, right, center = align_template_to_symmetry_axis(v_template, sym_idx)
    jreg = as_ndarray(dd["J_regressor"])
    if jreg.shape[0] != nverts:                                            # stored (35, V)
        jreg = jreg.T
    pd = as_ndarray(dd["posedirs"])
    posedirs = np.reshape(pd, [-1, pd.shape[-1]]).T                        # (306, 3V)  smal_torch.py:85-86
    # kintree root parent is uint32 max -> -1 after the int32 cast (smal_torch.py:91)
    parents = as_ndarray(dd["kintree_table"])[0].astype(np.int64).astype(np.uint32).astype(np.int32)
    parents = parents.copy()
    parents[0] = -1
    if not all(0 <= parents[i] < i for i in range(1, len(parents))):
        raise ValueError("kinematic tree is not topologically ordered (parent[i] < i required)")
    return SMALModelData(
        v_template=np.ascontiguousarray(v_sym, dtype=np.float32),
        shapedirs=np.ascontiguousarray(shapedir, dtype=np.float32),
        posedirs=np.ascontiguousarray(posedirs, dtype=np.float32),
        J_regressor=np.ascontiguousarray(jreg, dtype=np.float32),
        weights=np.ascontiguousarray(as_ndarray(dd["weights"]), dtype=np.float32),
        parents=parents,
        faces=np.ascontiguousarray(faces),
        left_inds=left, right_inds=right, center_inds=center)


def load_smal_model(smal_file, smal_data_file, smal_sym_file, shape_family_id=-1):
    """Reads the user's SMAL files (paths as in reference config.py:32-49)."""
    dd = load_pickle(smal_file)
    data = load_pickle(smal_data_file) if shape_family_id != -1 else None
    sym = load_pickle(smal_sym_file)
    return prepare_model(dd, data, sym, shape_family_id)


# ------------------------------------------------------------------------------------------------
# shape / pose prior statistics (one-time host maths)
# ------------------------------------------------------------------------------------------------

def shape_prior_from_cov(cov, mean):
    """precision factor and mean as the reference builds them (smal_fitter.py:53-54 / :65-66).

    P = cholesky(inv(cov + 1e-5 I))  (lower), computed in float64, cast to float32.
    """
    cov = np.asarray(cov, dtype=np.float64)
    invcov = np.linalg.inv(cov + 1e-5 * np.eye(cov.shape[0]))
    prec = np.linalg.cholesky(invcov)
    return prec.astype(np.float32), np.asarray(mean, dtype=np.float32)


def unity_shape_prior(unity_npz_path):
    """(P (26,26) f32, mean (26,) f32) from unity_betas.npz  (smal_fitter.py:48-54)."""
    u = np.load(unity_npz_path)
    return shape_prior_from_cov(u["cov"][:-1, :-1], u["mean"][:-1])


def family_shape_prior(smal_data, shape_family, n_betas=topo.NUM_BETAS):
    """(P (20,20), mean (20,)) from the SMAL data pickle's cluster statistics (smal_fitter.py:62-69)."""
    cov = np.array(smal_data["cluster_cov"])[[shape_family]][0]
    prec, _ = shape_prior_from_cov(cov, np.zeros(cov.shape[0]))
    mean = np.asarray(smal_data["cluster_means"])[[shape_family]][0][:n_betas].astype(np.float32)
    return np.ascontiguousarray(prec[:n_betas, :n_betas]), mean


def load_pose_prior(prior_path):
    """(P (105,105) f32, mean (105,) f32, mask (105,) f32) as reference priors/pose_prior_35.py:51-92.

    The mask is built *before* the ignore-list edit (pose_prior_35.py:78-92), i.e. only the three
    global-rotation entries are masked.
    """
    res = load_pickle(prior_path)
    prec = as_ndarray(res["pic"]).astype(np.float32)
    mean = as_ndarray(res["mean_pose"]).astype(np.float32)
    mask = np.ones(mean.shape[0], dtype=np.float32)
    mask[:3] = 0.0
    return np.ascontiguousarray(prec), mean, mask


def initial_global_rotation():
    """eul_to_axis([-pi/2, 0, -pi/2])  (reference smal_fitter.py:81, utils.py:61-63).

    nibabel's euler2angle_axis(z, y, x) composes M = Rx(x) Ry(y) Rz(z); with the reference's argument
    order (euler[2], euler[1], euler[0]) that is Rx(-pi/2) Rz(-pi/2), a 120 deg turn about
    -(1,1,1)/sqrt(3): axis*angle = (-1.20919958,)*3 (SURVEY §8c known-answer).
    """
    z, y, x = -np.pi / 2, 0.0, -np.pi / 2
    cz, sz, cy, sy, cx, sx = np.cos(z), np.sin(z), np.cos(y), np.sin(y), np.cos(x), np.sin(x)
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]])
    ry = np.array([[cy, 0, sy], [0, 1.0, 0], [-sy, 0, cy]])
    rx = np.array([[1.0, 0, 0], [0, cx, -sx], [0, sx, cx]])
    m = rx @ ry @ rz
    angle = np.arccos(np.clip((np.trace(m) - 1.0) / 2.0, -1.0, 1.0))
    axis = np.array([m[2, 1] - m[1, 2], m[0, 2] - m[2, 0], m[1, 0] - m[0, 1]])
    axis = axis / np.linalg.norm(axis)
    return (axis * angle).astype(np.float64)


# reference smal_fitter/priors/joint_limits_prior.py:3-37 `Ranges` in the order of its part ids 0..31 (:41-74) = SMAL joints
# 1..32 = rows 0..31 of joint_rotations: [x_min, x_max], [y_min, y_max], [z_min, z_max] per joint
_JOINT_RANGES = (
    ((-0.3, 0.3), (-1.2, 0.5), (-0.1, 0.1)),      # pelvis0
    ((-0.4, 0.4), (-1.0, 0.9), (-0.8, 0.8)),      # spine
    ((-0.4, 0.4), (-1.0, 0.9), (-0.8, 0.8)),      # spine0
    ((-0.4, 0.4), (-0.5, 1.2), (-0.4, 0.4)),      # spine1
    ((-0.5, 0.5), (-0.4, 1.4), (-0.5, 0.5)),      # spine2
    ((-0.5, 0.5), (-0.6, 1.4), (-0.8, 0.8)),      # spine3
    ((-0.05, 0.05), (-1.3, 0.8), (-0.6, 0.6)),    # LLeg1
    ((-0.05, 0.05), (-1.0, 1.1), (-0.6, 0.6)),    # LLeg2
    ((-0.4, 0.1), (-0.3, 1.4), (-0.7, 0.4)),      # LLeg3
    ((-0.3, 0.1), (-0.4, 1.5), (-0.7, 0.3)),      # LFoot
    ((-0.05, 0.05), (-1.3, 0.8), (-0.6, 0.6)),    # RLeg1
    ((-0.05, 0.05), (-1.0, 0.9), (-0.6, 0.6)),    # RLeg2
    ((-0.1, 0.4), (-0.3, 1.4), (-0.4, 0.7)),      # RLeg3
    ((-0.1, 0.3), (-0.4, 1.5), (-0.3, 0.7)),      # RFoot
    ((-0.8, 0.8), (-1.0, 1.0), (-1.1, 1.1)),      # Neck
    ((-0.5, 0.5), (-1.0, 0.9), (-0.9, 0.9)),      # Head
    ((-0.2, 0.3), (-0.5, 0.8), (-0.5, 0.4)),      # LLegBack1
    ((-0.2, 0.3), (-0.6, 0.8), (-0.6, 0.5)),      # LLegBack2
    ((-0.3, 0.2), (-0.8, 0.2), (-0.5, 0.4)),      # LLegBack3
    ((-0.3, 0.2), (-0.3, 1.1), (-0.5, 0.3)),      # LFootBack
    ((-0.3, 0.2), (-0.5, 0.8), (-0.4, 0.5)),      # RLegBack1
    ((-0.3, 0.2), (-0.6, 0.8), (-0.5, 0.6)),      # RLegBack2
    ((-0.2, 0.3), (-0.8, 0.2), (-0.4, 0.5)),      # RLegBack3
    ((-0.2, 0.3), (-0.3, 1.1), (-0.3, 0.5)),      # RFootBack
    ((-0.1, 0.1), (-1.5, 1.4), (-1.2, 1.2)),      # Tail1
    ((-0.1, 0.1), (-1.0, 1.0), (-0.8, 0.8)),      # Tail2
    ((-0.1, 0.1), (-1.0, 1.0), (-0.8, 0.8)),      # Tail3
    ((-0.1, 0.1), (-1.0, 1.0), (-0.8, 0.8)),      # Tail4
    ((-0.1, 0.1), (-1.0, 1.0), (-0.8, 0.8)),      # Tail5
    ((-0.1, 0.1), (-1.4, 1.4), (-1.0, 1.0)),      # Tail6
    ((-0.1, 0.1), (-0.7, 1.1), (-0.9, 0.8)),      # Tail7
    ((-0.1, 0.1), (-1.1, 0.5), (-0.1, 0.1)),      # Mouth
)


def joint_limit_table():
    """-> (min (34,3), max (34,3)) float32 for SMALFitter's w_limit term (reference smal_fitter.py:76-79, 146-151).

    The reference builds LimitPrior().min_values / max_values (32 joints x 3 = 96 numbers, joint_limits_prior.py:76-83) and
    views them as (N_POSE, 3) = (34, 3) -- which cannot work, and the code is commented out.  Here the 32 limited joints
    keep the prior's own order (part ids 0..31 = SMAL joints 1..32 = joint_rotations rows 0..31) and the two joints the
    table does not know (33, 34: the ears) are unconstrained."""
    lo = np.full((34, 3), -np.inf, np.float32)
    hi = np.full((34, 3), np.inf, np.float32)
    r = np.asarray(_JOINT_RANGES, np.float32)            # (32, 3, 2)
    lo[:32], hi[:32] = r[:, :, 0], r[:, :, 1]
    return lo, hi
