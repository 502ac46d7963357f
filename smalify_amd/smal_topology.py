"""Fixed dimensions and index tables of the SMAL quadruped model as the reference hard-codes them.

These are constants of the SMAL topology / the reference's fitting contract, not code:
  * dimensions                    reference smal_model/smal_torch.py:107,135,138 ; config.py:131-132
  * symmetry-plane vertex ids     reference smal_model/smal_basics.py:9
  * posed-mesh landmark vertices  reference smal_model/smal_torch.py:176-184
  * limb-scale joint groups       reference smal_model/batch_lbs.py:107-121
"""
import numpy as np

NUM_VERTS = 3889
NUM_FACES = 7774
NUM_JOINTS = 35           # 34 articulated joints + root
NUM_POSE = 34             # config.N_POSE
NUM_BETAS = 20            # config.N_BETAS
NUM_LOGSCALES = 6
NUM_POSE_FEATURES = 9 * (NUM_JOINTS - 1)   # 306
NUM_MODEL_JOINTS = NUM_JOINTS + 6          # 35 regressed + 6 landmark vertices = 41
NUM_KEYPOINTS = 25


def _ranges(*parts):
    out = []
    for p in parts:
        if isinstance(p, tuple):
            out.extend(range(p[0], p[1] + 1))
        else:
            out.append(p)
    return out


# 135 vertices on the sagittal (y = 0) plane, smal_basics.py:9
CENTER_VERTEX_IDS = _ranges(
    (0, 32), 37, 55, 119, 120, 163, 209, 210, 211, 213, 216, 227, 326, 395, 452, 578, 910, 959,
    964, 975, 976, 977, 1172, 1175, 1176, 1178, 1194, 1243, 1739, (1796, 1840), (1842, 1863),
    1870, 1919, 1960, 1961, 1965, 1967, 2003)
assert len(CENTER_VERTEX_IDS) == 135

# nose tip, chin, right ear tip, left ear tip, left eye, right eye  (smal_torch.py:176-184)
LANDMARK_VERTEX_IDS = (1863, 26, 2124, 150, 3055, 1097)

# limb-scale groups (batch_lbs.py:107-109)
LEG_JOINTS = list(range(7, 15)) + list(range(17, 25))
TAIL_JOINTS = list(range(25, 32))
EAR_JOINTS = [33, 34]


def limb_scale_mask():
    """(6, 105) matrix M with s = exp(log_scales @ M).reshape(35, 3)   (batch_lbs.py:111-124).

    legs : z <- ls0 (length), x,y <- ls1 (fatness);  tail: x <- ls2, y,z <- ls3;  ears: y <- ls4, z <- ls5
    """
    m = np.zeros((NUM_JOINTS, 3, NUM_LOGSCALES), dtype=np.float32)
    m[LEG_JOINTS, 2, 0] = 1.0
    m[LEG_JOINTS, 0, 1] = 1.0
    m[LEG_JOINTS, 1, 1] = 1.0
    m[TAIL_JOINTS, 0, 2] = 1.0
    m[TAIL_JOINTS, 1, 3] = 1.0
    m[TAIL_JOINTS, 2, 3] = 1.0
    m[EAR_JOINTS, 1, 4] = 1.0
    m[EAR_JOINTS, 2, 5] = 1.0
    return np.ascontiguousarray(m.reshape(NUM_JOINTS * 3, NUM_LOGSCALES).T)


def limb_scale_index():
    """(35, 3) int table: which log-scale drives axis a of joint j, or -1 (the mask has one 1 per row)."""
    m = limb_scale_mask()                       # (6, 105)
    idx = np.full(NUM_JOINTS * 3, -1, dtype=np.int32)
    for c in range(NUM_LOGSCALES):
        idx[m[c] > 0] = c
    return idx.reshape(NUM_JOINTS, 3)


# Kinematic tree used by the *synthetic* stand-in model (the real one comes from the model pickle's
# kintree_table, smal_torch.py:91).  Joint naming follows priors/pose_prior_35.py:17 (name2id35).
SYNTH_PARENTS = np.array(
    [-1, 0, 1, 2, 3, 4, 5,            # root, pelvis0, spine, spine0..spine3
     6, 7, 8, 9,                      # LLeg1..LFoot
     6, 11, 12, 13,                   # RLeg1..RFoot
     6, 15,                           # Neck, Head
     0, 17, 18, 19,                   # LLegBack1..LFootBack
     0, 21, 22, 23,                   # RLegBack1..RFootBack
     0, 25, 26, 27, 28, 29, 30,       # Tail1..Tail7
     16, 16, 16], dtype=np.int32)     # Mouth, LEar, REar
assert SYNTH_PARENTS.shape[0] == NUM_JOINTS
