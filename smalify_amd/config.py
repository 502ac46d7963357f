"""Fitting configuration: same constant names and values as the reference's module-level config
(reference config.py:7-132), so driver code written against `config.X` keeps working.

Only what the fitting hot path reads is mirrored; marker / colour tables used by the reference's cv2
visualiser are out of scope.
"""
import os
import time
from os.path import join

data_path = os.environ.get("SMALIFY_DATA", "data")           # reference config.py:7
BADJA_PATH = join(data_path, "BADJA")
STANFORD_EXTRA_PATH = join(data_path, "StanfordExtra")
OUTPUT_DIR = "checkpoints/{0}".format(time.strftime("%Y%m%d-%H%M%S"))

CROP_SIZE = 256
VIS_FREQUENCY = 100
GPU_IDS = "0"

FORCE_SMAL_PRIOR = False
ALLOW_LIMB_SCALING = True

SHAPE_FAMILY = 1
SEQUENCE_OR_IMAGE_NAME = "badja:rs_dog"
IMAGE_RANGE = range(0, 1)
WINDOW_SIZE = 10

CHECKPOINT_NAME = "20201001-125009"
EPOCH_NAME = "st10_ep0"

SMAL_MODEL_PATH = join(data_path, "SMALST", "smpl_models")
SMAL_FILE = join(SMAL_MODEL_PATH, "my_smpl_00781_4_all.pkl")
_win = "_WIN" if os.name == "nt" else ""
SMAL_DATA_FILE = join(SMAL_MODEL_PATH, "my_smpl_data_00781_4_all%s.pkl" % _win)
SMAL_UV_FILE = join(SMAL_MODEL_PATH, "my_smpl_00781_4_all_template_w_tex_uv_001%s.pkl" % _win)
SMAL_SYM_FILE = join(SMAL_MODEL_PATH, "symIdx%s.pkl" % _win)
WALKING_PRIOR_FILE = join(data_path, "priors",
                          "walking_toy_symmetric_pose_prior_with_cov_35parts%s.pkl" % _win)
UNITY_SHAPE_PRIOR = join(data_path, "priors", "unity_betas.npz")

IMG_RES = 224
MESH_COLOR = [0, 172, 223]

# rows: w_j2d, w_sil, w_betas, w_pose, w_limit (unused), w_splay, w_temporal, iterations, lr
# columns: the four stages                                            (reference config.py:63-72)
OPT_WEIGHTS = [
    [25.0, 10.0, 7.5, 5.0],
    [0.0, 500.0, 5000.0, 5000.0],
    [0.0, 1.0, 1.0, 1.0],
    [0.0, 1.0, 1.0, 1.0],
    [0.0, 100.0, 100.0, 100.0],
    [0.0, 0.1, 0.1, 0.1],
    [500.0, 100.0, 100.0, 100.0],
    [150, 400, 600, 800],
    [5e-3, 5e-3, 5e-4, 1e-4]]

TORSO_JOINTS = [2, 5, 8, 11, 12, 23]                          # reference config.py:75

# 25 keypoints as indices into the 41 model joints; 15 appears twice   (reference config.py:77-88)
CANONICAL_MODEL_JOINTS = [
    10, 9, 8, 20, 19, 18, 14, 13, 12, 24, 23, 22, 25, 31, 33, 34, 35, 36, 38, 37, 39, 40, 15, 15, 28]

BADJA_ANNOTATED_CLASSES = [                                    # reference config.py:91-101
    14, 13, 12, 24, 23, 22, 10, 9, 8, 20, 19, 18, 25, 31, -1, -1, 33, -1, 36, 35, -1, -1, -1, 15, 28]

N_POSE = 34
N_BETAS = 20
