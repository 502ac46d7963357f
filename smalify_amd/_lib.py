"""ctypes binding of libsmalfit.so (C-ABI declared in include/smalfit.h).

The library is built in-tree by `build_library()` (hipcc --offload-arch=gfx950) and loaded from
`smalify_amd/libsmalfit.so`.  There is no fallback: if the shared object is missing or cannot be
loaded, `load()` raises and every product entry point fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 5                    # SMALFIT_ABI_VERSION of include/smalfit.h this binding mirrors
LIB_PATH = os.environ.get("SMALFIT_LIB") or os.path.join(_HERE, "libsmalfit.so")   # override: development builds
CSRC = os.path.join(_HERE, "csrc")

_lib = None

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)


class ModelDesc(C.Structure):
    _fields_ = [("num_verts", C.c_int), ("num_faces", C.c_int), ("num_betas", C.c_int),
                ("v_template", C.c_void_p), ("shapedirs", C.c_void_p), ("posedirs", C.c_void_p),
                ("J_regressor", C.c_void_p), ("weights", C.c_void_p), ("parents", C.c_void_p),
                ("faces", C.c_void_p)]


class FitArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint), ("num_frames", C.c_int), ("window", C.c_int), ("logscale_mode", C.c_int),
                ("temporal", C.c_int), ("shape_prior_dim", C.c_int),
                ("w_j2d", C.c_float), ("w_sil", C.c_float), ("w_betas", C.c_float),
                ("w_pose", C.c_float), ("w_splay", C.c_float), ("w_temp", C.c_float),
                ("betas", C.c_void_p), ("log_beta_scales", C.c_void_p), ("global_rotation", C.c_void_p),
                ("joint_rotations", C.c_void_p), ("trans", C.c_void_p), ("global_mask", C.c_void_p),
                ("rotation_mask", C.c_void_p), ("target_joints", C.c_void_p),
                ("target_visibility", C.c_void_p), ("target_sil", C.c_void_p),
                ("halo_prev", C.c_void_p), ("halo_next", C.c_void_p), ("losses", C.c_void_p),
                ("g_betas", C.c_void_p), ("g_log_beta_scales", C.c_void_p),
                ("g_global_rotation", C.c_void_p), ("g_joint_rotations", C.c_void_p),
                ("g_trans", C.c_void_p), ("sil_out", C.c_void_p), ("proj_out", C.c_void_p),
                ("verts_out", C.c_void_p), ("target_sil_u8", C.c_void_p), ("w_limit", C.c_float),
                ("frame_offset", C.c_int), ("total_frames", C.c_int)]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(FitArgs)


class LbsArgs(C.Structure):
    _fields_ = [("num_frames", C.c_int), ("num_betas", C.c_int)] + [(n, C.c_void_p) for n in (
        "beta", "theta", "Rs", "logscale", "v_offset", "verts", "joints", "Rs_out", "v_shaped", "dverts", "djoints",
        "dbeta", "dtheta", "dRs", "dlogscale", "dv_offset")]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)     # smalfit_allgather_fn


class ShardArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint), ("world_size", C.c_int), ("rank", C.c_int), ("num_shared", C.c_int),
                ("num_trainable_shared", C.c_int), ("shared_grad", C.c_void_p), ("record", C.c_void_p), ("gathered", C.c_void_p),
                ("allgather", C.c_void_p), ("allgather_ctx", C.c_void_p)]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(ShardArgs)


class RcclCtx(C.Structure):
    _fields_ = [("comm", C.c_void_p), ("nccl_all_gather", C.c_void_p)]


class AdamArgs(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("num_segments", C.c_int), ("seg_begin", C.c_int * 4), ("seg_end", C.c_int * 4),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("step", C.c_int)]


class Fit3dArgs(C.Structure):
    _fields_ = [("num_meshes", C.c_int), ("num_betas", C.c_int), ("num_points", C.c_int),
                ("betas", C.c_void_p), ("log_beta_scales", C.c_void_p), ("global_rot", C.c_void_p),
                ("joint_rot", C.c_void_p), ("trans", C.c_void_p), ("deform_verts", C.c_void_p),
                ("lr_betas", C.c_float), ("lr_global_rot", C.c_float), ("lr_joint_rot", C.c_float),
                ("lr_trans", C.c_float), ("lr_deform_verts", C.c_float),
                ("m_betas", C.c_void_p), ("v_betas", C.c_void_p), ("m_global_rot", C.c_void_p),
                ("v_global_rot", C.c_void_p), ("m_joint_rot", C.c_void_p), ("v_joint_rot", C.c_void_p),
                ("m_trans", C.c_void_p), ("v_trans", C.c_void_p), ("m_deform_verts", C.c_void_p),
                ("v_deform_verts", C.c_void_p),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("adam_t", C.c_int),
                ("weights", C.c_float * 4), ("points", C.c_void_p), ("seed", C.c_ulonglong),
                ("iteration", C.c_uint), ("points_out", C.c_void_p), ("losses", C.c_void_p),
                ("verts_out", C.c_void_p)]


# every symbol include/smalfit.h declares: (restype, argtypes)
_VP, _I, _F = C.c_void_p, C.c_int, C.c_float
SIGNATURES = {
    "smalfit_version": (_I, []),
    "smalfit_last_error": (C.c_char_p, []),
    "smalfit_model_create": (_I, [C.POINTER(ModelDesc), C.POINTER(_VP)]),
    "smalfit_model_destroy": (None, [_VP]),
    "smalfit_engine_create": (_I, [_VP, _I, _I, C.POINTER(_VP)]),
    "smalfit_engine_destroy": (None, [_VP]),
    "smalfit_engine_status": (_I, [_VP, _VP, c_int_p]),
    "smalfit_engine_reset_raster_cache": (_I, [_VP, _VP]),
    "smalfit_engine_profile_begin": (_I, [_VP, _I, _I]),
    "smalfit_engine_profile_end": (_I, [_VP, _VP, _VP, _VP]),
    "smalfit_engine_set_pose_prior": (_I, [_VP, _VP, _VP, _VP]),
    "smalfit_engine_set_shape_prior": (_I, [_VP, _VP, _VP, _I]),
    "smalfit_engine_set_joint_limits": (_I, [_VP, _VP, _VP]),
    "smalfit_engine_clear_joint_limits": (_I, [_VP]),
    "smalfit_engine_set_option": (_I, [_VP, _I, _I]),
    "smalfit_lbs_forward": (_I, [_VP, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "smalfit_lbs_backward": (_I, [_VP, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "smalfit_lbs_forward_ex": (_I, [_VP, _VP, C.POINTER(LbsArgs)]),
    "smalfit_lbs_backward_ex": (_I, [_VP, _VP, C.POINTER(LbsArgs)]),
    "smalfit_rodrigues": (_I, [_VP, _I, _VP, _VP]),
    "smalfit_rodrigues_backward": (_I, [_VP, _I, _VP, _VP, _VP]),
    "smalfit_global_rigid_transformation": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "smalfit_global_rigid_transformation_backward": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "smalfit_render_forward": (_I, [_VP, _VP, _I, _VP, _VP, _I, _VP, _VP]),
    "smalfit_render_color": (_I, [_VP, _VP, _I, _VP, _VP, _VP]),
    "smalfit_render_backward": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP]),
    "smalfit_project_points_backward": (_I, [_VP, _I, _I, _VP, _VP, _VP]),
    "smalfit_fit_eval": (_I, [_VP, _VP, C.POINTER(FitArgs)]),
    "smalfit_fit_run": (_I, [_VP, _VP, C.POINTER(FitArgs), C.POINTER(AdamArgs), _I]),
    "smalfit_engine_set_graph": (_I, [_VP, _I]),
    "smalfit_adam_segments": (_I, [_VP, C.POINTER(AdamArgs)]),
    "smalfit_shard_local_step": (_I, [_VP, _VP, C.POINTER(FitArgs), C.POINTER(AdamArgs), _I, _VP, _VP]),
    "smalfit_shard_record": (_I, [_VP, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "smalfit_shard_reduce_step": (_I, [_VP, _I, _I, _VP, _I, _I, C.POINTER(AdamArgs)]),
    "smalfit_shard_run": (_I, [_VP, _VP, C.POINTER(FitArgs), C.POINTER(AdamArgs), C.POINTER(AdamArgs), C.POINTER(ShardArgs), _I]),
    "smalfit_rccl_allgather": (_I, [_VP, _VP, _VP, _I, _VP]),
    "smalfit_pose_prior": (_I, [_VP, _VP, _I, _VP, _VP]),
    "smalfit_pose_prior_backward": (_I, [_VP, _VP, _I, _VP, _VP, _VP]),
    "smalfit_temporal": (_I, [_VP, _VP, _I, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "smalfit_adam_step": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _F, _F, _F, _F, _I]),
    "smalfit_mesh_objective_create": (_I, [_I, _I, _VP, _I, _I, C.POINTER(_VP)]),
    "smalfit_mesh_objective_destroy": (None, [_VP]),
    "smalfit_mesh_objective_counts": (_I, [_VP, c_int_p, c_int_p]),
    "smalfit_mesh_objective_eval": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP]),
    "smalfit_mesh_targets_create": (_I, [_I, _VP, _VP, _VP, _VP, C.POINTER(_VP)]),
    "smalfit_mesh_targets_destroy": (None, [_VP]),
    "smalfit_mesh_targets_sample": (_I, [_VP, _VP, _I, C.c_ulonglong, C.c_uint, _VP]),
    "smalfit_fit3d_step": (_I, [_VP, _VP, _VP, _VP, C.POINTER(Fit3dArgs)]),
}


class SmalfitError(RuntimeError):
    pass


def build_library(verbose=False):
    """Compile smalify_amd/csrc for gfx950 into smalify_amd/libsmalfit.so (cross-compiles without a GPU)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -fno-slp-vectorize: the SLP vectoriser pairs scalar float32 multiply-adds into v_pk_fma_f32 / v_pk_mul_f32, which issue at 7.7
    # cycles per wave-instruction on gfx950 against 2.6 for the scalar form (profiles/r4_ubench_valu_issue_costs.txt): two scalar
    # instructions are faster than the packed one (the matrix-core kernels use builtins and are not affected)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-shared",
           os.path.join(CSRC, "smalfit_kernels.hip"), "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


def needs_rebuild():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    srcs.append(os.path.join(_HERE, "..", "include", "smalfit.h"))
    return any(os.path.getmtime(s) > t for s in srcs if os.path.exists(s))


def load():
    """Load libsmalfit.so and attach signatures. Raises SmalfitError if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmalfitError(
            "libsmalfit.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "— smalify_amd has no CPU fallback." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:
        raise SmalfitError("cannot load %s: %s" % (LIB_PATH, exc)) from exc
    # the ABI version first: a stale library lacks newer entry points, and "rebuild" is the message the user needs
    try:
        lib.smalfit_version.restype = C.c_int
        lib.smalfit_version.argtypes = []
        version = lib.smalfit_version()
    except AttributeError as exc:
        raise SmalfitError("%s exports no smalfit_version(): not a libsmalfit of this source tree, rebuild it" % LIB_PATH) from exc
    if version != ABI_VERSION:
        raise SmalfitError("%s has ABI version %d, this binding mirrors version %d of include/smalfit.h: rebuild the library"
                           % (LIB_PATH, version, ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise SmalfitError("%s does not export %s although it reports ABI version %d: header / library mismatch, rebuild the library"
                               % (LIB_PATH, name, version)) from exc
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().smalfit_last_error()
        raise SmalfitError("%s failed: %s" % (what or "smalfit call", msg.decode() if msg else "unknown error"))
