"""Thin object layer over the C-ABI: device model, workspace engine, typed calls on torch tensors.

PyTorch is plumbing here (HBM allocations, streams); every computation is a HIP kernel in
libsmalfit.so.  All tensors handed to these methods must be contiguous float32 CUDA(HIP) tensors.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import AdamArgs, FitArgs, LbsArgs, ModelDesc, ShardArgs, SmalfitError, check

LOSS_NAMES = ("joint", "pose", "splay", "betas", "sil_reproj", "temp_joint", "temp_global", "temp_trans", "limit")
NUM_LOSS_TERMS = len(LOSS_NAMES)


def _ptr(t):
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise SmalfitError("expected a contiguous float32 device tensor, got %r" % (
            (t.dtype, t.device, t.is_contiguous()) if isinstance(t, torch.Tensor) else type(t),))
    return C.c_void_p(t.data_ptr())


def _ptr_u8(t):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()):
        raise SmalfitError("expected a contiguous uint8 device tensor")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _host(a, dtype):
    return np.ascontiguousarray(np.asarray(a), dtype=dtype)


class DeviceModel:
    """SMAL constants resident in HBM (smalfit_model)."""

    def __init__(self, model_data):
        if not torch.cuda.is_available():
            raise SmalfitError("no HIP device available: smalify_amd has no CPU fallback")
        self.lib = _lib.load()
        md = model_data
        self.data = md
        self.num_verts = int(md.v_template.shape[0])
        self.num_faces = int(md.faces.shape[0])
        self.num_betas = int(md.shapedirs.shape[0])
        keep = dict(vt=_host(md.v_template, np.float32), sd=_host(md.shapedirs, np.float32),
                    pd=_host(md.posedirs, np.float32), jr=_host(md.J_regressor, np.float32),
                    w=_host(md.weights, np.float32), par=_host(md.parents, np.int32),
                    f=_host(md.faces, np.int32))
        desc = ModelDesc(self.num_verts, self.num_faces, self.num_betas,
                         keep["vt"].ctypes.data, keep["sd"].ctypes.data, keep["pd"].ctypes.data,
                         keep["jr"].ctypes.data, keep["w"].ctypes.data, keep["par"].ctypes.data,
                         keep["f"].ctypes.data)
        torch.cuda.init()
        torch.cuda.current_device()
        h = C.c_void_p()
        check(self.lib.smalfit_model_create(C.byref(desc), C.byref(h)), "smalfit_model_create")
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.smalfit_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class Engine:
    """Workspace for up to `max_frames` frames at `image_size`^2 (smalfit_engine)."""

    def __init__(self, model: DeviceModel, max_frames: int, image_size: int):
        self.lib = model.lib
        self.model = model
        self.max_frames = int(max_frames)
        self.image_size = int(image_size)
        h = C.c_void_p()
        check(self.lib.smalfit_engine_create(model.handle, self.max_frames, self.image_size, C.byref(h)),
              "smalfit_engine_create")
        self.handle = h
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.joint_limits_owner = None      # whoever set the engine's joint-limit table last (None: no table)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.smalfit_engine_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- priors ------------------------------------------------------------------------------
    def set_pose_prior(self, prec, mean, mask):
        p, m, k = _host(prec, np.float32), _host(mean, np.float32), _host(mask, np.float32)
        assert p.shape == (105, 105) and m.shape == (105,) and k.shape == (105,)
        check(self.lib.smalfit_engine_set_pose_prior(self.handle, p.ctypes.data, m.ctypes.data, k.ctypes.data),
              "smalfit_engine_set_pose_prior")

    def set_shape_prior(self, prec, mean):
        p, m = _host(prec, np.float32), _host(mean, np.float32)
        assert p.shape == (m.shape[0], m.shape[0])
        check(self.lib.smalfit_engine_set_shape_prior(self.handle, p.ctypes.data, m.ctypes.data, int(m.shape[0])),
              "smalfit_engine_set_shape_prior")
        self.shape_prior_dim = int(m.shape[0])

    def set_graph(self, enable=True):
        """replay one captured iteration per fit_run step (HIP graph) instead of enqueueing its launches one by one"""
        check(self.lib.smalfit_engine_set_graph(self.handle, int(bool(enable))), "smalfit_engine_set_graph")

    OPT_UNCLAMPED_EDGE_T = 1      # SMALFIT_OPT_UNCLAMPED_EDGE_T of include/smalfit.h

    def set_option(self, option, value):
        """engine options (include/smalfit.h): OPT_UNCLAMPED_EDGE_T = 1 selects the silhouette adjoint with the edge parameter left
        unclamped (what some pytorch3d 0.2.x sources are recalled to do); the default 0 is the exact gradient"""
        check(self.lib.smalfit_engine_set_option(self.handle, int(option), int(value)), "smalfit_engine_set_option")

    def clear_joint_limits(self):
        """back to the reference's behaviour: the w_limit column is ignored (its term is commented out upstream)"""
        check(self.lib.smalfit_engine_clear_joint_limits(self.handle), "smalfit_engine_clear_joint_limits")
        self.joint_limits_owner = None

    def set_joint_limits(self, min_values, max_values, owner=None):
        """(34,3) lower / upper limits of the joint rotations for the w_limit term (reference smal_fitter.py:76-79,146-151).
        The table is engine state; `owner` records whose it is, so that a fitter sharing the engine can tell that its table
        was replaced or cleared behind its back and put it back (FusedFitter.assert_joint_limits)."""
        lo, hi = _host(min_values, np.float32).reshape(-1), _host(max_values, np.float32).reshape(-1)
        assert lo.shape == (102,) and hi.shape == (102,)
        check(self.lib.smalfit_engine_set_joint_limits(self.handle, lo.ctypes.data, hi.ctypes.data), "smalfit_engine_set_joint_limits")
        self.joint_limits_owner = owner

    SECTIONS = ("lbs_fwd", "raster_sweep", "raster_select", "raster_bwd", "lbs_bwd", "raster_resolve", "raster_bbox")

    def reset_raster_cache(self):
        """Forget the rasteriser's cached per-pixel depth bounds (affects time only, never results)."""
        check(self.lib.smalfit_engine_reset_raster_cache(self.handle, _stream()), "smalfit_engine_reset_raster_cache")

    def profile_begin(self, max_evals, stride=1):
        """Time the sections of every `stride`-th fit_eval with HIP events on the launch stream (up to max_evals)."""
        check(self.lib.smalfit_engine_profile_begin(self.handle, int(max_evals), int(stride)), "smalfit_engine_profile_begin")

    def profile_end(self):
        """-> {section: (total_ms, count)} measured with HIP events on the current stream."""
        ms = (C.c_float * len(self.SECTIONS))()
        cnt = (C.c_int * len(self.SECTIONS))()
        check(self.lib.smalfit_engine_profile_end(self.handle, _stream(), ms, cnt), "smalfit_engine_profile_end")
        return {n: (float(ms[i]), int(cnt[i])) for i, n in enumerate(self.SECTIONS)}

    def status(self):
        """Synchronises the current stream; returns and clears the sticky status bits."""
        bits = C.c_int(0)
        check(self.lib.smalfit_engine_status(self.handle, _stream(), C.byref(bits)), "smalfit_engine_status")
        return bits.value

    # ---- fused fitter evaluation ---------------------------------------------------------------
    def fit_eval(self, *, betas, log_beta_scales, global_rotation, joint_rotations, trans,
                 target_joints, target_visibility, target_sil, weights, w_temp, window,
                 temporal=True, global_mask=None, rotation_mask=None, halo_prev=None, halo_next=None,
                 losses=None, grads=None, want=("betas", "log_beta_scales", "global_rotation",
                                                "joint_rotations", "trans"),
                 sil_out=None, proj_out=None, verts_out=None, frame_offset=0, total_frames=0):
        """One evaluation of sum_windows SMALFitter.forward + get_temporal and its gradient.

        weights = (w_j2d, w_sil, w_betas, w_pose, w_limit, w_splay) as in the reference's OPT_WEIGHTS columns
        (w_limit > 0 needs set_joint_limits).  target_sil: float32 (M,S,S), or uint8 bytes b = 255 t.
        Returns (losses (9,) device tensor in LOSS_NAMES order, grads dict)."""
        a, losses, grads, _keep = self.build_fit_args(
            betas=betas, log_beta_scales=log_beta_scales, global_rotation=global_rotation,
            joint_rotations=joint_rotations, trans=trans, target_joints=target_joints,
            target_visibility=target_visibility, target_sil=target_sil, weights=weights, w_temp=w_temp, window=window,
            temporal=temporal, global_mask=global_mask, rotation_mask=rotation_mask, halo_prev=halo_prev,
            halo_next=halo_next, losses=losses, grads=grads, want=want, sil_out=sil_out, proj_out=proj_out,
            verts_out=verts_out, frame_offset=frame_offset, total_frames=total_frames)
        check(self.lib.smalfit_fit_eval(self.handle, _stream(), C.byref(a)), "smalfit_fit_eval")
        return losses, grads

    def build_fit_args(self, *, betas, log_beta_scales, global_rotation, joint_rotations, trans,
                       target_joints, target_visibility, target_sil, weights, w_temp, window,
                       temporal=True, global_mask=None, rotation_mask=None, halo_prev=None, halo_next=None,
                       losses=None, grads=None, want=("betas", "log_beta_scales", "global_rotation",
                                                      "joint_rotations", "trans"),
                       sil_out=None, proj_out=None, verts_out=None, frame_offset=0, total_frames=0):
        """-> (smalfit_fit_args, losses, grads, keep-alive list): the argument block of smalfit_fit_eval / smalfit_fit_run.
        The block holds raw device pointers: the caller keeps the tensors alive for as long as it uses it."""
        M = int(global_rotation.shape[0])
        w_j2d, w_sil, w_betas, w_pose, w_limit, w_splay = [float(w) for w in weights]
        dev = global_rotation.device
        if losses is None:
            losses = torch.empty(NUM_LOSS_TERMS, device=dev, dtype=torch.float32)
        elif losses.numel() < NUM_LOSS_TERMS:
            raise SmalfitError("losses must hold %d floats" % NUM_LOSS_TERMS)
        if grads is None:
            grads = {}
        params = dict(betas=betas, log_beta_scales=log_beta_scales, global_rotation=global_rotation,
                      joint_rotations=joint_rotations, trans=trans)
        for k in want:
            if k not in grads and params[k] is not None:
                grads[k] = torch.empty_like(params[k])
        if log_beta_scales is None:
            mode = 0
        elif log_beta_scales.dim() == 1:
            mode = 1
        else:
            mode = 2
        a = FitArgs()
        a.num_frames, a.window, a.logscale_mode, a.temporal = M, int(window), mode, int(bool(temporal))
        a.shape_prior_dim = 0
        # where these M frames sit in their sequence (shards; zeros = they are the whole sequence)
        a.frame_offset, a.total_frames = int(frame_offset), int(total_frames)
        a.w_j2d, a.w_sil, a.w_betas, a.w_pose, a.w_splay, a.w_temp = w_j2d, w_sil, w_betas, w_pose, w_splay, float(w_temp)
        a.betas, a.log_beta_scales = _ptr(betas), _ptr(log_beta_scales)
        a.global_rotation, a.joint_rotations, a.trans = _ptr(global_rotation), _ptr(joint_rotations), _ptr(trans)
        a.global_mask, a.rotation_mask = _ptr(global_mask), _ptr(rotation_mask)
        a.target_joints, a.target_visibility = _ptr(target_joints), _ptr(target_visibility)
        if target_sil is not None and target_sil.dtype == torch.uint8:
            a.target_sil, a.target_sil_u8 = None, _ptr_u8(target_sil)
        else:
            a.target_sil, a.target_sil_u8 = _ptr(target_sil), None
        a.w_limit = w_limit
        a.halo_prev, a.halo_next = _ptr(halo_prev), _ptr(halo_next)
        a.losses = _ptr(losses)
        a.g_betas = _ptr(grads.get("betas")) if "betas" in want else None
        a.g_log_beta_scales = _ptr(grads.get("log_beta_scales")) if "log_beta_scales" in want else None
        a.g_global_rotation = _ptr(grads.get("global_rotation")) if "global_rotation" in want else None
        a.g_joint_rotations = _ptr(grads.get("joint_rotations")) if "joint_rotations" in want else None
        a.g_trans = _ptr(grads.get("trans")) if "trans" in want else None
        a.sil_out, a.proj_out, a.verts_out = _ptr(sil_out), _ptr(proj_out), _ptr(verts_out)
        keep = [betas, log_beta_scales, global_rotation, joint_rotations, trans, target_joints, target_visibility,
                target_sil, global_mask, rotation_mask, halo_prev, halo_next, losses, grads, sil_out, proj_out, verts_out]
        return a, losses, grads, keep

    def fit_run(self, fit_args, adam_args, iterations):
        """`iterations` x (evaluation + backward + Adam step) in one call (smalfit_fit_run; reference
        optimize_to_joints.py:113-137).  Advances adam_args.step."""
        check(self.lib.smalfit_fit_run(self.handle, _stream(), C.byref(fit_args), C.byref(adam_args), int(iterations)),
              "smalfit_fit_run")
        adam_args.step += int(iterations)

    # ---- SMAL.__call__ ---------------------------------------------------------------------------
    def lbs_forward(self, beta, theta, logscale=None, want_Rs=True, want_v_shaped=True, v_offset=None):
        """SMAL.__call__ (smalfit_lbs_forward_ex): theta (M,35,3) axis-angles or (M,35,3,3) rotation matrices; v_offset (M,V,3)
        is the reference's del_v (+ a per-call v_template's difference to the model's)"""
        M, nb = int(theta.shape[0]), int(beta.shape[1])
        V = self.model.num_verts
        dev = theta.device
        verts = torch.empty(M, V, 3, device=dev)
        joints = torch.empty(M, 41, 3, device=dev)
        Rs = torch.empty(M, 35, 3, 3, device=dev) if want_Rs else None
        vs = torch.empty(M, V, 3, device=dev) if want_v_shaped else None
        a = LbsArgs()
        a.num_frames, a.num_betas = M, nb
        a.beta, a.logscale, a.v_offset = _ptr(beta), _ptr(logscale), _ptr(v_offset)
        if theta.dim() == 4:
            a.Rs = _ptr(theta)
        else:
            a.theta = _ptr(theta)
        a.verts, a.joints, a.Rs_out, a.v_shaped = _ptr(verts), _ptr(joints), _ptr(Rs), _ptr(vs)
        check(self.lib.smalfit_lbs_forward_ex(self.handle, _stream(), C.byref(a)), "smalfit_lbs_forward_ex")
        return verts, joints, Rs, vs

    def lbs_backward(self, beta, theta, logscale, dverts, djoints, v_offset=None):
        """-> (dbeta, dtheta (or dRs for matrix input), dlogscale) and, when v_offset is given, dv_offset as a 4th value"""
        M, nb = int(theta.shape[0]), int(beta.shape[1])
        dev = theta.device
        dbeta = torch.empty(M, nb, device=dev)
        dth = torch.empty_like(theta)
        dls = torch.empty(M, 6, device=dev) if logscale is not None else None
        doff = torch.empty_like(v_offset) if v_offset is not None else None
        a = LbsArgs()
        a.num_frames, a.num_betas = M, nb
        a.beta, a.logscale, a.v_offset = _ptr(beta), _ptr(logscale), _ptr(v_offset)
        if theta.dim() == 4:
            a.Rs, a.dRs = _ptr(theta), _ptr(dth)
        else:
            a.theta, a.dtheta = _ptr(theta), _ptr(dth)
        a.dverts, a.djoints = _ptr(dverts), _ptr(djoints)
        a.dbeta, a.dlogscale, a.dv_offset = _ptr(dbeta), _ptr(dls), _ptr(doff)
        check(self.lib.smalfit_lbs_backward_ex(self.handle, _stream(), C.byref(a)), "smalfit_lbs_backward_ex")
        if v_offset is not None:
            return dbeta, dth, dls, doff
        return dbeta, dth, dls

    # ---- Renderer -----------------------------------------------------------------------------------
    def render_forward(self, verts, points=None, want_sil=True):
        M = int(verts.shape[0])
        S = self.image_size
        sil = torch.empty(M, S, S, device=verts.device) if want_sil else None
        proj = None
        P = 0
        if points is not None:
            P = int(points.shape[1])
            proj = torch.empty(M, P, 2, device=verts.device)
        check(self.lib.smalfit_render_forward(self.handle, _stream(), M, _ptr(verts), _ptr(points), P, _ptr(sil),
                                              _ptr(proj)), "smalfit_render_forward")
        return sil, proj

    def render_color(self, verts, rgb):
        """Hard-Phong colour render (visualisation, no gradient): verts (M,V,3) world incl. translation, rgb 3 floats in
        [0,1] -> (M,3,S,S)."""
        M = int(verts.shape[0])
        S = self.image_size
        image = torch.empty(M, 3, S, S, device=verts.device)
        col = np.ascontiguousarray(np.asarray(rgb, dtype=np.float32).reshape(3))
        check(self.lib.smalfit_render_color(self.handle, _stream(), M, _ptr(verts), col.ctypes.data, _ptr(image)),
              "smalfit_render_color")
        return image

    def render_backward(self, verts, sil, dsil):
        dverts = torch.empty_like(verts)
        check(self.lib.smalfit_render_backward(self.handle, _stream(), int(verts.shape[0]), _ptr(verts), _ptr(sil),
                                               _ptr(dsil), _ptr(dverts)), "smalfit_render_backward")
        return dverts

    def project_points_backward(self, points, dproj):
        dpoints = torch.empty_like(points)
        count = int(points.numel() // 3)
        check(self.lib.smalfit_project_points_backward(_stream(), count, self.image_size, _ptr(points), _ptr(dproj),
                                                       _ptr(dpoints)), "smalfit_project_points_backward")
        return dpoints


def pose_prior(engine, x):
    N = int(x.shape[0])
    out = torch.empty(N, 105, device=x.device)
    check(engine.lib.smalfit_pose_prior(engine.handle, _stream(), N, _ptr(x), _ptr(out)), "smalfit_pose_prior")
    return out


def pose_prior_backward(engine, x, dout):
    N = int(x.shape[0])
    dx = torch.empty(N, 105, device=x.device)
    check(engine.lib.smalfit_pose_prior_backward(engine.handle, _stream(), N, _ptr(x), _ptr(dout), _ptr(dx)),
          "smalfit_pose_prior_backward")
    return dx


def temporal(engine, w_temp, global_rotation, joint_rotations, trans, global_mask=None, rotation_mask=None):
    """-> (losses (3,) joint/global/trans, g_global_rotation, g_joint_rotations, g_trans)"""
    N = int(global_rotation.shape[0])
    losses = torch.empty(3, device=trans.device)
    gg, gj, gt = torch.empty_like(global_rotation), torch.empty_like(joint_rotations), torch.empty_like(trans)
    check(engine.lib.smalfit_temporal(engine.handle, _stream(), N, float(w_temp), _ptr(global_rotation),
                                      _ptr(joint_rotations), _ptr(trans), _ptr(global_mask), _ptr(rotation_mask),
                                      _ptr(losses), _ptr(gg), _ptr(gj), _ptr(gt)), "smalfit_temporal")
    return losses, gg, gj, gt


def rodrigues(theta):
    lib = _lib.load()
    count = int(theta.shape[0])
    R = torch.empty(count, 3, 3, device=theta.device)
    check(lib.smalfit_rodrigues(_stream(), count, _ptr(theta), _ptr(R)), "smalfit_rodrigues")
    return R


def rodrigues_backward(theta, dR):
    lib = _lib.load()
    count = int(theta.shape[0])
    dth = torch.empty_like(theta)
    check(lib.smalfit_rodrigues_backward(_stream(), count, _ptr(theta), _ptr(dR), _ptr(dth)), "smalfit_rodrigues_backward")
    return dth


def global_rigid_transformation(Rs, Js, parents, logscale=None):
    """Rs (N,35,3,3), Js (N,35,3), parents (35,) host ints, logscale (N,6) or None -> new_J (N,35,3), A (N,35,4,4)
    (reference batch_lbs.py:75-170; adjoint: global_rigid_transformation_backward)"""
    lib = _lib.load()
    N = int(Rs.shape[0])
    if tuple(Rs.shape) != (N, 35, 3, 3) or tuple(Js.shape) != (N, 35, 3):
        raise SmalfitError("Rs must be (N,35,3,3) and Js (N,35,3)")
    if logscale is not None and tuple(logscale.shape) != (N, 6):
        raise SmalfitError("betas_logscale must be (N,6)")
    par = _host(parents, np.int32).reshape(-1)
    if par.shape[0] != 35:
        raise SmalfitError("parents must hold 35 entries")
    new_J = torch.empty(N, 35, 3, device=Rs.device)
    A = torch.empty(N, 35, 4, 4, device=Rs.device)
    check(lib.smalfit_global_rigid_transformation(_stream(), N, _ptr(Rs), _ptr(Js), par.ctypes.data, _ptr(logscale),
                                                  _ptr(new_J), _ptr(A)), "smalfit_global_rigid_transformation")
    return new_J, A


def global_rigid_transformation_backward(Rs, Js, parents, logscale, d_new_J, d_A):
    """adjoint of global_rigid_transformation -> (dRs (N,35,3,3), dJs (N,35,3), dlogscale (N,6) or None)"""
    lib = _lib.load()
    N = int(Rs.shape[0])
    par = _host(parents, np.int32).reshape(-1)
    dRs, dJs = torch.empty_like(Rs), torch.empty_like(Js)
    dls = torch.empty_like(logscale) if logscale is not None else None
    scratch = torch.empty(N * 840, device=Rs.device)
    check(lib.smalfit_global_rigid_transformation_backward(_stream(), N, _ptr(Rs), _ptr(Js), par.ctypes.data, _ptr(logscale),
                                                           _ptr(d_new_J), _ptr(d_A), _ptr(scratch), _ptr(dRs), _ptr(dJs),
                                                           _ptr(dls)), "smalfit_global_rigid_transformation_backward")
    return dRs, dJs, dls


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, t, beta1=0.5, beta2=0.999, eps=1e-8):
    """In-place torch.optim.Adam step on a flat float32 device tensor (reference optimize_to_joints.py:96,137)."""
    lib = _lib.load()
    check(lib.smalfit_adam_step(_stream(), int(param.numel()), _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
                                float(lr), float(beta1), float(beta2), float(eps), int(t)), "smalfit_adam_step")


def make_adam_args(param, grad, exp_avg, exp_avg_sq, segments, lr, step=0, beta1=0.5, beta2=0.999, eps=1e-8):
    """smalfit_adam_args over flat float32 device tensors; segments = [(begin, end), ...] (at most 4)"""
    if len(segments) > 4:
        raise SmalfitError("at most 4 trainable ranges")
    a = AdamArgs()
    a.param, a.grad, a.exp_avg, a.exp_avg_sq = _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq)
    a.num_segments = len(segments)
    for k, (b, e) in enumerate(segments):
        a.seg_begin[k], a.seg_end[k] = int(b), int(e)
    a.lr, a.beta1, a.beta2, a.eps, a.step = float(lr), float(beta1), float(beta2), float(eps), int(step)
    return a


def adam_segments(adam_args):
    """one fused Adam launch over the ranges (t = step + 1); advances adam_args.step"""
    check(_lib.load().smalfit_adam_segments(_stream(), C.byref(adam_args)), "smalfit_adam_segments")
    adam_args.step += 1


def shard_record(num_shared, shared_grad, num_frames, global_rotation, joint_rotations, trans, global_mask, rotation_mask, record):
    check(_lib.load().smalfit_shard_record(_stream(), int(num_shared), _ptr(shared_grad), int(num_frames), _ptr(global_rotation),
                                           _ptr(joint_rotations), _ptr(trans), _ptr(global_mask), _ptr(rotation_mask),
                                           _ptr(record)), "smalfit_shard_record")


def shard_local_step(engine, fit_args, adam_args, num_shared, shared_grad, record):
    """evaluation + Adam on the per-frame ranges + this rank's record, one library call (smalfit_shard_local_step); does not
    advance adam_args.step (shard_reduce_step closes the iteration)"""
    check(engine.lib.smalfit_shard_local_step(engine.handle, _stream(), C.byref(fit_args), C.byref(adam_args), int(num_shared),
                                              _ptr(shared_grad), _ptr(record)), "smalfit_shard_local_step")


def shard_run(engine, fit_args, adam_local, adam_shared, shard_args, iterations):
    """`iterations` x [evaluation + per-frame Adam + record -> all-gather -> rank-ordered sum + shared Adam] in ONE library call
    (smalfit_shard_run); advances both step counts"""
    check(engine.lib.smalfit_shard_run(engine.handle, _stream(), C.byref(fit_args), C.byref(adam_local), C.byref(adam_shared),
                                       C.byref(shard_args), int(iterations)), "smalfit_shard_run")
    adam_local.step += int(iterations)
    adam_shared.step += int(iterations)


def shard_reduce_step(world_size, record_stride, gathered, num_shared, num_trainable, adam_args):
    """sum of the gathered partial shape gradients (rank order) + Adam on the first num_trainable shared parameters;
    does not advance adam_args.step (the caller does, once per iteration)"""
    check(_lib.load().smalfit_shard_reduce_step(_stream(), int(world_size), int(record_stride), _ptr(gathered), int(num_shared),
                                                int(num_trainable), C.byref(adam_args)), "smalfit_shard_reduce_step")


# ---- fitter_3d: SMAL-to-mesh objective ------------------------------------------------------
MESH_LOSS_NAMES = ("chamfer", "edge", "normal", "laplacian", "total")


class MeshObjective:
    """Chamfer + edge + normal-consistency + uniform-Laplacian objective of N deforming meshes of one topology and its
    gradient (smalfit_mesh_objective; reference fitter_3d/trainer.py:205-227 + the PyTorch3D v0.2.5 losses)."""

    def __init__(self, num_verts, faces, max_meshes, max_points):
        if not torch.cuda.is_available():
            raise SmalfitError("no HIP device available: smalify_amd has no CPU fallback")
        self.lib = _lib.load()
        f = _host(faces, np.int32).reshape(-1, 3)
        self.num_verts, self.num_faces = int(num_verts), int(f.shape[0])
        self.max_meshes, self.max_points = int(max_meshes), int(max_points)
        torch.cuda.init()
        torch.cuda.current_device()
        h = C.c_void_p()
        check(self.lib.smalfit_mesh_objective_create(self.num_verts, self.num_faces, f.ctypes.data, self.max_meshes,
                                                     self.max_points, C.byref(h)), "smalfit_mesh_objective_create")
        self.handle = h
        ne, npairs = C.c_int(), C.c_int()
        check(self.lib.smalfit_mesh_objective_counts(h, C.byref(ne), C.byref(npairs)), "smalfit_mesh_objective_counts")
        self.num_edges, self.num_face_pairs = ne.value, npairs.value

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.smalfit_mesh_objective_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def eval(self, lbs_verts, trans, deform_verts, points, weights, out=None):
        """-> dict(losses (5,), verts (N,V,3), dverts (N,V,3), dtrans (N,3)); `weights` = (w_chamfer, w_edge, w_normal,
        w_laplacian) host floats; `out` may carry preallocated tensors under the same keys"""
        N, V = int(lbs_verts.shape[0]), self.num_verts
        if tuple(lbs_verts.shape) != (N, V, 3) or tuple(trans.shape) != (N, 3):
            raise SmalfitError("lbs_verts must be (N,%d,3) and trans (N,3)" % V)
        if deform_verts is not None and tuple(deform_verts.shape) != (N, V, 3):
            raise SmalfitError("deform_verts must be (N,%d,3)" % V)
        S = 0
        if points is not None:
            if points.dim() != 3 or int(points.shape[0]) != N or int(points.shape[2]) != 3:
                raise SmalfitError("points must be (N,S,3)")
            S = int(points.shape[1])
        dev = lbs_verts.device
        o = out if out is not None else {}
        for k, shape in (("losses", (5,)), ("verts", (N, V, 3)), ("dverts", (N, V, 3)), ("dtrans", (N, 3))):
            if k not in o or tuple(o[k].shape) != shape:
                o[k] = torch.empty(shape, device=dev)
        w = _host(weights, np.float32).reshape(4)
        check(self.lib.smalfit_mesh_objective_eval(self.handle, _stream(), N, _ptr(lbs_verts), _ptr(trans),
                                                   _ptr(deform_verts), _ptr(points), S, w.ctypes.data, _ptr(o["verts"]),
                                                   _ptr(o["losses"]), _ptr(o["dverts"]), _ptr(o["dtrans"])),
              "smalfit_mesh_objective_eval")
        return o


class MeshTargets:
    """Target meshes resident in HBM + the area-weighted point sampler (smalfit_mesh_targets; reference
    fitter_3d/utils.py:253 Meshes(...) and trainer.py:209 sample_points_from_meshes)."""

    def __init__(self, verts_list, faces_list):
        if not torch.cuda.is_available():
            raise SmalfitError("no HIP device available: smalify_amd has no CPU fallback")
        if len(verts_list) == 0 or len(verts_list) != len(faces_list):
            raise SmalfitError("need one (verts, faces) pair per target mesh")
        self.lib = _lib.load()
        self.verts_list = [_host(v, np.float32).reshape(-1, 3) for v in verts_list]
        self.faces_list = [_host(f, np.int32).reshape(-1, 3) for f in faces_list]
        vc = np.array([len(v) for v in self.verts_list], np.int32)
        fc = np.array([len(f) for f in self.faces_list], np.int32)
        vv = np.ascontiguousarray(np.concatenate(self.verts_list, axis=0))
        ff = np.ascontiguousarray(np.concatenate(self.faces_list, axis=0))
        torch.cuda.init()
        torch.cuda.current_device()
        h = C.c_void_p()
        check(self.lib.smalfit_mesh_targets_create(len(vc), vc.ctypes.data, fc.ctypes.data, vv.ctypes.data,
                                                   ff.ctypes.data, C.byref(h)), "smalfit_mesh_targets_create")
        self.handle = h
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.joint_limits_owner = None      # whoever set the engine's joint-limit table last (None: no table)

    def __len__(self):
        return len(self.verts_list)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.smalfit_mesh_targets_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def sample(self, num_points, seed, iteration, out=None):
        """(N, num_points, 3) points; the same (seed, iteration) always gives the same points"""
        N = len(self)
        if out is None or tuple(out.shape) != (N, int(num_points), 3):
            out = torch.empty(N, int(num_points), 3, device=self.device)
        check(self.lib.smalfit_mesh_targets_sample(self.handle, _stream(), int(num_points), int(seed) & (2 ** 64 - 1),
                                                   int(iteration) & 0xFFFFFFFF, _ptr(out)), "smalfit_mesh_targets_sample")
        return out
