"""CPU check of the per-element device maths (smalify_amd/csrc/smalfit_math.h compiled for the host by
g++, test-only shim) against the oracle: Rodrigues fwd/bwd, camera fwd/bwd, and the per (pixel, face)
silhouette evaluation incl. the K-nearest depth threshold, on the full-size synthetic mesh."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import raster_naive as rn
from oracle import smal_oracle as so
from smalify_amd import model_io

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_math_shim.cpp")
SO = os.path.join(HERE, "_build", "libhost_math_shim.so")


@pytest.fixture(scope="module")
def shim():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    hdr = os.path.join(HERE, "..", "smalify_amd", "csrc", "smalfit_math.h")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", SRC, "-o", SO], check=True)
    return C.CDLL(SO)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_rodrigues_host(shim):
    rs = np.random.RandomState(1)
    th = rs.randn(40, 3).astype(np.float32)
    th[0] = 0
    th[1] = [2e-6, 0, -1e-6]
    th[2] = [0, 3.0, 0.5]
    G = rs.randn(40, 9).astype(np.float32)
    R = np.zeros((40, 9), np.float32)
    dth = np.zeros((40, 3), np.float32)
    shim.hm_rodrigues(40, _p(th), _p(G), _p(R), _p(dth))
    t = torch.from_numpy(th).double().requires_grad_(True)
    Ro = so.rodrigues(t)
    (Ro.reshape(40, 9) * torch.from_numpy(G).double()).sum().backward()
    assert np.abs(R - Ro.detach().numpy().reshape(40, 9)).max() < 2e-6
    err = np.linalg.norm(dth - t.grad.numpy()) / np.linalg.norm(t.grad.numpy())
    assert err < 2e-5, err
    assert np.abs(dth[0] - t.grad.numpy()[0]).max() < 1e-5       # theta = 0: generators, finite


def test_camera_host(shim):
    rs = np.random.RandomState(2)
    p = (rs.randn(50, 3) * 0.4).astype(np.float32)
    g2 = rs.randn(50, 2).astype(np.float32)
    ndc = np.zeros((50, 3), np.float32)
    g3 = np.zeros((50, 3), np.float32)
    shim.hm_camera(50, _p(p), _p(g2), _p(ndc), _p(g3))
    t = torch.from_numpy(p).double().requires_grad_(True)
    xn, yn, zv = so.world_to_ndc(t)
    (xn * torch.from_numpy(g2[:, 0]).double() + yn * torch.from_numpy(g2[:, 1]).double()).sum().backward()
    assert np.abs(ndc - torch.stack([xn, yn, zv], 1).detach().numpy()).max() < 1e-6
    assert np.linalg.norm(g3 - t.grad.numpy()) / np.linalg.norm(t.grad.numpy()) < 1e-6


@pytest.mark.parametrize("S,z", [(32, 0.0), (40, 1.45)])
def test_raster_math_host(shim, synth_model, S, z):
    """device maths (host build) == torch oracle == literal C naive rasteriser, incl. K=100 truncation
    (z = 0 is the reference's head-on start: up to ~870 candidate faces per pixel)."""
    md = synth_model
    om = so.OracleModel(md)
    rs = np.random.RandomState(0)
    theta = np.concatenate([model_io.initial_global_rotation()[None, None], 0.2 * rs.randn(1, 34, 3)], 1)
    with torch.no_grad():
        v, _, _, _ = so.smal_forward(om, torch.zeros(1, 20).double(), torch.from_numpy(theta).double(),
                                     torch.zeros(1, 6).double())
        v = (v + torch.tensor([0.02, -0.01, z]).double()).float().double()
    v.requires_grad_(True)
    sil_o, st = so.soft_silhouette(v, om.faces, S, return_stats=True)
    w = rs.randn(S, S).astype(np.float32)
    (sil_o[0] * torch.from_numpy(w).double()).sum().backward()
    xn, yn, zv = so.world_to_ndc(v.detach()[0].float())
    vn = np.ascontiguousarray(torch.stack([xn, yn, zv], 1).numpy(), np.float32)
    faces = np.ascontiguousarray(md.faces, np.int32)
    sil = np.zeros((S, S), np.float32)
    zthr = np.zeros((S, S), np.float32)
    gv = np.zeros((vn.shape[0], 2), np.float64)
    shim.hm_raster(_p(vn), vn.shape[0], _p(faces), faces.shape[0], S, _p(w), _p(sil), _p(zthr), _p(gv))
    d = np.abs(sil - sil_o[0].detach().numpy())
    assert st["max_faces_per_pixel"] > 100            # the truncation is exercised
    assert d.max() < 2e-4 and (d > 2e-5).mean() < 2e-3, (d.max(), (d > 2e-5).mean())
    # naive C restatement agrees as well
    s2, p2f, zb, di, _ = rn.forward(vn, faces, S)
    assert np.abs(s2 - sil).max() < 2e-4
    # gradients: chain the ndc gradient to world space and compare with the oracle's autograd
    xn, yn, zv = [t.double().numpy() for t in (xn, yn, zv)]
    s = 1.0 / np.tan(np.radians(30.0))
    gw = np.stack([-s / zv * gv[:, 0], s / zv * gv[:, 1], (xn * gv[:, 0] + yn * gv[:, 1]) / zv], 1)
    go = v.grad[0].numpy()
    err = np.linalg.norm(gw - go) / np.linalg.norm(go)
    assert err < 2e-3, err


@pytest.mark.parametrize("scaled", [False, True])
def test_global_rigid_transformation_host_against_reference_golden(shim, golden, scaled):
    """the free-standing kinematic chain (smalfit_global_rigid_transformation's per-frame function, compiled for the host)
    against outputs of the reference's batch_global_rigid_transformation (tests/golden, G2), with and without limb scales"""
    th = golden["g2_theta"]
    Rs = np.zeros((len(th), 9), np.float32)
    dummy = np.zeros((len(th), 3), np.float32)
    shim.hm_rodrigues(len(th), _p(np.ascontiguousarray(th)), _p(np.zeros((len(th), 9), np.float32)), _p(Rs), _p(dummy))
    n = 3
    Js = np.ascontiguousarray(golden["g2_Js"], np.float32)
    ls = np.ascontiguousarray(golden["g2_ls"], np.float32) if scaled else None
    parents = np.ascontiguousarray(golden["parents"], np.int32)
    newJ, A = np.zeros((n, 35, 3), np.float32), np.zeros((n, 35, 4, 4), np.float32)
    shim.hm_global_rigid(n, _p(Rs), _p(Js), _p(parents), None if ls is None else _p(ls), _p(newJ), _p(A))
    tag = "scale" if scaled else "noscale"
    want_J, want_A = golden["g2_newJ_" + tag], golden["g2_A_" + tag]
    assert np.linalg.norm(newJ - want_J) / np.linalg.norm(want_J) < 2e-6
    assert np.linalg.norm(A - want_A) / np.linalg.norm(want_A) < 2e-6
    assert (A[:, :, 3, :] == np.array([0, 0, 0, 1.0], np.float32)).all()


@pytest.mark.parametrize("S,z", [(48, 0.0), (100, 1.45)])
def test_candidate_form_agrees_bitwise_with_the_full_evaluation(shim, synth_model, S, z):
    """the forward sweep and the selection decide candidates with face_pixel_candidate, the backward with
    face_pixel_eval: on every (face, pixel) pair of the posed mesh the decision and the bits of the signed distance and the
    depth must be identical (the K-nearest bookkeeping relies on all kernels seeing the same candidates)"""
    md = synth_model
    om = so.OracleModel(md)
    rs = np.random.RandomState(1)
    theta = np.concatenate([model_io.initial_global_rotation()[None, None], 0.2 * rs.randn(1, 34, 3)], 1)
    with torch.no_grad():
        v, _, _, _ = so.smal_forward(om, torch.zeros(1, 20).double(), torch.from_numpy(theta).double(), torch.zeros(1, 6).double())
        v = (v + torch.tensor([0.02, -0.01, z]).double()).float()
    xn, yn, zv = so.world_to_ndc(v[0])
    vn = np.ascontiguousarray(torch.stack([xn, yn, zv], 1).numpy(), np.float32)
    faces = np.ascontiguousarray(md.faces, np.int32)
    pairs = C.c_longlong(0)
    shim.hm_candidate_mismatches.restype = C.c_int
    bad = shim.hm_candidate_mismatches(_p(vn), _p(faces), faces.shape[0], S, C.byref(pairs))
    assert pairs.value > 1e7 and bad == 0, (bad, pairs.value)


@pytest.mark.parametrize("scaled", [False, True])
def test_global_rigid_transformation_adjoint_host(shim, synth_model, scaled):
    """global_rigid_frame_bwd (the free-standing adjoint of batch_global_rigid_transformation, batch_lbs.py:75-170) against
    the oracle's autograd through its restatement of the chain"""
    rs = np.random.RandomState(5)
    n = 3
    th = (0.5 * rs.randn(n * 35, 3)).astype(np.float32)
    Rs = so.rodrigues(torch.from_numpy(th).double()).reshape(n, 35, 3, 3).float().numpy().copy()
    Js = (0.3 * rs.randn(n, 35, 3)).astype(np.float32)
    ls = (0.3 * rs.randn(n, 6)).astype(np.float32) if scaled else None
    dnewJ = rs.randn(n, 35, 3).astype(np.float32)
    dA = rs.randn(n, 35, 4, 4).astype(np.float32)
    parents = np.ascontiguousarray(synth_model.parents, np.int32)
    dRs, dJs, dls = np.zeros_like(Rs), np.zeros_like(Js), np.zeros((n, 6), np.float32)
    shim.hm_global_rigid_bwd(n, _p(Rs), _p(Js), _p(parents), _p(ls) if scaled else None, _p(dnewJ), _p(dA), _p(dRs), _p(dJs), _p(dls))
    R64 = torch.from_numpy(Rs).double().requires_grad_(True)
    J64 = torch.from_numpy(Js).double().requires_grad_(True)
    l64 = torch.from_numpy(ls).double().requires_grad_(True) if scaled else None
    g_t, g_r, a_t = so.kinematic_chain(R64, J64, [int(p) for p in parents], l64)
    dA64 = torch.from_numpy(dA).double()
    ((g_t * torch.from_numpy(dnewJ).double()).sum() + (g_r * dA64[:, :, :3, :3]).sum() + (a_t * dA64[:, :, :3, 3]).sum()).backward()
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))  # noqa: E731
    assert rel(dRs, R64.grad.numpy()) < 5e-6 and rel(dJs, J64.grad.numpy()) < 5e-6
    if scaled:
        assert rel(dls, l64.grad.numpy()) < 5e-6
