"""ctypes wrapper of oracle/raster_naive.c (ORACLE, test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libraster_naive.so")
_lib = None

SIGMA = 1e-4
BLUR = float(np.log(1.0 / 1e-4 - 1.0) * 1e-4)


def build():
    src = os.path.join(_HERE, "raster_naive.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return _SO


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def forward(v_ndc, faces, S, K=100, sigma=SIGMA, blur=BLUR, want_fragments=True):
    """v_ndc (V,3) float32 (x_ndc, y_ndc, z_view); faces (F,3) int32 -> sil (S,S) [, p2f, zbuf, dists], max_cand"""
    lib = _load()
    v = np.ascontiguousarray(v_ndc, np.float32)
    f = np.ascontiguousarray(faces, np.int32)
    sil = np.zeros((S, S), np.float32)
    p2f = np.zeros((S, S, K), np.int32)
    zbuf = np.zeros((S, S, K), np.float32)
    dists = np.zeros((S, S, K), np.float32)
    maxc = C.c_int(0)
    lib.raster_naive_forward(v.ctypes.data_as(C.c_void_p), C.c_int(v.shape[0]), f.ctypes.data_as(C.c_void_p),
                             C.c_int(f.shape[0]), C.c_int(S), C.c_float(blur), C.c_int(K), C.c_float(sigma),
                             p2f.ctypes.data_as(C.c_void_p), zbuf.ctypes.data_as(C.c_void_p),
                             dists.ctypes.data_as(C.c_void_p), sil.ctypes.data_as(C.c_void_p), C.byref(maxc))
    return sil, p2f, zbuf, dists, maxc.value


def backward(v_ndc, faces, S, p2f, dists, grad_sil, K=100, sigma=SIGMA, unclamped_t=False):
    """-> d(sum grad_sil*sil)/d(v_ndc[:, :2])  (V,2) float64.  unclamped_t: the adjoint with the edge parameter left unclamped
    (smal_oracle.EDGE_T_UNCLAMPED, SMALFIT_OPT_UNCLAMPED_EDGE_T) instead of the exact gradient"""
    lib = _load()
    v = np.ascontiguousarray(v_ndc, np.float32)
    f = np.ascontiguousarray(faces, np.int32)
    g = np.ascontiguousarray(grad_sil, np.float32)
    gv = np.zeros((v.shape[0], 2), np.float64)
    lib.raster_naive_backward(v.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p), C.c_int(S), C.c_int(K),
                              C.c_float(sigma), np.ascontiguousarray(p2f, np.int32).ctypes.data_as(C.c_void_p),
                              np.ascontiguousarray(dists, np.float32).ctypes.data_as(C.c_void_p),
                              g.ctypes.data_as(C.c_void_p), gv.ctypes.data_as(C.c_void_p), C.c_int(1 if unclamped_t else 0))
    return gv
