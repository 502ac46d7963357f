"""Process-wide registry of device models and engines used by the drop-in classes.

The reference's Renderer is constructed without a model (`Renderer(image_size, device)`,
p3d_renderer.py:18) and receives faces per call; the HIP rasteriser needs the topology when its
workspace is created, so the most recently constructed SMAL registers itself here and Renderer /
Prior look it up lazily.

Concurrency contract: an engine is a single workspace -- every call on it enqueues kernels that read and write the same
buffers, so two callers must not use one engine at the same time from different host threads or on different streams
(handles are thread-compatible, not thread-safe, include/smalfit.h).  The registry itself is guarded by a lock (creation
and growth of engines is atomic); callers that fit from several threads give each thread its own SMAL / SMALFitter
(each SMAL owns a DeviceModel, engines are keyed by it) and keep each one on one stream."""
from __future__ import annotations

import threading

from . import engine as eng

_current_model = None
_engines = {}
_lock = threading.RLock()


def set_current_model(device_model):
    global _current_model
    _current_model = device_model


def current_model():
    if _current_model is None:
        raise eng.SmalfitError("no SMAL model has been created yet (construct smalify_amd SMAL first)")
    return _current_model


def get_engine(device_model=None, max_frames=16, image_size=16):
    """engine with capacity >= (max_frames, exactly image_size); grows by re-creating"""
    dm = device_model or current_model()
    key = (id(dm), int(image_size))
    with _lock:
        return _get_engine_locked(dm, key, max_frames, image_size)


def _get_engine_locked(dm, key, max_frames, image_size):
    e = _engines.get(key)
    if e is None or e.max_frames < max_frames:
        cap = max(int(max_frames), 16 if e is None else 2 * e.max_frames)
        new = eng.Engine(dm, cap, int(image_size))
        if e is not None:
            for attr in ("_pose_prior", "_shape_prior"):
                if hasattr(e, attr):
                    setattr(new, attr, getattr(e, attr))
            if hasattr(e, "_pose_prior"):
                new.set_pose_prior(*e._pose_prior)
            if hasattr(e, "_shape_prior"):
                new.set_shape_prior(*e._shape_prior)
        _engines[key] = e = new
    return e
