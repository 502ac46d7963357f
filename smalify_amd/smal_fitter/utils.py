"""Counterpart of reference smal_fitter/utils.py (crop_to_silhouette) without cv2.

cv2.resize is restated for the two modes the reference uses on float images:
  INTER_NEAREST  source index = floor(dst_index * (1 / (dst_size / src_size)))  (OpenCV's legacy nearest, no half-pixel shift)
  INTER_LINEAR   source coordinate = (float32)((dst_index + 0.5) * (1 / (dst_size / src_size)) - 0.5), bilinear, border replicated,
                 float32 interpolation weights as in OpenCV's resize for CV_64F (opencv imgproc/src/resize.cpp: resizeNN, resize_)
cv2 is not installed here, so these restatements are **parity unpinned** (SURVEY §8f row 2)."""
from __future__ import annotations

import numpy as np


def resize_nearest(img, out_h, out_w):
    # OpenCV resizeNN: ifx = 1. / (dsize / ssize) in double (NOT ssize / dsize: the two differ in the last bit, which decides
    # floor() where dst_index * ssize / dsize is an integer), sx = min(cvFloor(x * ifx), ssize - 1)
    h, w = img.shape[:2]
    ify, ifx = 1.0 / (float(out_h) / float(h)), 1.0 / (float(out_w) / float(w))
    ys = np.minimum(np.floor(np.arange(out_h) * ify).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(out_w) * ifx).astype(np.int64), w - 1)
    return img[ys][:, xs]


def _linear_taps(out_n, in_n):
    # OpenCV resize(INTER_LINEAR) coordinate set-up: scale = 1. / (dsize / ssize) in double, the source coordinate is rounded to
    # float32 BEFORE it is split (fx = (float)((dx + 0.5) * scale - 0.5); sx = cvFloor(fx); fx -= sx), clamped at both borders
    # (fx = 0, sx = 0 / ssize - 1), and the two weights are the float32 values 1.f - fx and fx
    scale = 1.0 / (float(out_n) / float(in_n))
    f = ((np.arange(out_n) + 0.5) * scale - 0.5).astype(np.float32)
    i0 = np.floor(f).astype(np.int64)
    frac = (f - i0.astype(np.float32)).astype(np.float32)
    frac = np.where(i0 < 0, np.float32(0.0), frac)
    i0 = np.maximum(i0, 0)
    over = i0 >= in_n - 1
    frac = np.where(over, np.float32(0.0), frac)
    i0 = np.where(over, in_n - 1, i0)
    i1 = np.minimum(i0 + 1, in_n - 1)
    w1 = frac.astype(np.float32)
    w0 = (np.float32(1.0) - w1).astype(np.float32)
    return i0, i1, w0.astype(np.float64), w1.astype(np.float64)


def resize_linear(img, out_h, out_w):
    h, w = img.shape[:2]
    y0, y1, wy0, wy1 = _linear_taps(out_h, h)
    x0, x1, wx0, wx1 = _linear_taps(out_w, w)
    img = np.asarray(img, np.float64)
    extra = (1,) * (img.ndim - 2)
    rows = img[y0] * wy0.reshape((-1, 1) + extra) + img[y1] * wy1.reshape((-1, 1) + extra)
    return rows[:, x0] * wx0.reshape((1, -1) + extra) + rows[:, x1] * wx1.reshape((1, -1) + extra)


def crop_to_silhouette(sil_img, rgb_img, joints, target_size):
    """reference utils.py:5-36: pad 4x, square box 1.05x the silhouette's larger extent around its centre, nearest
    resize for the silhouette, bilinear for the image; joints (row, col) follow the crop and the scale."""
    assert sil_img.ndim == 2, "Silhouette image is not HxW"
    assert rgb_img.ndim == 3, "RGB image is not HxWx3"
    sil_h, sil_w = sil_img.shape
    pad_sil = np.zeros((sil_h * 4, sil_w * 4))
    pad_rgb = np.ones((sil_h * 4, sil_w * 4, 3))
    pad_sil[sil_h * 2: sil_h * 3, sil_w * 2: sil_w * 3] = sil_img
    pad_rgb[sil_h * 2: sil_h * 3, sil_w * 2: sil_w * 3, :] = rgb_img
    fg = np.where(pad_sil > 0)
    y_min, y_max, x_min, x_max = np.amin(fg[0]), np.amax(fg[0]), np.amin(fg[1]), np.amax(fg[1])
    half = int(1.05 * (max(x_max - x_min, y_max - y_min) / 2))
    centre_y = y_min + int((y_max - y_min) / 2)
    centre_x = x_min + int((x_max - x_min) / 2)
    square_sil = pad_sil[centre_y - half: centre_y + half, centre_x - half: centre_x + half]
    square_rgb = pad_rgb[centre_y - half: centre_y + half, centre_x - half: centre_x + half]
    sil_resize = resize_nearest(square_sil, target_size, target_size)
    rgb_resize = resize_linear(square_rgb, target_size, target_size)
    joints = np.asarray(joints, np.float64)
    scaled = np.zeros_like(joints)
    scaled[:, 0] = joints[:, 0] + (sil_h * 2) - (centre_y - half)
    scaled[:, 1] = joints[:, 1] + (sil_w * 2) - (centre_x - half)
    scaled = scaled * (target_size / (half * 2.0))
    return sil_resize, rgb_resize, scaled
