"""Counterpart of reference smal_fitter/data_loader.py without cv2 / imageio / pycocotools.

    load_badja_sequence(BADJA_PATH, sequence_name, crop_size, image_range=None) -> (rgb, sil, joints, visibility), file_names
    load_stanford_sequence(STANFORD_EXTRA, image_name, crop_size)               -> (rgb, sil, joints, visibility), file_names

Same outputs as the reference (data_loader.py:21-127): rgb float32 (N,3,S,S) in [0,1], sil float32 (N,1,S,S),
joints float32 (N,25,2) as (row, col) in the crop, visibility float32 (N,25).  Images are read with PIL (what imageio
uses underneath), the COCO run-length masks of StanfordExtra are decoded here, resizing follows cv2's conventions (see
utils.py).  None of the original packages is installed in this environment: **parity unpinned**."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from .. import config
from .utils import crop_to_silhouette, resize_linear


def _imread(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)


def decode_rle(counts, height, width):
    """pycocotools.mask.decode for one compressed RLE string (rleFrString + rleDecode): 6-bit groups offset by 48, bit 5 =
    continuation, bit 4 of the last group = sign, runs after the second are deltas to the run two back; runs alternate
    0 / 1 starting with 0 in column-major order."""
    if isinstance(counts, str):
        counts = counts.encode("ascii")
    runs = []
    p = 0
    while p < len(counts):
        x, k, more = 0, 0, True
        while more:
            c = counts[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(runs) > 2:
            x += runs[-2]
        runs.append(x)
    flat = np.zeros(height * width, np.uint8)
    pos, val = 0, 0
    for r in runs:
        if val:
            flat[pos:pos + r] = 1
        pos += r
        val ^= 1
    return flat.reshape((width, height)).T.copy()           # column-major fill


def encode_rle(mask):
    """inverse of decode_rle (pycocotools rleEncode + rleToString); used by the tests and for writing fixtures"""
    flat = np.asarray(mask, np.uint8).T.reshape(-1)
    change = np.flatnonzero(np.diff(flat)) + 1
    edges = np.concatenate([[0], change, [len(flat)]])
    runs = list(np.diff(edges))
    if flat[0] == 1:
        runs = [0] + runs
    out = bytearray()
    for i, x in enumerate(runs):
        x = int(x)
        if i > 2:
            x -= int(runs[i - 2])
        more = True
        while more:
            c = x & 0x1f
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(c + 48)
    return out.decode("ascii")


def load_badja_sequence(BADJA_PATH, sequence_name, crop_size, image_range=None):
    file_names, rgb_imgs, sil_imgs, joints, visibility = [], [], [], [], []
    json_path = os.path.join(BADJA_PATH, "joint_annotations", "{0}.json".format(sequence_name))
    with open(json_path) as fh:
        annotation = np.array(json.load(fh))
    if image_range is not None:
        annotation = annotation[image_range]
    for image_annotation in annotation:
        file_name = os.path.join(BADJA_PATH, image_annotation["image_path"])
        seg_name = os.path.join(BADJA_PATH, image_annotation["segmentation_path"])
        if os.path.exists(file_name) and os.path.exists(seg_name):
            landmarks = np.array(image_annotation["joints"])[config.BADJA_ANNOTATED_CLASSES]
            visibility.append(np.array(image_annotation["visibility"])[config.BADJA_ANNOTATED_CLASSES])
            rgb_img = _imread(file_name) / 255.0
            seg = _imread(seg_name)
            sil_img = (seg[:, :, 0] if seg.ndim == 3 else seg) / 255.0
            rgb_h, rgb_w, _ = rgb_img.shape
            # the reference passes cv2.INTER_NEAREST in the `dst` position of cv2.resize (data_loader.py:48), so the
            # interpolation actually used is the default, bilinear
            sil_img = resize_linear(sil_img, rgb_h, rgb_w)
            sil_img, rgb_img, landmarks = crop_to_silhouette(sil_img, rgb_img, landmarks, crop_size)
            rgb_imgs.append(rgb_img)
            sil_imgs.append(sil_img)
            joints.append(landmarks)
            file_names.append(os.path.basename(image_annotation["image_path"]))
        elif os.path.exists(file_name):
            print("BADJA SEGMENTATION file path: {0} is missing".format(seg_name))
        else:
            print("BADJA IMAGE file path: {0} is missing".format(file_name))
    rgb = torch.from_numpy(np.stack(rgb_imgs, 0)).float().permute(0, 3, 1, 2)
    sil = torch.from_numpy(np.stack(sil_imgs, 0)).float()[:, None, :, :]
    joints_t = torch.from_numpy(np.stack(joints, 0)).float()
    vis = torch.from_numpy(np.stack(visibility, 0).astype(np.float64)).float()
    vis[:, np.array(config.BADJA_ANNOTATED_CLASSES) == -1] = 0.0       # unlabelled classes are invisible
    return (rgb, sil, joints_t, vis), file_names


def load_stanford_sequence(STANFORD_EXTRA, image_name, crop_size):
    img_dir = os.path.join(STANFORD_EXTRA, "sample_imgs")
    with open(os.path.join(STANFORD_EXTRA, "StanfordExtra_sample.json")) as fh:
        entries = {i["img_path"]: i for i in json.load(fh)}
    data = entries[image_name]
    img = _imread(os.path.join(img_dir, data["img_path"]))
    seg = decode_rle(data["seg"], data["img_height"], data["img_width"])
    # tail_mid was not annotated in StanfordExtra: an extra invisible joint (data_loader.py:111-113)
    raw_joints = np.concatenate([np.array(data["joints"], np.float64), [[0.0, 0.0, 0.0]]], axis=0)
    sil_img, rgb_img, landmarks = crop_to_silhouette(seg, img / 255.0, raw_joints[:, [1, 0]], crop_size)
    rgb = torch.from_numpy(rgb_img).float()[None].permute(0, 3, 1, 2)
    sil = torch.from_numpy(np.asarray(sil_img, np.float64)).float()[None, None]
    joints = torch.from_numpy(landmarks).float()[:, :2].unsqueeze(0)
    visibility = torch.from_numpy(raw_joints).float()[:, -1].unsqueeze(0)
    return (rgb, sil, joints, visibility), [os.path.basename(data["img_path"])]
