#!/usr/bin/env python3
"""GUARDED generator of tests/golden/reference_golden_loaders.npz: vectors for the data-loader restatements (SURVEY section 8f
row 2) produced by the packages the reference itself calls -- cv2 and pycocotools -- neither of which is installed in the
build container or on the GPU box (no network).  Anyone who has them pins smalify_amd/smal_fitter/{utils,data_loader}.py with

    python tests/golden/make_golden_loaders.py        (needs `import cv2` and `from pycocotools.mask import decode`)

What it stores, for a handful of seeded synthetic inputs (arrays only; sizes chosen so that the scale factors are not
representable exactly and several destination pixels fall on source-pixel borders):

  resize   cv2.resize(img, (w, h), interpolation=cv2.INTER_NEAREST)   on float64 masks -- crop_to_silhouette's silhouette resize
           (reference smal_fitter/utils.py:27)
           cv2.resize(img, (w, h))                                     on float64 HxWx3 images -- its image resize (utils.py:28)
           cv2.resize(img, (w, h), cv2.INTER_NEAREST)                  the BADJA loader's call with the flag in the `dst` slot
           (data_loader.py:48): bilinear, whatever the flag says -- the quirk the restatement keeps
  crop     the reference's OWN crop_to_silhouette (imported from $SMALIFY_REFERENCE or /root/reference when its `utils` module is
           importable; otherwise the same statements executed here on cv2) on binary and soft masks, odd sizes, an animal touching
           the image border: silhouette, image and scaled joints (utils.py:5-36)
  rle      pycocotools.mask.decode of compressed COCO run-length strings made by pycocotools.mask.encode from random blobs, incl.
           an empty mask, a full mask and a mask starting with foreground (data_loader.py:88-97)

tests/test_data_loader_cpu.py consumes the file when it exists and reports the tests as skipped otherwise.  Without cv2 or
pycocotools this script exits with a message and writes nothing.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
OUT = os.path.join(HERE, "reference_golden_loaders.npz")


def reference_crop(cv2):
    """the reference's crop_to_silhouette if its module imports here (it also imports nibabel), else None"""
    ref = os.environ.get("SMALIFY_REFERENCE", "/root/reference")
    path = os.path.join(ref, "smal_fitter")
    if not os.path.isdir(path):
        return None
    sys.path.insert(0, path)
    try:
        import utils as ref_utils                      # reference smal_fitter/utils.py
        return ref_utils.crop_to_silhouette
    except Exception as exc:
        print("reference utils.py not importable (%s): running its crop statements on cv2 directly" % exc)
        return None
    finally:
        sys.path.pop(0)


def blob(rs, h, w, soft=False):
    """a random connected-looking foreground: union of a few discs; `soft`: values in (0, 1] like a resampled mask"""
    yy, xx = np.mgrid[0:h, 0:w]
    m = np.zeros((h, w))
    for _ in range(4):
        cy, cx, r = rs.uniform(0.25, 0.75) * h, rs.uniform(0.25, 0.75) * w, rs.uniform(0.08, 0.22) * min(h, w)
        m = np.maximum(m, ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r).astype(np.float64))
    if soft:
        m = m * rs.uniform(0.2, 1.0, size=m.shape)
    return m


def main():
    try:
        import cv2
        from pycocotools import mask as coco
    except ImportError as exc:
        sys.exit("cv2 / pycocotools are not importable here (%s): nothing written.  Run this where the reference's own "
                 "requirements are installed." % exc)
    rs = np.random.RandomState(20240917)
    out = {"cv2_version": np.array(cv2.__version__)}

    # ---- resize
    cases = [(37, 53, 64, 64), (60, 47, 128, 128), (64, 64, 48, 80), (101, 80, 96, 96), (17, 23, 5, 7)]
    out["resize_cases"] = np.array(cases)
    for i, (h, w, oh, ow) in enumerate(cases):
        mask = blob(rs, h, w, soft=(i % 2 == 1))
        img = np.round(rs.uniform(0.0, 1.0, size=(h, w, 3)) * 255.0) / 255.0      # 8-bit images / 255, as the loaders read them
        out["resize%d_mask" % i], out["resize%d_img" % i] = mask, img
        out["resize%d_mask_nearest" % i] = cv2.resize(mask, (ow, oh), interpolation=cv2.INTER_NEAREST)
        out["resize%d_img_linear" % i] = cv2.resize(img, (ow, oh))
        out["resize%d_mask_flag_in_dst_slot" % i] = cv2.resize(mask, (ow, oh), cv2.INTER_NEAREST)      # data_loader.py:48

    # ---- crop_to_silhouette
    crop = reference_crop(cv2)
    out["crop_source"] = np.array("reference smal_fitter/utils.py::crop_to_silhouette" if crop else "utils.py:5-36 restated on cv2")
    if crop is None:
        def crop(sil_img, rgb_img, joints, target_size):
            sil_h, sil_w = sil_img.shape
            pad_sil = np.zeros((sil_h * 4, sil_w * 4))
            pad_rgb = np.ones((sil_h * 4, sil_w * 4, 3))
            pad_sil[sil_h * 2: sil_h * 3, sil_w * 2: sil_w * 3] = sil_img
            pad_rgb[sil_h * 2: sil_h * 3, sil_w * 2: sil_w * 3, :] = rgb_img
            fg = np.where(pad_sil > 0)
            y_min, y_max, x_min, x_max = np.amin(fg[0]), np.amax(fg[0]), np.amin(fg[1]), np.amax(fg[1])
            half = int(1.05 * (max(x_max - x_min, y_max - y_min) / 2))
            cy, cx = y_min + int((y_max - y_min) / 2), x_min + int((x_max - x_min) / 2)
            sq_sil = pad_sil[cy - half: cy + half, cx - half: cx + half]
            sq_rgb = pad_rgb[cy - half: cy + half, cx - half: cx + half]
            s = cv2.resize(sq_sil, (target_size, target_size), interpolation=cv2.INTER_NEAREST)
            r = cv2.resize(sq_rgb, (target_size, target_size))
            sj = np.zeros_like(joints)
            sj[:, 0] = joints[:, 0] + (sil_h * 2) - (cy - half)
            sj[:, 1] = joints[:, 1] + (sil_w * 2) - (cx - half)
            return s, r, sj * (target_size / (half * 2.0))
    crops = [(45, 65, 128, False), (38, 31, 64, True), (100, 100, 96, False), (33, 47, 64, False)]
    out["crop_cases"] = np.array([(h, w, t, int(s)) for h, w, t, s in crops])
    for i, (h, w, target, soft) in enumerate(crops):
        mask = blob(rs, h, w, soft=soft)
        if i == 3:
            mask[:, :3] = 1.0                          # foreground touching the image border: the 4x padding is what saves the crop
        img = np.round(rs.uniform(0.0, 1.0, size=(h, w, 3)) * 255.0) / 255.0
        joints = np.stack([rs.uniform(0, h, 25), rs.uniform(0, w, 25)], 1)
        s, r, j = crop(mask, img, joints.copy(), target)
        out["crop%d_mask" % i], out["crop%d_img" % i], out["crop%d_joints" % i] = mask, img, joints
        out["crop%d_sil_out" % i], out["crop%d_img_out" % i], out["crop%d_joints_out" % i] = np.asarray(s), np.asarray(r), np.asarray(j)

    # ---- RLE
    rles = []
    for i, (h, w) in enumerate([(40, 60), (123, 77), (64, 64), (31, 45), (50, 50), (9, 200)]):
        m = (blob(rs, h, w) > 0).astype(np.uint8)
        if i == 2:
            m[:] = 0
        if i == 3:
            m[:] = 1
        if i == 4:
            m[0, 0] = 1                                # column-major run list starts with an empty background run
        enc = coco.encode(np.asfortranarray(m))
        counts = enc["counts"].decode("ascii") if isinstance(enc["counts"], bytes) else enc["counts"]
        dec = coco.decode({"size": [h, w], "counts": counts})
        assert dec.shape == (h, w) and (dec == m).all()
        out["rle%d_counts" % i], out["rle%d_size" % i], out["rle%d_mask" % i] = np.array(counts), np.array([h, w]), dec.astype(np.uint8)
        rles.append(i)
    out["rle_count"] = np.array(len(rles))
    np.savez_compressed(OUT, **out)
    print("wrote %s (%d arrays, %.0f KB); cv2 %s; crop: %s" % (OUT, len(out), os.path.getsize(OUT) / 1024.0, cv2.__version__, out["crop_source"]))


if __name__ == "__main__":
    main()
