"""Counterpart of reference smal_fitter/optimize_to_joints.py: the stage / epoch driver.

`main()` loads the configured sequence (data_loader.py), runs the fused on-device loop (smalify_amd.fitter.FusedFitter)
with the reference's schedule and writes the same per-frame checkpoint files (st{stage}_ep{epoch}.pkl / .ply, final
st10_ep0).  `fit_sequence` takes the loader's output tuple `(rgb, sil, joints, visibility), filenames` directly."""
from __future__ import annotations

import os
import pickle as pkl

import numpy as np
import torch

from .. import config, engine as eng, fitter as fit, model_io


def write_ply(path, vertices, faces):
    """binary little-endian PLY (replaces trimesh.Trimesh(...).export, optimize_to_joints.py:51-53)"""
    v = np.ascontiguousarray(vertices, dtype="<f4")
    f = np.asarray(faces).astype("<i4")
    with open(path, "wb") as fh:
        fh.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                  "property float z\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n"
                  % (len(v), len(f))).encode())
        fh.write(v.tobytes())
        rec = np.empty(len(f), dtype=[("n", "u1"), ("i", "<i4", (3,))])
        rec["n"] = 3
        rec["i"] = f
        fh.write(rec.tobytes())


def write_png(path, image):
    """8-bit RGB PNG with nothing but zlib (replaces imageio.imsave, optimize_to_joints.py:45)"""
    import struct
    import zlib
    image = np.asarray(image)
    if np.issubdtype(image.dtype, np.floating):          # float images are taken as [0, 1]
        image = np.clip(image, 0.0, 1.0) * 255.0 + 0.5
    img = np.ascontiguousarray(image, dtype=np.uint8)
    h, w, c = img.shape
    assert c == 3
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, w * 3)], axis=1).tobytes()   # filter type 0 per row

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                 + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


class ImageExporter:
    """same directory layout and file names as the reference's exporter (optimize_to_joints.py:25-53)"""

    def __init__(self, output_dir, filenames):
        os.makedirs(output_dir, exist_ok=True)
        self.output_dirs = []
        for filename in filenames:
            d = os.path.join(output_dir, os.path.splitext(filename)[0])
            os.makedirs(d, exist_ok=True)
            self.output_dirs.append(d)
        self.stage_id = 0
        self.epoch_name = 0

    def export(self, collage_np, batch_id, global_id, img_parameters, vertices, faces):
        stem = os.path.join(self.output_dirs[global_id], "st{0}_ep{1}".format(self.stage_id, self.epoch_name))
        write_png(stem + ".png", collage_np)
        with open(stem + ".pkl", "wb") as f:
            pkl.dump(img_parameters, f)
        v = vertices[batch_id]
        write_ply(stem + ".ply", v.cpu().numpy() if isinstance(v, torch.Tensor) else v, faces)


def fit_sequence(data, filenames, model_data, pose_prior, shape_prior, use_unity_prior=True, output_dir=None,
                 window_size=None, opt_weights=None, iters_scale=1.0):
    """Runs the complete schedule on one sequence; returns the FusedFitter (parameters stay on the GPU)."""
    rgb, sil, joints, vis = data
    S = int(sil.shape[-1])
    dm = eng.DeviceModel(model_data)
    engine = eng.Engine(dm, int(joints.shape[0]), S)
    engine.set_pose_prior(*pose_prior)
    engine.set_shape_prior(*shape_prior)
    f = fit.FusedFitter(engine, joints, vis, sil, window_size or config.WINDOW_SIZE, use_unity_prior,
                        mean_betas=shape_prior[1][:20], mean_log_scales=shape_prior[1][20:26] if use_unity_prior else None,
                        allow_limb_scaling=config.ALLOW_LIMB_SCALING)
    exporter = ImageExporter(output_dir, filenames) if output_dir else None
    rgb_dev = torch.as_tensor(np.asarray(rgb), dtype=torch.float32, device=engine.device)
    rot_y180 = torch.tensor([[-1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0]], device=engine.device)
    colour = [c / 255.0 for c in config.MESH_COLOR]

    def export(fitter, stage_id, epoch_id):
        """the reference's ImageExporter output per frame (optimize_to_joints.py:43-53): st{stage}_ep{epoch}.png (the
        five-panel collage of smal_fitter.py:226-266), .pkl, .ply -- from a forward-only snapshot, the fit's losses untouched"""
        if exporter is None:
            return
        from .smal_fitter import SMALFitter
        verts, sil_r, proj = fitter.snapshot()
        rendered = engine.render_color(verts, colour)
        centre = verts.mean(dim=1, keepdim=True)
        back = ((verts - centre) @ rot_y180.T).contiguous()
        rev_rendered = engine.render_color(back, colour)
        # keypoints of the turned mesh: the canonical joints are vertices-independent here, so mark the front view's only
        overlay = rendered * 0.8 + rgb_dev * 0.2
        sil_err = (1.0 - (fitter.target_sil_float() - sil_r).abs()).unsqueeze(1).expand(-1, 3, -1, -1).cpu()   # smal_fitter.py:243, in [0, 1]
        vis_t = fitter.visibility_full
        collage = torch.cat([SMALFitter._draw_joints(rgb_dev, fitter.target_joints, vis_t),
                             SMALFitter._draw_joints(rendered, proj, vis_t), SMALFitter._draw_joints(overlay, proj, vis_t),
                             sil_err, rev_rendered.cpu()], dim=3)
        v_np = verts.cpu().numpy()
        for i, (d, params) in enumerate(zip(exporter.output_dirs, fitter.frame_parameters())):
            stem = os.path.join(d, "st{0}_ep{1}".format(stage_id, epoch_id))
            write_png(stem + ".png", (np.transpose(collage[i].numpy(), (1, 2, 0)) * 255.0).astype(np.uint8))
            with open(stem + ".pkl", "wb") as fh:
                pkl.dump(params, fh)
            write_ply(stem + ".ply", v_np[i], model_data.faces)

    f.run_schedule(opt_weights, iters_scale, on_visualize=export)
    export(f, 10, 0)                                   # final stage (optimize_to_joints.py:142-144)
    return f


def main():
    """reference optimize_to_joints.py:55-144: dataset from config.SEQUENCE_OR_IMAGE_NAME, model / priors from the config
    paths (data root: $SMALIFY_DATA), the full schedule, checkpoints under config.OUTPUT_DIR."""
    from .data_loader import load_badja_sequence, load_stanford_sequence
    os.makedirs(config.OUTPUT_DIR, exist_ok=True)
    dataset, name = config.SEQUENCE_OR_IMAGE_NAME.split(":")
    if dataset == "badja":
        data, filenames = load_badja_sequence(config.BADJA_PATH, name, config.CROP_SIZE, image_range=config.IMAGE_RANGE)
    else:
        data, filenames = load_stanford_sequence(config.STANFORD_EXTRA_PATH, name, config.CROP_SIZE)
    print("Dataset size: {0}".format(len(filenames)))
    assert config.SHAPE_FAMILY >= 0, "Shape family should be greater than 0"
    use_unity_prior = config.SHAPE_FAMILY == 1 and not config.FORCE_SMAL_PRIOR
    model_data = model_io.load_smal_model(config.SMAL_FILE, config.SMAL_DATA_FILE, config.SMAL_SYM_FILE, config.SHAPE_FAMILY)
    pose_prior = model_io.load_pose_prior(config.WALKING_PRIOR_FILE)
    shape_prior = (model_io.unity_shape_prior(config.UNITY_SHAPE_PRIOR) if use_unity_prior
                   else model_io.family_shape_prior(model_io.load_pickle(config.SMAL_DATA_FILE), config.SHAPE_FAMILY))
    return fit_sequence(data, filenames, model_data, pose_prior, shape_prior, use_unity_prior=use_unity_prior,
                        output_dir=config.OUTPUT_DIR)


if __name__ == "__main__":
    main()
