"""Drop-in for reference smal_fitter/generate_video.py: re-render a fitted sequence from its checkpoints.

Loads the dataset named by config.SEQUENCE_OR_IMAGE_NAME, builds a SMALFitter, reads the per-frame `.pkl` checkpoints of
`checkpoints/<CHECKPOINT_NAME>/<frame>/<EPOCH_NAME>.pkl` (SMALFitter.load_checkpoint, reference smal_fitter.py:192-207)
and writes one five-panel collage `NNNN.png` plus the frame's parameter dict `NNNN.pkl` per frame to
`exported/<CHECKPOINT_NAME>/<EPOCH_NAME>/` (reference generate_video.py:24-72).  Frames are numbered consecutively so
that `ffmpeg -framerate 50 -i %04d.png -pix_fmt yuv420p out.gif` works as the reference's header comment suggests.
As in the reference, load_checkpoint looks for `<frame index:04>/<EPOCH_NAME>.pkl` while the fit exporter names its
directories after the image stems (optimize_to_joints.py:37): the two agree only for frames named 0000.png, 0001.png …
(BADJA crops are).  No cv2 / imageio / PyTorch3D: the collage is drawn by SMALFitter.generate_visualization (HIP colour render + numpy)."""
from __future__ import annotations

import os
import pickle as pkl

from .. import config, model_io
from .optimize_to_joints import write_png


class ImageExporter:
    """generate_video.py:24-32: `<global_id:04>.png` + `.pkl` in one flat directory"""

    def __init__(self, output_dir):
        self.output_dir = output_dir
        os.makedirs(output_dir, exist_ok=True)

    def export(self, collage_np, batch_id, global_id, img_parameters, vertices, faces):
        write_png(os.path.join(self.output_dir, "{0:04}.png".format(global_id)), collage_np)
        with open(os.path.join(self.output_dir, "{0:04}.pkl".format(global_id)), "wb") as f:
            pkl.dump(img_parameters, f)


def render_checkpoints(data, checkpoint_dir, epoch_name, output_dir, shape_family, use_unity_prior, window_size,
                       model_data=None, pose_prior_data=None, shape_prior_data=None):
    """the body of the reference's main() for data already in memory; returns the fitter"""
    import torch
    from .smal_fitter import SMALFitter
    model = SMALFitter(torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu", data,
                       window_size, shape_family, use_unity_prior, model_data=model_data, pose_prior_data=pose_prior_data,
                       shape_prior_data=shape_prior_data)
    model.load_checkpoint(checkpoint_dir, epoch_name)
    model.generate_visualization(ImageExporter(output_dir))
    return model


def main():
    from .data_loader import load_badja_sequence, load_stanford_sequence
    output_dir = os.path.join("exported", config.CHECKPOINT_NAME, config.EPOCH_NAME)
    dataset, name = config.SEQUENCE_OR_IMAGE_NAME.split(":")
    if dataset == "badja":
        data, filenames = load_badja_sequence(config.BADJA_PATH, name, config.CROP_SIZE, image_range=config.IMAGE_RANGE)
    else:
        data, filenames = load_stanford_sequence(config.STANFORD_EXTRA_PATH, name, config.CROP_SIZE)
    print("Dataset size: {0}".format(len(filenames)))
    use_unity_prior = config.SHAPE_FAMILY == 1 and not config.FORCE_SMAL_PRIOR
    if not use_unity_prior and not config.ALLOW_LIMB_SCALING:
        print("WARNING: Limb scaling is only recommended for the new Unity prior.")
    model_data = model_io.load_smal_model(config.SMAL_FILE, config.SMAL_DATA_FILE, config.SMAL_SYM_FILE, config.SHAPE_FAMILY)
    return render_checkpoints(data, os.path.join("checkpoints", config.CHECKPOINT_NAME), config.EPOCH_NAME, output_dir,
                              config.SHAPE_FAMILY, use_unity_prior, config.WINDOW_SIZE, model_data=model_data)


if __name__ == "__main__":
    main()
