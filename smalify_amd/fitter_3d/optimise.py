"""Drop-in for reference fitter_3d/optimise.py: fit SMAL to a directory of .obj meshes, stages from YAML or arguments.

    python -m smalify_amd.fitter_3d.optimise --mesh_dir <dir> [--yaml_src cfg.yaml] [--scheme default --lr 1e-3 --nits 100]

Same arguments and YAML layout (`stages: {name: {scheme, nits, lr, loss_weights, custom_lrs}}`, `args: {...}` overriding
the command line) as fitter_3d/optimise.py:18-90 and fitter_3d/example_cfg.yaml."""
from __future__ import annotations

import argparse
import os

from .trainer import SMAL3DFitter, SMALParamGroup, Stage, StageManager
from .utils import load_meshes


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--results_dir", type=str, default="fit3d_results", help="Directory in which results are stored")
    parser.add_argument("--mesh_dir", type=str, default="fitter_3d/example_meshes", help="Directory in which meshes are stored")
    parser.add_argument("--frame_step", type=int, default=1,
                        help="If directory is a sequence of animated frames, only take every nth frame")
    parser.add_argument("--shape_family_id", type=int, default=-1,
                        help="Shape family to use for optimisation (-1 to use default SMAL mesh)")
    parser.add_argument("--yaml_src", type=str, default=None, help="YAML source for experimental set-up")
    parser.add_argument("--scheme", type=str, default="default", choices=list(SMALParamGroup.param_map.keys()),
                        help="Optimisation scheme")
    parser.add_argument("--lr", type=float, default=1e-3)
    parser.add_argument("--nits", type=int, default=100)
    parser.add_argument("--seed", type=int, default=0, help="seed of the target-point sampler")
    parser.add_argument("--no_plots", action="store_true", help="skip the per-stage mesh figures")
    return parser


def main(args, model_data=None, smal_data=None):
    stage_options = None
    if args.yaml_src is not None:
        import yaml
        try:
            with open(args.yaml_src) as infile:
                yaml_cfg = yaml.load(infile, Loader=yaml.FullLoader)
        except FileNotFoundError:
            raise FileNotFoundError(f"No YAML file found at {args.yaml_src}.")
        stage_options = yaml_cfg["stages"]
        for arg, val in (yaml_cfg.get("args") or {}).items():       # YAML args overwrite the command line
            setattr(args, arg, val)

    mesh_names, target_meshes = load_meshes(mesh_dir=args.mesh_dir, frame_step=args.frame_step)
    n_batch = len(target_meshes)
    os.makedirs(args.results_dir, exist_ok=True)
    manager = StageManager(out_dir=args.results_dir, labels=mesh_names)
    smal_model = SMAL3DFitter(batch_size=n_batch, shape_family=args.shape_family_id, model_data=model_data,
                              smal_data=smal_data)
    stage_kwargs = dict(target_meshes=target_meshes, smal_3d_fitter=smal_model, out_dir=args.results_dir,
                        mesh_names=mesh_names, seed=getattr(args, "seed", 0))
    if stage_options is not None:
        for stage_name, kwargs in stage_options.items():
            manager.add_stage(Stage(name=stage_name, **kwargs, **stage_kwargs))
    else:
        print("No YAML provided. Loading from system args. ")
        manager.add_stage(Stage(scheme=args.scheme, nits=args.nits, lr=args.lr, **stage_kwargs))
    manager.run(plot=not getattr(args, "no_plots", False))
    return manager


if __name__ == "__main__":
    main(build_parser().parse_args())
