"""Drop-in for the reference's fitter_3d package (SMAL fitted to 3D target meshes): same classes, stage schemes,
YAML layout and .npz output as fitter_3d/{trainer,utils,optimise}.py, with the objective, its gradient, the point
sampler and Adam running as HIP kernels behind the C-ABI (smalfit_mesh_objective_* / smalfit_mesh_targets_*)."""
from .trainer import SMAL3DFitter, SMALParamGroup, Stage, StageManager, default_weights  # noqa: F401
from .utils import TargetMeshes, load_meshes, load_obj  # noqa: F401
