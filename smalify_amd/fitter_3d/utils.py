"""Drop-in for reference fitter_3d/utils.py: .obj loading with the reference's normalisation, the target-mesh container
and the matplotlib figures -- without PyTorch3D (load_obj / Meshes)."""
from __future__ import annotations

import os

import numpy as np

from .. import engine as eng


def try_mkdir(loc):
    if not os.path.isdir(loc):
        os.mkdir(loc)


def try_mkdirs(locs):
    for loc in locs:
        try_mkdir(loc)


def load_obj(path):
    """Wavefront .obj -> (verts (V,3) float32, faces (F,3) int64).  What pytorch3d.io.load_obj(load_textures=False)
    returns as `verts` and `faces.verts_idx` (fitter_3d/utils.py:232-235): `v x y z [w]` lines, `f` lines with
    v, v/vt, v//vn or v/vt/vn corners, 1-based or negative (relative) indices, polygons fan-triangulated (0, i+1, i+2)."""
    verts, faces = [], []
    with open(path, "r", errors="replace") as fh:
        for ln, line in enumerate(fh, 1):
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                if len(tok) < 4:
                    raise ValueError("%s:%d: vertex with fewer than 3 coordinates" % (path, ln))
                verts.append([float(tok[1]), float(tok[2]), float(tok[3])])
            elif tok[0] == "f":
                corner = []
                for t in tok[1:]:
                    i = int(t.split("/")[0])
                    corner.append(i - 1 if i > 0 else len(verts) + i)
                if len(corner) < 3:
                    raise ValueError("%s:%d: face with fewer than 3 corners" % (path, ln))
                for k in range(len(corner) - 2):
                    faces.append([corner[0], corner[k + 1], corner[k + 2]])
    v = np.asarray(verts, np.float32).reshape(-1, 3)
    f = np.asarray(faces, np.int64).reshape(-1, 3)
    if len(v) == 0 or len(f) == 0:
        raise ValueError("%s: no geometry" % path)
    if f.min() < 0 or f.max() >= len(v):
        raise ValueError("%s: face index out of range" % path)
    return v, f


def normalise_verts(verts):
    """centre on the vertex mean and scale by the largest absolute coordinate (fitter_3d/utils.py:237-241)"""
    v = np.asarray(verts, np.float32)
    v = v - v.mean(0)
    return v / np.abs(v).max(0).max()


class TargetMeshes:
    """The target meshes of a fit: host copies for plotting/saving + the device-resident sampler
    (the role pytorch3d.structures.Meshes plays in fitter_3d/utils.py:253 and trainer.py:209)."""

    def __init__(self, verts_list, faces_list):
        if len(verts_list) == 0 or len(verts_list) != len(faces_list):
            raise ValueError("need one (verts, faces) pair per target mesh")
        self.verts = [np.ascontiguousarray(np.asarray(v), np.float32).reshape(-1, 3) for v in verts_list]
        self.faces = [np.ascontiguousarray(np.asarray(f), np.int32).reshape(-1, 3) for f in faces_list]
        self._device_targets = None

    @property
    def _dev(self):
        """the device-resident copy + sampler, uploaded on first use (loading and plotting need no GPU)"""
        if self._device_targets is None:
            self._device_targets = eng.MeshTargets(self.verts, self.faces)
        return self._device_targets

    def __len__(self):
        return len(self.verts)

    def __getitem__(self, n):
        return self.verts[n], self.faces[n]

    def verts_list(self):
        return self.verts

    def faces_list(self):
        return self.faces

    def sample(self, num_points, seed, iteration, out=None):
        return self._dev.sample(num_points, seed, iteration, out=out)


def load_meshes(mesh_dir: str, sorting=lambda arr: arr, n_meshes=None, frame_step=1, device="cuda:0"):
    """Given a dir of .obj files, loads all and returns (mesh_names, target_meshes) like fitter_3d/utils.py:204-255.
    NOTE as in the reference, the default `sorting` keeps os.listdir's order."""
    file_list = [f for f in os.listdir(mesh_dir) if ".obj" in f]
    obj_list = sorting(file_list)[::frame_step]
    if n_meshes is not None:
        obj_list = obj_list[:n_meshes]
    if not obj_list:
        raise FileNotFoundError("no .obj files in %s" % mesh_dir)
    mesh_names, all_verts, all_faces = [], [], []
    for obj_file in obj_list:
        mesh_names.append(obj_file[:-4])
        v, f = load_obj(os.path.join(mesh_dir, obj_file))
        all_verts.append(normalise_verts(v))
        all_faces.append(f)
    print(f"{len(all_verts)} target meshes loaded.")
    return mesh_names, TargetMeshes(all_verts, all_faces)


# ---- figures (matplotlib; host side, outside the hot path) --------------------------------------------------------
def equal_3d_axes(ax, X, Y, Z, zoom=1.0):
    """same length scale on all three axes (fitter_3d/utils.py:20-33)"""
    max_range = np.array([X.max() - X.min(), Y.max() - Y.min(), Z.max() - Z.min()]).max() / (2.0 * zoom)
    mid = [(a.max() + a.min()) * 0.5 for a in (X, Y, Z)]
    ax.set_xlim(mid[0] - max_range, mid[0] + max_range)
    ax.set_ylim(mid[1] - max_range, mid[1] + max_range)
    ax.set_zlim(mid[2] - max_range, mid[2] + max_range)


def plot_mesh(ax, verts, faces, label="", colour="blue", equalize=True, zoom=1.5, alpha=1.0):
    X, Y, Z = np.asarray(verts).T
    surf = ax.plot_trisurf(X, Y, Z, triangles=np.asarray(faces), alpha=alpha, color=colour, shade=True)
    if equalize:
        equal_3d_axes(ax, X, Y, Z, zoom=zoom)
    ax.plot([], [], color=colour, label=label)
    return surf


def plot_meshes(target_meshes, src_verts, src_faces, mesh_names=[], title="", figtitle="",
                out_dir="static_fits_output/pointclouds"):
    """one figure per mesh with three panels: target, SMAL, both (fitter_3d/utils.py:75-112)"""
    import matplotlib
    matplotlib.use("Agg")
    from matplotlib import pyplot as plt
    os.makedirs(out_dir, exist_ok=True)
    for n in range(len(target_meshes)):
        fig = plt.figure(figsize=(15, 5))
        axes = [fig.add_subplot(1, 3, k, projection="3d") for k in range(1, 4)]
        for ax in axes:
            ax.set_xlabel("x")
            ax.set_ylabel("y")
            ax.set_zlabel("z")
        tv, tf = target_meshes[n]
        for i, (v, f, colour, label) in enumerate(((tv, tf, "green", "target"), (src_verts[n], src_faces, "blue", "SMAL"))):
            for j, ax in enumerate((axes[i], axes[2])):
                plot_mesh(ax, v, f, colour=colour, label=label, alpha=[1, 0.5][j])
        fig.suptitle(figtitle)
        for ax in axes:
            ax.legend()
        name = mesh_names[n] if mesh_names else n
        plt.savefig(f"{out_dir}/{name} - {title}.png")
        plt.close(fig)
