"""CPU tests of the visualisation helpers: the oracle's hard-Phong restatement and the cv2 / imageio /
trimesh-free exporters."""
import os
import struct
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _posed_mesh(n=1):
    from oracle import smal_oracle as so
    from smalify_amd import synthetic
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    om = so.OracleModel(md)
    sp = synthetic.synthetic_shape_prior()
    gt = synthetic.ground_truth_params(n, seed=7, mean_betas=sp[1][:20], mean_logscale=sp[1][20:26])
    gt["trans"][:, 2] += 1.0
    theta = np.concatenate([gt["global_rotation"][:, None], gt["joint_rotations"]], 1)
    with torch.no_grad():
        vo, _, _, _ = so.smal_forward(om, torch.from_numpy(np.tile(gt["betas"], (n, 1))).double(),
                                      torch.from_numpy(theta).double(),
                                      torch.from_numpy(np.tile(gt["log_beta_scales"], (n, 1))).double())
    return so, om, vo + torch.from_numpy(gt["trans"]).double()[:, None]


def test_oracle_hard_phong_invariants():
    so, om, verts = _posed_mesh()
    S = 48
    img = so.hard_phong_render(verts, om.faces, S, (0.0, 172 / 255.0, 223 / 255.0))
    assert img.shape == (1, 3, S, S)
    covered = (img < 1.0).any(axis=1)
    sil = so.soft_silhouette(verts, om.faces, S).numpy()
    # a pixel strictly inside some face is also a (strong) soft-silhouette pixel; the background is exactly white
    assert covered.mean() > 0.03 and (sil[covered] > 0.4).all()
    assert (img[:, :, ~covered[0]] == 1.0).all()
    # ambient 0.5 + diffuse <= 0.3 on a colour <= 1, specular <= 0.2
    assert img.min() >= 0.0 and img[:, :, covered[0]].max() <= 1.0 + 1e-9
    # the red channel of the mesh colour is 0: red is pure specular there, so never above 0.2
    assert img[0, 0][covered[0]].max() <= 0.2 + 1e-9
    again = so.hard_phong_render(verts, om.faces, S, (0.0, 172 / 255.0, 223 / 255.0))
    assert np.array_equal(img, again)


def test_vertex_normals_are_unit_and_outward_on_a_tetrahedron():
    from oracle import smal_oracle as so
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float64)
    f = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]])
    vn = so.vertex_normals(v, f)
    assert np.allclose(np.linalg.norm(vn, axis=1), 1.0)
    assert ((vn * (v - v.mean(0))).sum(1) > 0).all()


def test_png_and_ply_writers_round_trip(tmp_path):
    from smalify_amd.smal_fitter.optimize_to_joints import write_png, write_ply
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, size=(13, 29, 3)).astype(np.uint8)
    p = str(tmp_path / "a.png")
    write_png(p, img)
    blob = open(p, "rb").read()
    assert blob[:8] == b"\x89PNG\r\n\x1a\n" and blob[12:16] == b"IHDR"
    w, h, depth, ctype = struct.unpack(">IIBB", blob[16:26])
    assert (w, h, depth, ctype) == (29, 13, 8, 2)
    n = struct.unpack(">I", blob[33:37])[0]
    assert blob[37:41] == b"IDAT"
    rows = np.frombuffer(zlib.decompress(blob[41:41 + n]), np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:].reshape(h, w, 3), img)
    assert struct.unpack(">I", blob[41 + n:45 + n])[0] == zlib.crc32(blob[37:41 + n]) & 0xFFFFFFFF
    verts = rs.randn(5, 3).astype(np.float32)
    faces = np.array([[0, 1, 2], [2, 3, 4]])
    q = str(tmp_path / "m.ply")
    write_ply(q, verts, faces)
    data = open(q, "rb").read()
    head, body = data.split(b"end_header\n")
    assert b"element vertex 5" in head and b"element face 2" in head and b"binary_little_endian" in head
    assert np.array_equal(np.frombuffer(body[:60], "<f4").reshape(5, 3), verts)
    rec = np.frombuffer(body[60:], dtype=[("n", "u1"), ("i", "<i4", (3,))])
    assert (rec["n"] == 3).all() and np.array_equal(rec["i"], faces)


def test_generate_video_exporter_writes_numbered_frames(tmp_path):
    """reference generate_video.py:24-32: one flat directory, `NNNN.png` + `NNNN.pkl` per frame (ffmpeg's %04d pattern)"""
    import pickle
    from PIL import Image
    from smalify_amd.smal_fitter.generate_video import ImageExporter
    ex = ImageExporter(str(tmp_path / "exported" / "ckpt" / "st10_ep0"))
    rs = np.random.RandomState(0)
    for gid in (0, 7):
        collage = rs.rand(32, 160, 3).astype(np.float32)          # float in [0, 1]; uint8 collages pass through unchanged
        params = {"global_rotation": rs.randn(3).astype(np.float32), "betas": rs.randn(20).astype(np.float32)}
        ex.export(collage, gid % 4, gid, params, None, None)
        stem = tmp_path / "exported" / "ckpt" / "st10_ep0" / ("%04d" % gid)
        img = np.asarray(Image.open(str(stem) + ".png"))
        assert img.shape == (32, 160, 3)
        assert np.abs(img.astype(np.float32) / 255.0 - collage).max() <= 0.5 / 255.0 + 1e-6
        with open(str(stem) + ".pkl", "rb") as fh:
            back = pickle.load(fh)
        assert np.array_equal(back["betas"], params["betas"])
    u8 = (rs.rand(8, 8, 3) * 255).astype(np.uint8)
    ex.export(u8, 0, 9, {}, None, None)
    assert np.array_equal(np.asarray(Image.open(str(tmp_path / "exported" / "ckpt" / "st10_ep0" / "0009.png"))), u8)
