"""Real-data ingestion on the CPU: the HMR-track class against a fixture captured from the
reference's own ImgSmpl, the C3D reader / writer against the published layout, the AVI frame rate, and the licensed-model
pickle loader against a file written in smplx's SMPL_NEUTRAL.pkl layout."""
import os
import pickle
import struct

import numpy as np
import pytest
import torch

from uuo_mocap_amd import ingest


def _data_from_fixture(g):
    F = g["rot"].shape[0]
    missing = set(int(v) for v in g["missing"])
    data = {}
    for f in range(F):
        key = "frame_%05d.jpg" % f
        if f in missing:
            data[key] = {"tracked_ids": [], "smpl": [], "3d_joints": [], "camera_bbox": [], "center": [], "scale": [],
                         "size": [], "2d_joints": []}
        else:
            data[key] = {"tracked_ids": [1], "smpl": [{"global_orient": g["rot"][f, :1], "body_pose": g["rot"][f, 1:],
                                                        "betas": g["betas"][f]}],
                         "3d_joints": [g["j3d"][f]], "camera_bbox": [g["cam"][f]], "center": [g["center"][f]],
                         "scale": [g["scale"][f]], "size": [g["size"][f]], "2d_joints": [g["j2d"][f].reshape(-1)]}
    return data


def test_img_smpl_matches_the_reference_class(golden):
    g = golden("ingest_img_smpl.npz")
    img = ingest.ImgSmpl(_data_from_fixture(g), float(g["freq"]))
    np.testing.assert_array_equal(img.img_mask.numpy(), g["out_img_mask"])
    for k in ("camera_bbox", "center", "scale", "size", "foot_contacts"):
        np.testing.assert_array_equal(getattr(img, k).numpy(), g["out_" + k], err_msg=k)
    # detected frames are copied, gaps interpolated: lerp exactly the reference's arithmetic, slerp to rounding
    np.testing.assert_allclose(img.trans.numpy(), g["out_trans"], atol=1e-6)
    np.testing.assert_allclose(img.betas.numpy(), g["out_betas"], atol=1e-6)
    for k in ("root_orient", "hmr_root_orient", "pose_body"):
        np.testing.assert_allclose(getattr(img, k).numpy(), g["out_" + k], atol=2e-6, err_msg=k)
    valid = g["out_img_mask"]
    np.testing.assert_array_equal(img.pose_body.numpy()[valid], g["rot"][valid][:, 1:])
    # attribute contract of the orchestrator (multimodal.py:88-100)
    F = valid.shape[0]
    assert img.trans.shape == (F, 3) and img.root_orient.shape == (F, 1, 3, 3) and img.pose_body.shape == (F, 23, 3, 3)
    assert img.betas.shape == (F, 10) and img.foot_contacts.shape == (F, 2) and img.img_mask.dtype == torch.bool
    sm = img.get_smpl()
    assert sm["poses"].shape == (F, 72) and sm["betas"].shape == (10,)


def test_img_smpl_edge_cases():
    g = {"tracked_ids": [1], "smpl": [{"global_orient": np.eye(3, dtype=np.float32)[None],
                                      "body_pose": np.tile(np.eye(3, dtype=np.float32), (23, 1, 1)),
                                      "betas": np.arange(10, dtype=np.float32)}],
         "3d_joints": [np.ones((45, 3), np.float32)], "camera_bbox": [np.zeros(3)], "center": [np.zeros(2)],
         "scale": [1.0], "size": [np.array([4.0, 5.0])], "2d_joints": [np.zeros(90)]}
    empty = {"tracked_ids": [], "smpl": [], "3d_joints": [], "camera_bbox": [], "center": [], "scale": [], "size": [],
             "2d_joints": []}
    # a single detection: every frame takes it; no detection at all: zeros, mask all False
    one = ingest.ImgSmpl({"a": empty, "b": g, "c": empty}, 25.0)
    assert one.img_mask.tolist() == [False, True, False]
    assert torch.equal(one.betas[0], one.betas[1]) and torch.equal(one.betas[2], one.betas[1])
    none = ingest.ImgSmpl({"a": empty, "b": empty}, 25.0)
    assert not none.img_mask.any() and float(none.trans.abs().sum()) == 0.0


@pytest.mark.parametrize("units,factor", [("mm", 1000.0), ("cm", 100.0), ("m", 1.0)])
def test_c3d_round_trip_and_markers_semantics(tmp_path, units, factor):
    rng = np.random.default_rng(0)
    F, M = 37, 41
    pts_m = rng.normal(size=(F, M, 3))
    pts_m[3, 5] = np.nan
    pts_m[10:14, 0] = np.nan
    labels = ["LFHD", "RFHD"] + ["MK%02d" % i for i in range(M - 2)]
    fn = str(tmp_path / "seq.c3d")
    ingest.write_c3d(fn, pts_m * factor, rate=120.0, units=units, labels=labels)
    raw = open(fn, "rb").read()
    assert raw[1] == 0x50 and len(raw) % 512 == 0
    c = ingest.read_c3d(fn)
    assert c["rate"] == 120.0 and c["units"] == units and c["labels"] == labels
    mk = ingest.Markers(fn)
    assert mk.get_frequency() == 120 and mk.get_num_markers() == M and len(mk) == F
    got = mk.get_points()
    assert got.shape == (F, M, 3)
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(pts_m), rtol=1e-6, atol=1e-6)
    assert np.isnan(got[3, 5]).all() and np.isnan(got[10:14, 0]).all() and not np.isnan(got[0]).any()
    # the runner's clean-up as the reference calls it: a no-op on real data
    z = np.nan_to_num(got, nan=0.0)
    assert ingest.cleanup_markers(z).shape == z.shape


def test_c3d_integer_format(tmp_path):
    """16-bit integer files (POINT:SCALE > 0): coordinates are stored / SCALE."""
    F, M, scale = 5, 3, 0.05
    pts = np.arange(F * M * 3, dtype=np.float64).reshape(F, M, 3)
    fn = str(tmp_path / "int.c3d")
    ingest.write_c3d(fn, pts, rate=60.0, units="mm")
    raw = bytearray(open(fn, "rb").read())
    # rewrite the same file as an integer file: patch POINT:SCALE and the header scale, re-encode the data section
    data_start = struct.unpack_from("<H", raw, 16)[0]
    i = raw.find(b"SCALE")
    struct.pack_into("<f", raw, i + 5 + 2 + 2, scale)   # name | offset(2) | type(1) ndim(1) | value
    struct.pack_into("<f", raw, 12, scale)
    ints = np.zeros((F, M, 4), "<i2")
    ints[..., :3] = np.round(pts / scale)
    body = ints.tobytes()
    raw = raw[:(data_start - 1) * 512] + body.ljust((len(body) + 511) // 512 * 512, b"\x00")
    open(fn, "wb").write(bytes(raw))
    got = ingest.read_c3d(fn)["points"]
    np.testing.assert_allclose(got, np.round(pts / scale) * scale, atol=1e-9)


def test_c3d_rejects_foreign_byte_orders(tmp_path):
    fn = str(tmp_path / "dec.c3d")
    ingest.write_c3d(fn, np.zeros((2, 2, 3)), rate=30.0)
    raw = bytearray(open(fn, "rb").read())
    raw[512 + 3] = 85  # DEC
    open(fn, "wb").write(bytes(raw))
    with pytest.raises(NotImplementedError, match="processor type"):
        ingest.read_c3d(fn)


def test_video_frame_rate_from_avi_headers(tmp_path):
    def chunk(tag, payload):
        return tag + struct.pack("<I", len(payload)) + payload

    avih = struct.pack("<IIIIIIIIII", 33367, 0, 0, 0, 450, 0, 1, 0, 640, 480) + b"\x00" * 16
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, 1001, 30000, 0, 450, 0, 0, 0) + b"\x00" * 8
    hdrl = b"hdrl" + chunk(b"avih", avih) + chunk(b"LIST", b"strl" + chunk(b"strh", strh))
    body = b"AVI " + chunk(b"LIST", hdrl)
    fn = str(tmp_path / "v.avi")
    open(fn, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)
    assert ingest.video_frame_rate(fn) == pytest.approx(30000 / 1001)
    with pytest.raises(ValueError):
        open(fn, "wb").write(b"not a video")
        ingest.video_frame_rate(fn)


def test_load_smpl_pkl_reads_the_smplx_layout(tmp_path, tables):
    """SMPL_NEUTRAL.pkl as smplx reads it: a latin1 pickle of a dict with v_template [V,3], shapedirs [V,3,300 or 10],
    posedirs [V,3,207], J_regressor (scipy sparse), weights [V,24], kintree_table [2,24] (root's parent 2^32-1), f [13776,3]
    -- written here from the synthetic tables, read back through the product's loader."""
    import scipy.sparse as sp

    from uuo_mocap_amd.body_model import load_model, load_smpl_pkl

    V = tables.v_template.shape[0]
    shapedirs = np.zeros((V, 3, 300), np.float64)
    shapedirs[:, :, :10] = tables.shapedirs
    kintree = np.stack([np.asarray(tables.parents, np.int64), np.arange(24, dtype=np.int64)]).astype(np.uint32)
    kintree[0, 0] = 4294967295
    payload = {
        "v_template": tables.v_template.astype(np.float64),
        "shapedirs": shapedirs,
        "posedirs": tables.posedirs.T.reshape(V, 3, 207).astype(np.float64),
        "J_regressor": sp.csc_matrix(tables.J_regressor.astype(np.float64)),
        "weights": tables.lbs_weights.astype(np.float64),
        "kintree_table": kintree,
        "f": np.asarray(tables.faces, np.uint32),
        "bs_type": "lrotmin", "bs_style": "lbs",
    }
    os.makedirs(tmp_path / "smpl")
    fn = str(tmp_path / "smpl" / "SMPL_NEUTRAL.pkl")
    with open(fn, "wb") as fh:
        pickle.dump(payload, fh, protocol=2)
    got = load_smpl_pkl(fn)
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
        np.testing.assert_array_equal(getattr(got, k), getattr(tables, k), err_msg=k)
        assert getattr(got, k).dtype == np.float32
    np.testing.assert_array_equal(got.parents[1:], np.asarray(tables.parents)[1:])
    assert got.parents[0] == -1
    np.testing.assert_array_equal(got.faces, np.asarray(tables.faces))
    assert got.checksum() == tables.checksum()
    # and through the smplx.create-style entry point
    assert load_model(str(tmp_path), "neutral", allow_synthetic=False).checksum() == tables.checksum()
