"""Real-data ingestion next to the fitted path: the three readers the reference's runner needs
(reference src/video_mocap/test/test.py:87-102) without its third-party dependencies.

* ``ImgSmpl`` -- reference src/video_mocap/img_smpl/img_smpl.py:12-147: the per-frame dictionary that 4D-Humans / PHALP
  writes (``demo_<sequence>.pkl``, read with joblib) -> dense per-frame SMPL parameters; frames without a tracked person
  are filled from their valid neighbours (nearest at the ends; translation and shape linearly, rotations by shortest-arc
  quaternion slerp in between), ``img_mask`` marks the frames that had a detection.  Same constructor, same attributes.
  The gap filling is one batched expression (on the GPU when a device is given) instead of a Python loop over frames with
  five tensor assignments each.
* ``Markers`` -- reference src/video_mocap/markers/markers.py:6-54 (a thin wrapper over ezc3d): here a self-contained reader
  of the C3D point section (https://www.c3d.org/HTML/default.htm: 512-byte blocks, parameter section with the POINT group,
  integer or floating-point frames, Intel byte order) and a writer of the same subset for tests and synthetic exports.
  ezc3d is absent here, so the parser is anchored on the published format only (parity unpinned for exotic files: DEC / MIPS
  byte orders are refused, not guessed).
* ``video_frame_rate`` -- the reference asks OpenCV for CAP_PROP_FPS of the ``.avi`` (test.py:87-88); here the rate is read
  from the RIFF headers (video stream ``strh`` dwRate / dwScale, else ``avih`` dwMicroSecPerFrame).
"""
from __future__ import annotations

import os
import struct
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .resample import quaternion_to_matrix, unitquat_slerp
from .transforms import matrix_to_axis_angle, matrix_to_quaternion

#: 4D-Humans joint order (reference utils/img_smpl_utils.py:4-52); only the entries used here
JOINT_PELVIS_LOW = 8
TOES = ((19, 20), (22, 23))  # (l_toe_in, l_toe_out), (r_toe_in, r_toe_out)

_HMR_TO_MOCAP = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0]], dtype=np.float32)  # img_smpl.py:40-44


def get_foot_contacts(joints_2d: np.ndarray, freq: float) -> np.ndarray:
    """reference utils/img_smpl_utils.py:57-98: a toe pair is in contact when both 2D toe key points move slower than
    1e-4 / (diagonal of the person's 2D bounding box) per second-squared-frame unit.  [F,45,2] -> [F,2]."""
    min_x, max_x = np.min(joints_2d[:, :, 0], axis=1), np.max(joints_2d[:, :, 0], axis=1)
    min_y, max_y = np.min(joints_2d[:, :, 1], axis=1), np.max(joints_2d[:, :, 1], axis=1)
    extent = np.sqrt(((max_x - min_x) ** 2) + ((max_y - min_y) ** 2))
    extent = np.maximum(extent, np.ones_like(extent) * 0.01)
    threshold = np.repeat(np.expand_dims(0.0001 / extent, axis=-1), repeats=joints_2d.shape[1], axis=-1)
    vel = np.concatenate((np.zeros((1,) + joints_2d.shape[1:]), joints_2d[1:] - joints_2d[:-1]), axis=0) / freq
    still = np.linalg.norm(vel, axis=-1) < threshold
    out = np.ones((joints_2d.shape[0], len(TOES)))
    for g, pair in enumerate(TOES):
        for t in pair:
            out[:, g] *= still[:, t]
    return out


def fill_gaps(values: Dict[str, torch.Tensor], img_mask: torch.Tensor) -> Dict[str, torch.Tensor]:
    """img_smpl.py:54-98 for all frames at once.  `values`: "trans" [F,3] and "betas" [F,10] (linear) and any number of
    rotation tracks [F,K,3,3] (shortest-arc slerp of pytorch3d quaternions, roma.utils.unitquat_slerp semantics); frames
    where img_mask is False take the nearest valid frame at the ends and the interpolation of their valid neighbours in
    between, alpha = (f - left) / (right - left) in Python-float arithmetic like the reference."""
    F = int(img_mask.shape[0])
    valid = torch.where(img_mask)[0].tolist()
    if len(valid) == 0 or len(valid) == F:
        return {k: v.clone() for k, v in values.items()}
    left, right, alpha = [], [], []
    vi = 0
    for f in range(F):
        while vi + 1 < len(valid) and valid[vi + 1] <= f:
            vi += 1
        if img_mask[f]:
            left.append(f); right.append(f); alpha.append(0.0)
        elif f < valid[0]:
            left.append(valid[0]); right.append(valid[0]); alpha.append(0.0)
        elif f > valid[-1]:
            left.append(valid[-1]); right.append(valid[-1]); alpha.append(0.0)
        else:
            lo, hi = valid[vi], valid[vi + 1]
            left.append(lo); right.append(hi); alpha.append((f - lo) / (hi - lo))
    dev = next(iter(values.values())).device
    li, ri = torch.tensor(left, device=dev), torch.tensor(right, device=dev)
    a = torch.tensor(alpha, dtype=torch.float32, device=dev)
    inner = (li != ri)
    out = {}
    for key, v in values.items():
        if v.dim() >= 3 and v.shape[-2:] == (3, 3):
            q = unitquat_slerp(matrix_to_quaternion(v[li]), matrix_to_quaternion(v[ri]),
                               a.reshape((-1,) + (1,) * (v.dim() - 3)))
            filled = torch.where(inner.reshape((-1,) + (1,) * (v.dim() - 1)), quaternion_to_matrix(q), v[li])
        else:
            w = a.reshape((-1,) + (1,) * (v.dim() - 1))
            filled = torch.where(inner.reshape(w.shape), v[li] * (1.0 - w) + v[ri] * w, v[li])
        out[key] = filled.to(v.dtype)
    return out


class ImgSmpl:
    """reference img_smpl/img_smpl.py:12-147.  `data`: {frame key -> {"tracked_ids": [...], "smpl": [{"global_orient"
    [1,3,3], "body_pose" [23,3,3], "betas" [10]}], "3d_joints": [[45,3]], "camera_bbox": [[3]], "center": [[2]],
    "scale": [s], "size": [[2]], "2d_joints": [[90]]}} as 4D-Humans writes it; `freq` the video frame rate."""

    def __init__(self, data: Dict, freq: float, device: Optional[torch.device] = None):
        self.data = data
        self.freq = freq
        keys = sorted(data.keys())
        F = len(keys)
        trans = np.zeros((F, 3), np.float32)
        root = np.zeros((F, 1, 3, 3), np.float32)
        hmr_root = np.zeros((F, 1, 3, 3), np.float32)
        pose = np.zeros((F, 23, 3, 3), np.float32)
        betas = np.zeros((F, 10), np.float32)
        mask = np.zeros(F, bool)
        camera_bbox, center = np.zeros((F, 3), np.float32), np.zeros((F, 2), np.float32)
        size, scale = np.zeros((F, 2), np.float32), np.zeros((F, 1), np.float32)
        joints_2d = np.zeros((F, 45, 2))
        for i, key in enumerate(keys):
            fr = data[key]
            if len(fr["tracked_ids"]) > 0:
                mask[i] = True
                smpl0 = fr["smpl"][0]
                ro = np.asarray(smpl0["global_orient"], np.float32).reshape(1, 3, 3)
                hmr_root[i] = ro
                trans[i] = np.asarray(fr["3d_joints"][0], np.float32)[JOINT_PELVIS_LOW]
                root[i] = _HMR_TO_MOCAP @ ro
                pose[i] = np.asarray(smpl0["body_pose"], np.float32).reshape(23, 3, 3)
                betas[i] = np.asarray(smpl0["betas"], np.float32).reshape(10)
            if len(fr.get("camera_bbox", [])) > 0:
                camera_bbox[i] = np.asarray(fr["camera_bbox"][0], np.float32)
                center[i] = np.asarray(fr["center"][0], np.float32)
                scale[i] = np.asarray(fr["scale"][0], np.float32)
                size[i] = np.asarray(fr["size"][0], np.float32)
            j2 = fr.get("2d_joints", [])
            if len(j2) > 0:
                flat = np.asarray(j2[0], np.float64).reshape(-1)
                n = min(45, flat.shape[0] // 2)
                joints_2d[i, :n] = flat[:2 * n].reshape(n, 2)
        dev = torch.device(device) if device is not None else torch.device("cpu")
        self.img_mask = torch.from_numpy(mask)
        filled = fill_gaps({"trans": torch.from_numpy(trans).to(dev), "betas": torch.from_numpy(betas).to(dev),
                            "root_orient": torch.from_numpy(root).to(dev), "hmr_root_orient": torch.from_numpy(hmr_root).to(dev),
                            "pose_body": torch.from_numpy(pose).to(dev)}, self.img_mask.to(dev))
        self.trans = filled["trans"].cpu()
        self.root_orient = filled["root_orient"].cpu()
        self.hmr_root_orient = filled["hmr_root_orient"].cpu()
        self.pose_body = filled["pose_body"].cpu()
        self.betas = filled["betas"].cpu()
        self.camera_bbox = torch.from_numpy(camera_bbox)
        self.center = torch.from_numpy(center)
        self.scale = torch.from_numpy(scale)
        self.size = torch.from_numpy(size)
        self.foot_contacts = torch.from_numpy(get_foot_contacts(joints_2d, freq).astype(np.float32))

    def get_smpl(self) -> Dict:
        poses = torch.flatten(matrix_to_axis_angle(torch.cat((self.root_orient, self.pose_body), dim=1)), start_dim=1)
        return {"betas": self.betas[0].numpy(), "gender": np.array("neutral"), "mocap_frame_rate": self.freq,
                "poses": poses.numpy(), "trans": self.trans.numpy()}


def load_hmr_pkl(filename: str, freq: float, device: Optional[torch.device] = None) -> ImgSmpl:
    """test.py:95-96: ``ImgSmpl(joblib.load(demo_<sequence>.pkl), video_freq)``."""
    import joblib

    return ImgSmpl(joblib.load(filename), freq, device=device)


# ---------------------------------------------------------------------------------------------------------------------
# C3D
# ---------------------------------------------------------------------------------------------------------------------
def _read_parameters(buf: bytes, start_block: int) -> Dict[str, Dict]:
    """The parameter section: header (4 bytes: 2 reserved, block count, processor type) then a chain of group / parameter
    records.  Returns {GROUP: {PARAM: value}} with values as numpy arrays (or lists of strings for character data)."""
    base = (start_block - 1) * 512
    proc = buf[base + 3]
    if proc != 84:
        raise NotImplementedError("C3D processor type %d (84 = Intel is supported; 85 DEC and 86 MIPS files are not)" % proc)
    pos = base + 4
    groups: Dict[int, str] = {}
    raw: List = []
    while True:
        n_name = struct.unpack_from("<b", buf, pos)[0]
        gid = struct.unpack_from("<b", buf, pos + 1)[0]
        n = abs(n_name)
        if n == 0:
            break
        name = buf[pos + 2:pos + 2 + n].decode("latin1").strip().upper()
        off_pos = pos + 2 + n
        next_off = struct.unpack_from("<h", buf, off_pos)[0]
        if gid < 0:
            groups[-gid] = name
        else:
            p = off_pos + 2
            dtype = struct.unpack_from("<b", buf, p)[0]
            ndim = buf[p + 1]
            dims = list(buf[p + 2:p + 2 + ndim])
            p += 2 + ndim
            count = int(np.prod(dims)) if ndim else 1
            if dtype == -1:
                chars = buf[p:p + count].decode("latin1")
                if ndim <= 1:
                    val = [chars.strip()]
                else:
                    w = dims[0]
                    val = [chars[i * w:(i + 1) * w].strip() for i in range(count // max(w, 1))]
            else:
                fmt = {1: "<i1", 2: "<i2", 4: "<f4"}[dtype]
                val = np.frombuffer(buf, dtype=fmt, count=count, offset=p).copy()
                if ndim > 1:
                    val = val.reshape(dims[::-1])
            raw.append((gid, name, val))
        if next_off == 0:
            break
        pos = off_pos + next_off
    out: Dict[str, Dict] = {}
    for gid, name, val in raw:
        out.setdefault(groups.get(gid, "GROUP%d" % gid), {})[name] = val
    return out


def read_c3d(filename: str) -> Dict:
    """-> {"points" [F,M,3] float64 in file units, "residuals" [F,M], "rate", "units", "labels", "parameters"}."""
    with open(filename, "rb") as fh:
        buf = fh.read()
    if len(buf) < 512 or buf[1] != 0x50:
        raise ValueError("%s is not a C3D file (key byte 0x50 missing)" % filename)
    param_block = buf[0]
    params = _read_parameters(buf, param_block)
    point = params.get("POINT", {})
    n_points, n_analog_total, first, last = struct.unpack_from("<HHHH", buf, 2)
    scale_hdr = struct.unpack_from("<f", buf, 12)[0]
    data_start = struct.unpack_from("<H", buf, 16)[0]
    analog_per_frame = struct.unpack_from("<H", buf, 18)[0]
    rate_hdr = struct.unpack_from("<f", buf, 20)[0]
    used = int(point["USED"][0]) if "USED" in point else n_points
    scale = float(point["SCALE"][0]) if "SCALE" in point else scale_hdr
    if "DATA_START" in point:
        data_start = int(np.asarray(point["DATA_START"]).astype(np.int64)[0]) & 0xFFFF
    n_frames = last - first + 1
    if "FRAMES" in point:
        fr = int(np.asarray(point["FRAMES"]).astype(np.int64)[0]) & 0xFFFF
        if fr > 0 and last - first + 1 <= 0:
            n_frames = fr
    is_float = scale < 0
    words = used * 4 + n_analog_total
    off = (data_start - 1) * 512
    if is_float:
        arr = np.frombuffer(buf, dtype="<f4", count=n_frames * words, offset=off).reshape(n_frames, words)
        pts = arr[:, :used * 4].reshape(n_frames, used, 4).astype(np.float64)
        xyz = pts[..., :3]
        resid = pts[..., 3]
    else:
        arr = np.frombuffer(buf, dtype="<i2", count=n_frames * words, offset=off).reshape(n_frames, words)
        pts = arr[:, :used * 4].reshape(n_frames, used, 4).astype(np.float64)
        xyz = pts[..., :3] * abs(scale)
        resid = pts[..., 3]
    invalid = resid < 0 if not is_float else pts[..., 3] < 0  # a negative fourth word marks an invalid point
    xyz = np.where(invalid[..., None], np.nan, xyz)
    rate = float(point["RATE"][0]) if "RATE" in point else rate_hdr
    units = point["UNITS"][0] if "UNITS" in point else "mm"
    return {"points": xyz, "residuals": resid, "rate": rate, "units": units, "labels": list(point.get("LABELS", [])),
            "parameters": params}


def write_c3d(filename: str, points: np.ndarray, rate: float, units: str = "mm", labels: Optional[Sequence[str]] = None):
    """Minimal Intel floating-point C3D with the POINT group (USED, FRAMES, SCALE, RATE, DATA_START, UNITS, LABELS):
    the subset `read_c3d` (and ezc3d / the reference's Markers) needs.  `points` [F,M,3] in `units`; NaN = invalid."""
    points = np.asarray(points, np.float64)
    F, M = points.shape[0], points.shape[1]
    labels = list(labels) if labels is not None else ["M%03d" % i for i in range(M)]

    def group(gid, name):
        b = name.encode()
        return struct.pack("<bb", len(b), -gid) + b + struct.pack("<hB", 3, 0)

    def param(gid, name, dtype, dims, payload):
        b = name.encode()
        body = struct.pack("<bB", dtype, len(dims)) + bytes(dims) + payload + b"\x00"
        return struct.pack("<bb", len(b), gid) + b + struct.pack("<h", 2 + len(body)) + body

    lab_w = max(4, max(len(s) for s in labels))
    lab_bytes = b"".join(s.ljust(lab_w).encode()[:lab_w] for s in labels)
    units_b = units.ljust(4).encode()[:4]
    recs = [group(1, "POINT"),
            param(1, "USED", 2, [], struct.pack("<h", M)),
            param(1, "FRAMES", 2, [], struct.pack("<H", F & 0xFFFF)),
            param(1, "SCALE", 4, [], struct.pack("<f", -1.0)),
            param(1, "RATE", 4, [], struct.pack("<f", float(rate))),
            param(1, "UNITS", -1, [4], units_b),
            param(1, "LABELS", -1, [lab_w, M], lab_bytes)]
    body = b"".join(recs)
    # DATA_START depends on the size of the section that contains it: fixed-size record, so compute first
    ds_len = len(param(1, "DATA_START", 2, [], struct.pack("<h", 0)))
    total = 4 + len(body) + ds_len + 2
    n_param_blocks = (total + 511) // 512
    data_start = 2 + n_param_blocks
    body += param(1, "DATA_START", 2, [], struct.pack("<H", data_start))
    section = struct.pack("<BBBB", 1, 0x50, n_param_blocks, 84) + body + struct.pack("<bb", 0, 0)
    section = section.ljust(n_param_blocks * 512, b"\x00")
    header = bytearray(512)
    header[0], header[1] = 2, 0x50
    struct.pack_into("<HHHH", header, 2, M, 0, 1, F & 0xFFFF)
    struct.pack_into("<H", header, 10, 0)
    struct.pack_into("<f", header, 12, -1.0)
    struct.pack_into("<H", header, 16, data_start)
    struct.pack_into("<H", header, 18, 0)
    struct.pack_into("<f", header, 20, float(rate))
    data = np.zeros((F, M, 4), "<f4")
    bad = np.isnan(points).any(axis=-1)
    data[..., :3] = np.nan_to_num(points, nan=0.0)
    data[..., 3] = np.where(bad, -1.0, 0.0)
    blob = data.tobytes()
    blob = blob.ljust((len(blob) + 511) // 512 * 512, b"\x00")
    with open(filename, "wb") as fh:
        fh.write(bytes(header) + section + blob)


class Markers:
    """reference markers/markers.py:6-54: points [F,M,3] in METRES (POINT:UNITS m / cm / mm -> /1, /100, /1000), invalid
    points NaN (the runner zero-fills them, test.py:99), the frame rate as int(POINT:RATE), POINT:LABELS."""

    def __init__(self, filename: str, shuffle: bool = False):
        self.filename = filename
        c3d = read_c3d(filename)
        self.units = c3d["units"]
        self.scale_factor = {"m": 1, "cm": 100, "mm": 1000}.get(self.units, None)
        if self.scale_factor is None:
            raise ValueError("POINT:UNITS %r (m, cm or mm expected, markers.py:12-18)" % self.units)
        self.points = c3d["points"] / self.scale_factor
        if shuffle:
            shuffled = np.zeros_like(self.points)
            for f in range(self.points.shape[0]):
                shuffled[f] = self.points[f, np.random.permutation(self.points.shape[1])]
            self.points = shuffled
        self.freq = int(c3d["rate"])
        self.labels = c3d["labels"]

    def get_points(self):
        return self.points

    def set_points(self, points):
        self.points = points

    def get_labels(self):
        return self.labels

    def get_num_markers(self):
        return self.points.shape[1]

    def __len__(self):
        return self.points.shape[0]

    def get_duration(self):
        return self.freq * self.points.shape[0]

    def get_frequency(self):
        return self.freq


def cleanup_markers(points: np.ndarray) -> np.ndarray:
    """reference datasets/preprocess_cmu_kitchen.py:32-39 AS THE RUNNER CALLS IT (test.py:98-101): written for [4,M,F] arrays
    but handed [F,M,3], it walks the LAST axis (x, y, z), looks at `points[:3, :, c]` and cuts the coordinate axis after the
    last column that is not entirely zero in the first three frames -- effectively a no-op on real data;
    reproduced literally so that the degenerate case (trailing all-zero coordinates) behaves the same."""
    frame = 0
    for frame in range(points.shape[2] - 1, -1, -1):
        count = np.count_nonzero(points[:3, :, frame] == 0)
        if count != points[:3, :, frame].size:
            break
    return points[:, :, :frame + 1]


# ---------------------------------------------------------------------------------------------------------------------
# video frame rate
# ---------------------------------------------------------------------------------------------------------------------
def video_frame_rate(filename: str) -> float:
    """Frame rate of an AVI container from its RIFF headers (what cv2.VideoCapture(...).get(CAP_PROP_FPS) reports for the
    reference, test.py:87-88): dwRate / dwScale of the first video stream header, else 1e6 / dwMicroSecPerFrame."""
    with open(filename, "rb") as fh:
        head = fh.read(1 << 16)
    if head[:4] != b"RIFF" or head[8:12] != b"AVI ":
        raise ValueError("%s is not an AVI file; pass the frame rate explicitly" % filename)
    i = head.find(b"strh")
    while i >= 0:
        if head[i + 8:i + 12] == b"vids":
            scale, rate = struct.unpack_from("<II", head, i + 8 + 20)
            if scale > 0 and rate > 0:
                return rate / scale
        i = head.find(b"strh", i + 4)
    j = head.find(b"avih")
    if j >= 0:
        us = struct.unpack_from("<I", head, j + 8)[0]
        if us > 0:
            return 1e6 / us
    raise ValueError("no frame rate found in %s" % filename)
