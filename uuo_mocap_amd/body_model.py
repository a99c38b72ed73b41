"""SMPL body-model tables: loader for a licensed ``SMPL_NEUTRAL.pkl`` and a
deterministic synthetic SMPL-shaped stand-in.

The reference builds its model with ``smplx.create("./body_models/",
model_type="smpl", gender="neutral")`` (reference utils/smpl.py:22-27).  The
licensed pickle is not redistributable and is absent from both the build
container and the GPU box, so every test and benchmark runs on
:func:`synthetic_smpl`, a table set with exactly SMPL's shapes and sparsity
structure (V=6890, J=24, 10 betas, 207 pose-blend coefficients, <=4 non-zero
skin weights per vertex, sparse row-stochastic joint regressor) generated from
an integer hash so it is bit-reproducible on any box.

Only *data* lives here; the arithmetic that consumes these tables is in
``csrc/`` (HIP) and, as the checker, ``oracle/``.
"""
from __future__ import annotations

import os
import pickle
from dataclasses import dataclass
from typing import Optional

import numpy as np

NUM_VERTS = 6890
NUM_JOINTS = 24
NUM_BETAS = 10
NUM_POSE_FEATS = 207  # 23 joints x 9
NUM_FACES = 13776

# SMPL kinematic tree (public smplx documentation; reference utils/smpl_utils.py:11-36 lists the joint order)
SMPL_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21], dtype=np.int64
)

SMPL_JOINT_NAMES = [
    "pelvis", "left_hip", "right_hip", "spine1", "left_knee", "right_knee", "spine2", "left_ankle",
    "right_ankle", "spine3", "left_foot", "right_foot", "neck", "left_collar", "right_collar", "head",
    "left_shoulder", "right_shoulder", "left_elbow", "right_elbow", "left_wrist", "right_wrist",
    "left_hand", "right_hand",
]

# smplx VertexJointSelector with VERTEX_IDS['smplh']: nose, reye, leye, rear, lear,
# LBigToe, LSmallToe, LHeel, RBigToe, RSmallToe, RHeel, then l/r x thumb,index,middle,ring,pinky tips.
SMPL_EXTRA_JOINT_VIDS = np.array(
    [332, 6260, 2800, 4071, 583,
     3216, 3226, 3387, 6617, 6624, 6787,
     2746, 2319, 2445, 2556, 2673,
     6191, 5782, 5905, 6016, 6133], dtype=np.int64
)


@dataclass
class SmplTables:
    """Host-side SMPL tables (float32 / int64 numpy), layouts as smplx holds them."""

    v_template: np.ndarray  # [V,3]
    shapedirs: np.ndarray  # [V,3,10]
    posedirs: np.ndarray  # [207, V*3]   (smplx: posedirs.reshape(V*3,207).T)
    J_regressor: np.ndarray  # [24,V]
    parents: np.ndarray  # [24] int64
    lbs_weights: np.ndarray  # [V,24]
    faces: np.ndarray  # [13776,3] int64
    extra_joint_vids: np.ndarray  # [21] int64
    name: str = "synthetic"

    def checksum(self) -> int:
        """Order-dependent 64-bit checksum of every float table (pins the generator across boxes)."""
        acc = np.uint64(0)
        for arr in (self.v_template, self.shapedirs, self.posedirs, self.J_regressor, self.lbs_weights):
            bits = np.ascontiguousarray(arr, dtype=np.float32).view(np.uint32).astype(np.uint64).ravel()
            idx = np.arange(1, bits.size + 1, dtype=np.uint64)
            with np.errstate(over="ignore"):
                acc = acc * np.uint64(0x9E3779B97F4A7C15) + np.sum(bits * (idx | np.uint64(1)), dtype=np.uint64)
        return int(acc)


# ----------------------------------------------------------------------------------------------
# deterministic hash -> floats
# ----------------------------------------------------------------------------------------------

def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(seed: int, *shape: int) -> np.ndarray:
    """float64 uniforms in [0,1) from (seed, flat index) via SplitMix64; no RNG state."""
    n = int(np.prod(shape)) if shape else 1
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _splitmix64(idx ^ _splitmix64(np.array([seed], dtype=np.uint64)))
    u = (key >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.reshape(shape) if shape else u[0]


def hash_normal(seed: int, *shape: int) -> np.ndarray:
    """float64 standard normals (Box-Muller on two hash streams)."""
    u1 = hash_uniform(seed * 2 + 1, *shape)
    u2 = hash_uniform(seed * 2 + 2, *shape)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


# ----------------------------------------------------------------------------------------------
# synthetic SMPL-shaped model
# ----------------------------------------------------------------------------------------------

# nominal T-pose joint centres (metres; x lateral, y up, z forward) -- SMPL-like proportions
_NOMINAL_JOINTS = np.array([
    [0.000, 0.000, 0.000],   # pelvis
    [0.070, -0.090, 0.000],  # l hip
    [-0.070, -0.090, 0.000],  # r hip
    [0.000, 0.110, -0.020],  # spine1
    [0.100, -0.470, 0.000],  # l knee
    [-0.100, -0.470, 0.000],  # r knee
    [0.000, 0.250, 0.000],   # spine2
    [0.090, -0.870, -0.030],  # l ankle
    [-0.090, -0.870, -0.030],  # r ankle
    [0.000, 0.300, 0.020],   # spine3
    [0.110, -0.930, 0.090],  # l foot
    [-0.110, -0.930, 0.090],  # r foot
    [0.000, 0.510, -0.020],  # neck
    [0.080, 0.420, 0.000],   # l collar
    [-0.080, 0.420, 0.000],  # r collar
    [0.000, 0.580, 0.030],   # head
    [0.180, 0.450, -0.010],  # l shoulder
    [-0.180, 0.450, -0.010],  # r shoulder
    [0.440, 0.440, -0.030],  # l elbow
    [-0.440, 0.440, -0.030],  # r elbow
    [0.690, 0.450, -0.030],  # l wrist
    [-0.690, 0.450, -0.030],  # r wrist
    [0.770, 0.440, -0.040],  # l hand
    [-0.770, 0.440, -0.040],  # r hand
], dtype=np.float64)

# per-joint limb radius (m) and relative surface share
_RADIUS = np.array([0.13, 0.075, 0.075, 0.12, 0.055, 0.055, 0.125, 0.04, 0.04, 0.13, 0.035, 0.035,
                    0.055, 0.06, 0.06, 0.09, 0.05, 0.05, 0.04, 0.04, 0.03, 0.03, 0.035, 0.035])


def _segment_ends(parents: np.ndarray, joints: np.ndarray) -> np.ndarray:
    """End point of the limb segment that starts at each joint."""
    ends = np.zeros_like(joints)
    for j in range(NUM_JOINTS):
        kids = np.where(parents == j)[0]
        if len(kids) == 1:
            ends[j] = joints[kids[0]]
        elif len(kids) > 1:
            ends[j] = joints[j] + 0.6 * (joints[kids].mean(axis=0) - joints[j])
        else:  # leaves: extend along the incoming bone
            p = parents[j]
            d = joints[j] - joints[p]
            d = d / np.linalg.norm(d)
            ext = {15: 0.16, 10: 0.10, 11: 0.10, 22: 0.09, 23: 0.09}.get(j, 0.08)
            ends[j] = joints[j] + ext * d
    return ends


def _point_segment_distance(p: np.ndarray, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """p [V,3], a/b [J,3] -> [V,J] distances to segments."""
    ab = b - a  # [J,3]
    ap = p[:, None, :] - a[None]  # [V,J,3]
    t = np.clip(np.sum(ap * ab[None], axis=-1) / np.sum(ab * ab, axis=-1)[None], 0.0, 1.0)
    closest = a[None] + t[..., None] * ab[None]
    return np.linalg.norm(p[:, None, :] - closest, axis=-1)


_SYNTH_CACHE: dict = {}


def synthetic_smpl(seed: int = 0) -> SmplTables:
    """Deterministic SMPL-shaped tables (see module docstring). Cached per seed."""
    if seed in _SYNTH_CACHE:
        return _SYNTH_CACHE[seed]
    parents = SMPL_PARENTS.copy()
    J0 = _NOMINAL_JOINTS
    ends = _segment_ends(parents, J0)
    seg_len = np.linalg.norm(ends - J0, axis=1)

    # vertices per joint ~ lateral surface area, fixed up to sum exactly to V
    area = seg_len * _RADIUS
    counts = np.floor(area / area.sum() * NUM_VERTS).astype(np.int64)
    counts = np.maximum(counts, 48)
    j = 0
    while counts.sum() != NUM_VERTS:
        step = 1 if counts.sum() < NUM_VERTS else -1
        counts[j % NUM_JOINTS] += step
        j += 1

    verts = np.zeros((NUM_VERTS, 3), dtype=np.float64)
    owner = np.zeros(NUM_VERTS, dtype=np.int64)
    faces = []
    base = 0
    for jn in range(NUM_JOINTS):
        n = int(counts[jn])
        ring = max(8, int(round(np.sqrt(n * 2.0 * np.pi * _RADIUS[jn] / max(seg_len[jn], 1e-3)))))
        ring = min(ring, n)
        rows = int(np.ceil(n / ring))
        d = ends[jn] - J0[jn]
        d = d / np.linalg.norm(d)
        ref = np.array([0.0, 0.0, 1.0]) if abs(d[2]) < 0.9 else np.array([1.0, 0.0, 0.0])
        u = np.cross(d, ref)
        u /= np.linalg.norm(u)
        w = np.cross(d, u)
        jit = hash_uniform(seed * 1000 + 10 + jn, n, 3)
        k = np.arange(n)
        r_idx = k // ring
        a_idx = k % ring
        t = (r_idx + 0.25 + 0.5 * jit[:, 0]) / rows
        theta = 2.0 * np.pi * (a_idx + 0.8 * jit[:, 1] + 0.5 * (r_idx % 2)) / ring
        # taper radius toward the segment ends so limbs close up
        taper = 0.55 + 0.45 * np.sin(np.pi * np.clip(t, 0.02, 0.98))
        rad = _RADIUS[jn] * taper * (0.92 + 0.16 * jit[:, 2])
        verts[base:base + n] = (J0[jn][None] + t[:, None] * (ends[jn] - J0[jn])[None]
                                + rad[:, None] * (np.cos(theta)[:, None] * u[None] + np.sin(theta)[:, None] * w[None]))
        owner[base:base + n] = jn
        # tube faces between consecutive rings
        for r in range(rows - 1):
            for a in range(ring):
                v00 = base + r * ring + a
                v01 = base + r * ring + (a + 1) % ring
                v10 = v00 + ring
                v11 = v01 + ring
                if v10 < base + n and v11 < base + n:
                    faces.append((v00, v01, v11))
                    faces.append((v00, v11, v10))
        base += n

    faces = np.array(faces, dtype=np.int64)
    if faces.shape[0] >= NUM_FACES:
        faces = faces[:NUM_FACES]
    else:  # pad with fan triangles so the array has SMPL's face count (faces are off the hot path)
        pad = NUM_FACES - faces.shape[0]
        k = np.arange(pad, dtype=np.int64)
        extra = np.stack([k % NUM_VERTS, (k + 1) % NUM_VERTS, (k + 2) % NUM_VERTS], axis=1)
        faces = np.concatenate([faces, extra], axis=0)

    # skin weights: <=4 non-zeros per vertex (owner + nearest segments), rows sum to 1
    dist = _point_segment_distance(verts, J0, ends)  # [V,24]
    sigma = 0.06
    score = np.exp(-(dist / sigma) ** 2)
    score[np.arange(NUM_VERTS), owner] += 1.0  # owner always dominant
    order = np.argsort(-score, axis=1, kind="stable")[:, :4]
    lbs = np.zeros((NUM_VERTS, NUM_JOINTS), dtype=np.float64)
    rows_i = np.arange(NUM_VERTS)[:, None]
    top = score[rows_i, order]
    top = np.where(top < 1e-3, 0.0, top)
    top = top / top.sum(axis=1, keepdims=True)
    lbs[rows_i, order] = top

    # joint regressor: 24 nearest vertices, positive hashed weights, row-stochastic
    Jreg = np.zeros((NUM_JOINTS, NUM_VERTS), dtype=np.float64)
    for jn in range(NUM_JOINTS):
        dj = np.linalg.norm(verts - J0[jn][None], axis=1)
        near = np.argsort(dj, kind="stable")[:24]
        wj = 0.5 + hash_uniform(seed * 1000 + 200 + jn, 24)
        Jreg[jn, near] = wj / wj.sum()

    # shape blend shapes
    S = np.zeros((NUM_VERTS, 3, NUM_BETAS), dtype=np.float64)
    x, y, z = verts[:, 0], verts[:, 1], verts[:, 2]
    S[:, :, 0] = 0.04 * verts
    S[:, 0, 1] = 0.03 * x
    S[:, 2, 1] = 0.03 * z
    S[:, 1, 2] = 0.03 * y
    S[:, 0, 3] = 0.03 * np.sign(x) * np.maximum(np.abs(x) - 0.2, 0.0)
    om = 2.0 + 10.0 * hash_uniform(seed * 1000 + 300, NUM_BETAS, 3, 3)
    ph = 2.0 * np.pi * hash_uniform(seed * 1000 + 301, NUM_BETAS, 3)
    for c in range(4, NUM_BETAS):
        for a in range(3):
            S[:, a, c] = 0.01 * np.sin(verts @ om[c, a] + ph[c, a])

    # pose blend shapes: local to the driving joint, smooth in space
    P = np.zeros((NUM_POSE_FEATS, NUM_VERTS, 3), dtype=np.float64)
    om_p = 3.0 + 12.0 * hash_uniform(seed * 1000 + 400, NUM_POSE_FEATS, 3, 3)
    ph_p = 2.0 * np.pi * hash_uniform(seed * 1000 + 401, NUM_POSE_FEATS, 3)
    for k in range(NUM_POSE_FEATS):
        jn = k // 9 + 1
        loc = lbs[:, jn] + 0.5 * lbs[:, parents[jn]] + 0.05
        for a in range(3):
            P[k, :, a] = 0.008 * loc * np.sin(verts @ om_p[k, a] + ph_p[k, a])

    tables = SmplTables(
        v_template=verts.astype(np.float32),
        shapedirs=S.astype(np.float32),
        posedirs=P.reshape(NUM_POSE_FEATS, NUM_VERTS * 3).astype(np.float32),
        J_regressor=Jreg.astype(np.float32),
        parents=parents,
        lbs_weights=lbs.astype(np.float32),
        faces=faces,
        extra_joint_vids=SMPL_EXTRA_JOINT_VIDS.copy(),
        name="synthetic-seed%d" % seed,
    )
    _SYNTH_CACHE[seed] = tables
    return tables


# ----------------------------------------------------------------------------------------------
# licensed model loader (best effort; mirrors what smplx reads from SMPL_NEUTRAL.pkl)
# ----------------------------------------------------------------------------------------------

class _ChStub:
    """Stands in for chumpy.ch.Ch when unpickling original SMPL files (install.sh:18 needs chumpy)."""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {})

    @property
    def r(self):
        return np.asarray(self.__dict__.get("x"))


class _SmplUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("chumpy"):
            return _ChStub
        return super().find_class(module, name)


def _to_np(x):
    if isinstance(x, _ChStub):
        return np.asarray(x.r)
    if hasattr(x, "toarray"):
        return np.asarray(x.toarray())
    return np.asarray(x)


def load_smpl_pkl(path: str) -> SmplTables:
    with open(path, "rb") as fh:
        data = _SmplUnpickler(fh, encoding="latin1").load()
    v_t = _to_np(data["v_template"]).astype(np.float32)
    V = v_t.shape[0]
    shapedirs = _to_np(data["shapedirs"])[:, :, :NUM_BETAS].astype(np.float32)
    posedirs = _to_np(data["posedirs"]).astype(np.float32)  # [V,3,207]
    posedirs = posedirs.reshape(V * 3, -1).T.copy()
    kintree = _to_np(data["kintree_table"]).astype(np.int64)
    parents = kintree[0].copy()
    parents[0] = -1
    return SmplTables(
        v_template=v_t,
        shapedirs=shapedirs,
        posedirs=posedirs,
        J_regressor=_to_np(data["J_regressor"]).astype(np.float32),
        parents=parents,
        lbs_weights=_to_np(data["weights"]).astype(np.float32),
        faces=_to_np(data["f"]).astype(np.int64),
        extra_joint_vids=SMPL_EXTRA_JOINT_VIDS.copy(),
        name=os.path.basename(path),
    )


def load_model(body_model_path: str = "./body_models/", gender: str = "neutral",
               allow_synthetic: bool = True) -> SmplTables:
    """smplx.create(path, model_type="smpl", gender=...) equivalent: real pickle iff present."""
    fn = os.path.join(body_model_path, "smpl", "SMPL_%s.pkl" % gender.upper())
    if os.path.isfile(fn):
        return load_smpl_pkl(fn)
    if not allow_synthetic:
        raise FileNotFoundError(fn)
    return synthetic_smpl(0)
