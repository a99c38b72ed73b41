"""Import shim used ONLY by oracle/make_golden.py, ONLY in the build container.  TEST INFRASTRUCTURE.

The reference's stage solvers (src/video_mocap/optimization.py, markers/markers_utils.py,
multimodal.py) import third-party packages that are not installed here (pytorch3d, smplx, roma,
igl, trimesh, moshpp, pybullet, ...: SURVEY.md 8c).  ``install()`` registers in-memory stand-in
modules that route the *numerical* entry points to the oracle restatements (oracle/p3d_ref.py,
oracle/smpl_ref.py) and leave everything else as inert placeholders, so the reference's own
orchestration code can be executed to capture golden fixtures that pin loss weights, parameter
packing, L-BFGS options and stage order.  No reference source is copied; /root/reference is only
put on sys.path.  Nothing here is used on the GPU box.
"""
from __future__ import annotations

import sys
import types

REFERENCE_SRC = "/root/reference/src"


class _Inert(types.ModuleType):
    """Module whose every missing attribute is an inert callable/class placeholder."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        class _Placeholder:  # noqa: D401 - stand-in
            def __init__(self, *a, **k):
                pass

            def __call__(self, *a, **k):
                raise RuntimeError("inert shim object called: %s.%s" % (self.__class__.__module__, name))

        _Placeholder.__name__ = name
        setattr(self, name, _Placeholder)
        return _Placeholder


def _mod(name: str, inert: bool = True, **attrs):
    m = (_Inert if inert else types.ModuleType)(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave as a package so submodule imports resolve through sys.modules
    sys.modules[name] = m
    return m


def _merge(dst, *srcs):
    for src in srcs:
        for k, v in src.items():
            if isinstance(v, dict) and isinstance(dst.get(k), dict):
                _merge(dst[k], v)
            elif isinstance(v, dict):
                dst[k] = _merge({}, v)
            else:
                dst[k] = v
    return dst


def install(tables=None):
    """Register the stand-ins and put the reference on sys.path. Returns the SMPL tables in use."""
    from oracle import p3d_ref, smpl_ref
    from uuo_mocap_amd.body_model import SMPL_JOINT_NAMES, synthetic_smpl

    if tables is None:
        tables = synthetic_smpl(0)

    _mod("pytorch3d")
    _mod("pytorch3d.loss", chamfer_distance=p3d_ref.chamfer_distance)
    _mod("pytorch3d.loss.chamfer", chamfer_distance=p3d_ref.chamfer_distance)
    _mod("pytorch3d.ops", knn_points=p3d_ref.knn_points)
    _mod(
        "pytorch3d.transforms",
        rotation_6d_to_matrix=p3d_ref.rotation_6d_to_matrix,
        matrix_to_rotation_6d=p3d_ref.matrix_to_rotation_6d,
        axis_angle_to_matrix=p3d_ref.axis_angle_to_matrix,
        quaternion_to_matrix=p3d_ref.quaternion_to_matrix,
        matrix_to_quaternion=p3d_ref.matrix_to_quaternion,
        so3_relative_angle=p3d_ref.so3_relative_angle,
    )

    def _create(model_path, model_type="smpl", gender="neutral", batch_size=1, **kw):
        assert model_type == "smpl"
        return smpl_ref.SMPLRef(tables)

    _mod("smplx", create=_create)
    _mod("smplx.joint_names", SMPL_JOINT_NAMES=list(SMPL_JOINT_NAMES))
    _mod("moshpp")
    _mod("moshpp.marker_layout")
    _mod("moshpp.marker_layout.marker_vids", all_marker_vids={"smpl": {}})
    _mod("mergedeep", merge=_merge)
    for name in ("roma", "roma.utils", "igl", "trimesh", "trimesh.triangles", "pybullet", "pybullet_data", "cv2",
                 "imageio", "pyrender", "seaborn", "ezc3d", "human_body_prior", "human_body_prior.tools",
                 "human_body_prior.tools.rotation_tools", "OpenGL", "OpenGL.GL", "matplotlib.pyplot_shim"):
        if name not in sys.modules:
            _mod(name)
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    return tables
