"""Drop-in shim: put the directory that holds this package (`uuo_mocap_amd/dropin`) in front of the reference's `src/` on
sys.path and every hot-path import -- `video_mocap.multimodal`, `video_mocap.optimization`, `video_mocap.utils.smpl`, ... --
resolves to the MI355X implementation; modules that are not on the hot path fall through to the reference's own package
(its directories are appended to the packages' `__path__`).

One table instead of one re-export file per module: a meta-path finder serves the module names below from the
`uuo_mocap_amd` module that implements them (reference file -> implementation: INTEGRATION.md section 1)."""
import importlib
import importlib.abc
import importlib.util
import os
import sys
import types

_HERE = os.path.dirname(os.path.abspath(__file__))

# reference module path -> (implementing module, names; None = its whole public surface)
_MODULES = {
    "video_mocap.multimodal": ("uuo_mocap_amd.multimodal", None),
    "video_mocap.optimization": ("uuo_mocap_amd.optimization", None),
    "video_mocap.losses.chamfer_distance": ("uuo_mocap_amd.losses", ["weighted_chamfer_distance"]),
    "video_mocap.losses.losses": ("uuo_mocap_amd.losses", ["MarkerLoss"]),
    "video_mocap.markers.markers": ("uuo_mocap_amd.ingest", ["Markers"]),
    "video_mocap.markers.markers_utils": ("uuo_mocap_amd.markers_utils", [
        "filter_rigid", "find_best_part_fits", "segment_rigid", "get_sub_hierachies",
        "remove_approximately_redundant_hierarchies", "get_aabb", "get_aabb_volume"]),
    "video_mocap.img_smpl.img_smpl": ("uuo_mocap_amd.ingest", ["ImgSmpl"]),
    "video_mocap.utils.smpl": ("uuo_mocap_amd.smpl", ["SmplInference", "SmplInferenceGender"]),
    "video_mocap.utils.config": ("uuo_mocap_amd.config", ["load_config"]),
    "video_mocap.utils.settings": ("uuo_mocap_amd.engine", ["MARKER_DISTANCE"]),
    "video_mocap.utils.hmr_utils": ("uuo_mocap_amd.reprojection", [
        "apply_matrix_33_to_vector_3", "convert_hmr_pos_to_mocap_pos", "convert_mocap_pos_to_hmr_pos",
        "get_3d_parameters", "optim_reprojection", "perspective_projection"]),
    "video_mocap.evaluation.metrics": ("uuo_mocap_amd.metrics", [
        "compute_MPJPE", "compute_MPJPE_joints", "compute_MPJVE", "compute_MPJVE_joints", "compute_PA_MPJPE",
        "compute_PA_MPJPE_joints", "compute_PA_MPJVE", "compute_PA_MPJVE_joints", "compute_V2V",
        "compute_marker_to_surface_distance", "compute_similarity_transform"]),
}
_PACKAGES = sorted({name.rsplit(".", 1)[0] for name in _MODULES} - {"video_mocap"})


def _reference_dirs(*sub):
    """The reference's own directories for a (sub)package, wherever `video_mocap` appears later on sys.path."""
    out = []
    for p in sys.path:
        cand = os.path.join(p, "video_mocap", *sub)
        if os.path.isdir(cand) and os.path.abspath(cand) != os.path.join(_HERE, *sub) and cand not in out:
            out.append(cand)
    return out


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname in _MODULES:
            return importlib.util.spec_from_loader(fullname, self)
        if fullname in _PACKAGES:
            return importlib.util.spec_from_loader(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return types.ModuleType(spec.name)

    def exec_module(self, module):
        name = module.__name__
        if name in _PACKAGES:  # a package of ours: its other modules come from the reference's directory
            module.__path__ = _reference_dirs(*name.split(".")[1:])
            return
        impl_name, names = _MODULES[name]
        impl = importlib.import_module(impl_name)
        if names is None:
            names = getattr(impl, "__all__", None) or [n for n in vars(impl) if not n.startswith("_")]
        for n in names:
            setattr(module, n, getattr(impl, n))
        module.__doc__ = "drop-in for the reference's %s: served by %s" % (name, impl_name)


__path__ = [_HERE] + _reference_dirs()
if not any(isinstance(f, _Finder) for f in sys.meta_path):
    sys.meta_path.insert(0, _Finder())
