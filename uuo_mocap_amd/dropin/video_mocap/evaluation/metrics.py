from uuo_mocap_amd.metrics import (compute_MPJPE, compute_MPJPE_joints, compute_MPJVE, compute_MPJVE_joints,  # noqa: F401
                                   compute_PA_MPJPE, compute_PA_MPJPE_joints, compute_PA_MPJVE,
                                   compute_PA_MPJVE_joints, compute_V2V, compute_marker_to_surface_distance,
                                   compute_similarity_transform)
