from uuo_mocap_amd.losses import weighted_chamfer_distance  # noqa: F401
