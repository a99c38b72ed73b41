from uuo_mocap_amd.losses import MarkerLoss  # noqa: F401
