from uuo_mocap_amd.smpl import SmplInference  # noqa: F401
