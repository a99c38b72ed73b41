from uuo_mocap_amd.config import load_config  # noqa: F401
