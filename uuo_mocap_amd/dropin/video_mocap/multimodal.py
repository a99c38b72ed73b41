from uuo_mocap_amd.multimodal import *  # noqa: F401,F403
from uuo_mocap_amd.multimodal import multimodal_video_mocap, pad, normalize_rot  # noqa: F401
