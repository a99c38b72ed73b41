from uuo_mocap_amd.smpl import SmplInference, SmplInferenceGender  # noqa: F401
