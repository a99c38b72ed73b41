from uuo_mocap_amd.ingest import ImgSmpl  # noqa: F401
