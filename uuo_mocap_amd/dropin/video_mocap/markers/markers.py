from uuo_mocap_amd.ingest import Markers  # noqa: F401
