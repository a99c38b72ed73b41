from uuo_mocap_amd.markers_utils import (filter_rigid, find_best_part_fits, segment_rigid, get_sub_hierachies,  # noqa: F401
                                         remove_approximately_redundant_hierarchies, get_aabb, get_aabb_volume)
