from uuo_mocap_amd.optimization import *  # noqa: F401,F403
from uuo_mocap_amd.optimization import (optim_root, optim_chamfer, optim_markers, compute_nearest_points,  # noqa: F401
                                        compute_marker_labels_from_coords, compute_root_orient_y, compute_root_orient_z,
                                        chamfer_distance_by_part, get_marker_mask, weighted_mse_loss,
                                        weighted_chamfer_distance, MarkerLoss)
