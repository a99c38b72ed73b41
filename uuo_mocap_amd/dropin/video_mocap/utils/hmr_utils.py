from uuo_mocap_amd.reprojection import (apply_matrix_33_to_vector_3, convert_hmr_pos_to_mocap_pos,  # noqa: F401
                                        convert_mocap_pos_to_hmr_pos, get_3d_parameters, optim_reprojection,
                                        perspective_projection)
