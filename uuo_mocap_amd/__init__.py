"""MI355X-native SMPL-to-unlabeled-marker fitting (hot path of UUO-Mocap), HIP kernels behind a C ABI."""
__version__ = "0.1.0"
