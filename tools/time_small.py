import sys, ctypes
sys.path.insert(0, '.')
from uuo_mocap_amd import _lib
lib = _lib.load_debug()
lib.uuo_debug_time_small.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
for k in (2, 50, 100):
    row = []
    for stop in (1, 2, 3, 4, 0, 201, 202, 203, 204, 205, 200):
        ms = ctypes.c_float()
        rc = lib.uuo_debug_time_small(k, 20, stop, ctypes.byref(ms))
        row.append("stop%d %.1fus" % (stop, ms.value * 1e3) if rc == 0 else "err %s" % lib.uuo_last_error())
    print("k=%d:" % k, "  ".join(row))
