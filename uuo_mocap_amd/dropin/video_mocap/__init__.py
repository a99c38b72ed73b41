"""Drop-in shim: put this directory in front of the reference's `src/` on sys.path and every hot-path import
(`video_mocap.multimodal`, `video_mocap.optimization`, ...) resolves to the MI355X implementation.  Modules that
are not on the hot path fall through to the reference's own package (its directory is appended to __path__)."""
import os
import sys

__path__ = [os.path.dirname(os.path.abspath(__file__))]
for _p in sys.path:
    _cand = os.path.join(_p, "video_mocap")
    if os.path.isdir(_cand) and os.path.abspath(_cand) != __path__[0] and _cand not in __path__:
        __path__.append(_cand)
