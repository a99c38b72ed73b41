import os
import sys

__path__ = [os.path.dirname(os.path.abspath(__file__))]
for _p in sys.path:
    _cand = os.path.join(_p, "video_mocap", "losses")
    if os.path.isdir(_cand) and os.path.abspath(_cand) != __path__[0] and _cand not in __path__:
        __path__.append(_cand)
