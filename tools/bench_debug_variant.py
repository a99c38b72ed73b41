#!/usr/bin/env python3
"""bench.py on the DEBUG flavour of the library with a kernel-variant knob set (timing ablations only: the numbers a knob
produces are not fits of anything).   python tools/bench_debug_variant.py UUO_SK2_VAR=4 -- --steps 4 --warmup 1 ..."""
import os
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
args = sys.argv[1:]
while args and args[0] != "--":
    k, v = args.pop(0).split("=", 1)
    os.environ[k] = v
if args and args[0] == "--":
    args.pop(0)
from uuo_mocap_amd import _lib  # noqa: E402

_lib.LIB_PATH = _lib.LIB_DEBUG_PATH
sys.argv = [os.path.join(root, "bench.py")] + args
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
