"""YAML configuration with single-parent inheritance (reference utils/config.py:6-18).

``parent:`` is a path; the reference resolves it relative to the CWD (``config/video_mocap.yaml``).  Here it is
tried relative to the CWD first and then relative to the directory that holds the packaged configs, so both
the reference's files and the packaged copies load.  Child keys deep-merge over the parent's (mergedeep
semantics for nested dicts: dicts merge recursively, everything else is replaced).
"""
from __future__ import annotations

import copy
import os
from typing import Dict

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")


def deep_merge(dst: Dict, *srcs: Dict) -> Dict:
    for src in srcs:
        for key, val in src.items():
            if isinstance(val, dict) and isinstance(dst.get(key), dict):
                deep_merge(dst[key], val)
            else:
                dst[key] = copy.deepcopy(val)
    return dst


def _resolve(path: str) -> str:
    if os.path.isfile(path):
        return path
    alt = os.path.join(os.path.dirname(CONFIG_DIR), path)
    if os.path.isfile(alt):
        return alt
    alt = os.path.join(CONFIG_DIR, os.path.basename(path))
    if os.path.isfile(alt):
        return alt
    raise FileNotFoundError(path)


def load_config(filename: str) -> Dict:
    with open(_resolve(filename), "r") as stream:
        try:
            output = yaml.safe_load(stream)
        except yaml.YAMLError as error:  # the reference prints and returns None (utils/config.py:16-18)
            print(error)
            return None
    if output.get("parent") is not None:
        output = deep_merge({}, load_config(output["parent"]), output)
    return output


def packaged_config(name: str = "video_mocap") -> Dict:
    """One of the packaged flag sets: video_mocap | hmr_full | hmr_part | mht_rotation."""
    return load_config(os.path.join(CONFIG_DIR, name + ".yaml"))
