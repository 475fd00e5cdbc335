"""Importable alias of the hyphenated package directory ``structure-from-motion_amd``:
``import sfm_amd`` gives the package object itself (``sfm_amd.processors``, ``sfm_amd.native`` ...)."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
sys.modules[__name__] = importlib.import_module("structure-from-motion_amd")
