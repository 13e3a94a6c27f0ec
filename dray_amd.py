"""Importable alias of the ``distributed-raytracer_amd`` package (its name has a hyphen)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)

_pkg = importlib.import_module("distributed-raytracer_amd")
sys.modules[__name__] = _pkg
