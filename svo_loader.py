"""Import helper for the hyphen-named package directory ``stereo-semantic-vo_amd``.

``load()`` returns the package module (registered as ``stereo_semantic_vo_amd``).
"""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(_ROOT, "stereo-semantic-vo_amd")
MOD_NAME = "stereo_semantic_vo_amd"


def load():
    if MOD_NAME in sys.modules:
        return sys.modules[MOD_NAME]
    spec = importlib.util.spec_from_file_location(
        MOD_NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[MOD_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
