"""Import shim: makes the on-disk package directory ``style-big-gan_amd/`` importable as ``style_big_gan_amd``.

The directory name carries a hyphen (it mirrors the upstream project's name), which Python cannot import
directly; this module replaces itself in ``sys.modules`` with the real package.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "style-big-gan_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
