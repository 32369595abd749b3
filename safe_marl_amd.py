"""Import shim: the package directory is ``safe-marl_amd/`` (hyphenated, as the
build contract names it), which Python cannot import by name.  Importing
``safe_marl_amd`` loads that directory as a regular package under this name.
"""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "safe-marl_amd")
_spec = importlib.util.spec_from_file_location(
    "safe_marl_amd",
    os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir],
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["safe_marl_amd"] = _mod
_spec.loader.exec_module(_mod)
