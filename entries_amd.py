"""Import shim: ``import entries_amd`` loads the package in ``2022-entries_amd/`` (whose name is not an identifier)."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "2022-entries_amd")
_spec = importlib.util.spec_from_file_location(
    "entries_amd_pkg", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["entries_amd_pkg"] = _mod
_spec.loader.exec_module(_mod)
globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("__")})
package = _mod
PACKAGE_DIR = _pkg_dir
