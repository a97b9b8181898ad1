"""`src` package of the drop-in: registers the kernel package under its import name `mia_amd` (the directory name
holds a hyphen, so it cannot be imported by name) before src.model / src.loss use it."""
import importlib.util
import os
import sys

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "mia_amd" not in sys.modules:
    _spec = importlib.util.spec_from_file_location("mia_amd", os.path.join(_PKG, "__init__.py"),
                                                   submodule_search_locations=[_PKG])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["mia_amd"] = _mod
    _spec.loader.exec_module(_mod)
