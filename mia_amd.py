"""Import alias: the package lives in `medical-image-analysis_amd/` (not a valid Python identifier),
so `import mia_amd` loads that directory as the package `mia_amd`."""
import importlib.util
import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "medical-image-analysis_amd")
_spec = importlib.util.spec_from_file_location("mia_amd", os.path.join(_PKG, "__init__.py"),
                                               submodule_search_locations=[_PKG])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mia_amd"] = _mod
_spec.loader.exec_module(_mod)
