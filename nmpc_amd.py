"""Import alias: the package directory carries the mandated long name (with hyphens), which
Python cannot import by statement; `import nmpc_amd` loads it under this short name."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "nonlinear-mpc-for-collision-free-and-deadlock-free-navigation-of-multiple-nonholonomic-mobile-robots_amd")
_spec = importlib.util.spec_from_file_location("nmpc_amd", os.path.join(_PKG_DIR, "__init__.py"),
                                               submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nmpc_amd"] = _mod
_spec.loader.exec_module(_mod)
