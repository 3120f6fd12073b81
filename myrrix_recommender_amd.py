"""Import alias: the package directory is named `myrrix-recommender_amd` (not a valid Python
identifier), so `import myrrix_recommender_amd` loads it from there."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "myrrix-recommender_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
