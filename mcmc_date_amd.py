"""Import alias: the package directory is `mcmc-date_amd/` (named after the upstream project, with a
hyphen Python cannot import).  `import mcmc_date_amd` loads that directory as this module."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mcmc-date_amd")
_spec = _ilu.spec_from_file_location(
    "mcmc_date_amd", _os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules["mcmc_date_amd"] = _mod
_spec.loader.exec_module(_mod)
