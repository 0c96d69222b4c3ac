"""Import helper: the package directory ``fenicsx-fus-gpu_amd/`` is not a valid
Python identifier, so it is registered under the alias ``fenicsx_fus_gpu_amd``."""

import importlib
import importlib.util
import os
import sys

ALIAS = "fenicsx_fus_gpu_amd"
ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "fenicsx-fus-gpu_amd")


def load():
    """Return the package module (imported once, cached in ``sys.modules``)."""
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    spec = importlib.util.spec_from_file_location(
        ALIAS, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod


def submodule(name: str):
    load()
    return importlib.import_module(f"{ALIAS}.{name}")
