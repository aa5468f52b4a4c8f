"""Loads the package directory `multiple-objects-gan_amd/` (a hyphenated name cannot be
imported directly) under the alias `mogan_amd`."""
import importlib.util
import os
import sys

_ALIAS = "mogan_amd"


def load():
    if _ALIAS in sys.modules:
        return sys.modules[_ALIAS]
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multiple-objects-gan_amd")
    spec = importlib.util.spec_from_file_location(
        _ALIAS, os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
