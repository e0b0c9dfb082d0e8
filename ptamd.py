"""Import shim: loads the package directory `owl-path-tracer_amd/` under the importable name
`owl_path_tracer_amd` (the hyphen in the directory name is mandated by the repo layout)."""
import importlib.util
import os
import sys

_NAME = "owl_path_tracer_amd"
ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "owl-path-tracer_amd")


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
