"""Import the hyphenated directory ``mc-gra_amd/`` as package ``mc_gra_amd``."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "mc-gra_amd")


def load():
    if "mc_gra_amd" in sys.modules:
        return sys.modules["mc_gra_amd"]
    spec = importlib.util.spec_from_file_location(
        "mc_gra_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["mc_gra_amd"] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop("mc_gra_amd", None)
        raise
    return mod
