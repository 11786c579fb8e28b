"""Makes the hyphenated package directory `chrono-wisc_amd/` importable as `chrono_wisc_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "chrono-wisc_amd")


def load():
    if "chrono_wisc_amd" in sys.modules:
        return sys.modules["chrono_wisc_amd"]
    spec = importlib.util.spec_from_file_location("chrono_wisc_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["chrono_wisc_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
