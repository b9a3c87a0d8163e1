"""Import shim: the product package lives in the directory ``leann-rs_amd/`` (name fixed by the
project layout); a hyphen is not importable, so this module registers that directory as the
package ``leann_rs_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "leann-rs_amd")
_spec = importlib.util.spec_from_file_location(
    "leann_rs_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["leann_rs_amd"] = _mod
_spec.loader.exec_module(_mod)
