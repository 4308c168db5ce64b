"""Test-side access to the package's ctypes binding (c-kzg-4844_amd/ckzg.py).  The package
directory name is not an importable identifier, so it is loaded by path."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PKG = os.path.join(ROOT, "c-kzg-4844_amd")


def _load():
    name = "ckzg_4844_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(_PKG, "__init__.py"),
                                                  submodule_search_locations=[_PKG])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_m = _load()
Kzg = _m.Kzg
KzgError = _m.KzgError
KZGSettings = _m.KZGSettings
HIP_SO = _m.HIP_SO
TRUSTED_SETUP = _m.TRUSTED_SETUP
BYTES_PER_BLOB = 131072
BYTES_PER_CELL = 2048
