"""Import alias: the package directory is named `self-attention-tacotron_amd` (not a Python identifier).
`import satt_amd` (and `satt_amd.<sub>`) resolve to the SAME module objects as
`self-attention-tacotron_amd(.<sub>)` through a meta-path finder, so there is exactly one instance of each module."""
import importlib
import importlib.abc
import importlib.machinery
import sys

_ALIAS = "satt_amd"
_REAL = "self-attention-tacotron_amd"


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(_ALIAS + "."):
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
