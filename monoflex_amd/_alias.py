"""Top-level import names of the reference (`config`, `model`, `solver`, `engine`, `data`, `utils`, `structures`).

`tools/plain_train_net.py` of the reference imports `from model.detector import KeypointDetector`, `from config import cfg`,
... (plain_train_net.py:9-26).  The tiny top-level packages at the repository root call `install(__name__)`: the name is
bound to the monoflex_amd sub-package of the same name, and a meta-path finder resolves every dotted name below it to the
SAME module object as `monoflex_amd.<name>` (no second copy of a module, relative imports keep working)."""
import importlib
import importlib.abc
import importlib.machinery
import sys

ALIASES = ("config", "model", "solver", "engine", "data", "utils", "structures")


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        head = fullname.partition(".")[0]
        if head in ALIASES and "." in fullname and sys.modules.get(head) is sys.modules.get("monoflex_amd." + head):
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module("monoflex_amd." + spec.name)

    def exec_module(self, module):
        pass


_finder = _AliasFinder()


def install(name):
    if name not in ALIASES:
        raise ImportError("no alias for %r" % name)
    if _finder not in sys.meta_path:
        sys.meta_path.insert(0, _finder)
    real = importlib.import_module("monoflex_amd." + name)
    sys.modules[name] = real
    return real
