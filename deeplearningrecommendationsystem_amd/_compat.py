"""Import-name boundary: lets the reference's own import lines (``from model.pnn import PNN``,
``from trainer.trainer import Trainer`` -- scripts/pnn.py:7-13) resolve to this package.

The mirrors use package-relative imports, so they cannot simply be put on ``sys.path`` under
another top-level name.  A shim package ``compat/<name>/__init__.py`` calls ``alias(...)``:
every submodule of the real package is registered in ``sys.modules`` under the reference's
dotted name (same module object, loaded once), and the shim's ``__path__`` is extended over
``sys.path`` so that submodules the build does not replace (``evaluator.ranking``) still come
from wherever the caller has them."""
from __future__ import annotations

import importlib
import pkgutil
import sys


def alias(shim_name: str, shim_globals: dict, real_name: str) -> None:
    real = importlib.import_module(real_name)
    for info in pkgutil.iter_modules(real.__path__):
        if info.name.startswith("_"):
            continue
        mod = importlib.import_module(f"{real_name}.{info.name}")
        sys.modules[f"{shim_name}.{info.name}"] = mod
        shim_globals[info.name] = mod
    for k in getattr(real, "__all__", ()):
        shim_globals[k] = getattr(real, k)
    shim_globals["__path__"] = pkgutil.extend_path(shim_globals["__path__"], shim_name)
