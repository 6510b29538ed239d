"""Emit pickles that stable-baselines3 / sb3-contrib / gym classes will revive — without those packages.

SB3's model zips and ``VecNormalize`` pickles reference classes by module path
(``gym.spaces.box.Box``, ``stable_baselines3.common.running_mean_std.RunningMeanStd`` …).  pickle
writes such a reference as (module, qualname) after checking that ``sys.modules[module].qualname`` is
the object being pickled, so a stand-in class registered under a temporary stand-in module produces
byte streams with exactly the references a real installation expects (SURVEY.md A.1/A.2 show the same
structure in the reference's artifacts).  Instances are written as NEWOBJ + state dict, which is what
the default ``object.__reduce_ex__`` of the real classes produced in those artifacts.
If the real module is importable it is used instead of a stand-in.
"""
from __future__ import annotations

import collections
import contextlib
import importlib
import sys
import types
from typing import Dict, Iterable, Tuple


def _real(module: str, name: str):
    try:
        return getattr(importlib.import_module(module), name)
    except Exception:
        return None


@contextlib.contextmanager
def stand_ins(specs: Iterable[Tuple[str, str, str]]):
    """specs: (module, qualname, kind) with kind in {"object", "namedtuple:<f1>,<f2>,..."}.
    Yields {(module, qualname): class}.  Stand-in modules are removed from sys.modules afterwards."""
    created, out = [], {}
    try:
        for module, name, kind in specs:
            cls = _real(module, name)
            if cls is None:
                parts = module.split(".")
                for i in range(1, len(parts) + 1):
                    m = ".".join(parts[:i])
                    if m not in sys.modules:
                        sys.modules[m] = types.ModuleType(m)
                        created.append(m)
                if kind.startswith("namedtuple:"):
                    cls = collections.namedtuple(name, kind.split(":", 1)[1].split(","))
                else:
                    cls = type(name, (object,), {})
                cls.__module__, cls.__qualname__ = module, name
                setattr(sys.modules[module], name, cls)
            out[(module, name)] = cls
        yield out
    finally:
        for m in reversed(created):
            sys.modules.pop(m, None)


def instance(cls, state: Dict):
    """An instance of ``cls`` carrying ``state`` as its __dict__ (pickles as NEWOBJ + BUILD)."""
    obj = cls.__new__(cls)
    obj.__dict__.update(state)
    return obj
