"""TEST INFRASTRUCTURE ONLY — import the upstream Python reference in the build container.

Used by ``oracle/gen_golden.py`` (fixture generation) and by nothing in the product
path.  ``/root/reference`` does not exist on the GPU box, so nothing under ``tests/``
(-m gpu), ``bench.py`` or ``__graft_entry__.smoke`` may import this module.

The reference targets Python >= 3.12 with numba installed; this container has
Python 3.10 without numba.  The five in-process compatibility shims below are the
ones SURVEY.md §8(c) lists; shim (1) is what the reference's own
``tests/conftest.py:74-113`` does (identity ``njit``).
"""
from __future__ import annotations

import dataclasses
import datetime
import enum
import sys
import types
import typing

REFERENCE_SRC = "/root/reference/src"


def _install_shims() -> None:
    sys.dont_write_bytecode = True
    # (1) identity numba.njit / numba.jit
    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")

        def _identity(*args, **kwargs):
            if len(args) == 1 and callable(args[0]) and not kwargs:
                return args[0]
            return lambda fn: fn

        nb.njit = _identity  # type: ignore[attr-defined]
        nb.jit = _identity  # type: ignore[attr-defined]
        sys.modules["numba"] = nb
    # (2) enum.StrEnum
    if not hasattr(enum, "StrEnum"):
        class StrEnum(str, enum.Enum):
            def __str__(self) -> str:
                return str(self.value)

        enum.StrEnum = StrEnum  # type: ignore[attr-defined]
    # (3) typing extras
    import typing_extensions as te

    for name in ("NotRequired", "Required", "Self", "override", "assert_never", "TypeAlias", "Unpack"):
        if not hasattr(typing, name) and hasattr(te, name):
            setattr(typing, name, getattr(te, name))
    # (4) datetime.UTC
    if not hasattr(datetime, "UTC"):
        datetime.UTC = datetime.timezone.utc  # type: ignore[attr-defined]
    # (5) dataclass(weakref_slot=...) is 3.11+
    if not getattr(dataclasses.dataclass, "_fk_shim", False):
        _orig = dataclasses.dataclass

        def dataclass(cls=None, /, **kwargs):
            if kwargs.pop("weakref_slot", False):
                kwargs["slots"] = False
            if cls is None:
                return lambda c: _orig(c, **kwargs)
            return _orig(cls, **kwargs)

        dataclass._fk_shim = True  # type: ignore[attr-defined]
        dataclasses.dataclass = dataclass  # type: ignore[assignment]


def import_reference():
    """Return the imported ``farkle`` reference package (build container only)."""
    _install_shims()
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import farkle  # noqa: F401

    return farkle
