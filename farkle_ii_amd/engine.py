"""Process-wide default engine (one context per process / GPU)."""
from __future__ import annotations

import os

from .backend import Engine

_ENGINE: Engine | None = None


def get_engine(device: int | None = None) -> Engine:
    """Return the process's engine, creating it on ``device`` (default: LOCAL_RANK or 0) on first use.

    There is no CPU fallback: this raises :class:`farkle_ii_amd.backend.FarkleHipError` without a GPU."""
    global _ENGINE
    if _ENGINE is None:
        ordinal = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)
        _ENGINE = Engine(ordinal)
    return _ENGINE


def set_engine(engine: Engine | None) -> None:
    global _ENGINE
    _ENGINE = engine
