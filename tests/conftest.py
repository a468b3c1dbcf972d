"""Shared pytest configuration.

Markers
-------
gpu : needs a real MI355X (run with ``-m gpu`` on the GPU box).  Everything else runs on CPU.
"""
from __future__ import annotations

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "oracle", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU (HIP path through the C-ABI)")


@pytest.fixture(scope="session")
def golden_dir() -> Path:
    return ROOT / "tests" / "golden"
