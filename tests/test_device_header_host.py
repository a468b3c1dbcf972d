"""CPU check of the scoring code the kernels use: `fk_device.h` is __host__ __device__, so its SWAR scorer, the
score-table entries and the discard table can be exercised on the host against the readable loop form
(`default_score_loops`, the statement of scoring.py:197-366 / 618-693) over the whole input space — 923 multisets x
the 144 valid flag combinations x 6 score thresholds x 7 dice thresholds x 7 turn scores.  No GPU, no oracle."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (shutil.which(HIPCC) or Path(HIPCC).exists()), reason="hipcc not available")
def test_swar_scorer_and_tables_match_loop_form_on_host(tmp_path):
    exe = tmp_path / "device_header_host_check"
    src = ROOT / "tests" / "native" / "device_header_host_check.hip"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-o", str(exe), str(src)], check=True,
                   capture_output=True, text=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "multisets 923" in out.stdout and "bad_swar 0 bad_table 0 bad_lut 0" in out.stdout and "bad_fast 0" in out.stdout
