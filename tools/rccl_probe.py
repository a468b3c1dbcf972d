"""Diagnostic: one-rank RCCL communicator through the C-ABI (NCCL_DEBUG=INFO to see RCCL's own log)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from farkle_ii_amd.backend import Engine
eng = Engine(0)
try:
    eng.comm_init(eng.comm_unique_id(), 0, 1)
    t = np.arange(64 * 26, dtype=np.int64)
    print("reduce ok:", np.array_equal(eng.reduce_tally(t, 0), t))
    eng.comm_destroy()
except Exception as exc:
    print("FAILED:", exc)
