"""Diagnostic: wall time of Engine.tournament vs device time, per call."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from bench import grid64
from farkle_ii_amd.backend import Engine

table = grid64()
eng = Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 312500
for i in range(5):
    t0 = time.perf_counter()
    r = eng.tournament(table, 2, 42, i * n, (i + 1) * n)
    dt = (time.perf_counter() - t0) * 1e3
    t = eng.timing()
    print(f"call {i}: wall {dt:.2f} ms  device total {t['total_ms']:.2f}  play {t['play_ms']:.2f} seed {t['seed_ms']:.2f} perm {t['perm_ms']:.2f}")
