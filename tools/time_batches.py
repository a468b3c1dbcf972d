"""Diagnostic: BASELINE config 2 launch (10^7 games) with per-batch tallies (global int64 atomics) vs one batch (LDS tally)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from bench import grid64
from farkle_ii_amd.backend import Engine

table = grid64()
eng = Engine(0)
n = 312500
ref = eng.tournament(table, 2, 42, 0, n)["tally"][0]
for spb in (n, 31250, 3125, 625, 125):
    for rep in range(3):
        r = eng.tournament(table, 2, 42, 0, n, shuffles_per_batch=spb)
        t = eng.timing()
    assert np.array_equal(r["tally"].sum(axis=0), ref)
    print(f"shuffles_per_batch={spb:7d} batches={r['tally'].shape[0]:5d}: play {t['play_ms']:.2f} ms total {t['total_ms']:.2f} ms lds {t['play_lds_bytes']}", flush=True)
