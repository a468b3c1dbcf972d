"""Diagnostic: BASELINE config 3 shape (k=4, default 5160-strategy grid) timing per kernel."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from farkle_ii_amd.backend import Engine
from farkle_ii_amd.strategies import STRATEGY_DTYPE, default_grid_tuples

tuples = default_grid_tuples()
table = np.zeros(len(tuples), dtype=STRATEGY_DTYPE)
for i, t in enumerate(tuples):
    table[i] = tuple(t)
eng = Engine(0)
for k, n_sh in ((4, 4000), (2, 2000), (8, 4000)):
    for rep in range(2):
        t0 = time.perf_counter()
        r = eng.tournament(table, k, 0, 0, n_sh)
        dt = time.perf_counter() - t0
        t = eng.timing()
    games = n_sh * (5160 // k)
    print(f"k={k} shuffles={n_sh} games={games}: wall {dt*1e3:.1f} ms -> {games/dt/1e6:.1f} M games/s | play {t['play_ms']:.1f} seed {t['seed_ms']:.2f} perm {t['perm_ms']:.2f} ms block {t['play_block']} grid {t['play_grid']} lds {t['play_lds_bytes']}", flush=True)
    tl = r["tally"][0]
    assert (tl[:, 1] == n_sh).all() and (tl[:, 1] == tl[:, 2] + tl[:, 3]).all()
