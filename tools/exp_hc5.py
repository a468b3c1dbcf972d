"""Diagnostic (round 3): the cold-in-LDS instance of the hot / cold kernel (32 bytes per seat and lane) at k = 3 .. 5 against
ten-dword LDS records and the register instances."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from farkle_ii_amd.backend import Engine
from tools.time_config import table_for

table = table_for(5160)
eng = Engine(0)
for k, n_sh in ((5, 15000), (4, 12000), (3, 9000)):
    ref = None
    table_k = table[:5160 - 5160 % k] if 5160 % k else table
    games = n_sh * (len(table_k) // k)
    for label, opts in (("lds-records (ten dwords)", dict(hot_cold=0)), ("hc, registers / plane", dict(hot_cold=1, hot_cold_lds=0)),
                        ("hc, cold in LDS", dict(hot_cold=1, hot_cold_lds=1)), ("lds-records again", dict(hot_cold=0))):
        for name, value in opts.items():
            eng.set_option(name, value)
        best = None
        for rep in range(3):
            r = eng.tournament(table_k, k, 0, 0, n_sh)
            t = eng.timing()
            best = t["play_ms"] if best is None else min(best, t["play_ms"])
        if ref is None:
            ref = r["tally"].copy()
        assert np.array_equal(ref, r["tally"]), (k, label)
        print(f"k={k} {label:30s} play {best:8.3f} ms  {games / best / 1e3:8.1f} M games/s  block {t['play_block']} grid {t['play_grid']} lds {t['play_lds_bytes']}", flush=True)
    eng.set_option("hot_cold", -1)
    eng.set_option("hot_cold_lds", -1)
