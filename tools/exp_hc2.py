"""Diagnostic (round 3): hot / cold kernel with LDS tables at k = 8 / 6 against resident blocks per CU (L2 capacity probe)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from farkle_ii_amd.backend import Engine
from tools.time_config import table_for

table = table_for(5160)
eng = Engine(0)
for k, n_sh in ((8, 24000), (7, 21070), (6, 18000), (5, 15000), (4, 12000), (3, 9000)):
    ref = None
    if 5160 % k:
        table_k = table[:5160 - 5160 % k]
    else:
        table_k = table
    games = n_sh * (len(table_k) // k)
    for label, opts in ([("lds-records", dict(hot_cold=0, blocks_per_cu=0))] +
                        [(f"hc+lt {b} blocks/CU", dict(hot_cold=1, blocks_per_cu=b, hot_cold_inc_regs=0)) for b in (3,)] +
                        [(f"hc+lt+regs {b} blocks/CU", dict(hot_cold=1, blocks_per_cu=b, hot_cold_inc_regs=1)) for b in (2, 3, 4)] +
                        [("lds-records again", dict(hot_cold=0, blocks_per_cu=0))]):
        for name, value in opts.items():
            eng.set_option(name, value)
        best = None
        for rep in range(2):
            r = eng.tournament(table_k, k, 0, 0, n_sh)
            t = eng.timing()
            best = t["play_ms"] if best is None else min(best, t["play_ms"])
        if ref is None:
            ref = r["tally"].copy()
        assert np.array_equal(ref, r["tally"])
        print(f"k={k} {label:30s} play {best:8.3f} ms  {games / best / 1e3:8.1f} M games/s  block {t['play_block']} grid {t['play_grid']} lds {t['play_lds_bytes']}", flush=True)
    eng.set_option("blocks_per_cu", 0)
    eng.set_option("hot_cold_inc_regs", 1)
