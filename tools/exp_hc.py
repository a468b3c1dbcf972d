"""Diagnostic (round 3): hot / cold game kernel variants against the LDS-record kernel on the 5 160-strategy grid, per k.
usage: python tools/exp_hc.py [shuffles_for_k4] [k ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from farkle_ii_amd.backend import Engine
from tools.time_config import table_for

table = table_for(5160)
eng = Engine(0)
base = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
ks = [int(x) for x in sys.argv[2:]] or [8, 6, 4]
variants = [("lds-records", dict(hot_cold=0))]
for block, waves in ((256, 5), (256, 4), (256, 3), (768, 6), (768, 3), (1024, 4)):
    for tables in (0, 1):
        variants.append((f"hc b{block} w{waves} t{tables}", dict(hot_cold=1, hot_cold_block=block, hot_cold_waves=waves, hot_cold_tables=tables)))
for k in ks:
    n_sh = base * k // 4  # the same number of seat exposures per launch
    ref = None
    games = n_sh * (5160 // k)
    for label, opts in variants:
        for name, value in opts.items():
            eng.set_option(name, value)
        best = None
        for rep in range(2):
            r = eng.tournament(table, k, 0, 0, n_sh)
            t = eng.timing()
            best = t["play_ms"] if best is None else min(best, t["play_ms"])
        if ref is None:
            ref = r["tally"].copy()
        assert np.array_equal(ref, r["tally"]), "tally changed"
        print(f"k={k} {label:18s} play {best:8.3f} ms  {games / best / 1e3:8.1f} M games/s  "
              f"block {t['play_block']} grid {t['play_grid']} lds {t['play_lds_bytes']}", flush=True)
