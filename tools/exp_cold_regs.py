"""Round 4: the cold records of the hot / cold kernel in REGISTERS (no plane: no store + load per turn, no 64-byte fabric write
per 16-byte store) against the plane instances, k = 8 .. 5 on the 5 160-strategy grid.  Tallies must agree."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from farkle_ii_amd.backend import Engine
from tools.time_config import table_for

table = table_for(5160)
eng = Engine(0)
ks = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 7, 6, 5]
for k in ks:
    n_sh = 3000 * k
    ref = None
    table_k = table[:5160 - 5160 % k] if 5160 % k else table
    games = n_sh * (len(table_k) // k)
    variants = [("plane (round 3)", dict(hot_cold_cold_regs=0)), ("cold regs, inc loaded", dict(hot_cold_cold_regs=2))]
    if k == 8:
        variants.append(("cold regs + inc regs", dict(hot_cold_cold_regs=1)))
    variants.append(("plane again", dict(hot_cold_cold_regs=0)))
    for label, opts in variants:
        for name, value in opts.items():
            eng.set_option(name, value)
        best = None
        for rep in range(3):
            r = eng.tournament(table_k, k, 0, 0, n_sh)
            t = eng.timing()
            best = t["play_ms"] if best is None else min(best, t["play_ms"])
        if ref is None:
            ref = r["tally"].copy()
        assert np.array_equal(ref, r["tally"]), label
        print(f"k={k} {label:24s} play {best:8.3f} ms  {games / best / 1e3:8.1f} M games/s  block {t['play_block']} grid {t['play_grid']} lds {t['play_lds_bytes']}", flush=True)
    eng.set_option("hot_cold_cold_regs", 0)
