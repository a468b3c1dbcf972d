"""Diagnostic (round 3): k = 8 as one 896-thread block per CU (3.5 waves per SIMD, 128 registers) against three 256-thread blocks."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from farkle_ii_amd.backend import Engine
from tools.time_config import table_for

table = table_for(5160)
eng = Engine(0)
ref = None
for label, opts in (("3 x 256 threads (three waves)", dict(hot_cold_inc_regs=1)), ("1 x 896 threads (3.5 waves)", dict(hot_cold_inc_regs=2)),
                    ("3 x 256 threads again", dict(hot_cold_inc_regs=1))):
    for name, value in opts.items():
        eng.set_option(name, value)
    best = None
    for rep in range(3):
        r = eng.tournament(table, 8, 0, 0, 24000)
        t = eng.timing()
        best = t["play_ms"] if best is None else min(best, t["play_ms"])
    if ref is None:
        ref = r["tally"].copy()
    assert np.array_equal(ref, r["tally"])
    print(f"k=8 {label:32s} play {best:8.3f} ms  block {t['play_block']} grid {t['play_grid']} lds {t['play_lds_bytes']}", flush=True)
