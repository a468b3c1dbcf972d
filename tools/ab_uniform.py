"""Diagnostic: scalar-flag (uniform_flags) kernel instance vs the generic one on a 64-strategy table whose
strategies all share the same flags (config-2 table with require_both / favor fixed)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from bench import grid64
from farkle_ii_amd.backend import Engine

table = grid64().copy()
table["require_both"] = 1
table["favor_score"] = 1
table["score_threshold"] = 250 + 25 * (np.arange(64) // 4)
table["dice_threshold"] = np.arange(64) % 4
eng = Engine(0)
n = 312500
eng.tournament(table, 2, 42, 0, n)
res = {0: [], -1: []}
ref = None
for i in range(6):
    for v in (0, -1):
        eng.set_option("uniform_flags", v)
        r = eng.tournament(table, 2, 42, 0, n)
        if ref is None: ref = r["tally"].copy()
        assert np.array_equal(ref, r["tally"])
        res[v].append(eng.timing()["play_ms"])
for v in (0, -1):
    a = np.array(res[v])
    print(f"uniform_flags={v}: play_ms min {a.min():.2f} median {np.median(a):.2f} max {a.max():.2f}")
