"""Diagnostic: A/B an fk_set_option setting on BASELINE config 2 (one launch = 10^7 games), interleaved calls.
usage: python tools/ab_option.py <option> <value_a> <value_b> [calls]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from bench import grid64
from farkle_ii_amd.backend import Engine

name, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 6
table = grid64()
eng = Engine(0)
n = 312500
eng.tournament(table, 2, 42, 0, n)
res = {va: [], vb: []}
ref = None
for i in range(calls):
    for v in (va, vb):
        eng.set_option(name, v)
        r = eng.tournament(table, 2, 42, 0, n)
        if ref is None: ref = r["tally"].copy()
        assert np.array_equal(ref, r["tally"])
        res[v].append(eng.timing()["play_ms"])
for v in (va, vb):
    a = np.array(res[v])
    print(f"{name}={v}: play_ms min {a.min():.2f} median {np.median(a):.2f} max {a.max():.2f}")
