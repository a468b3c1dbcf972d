"""Diagnostic (round 4): what the lag post-pass of fk_tournament_run_lags costs beside the plain call, BASELINE config 3 shape by
default (5 160 strategies, k = 4).  usage: python tools/time_lags.py [grid=5160] [k=4] [n_shuffles=77520]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from farkle_ii_amd.backend import Engine
from tools.time_config import table_for

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 5160
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_sh = int(sys.argv[3]) if len(sys.argv) > 3 else 77520
table = table_for(grid)
eng = Engine(0)
games = n_sh * (len(table) // k)
for label, lags in (("counts only", None), ("lags (1,)", (1,)), ("lags (1, 2, 5)", (1, 2, 5)), ("lags (1 .. 16)", tuple(range(1, 17))), ("counts only", None)):
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        r = eng.tournament(table, k, 0, rep * n_sh, (rep + 1) * n_sh) if lags is None else eng.tournament_lags(table, k, 0, rep * n_sh, (rep + 1) * n_sh, lags)
        dt = time.perf_counter() - t0
        t = eng.timing()
        rec = (dt * 1e3, t["total_ms"], t["play_ms"])
        best = rec if best is None or rec[0] < best[0] else best
    extra = "" if lags is None else f"  value matrix {n_sh * len(table) * 2 / 1e6:.0f} MB, pairs per strategy at lag 1: {int(r['lag_sums'][0, 0, 0])}"
    print(f"grid {len(table)} k={k} shuffles={n_sh} games={games:.3g}  {label:16s} wall {best[0]:8.2f} ms  device {best[1]:8.2f}  game kernel {best[2]:8.2f}{extra}", flush=True)
