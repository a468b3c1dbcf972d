"""Diagnostic: H2H block throughput (BASELINE config 5 shape) and rows-mode throughput."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from bench import grid64
from farkle_ii_amd.backend import Engine

table = grid64()
eng = Engine(0)
for target in (10**6, 10**7, 10**8):
    t0 = time.perf_counter()
    st = eng.h2h(table[[3, 40]], 42, 5, 0, target, 2 * target, 10**10)
    dt = time.perf_counter() - t0
    print(f"h2h target {target}: state {st.tolist()} in {dt*1e3:.1f} ms -> {int(st[0])/dt/1e6:.1f} M attempts/s", flush=True)
for n_sh in (31250, 312500):
    t0 = time.perf_counter()
    r = eng.tournament(table, 2, 42, 0, n_sh, want_rows=True)
    dt = time.perf_counter() - t0
    t = eng.timing()
    print(f"rows mode {n_sh*32} games: wall {dt*1e3:.1f} ms ({n_sh*32/dt/1e6:.1f} M games/s), play {t['play_ms']:.1f} ms, rows {r['rows'].nbytes/1e6:.0f} MB", flush=True)
