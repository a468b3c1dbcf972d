"""Diagnostic: batched H2H throughput — n blocks of `target` completed games in shared launches (fk_h2h_run_blocks).
usage: python tools/time_h2h_blocks.py [n_blocks=10000] [target=2191] [reps=3]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from farkle_ii_amd.backend import Engine
from tools.time_config import table_for

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
target = int(sys.argv[2]) if len(sys.argv) > 2 else 2191
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
table = table_for(5160)
rng = np.random.default_rng(1)
idx = rng.integers(0, len(table), (n, 2))
eng = Engine(0)
for rep in range(reps):
    t0 = time.perf_counter()
    st = eng.h2h_blocks(table[idx], 42 + rep, np.arange(n) // 2, np.arange(n) % 2, target, 2 * target)
    dt = time.perf_counter() - t0
    t = eng.timing()
    att = int(st[:, 0].sum())
    print(f"{n} blocks x target {target}: {att} attempts, {int(st[:, 1].sum())} completed, wall {dt*1e3:.1f} ms -> {att/dt/1e6:.1f} M attempts/s | "
          f"launches {t['play_launches']} play {t['play_ms']:.1f} seed {t['seed_ms']:.2f} ms | unfinished blocks {(st[:, 1] < target).sum()}", flush=True)
