"""Diagnostic (round 3): game-kernel time against resident waves per SIMD at k = 4 and k = 2 on the 5 160-strategy grid — ten-dword
LDS records (`max_waves`) and the cold-in-LDS hot / cold instance (`hot_cold_waves`)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from farkle_ii_amd.backend import Engine
from tools.time_config import table_for

table = table_for(5160)
eng = Engine(0)
def run(k, n_sh, label, opts):
    for name, value in opts.items():
        eng.set_option(name, value)
    best = None
    for rep in range(3):
        eng.tournament(table[:5160 - 5160 % k], k, 0, 0, n_sh)
        t = eng.timing()
        best = t["play_ms"] if best is None else min(best, t["play_ms"])
    lanes = t["play_block"] * t["play_grid"] / 256
    print(f"k={k} {label:34s} play {best:8.3f} ms  block {t['play_block']} grid {t['play_grid']} lds {t['play_lds_bytes']}  waves/SIMD {lanes / 256:.2f}", flush=True)
run(4, 12000, "lds records (1 x 1 024 threads)", dict(hot_cold=0))
run(4, 12000, "cold in LDS, 5 x 256 threads", dict(hot_cold=1, hot_cold_lds=2))
run(4, 12000, "cold in LDS, 4 x 320 threads", dict(hot_cold=1, hot_cold_lds=1))
run(4, 12000, "cold in LDS, 3 x 320 threads", dict(hot_cold=1, hot_cold_lds=1, blocks_per_cu=3))
eng.set_option("blocks_per_cu", 0)
run(3, 9000, "k = 3 lds records", dict(hot_cold=0))
run(3, 9000, "k = 3 cold in LDS, 6 x 256", dict(hot_cold=1, hot_cold_lds=1))
run(3, 9000, "k = 3 cold in LDS, 5 x 256", dict(hot_cold=1, hot_cold_lds=1, blocks_per_cu=5))
eng.set_option("blocks_per_cu", 0)
run(5, 15000, "k = 5 register instance", dict(hot_cold=1, hot_cold_lds=0))
run(5, 15000, "k = 5 cold in LDS, 4 x 256", dict(hot_cold=1, hot_cold_lds=1))
