"""A/B (GPU): the Fisher-Yates draws of a chunk's shuffles by one thread per shuffle (option perm_draw_wave = 0) and by a wave per shuffle
(1, csrc/fk_perm_wave.h) — permutations and tallies compared, call wall time and the engine's perm_ms / play_ms at the launch sizes of
rows mode (800 - 930 shuffles of the 5 160-strategy grid) and beyond.  usage: python tools/ab_perm_draw_wave.py"""
import sys, time, numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from farkle_ii_amd.backend import Engine
from farkle_ii_amd.strategies import generate_strategy_grid, pack_strategies, prepare_public_helper_strategies
table = pack_strategies(prepare_public_helper_strategies(generate_strategy_grid()[0]))
eng = Engine(0)
for k, n_sh in ((2, 800), (5, 900), (12, 930), (4, 5000), (2, 30000)):
    out = {}
    for wave in (0, 1):
        eng.set_option("perm_draw_wave", wave)
        res = eng.tournament(table, k, 102, 0, n_sh, want_perms=True)
        ts = []
        for g in range(1, 6):
            eng.hint_next((g + 1) * n_sh, (g + 2) * n_sh)
            t0 = time.perf_counter(); eng.tournament(table, k, 102, g * n_sh, (g + 1) * n_sh); ts.append((time.perf_counter() - t0) * 1e3)
        t = eng.timing()
        out[wave] = (res["perms"], res["tally"], min(ts), t["perm_ms"], t["play_ms"])
    same = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    print(f"k={k} n_sh={n_sh}: perms+tally equal {same}; call wall thread {out[0][2]:.2f} ms -> wave {out[1][2]:.2f} ms; perm_ms {out[0][3]:.2f} -> {out[1][3]:.2f}; play {out[0][4]:.2f} / {out[1][4]:.2f}", flush=True)
