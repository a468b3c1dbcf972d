"""Diagnostic: per-kernel timing of one tournament shape.
usage: python tools/time_config.py <grid: 64|5160|5148> <k> <n_shuffles> [reps] [root_seed] [rows: 0|1] [opt=value ...]
Prints one line per call; with `rows=1` also the measured R, T (rolls, turns per game) of the launch."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from farkle_ii_amd.backend import Engine
from farkle_ii_amd.strategies import STRATEGY_DTYPE, default_grid_tuples


def table_for(grid: int) -> np.ndarray:
    if grid == 64:
        from bench import grid64
        return grid64()
    tuples = default_grid_tuples()[:grid]  # (5148 = the first strategies of the default grid: divisible by 9 and 11)
    table = np.zeros(len(tuples), dtype=STRATEGY_DTYPE)
    for i, t in enumerate(tuples):
        table[i] = tuple(t)
    return table


if __name__ == "__main__":
    grid, k, n_sh = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    pos = [a for a in sys.argv[4:] if "=" not in a]
    opts = [a.split("=") for a in sys.argv[4:] if "=" in a]
    reps = int(pos[0]) if len(pos) > 0 else 3
    root = int(pos[1]) if len(pos) > 1 else (42 if grid == 64 else 0)
    rows = bool(int(pos[2])) if len(pos) > 2 else False
    table = table_for(grid)
    S = len(table)
    eng = Engine(0)
    for name, value in opts:
        eng.set_option(name, int(value))
    games = n_sh * (S // k)
    for rep in range(reps):
        t0 = time.perf_counter()
        r = eng.tournament(table, k, root, rep * n_sh, (rep + 1) * n_sh, want_rows=rows and rep == reps - 1)
        dt = time.perf_counter() - t0
        t = eng.timing()
        print(f"grid={S} k={k} shuffles={n_sh} games={games}: wall {dt*1e3:.2f} ms ({games/dt/1e6:.1f} M games/s) | device {t['total_ms']:.2f} "
              f"play {t['play_ms']:.2f} seed {t['seed_ms']:.2f} perm {t['perm_ms']:.2f} ms | block {t['play_block']} grid {t['play_grid']} "
              f"lds {t['play_lds_bytes']}" + (f" | clock {t['play_clock_mhz']} MHz blocks end p50 {t['play_block_end_p50_ms']:.2f} max "
                                              f"{t['play_block_end_max_ms']:.2f} ms" if t.get("play_clock_mhz") else ""), flush=True)
        tl = r["tally"][0]
        assert (tl[:, 1] == n_sh).all() and (tl[:, 1] == tl[:, 2] + tl[:, 3]).all()
    if rows:
        R = r["rows"]["seats"]["rolls"].astype(np.int64).sum(axis=1)
        T = r["rows"]["seats"]["n_turns"].astype(np.int64).sum(axis=1)
        W = 229 * R + 30 * T + 850 * k
        print(f"R={R.mean():.2f} T={T.mean():.2f} W={W.mean():.1f} total_rolls={int(R.sum())} max_R={int(R.max())} "
              f"safety={float((r['rows']['status'] == 1).mean()):.4f}", flush=True)
