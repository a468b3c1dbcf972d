"""Diagnostic: steps per second with 1 vs 2 engine contexts in flight on one GPU (two host threads, alternate steps).
usage: python tools/time_inflight.py <grid: 64|5160> <k> <n_shuffles> [steps]"""
import sys, time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from farkle_ii_amd.backend import Engine
from tools.time_config import table_for

if __name__ == "__main__":
    grid, k, n_sh = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    table = table_for(grid)
    games = n_sh * (len(table) // k)
    for n_eng in (1, 2, 1, 2, 3):
        engs = [Engine(0) for _ in range(n_eng)]

        def run(i):
            e = engs[i % n_eng]
            if i + n_eng <= steps:
                e.hint_next((i + n_eng) * n_sh, (i + n_eng + 1) * n_sh)
            return e.tournament(table, k, 7, i * n_sh, (i + 1) * n_sh)["tally"][0]

        for i in range(n_eng):
            engs[i].tournament(table, k, 7, 0, n_sh)  # warm
        with ThreadPoolExecutor(max_workers=n_eng) as pool:
            t0 = time.perf_counter()
            tot = sum(pool.map(run, range(1, steps + 1)))
            dt = (time.perf_counter() - t0) / steps
        assert int(tot[:, 1].sum()) == steps * games * k
        print(f"{n_eng} in flight: {dt * 1e3:.3f} ms per step ({games / dt / 1e6:.1f} M games/s)", flush=True)
        del engs
