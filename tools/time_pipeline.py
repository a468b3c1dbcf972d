"""Diagnostic: wall time per step of back-to-back tournament calls with hints, option "pipeline" 0 vs 1.
usage: python tools/time_pipeline.py <grid: 64|5160> <k> <n_shuffles> [steps]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from farkle_ii_amd.backend import Engine
from tools.time_config import table_for

if __name__ == "__main__":
    grid, k, n_sh = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    table = table_for(grid)
    eng = Engine(0)
    for pipeline in (0, 1, 0, 1):
        eng.set_option("pipeline", pipeline)
        eng.tournament(table, k, 7, 0, n_sh)
        t0 = time.perf_counter()
        for i in range(1, steps + 1):
            if i < steps:
                eng.hint_next((i + 1) * n_sh, (i + 2) * n_sh)
            eng.tournament(table, k, 7, i * n_sh, (i + 1) * n_sh)
        dt = (time.perf_counter() - t0) / steps
        print(f"pipeline={pipeline}: {dt * 1e3:.3f} ms per step ({n_sh * (len(table) // k) / dt / 1e6:.1f} M games/s)", flush=True)
