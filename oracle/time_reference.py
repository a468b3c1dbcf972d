"""TEST INFRASTRUCTURE ONLY — time the upstream Python reference in the build container (CPU
baseline plan step 1) and commit the numbers as ``tests/golden/reference_cpu_timing.json``.

Measured: (a) the batch leg of ``farkle time --players 2 --n-games 1000 --seed 42`` (time_farkle.py:49-106):
``simulate_many_games(n_games=1000, strategies=make_random_strategies(2, 42), seed=42, n_jobs=1)``;
(b) the tournament loop ``_play_one_shuffle`` (run_tournament.py:301-393) on BASELINE configs[1]'s 64-strategy grid
at k=2, root 42, shuffles 0..N-1 (32 games each), in 1 process and in 8 processes (one contiguous shuffle range
each, the reference's own per-worker decomposition).  Runs only here: ``/root/reference`` does not exist on the
GPU box, where bench.py times the C oracle instead.

    python oracle/time_reference.py [n_shuffles_per_process]
"""
from __future__ import annotations

import json
import multiprocessing as mp
import os
import sys
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))


def _setup():
    import ref_import

    ref_import.import_reference()
    from farkle.simulation import run_tournament as rt
    from farkle.simulation.simulation import generate_strategy_grid, simulate_many_games

    strategies, _ = generate_strategy_grid(
        score_thresholds=[250, 300, 350, 400], dice_thresholds=[0, 1, 2, 3], smart_five_opts=[True],
        smart_one_opts=[True], consider_score_opts=[True], consider_dice_opts=[True], auto_hot_dice_opts=[True],
        run_up_score_opts=[True], include_stop_at=False, include_stop_at_heuristic=False)
    return rt, strategies, simulate_many_games


def _tournament_range(args):
    first, count = args
    rt, strategies, _ = _setup()
    cfg = rt.TournamentConfig(n_players=2, n_strategies=len(strategies))
    rt._init_worker(strategies, cfg, None)
    t0 = time.perf_counter()
    wins_total = 0
    for sh in range(first, first + count):
        task = rt.ShuffleTask(root_seed=42, k=2, shuffle_index=sh, shuffle_seed=0, deterministic_batch_id=0)
        wins, _sums, _sqs = rt._play_one_shuffle(task)[:3]
        wins_total += sum(wins.values())
    return time.perf_counter() - t0, wins_total


def main() -> None:
    n_sh = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    rt, strategies, simulate_many_games = _setup()
    out = {"host": {"cpus": os.cpu_count(), "note": "build container (8 CPUs), Python 3.10, numba absent (identity njit)"}}

    t0 = time.perf_counter()
    from farkle.simulation.time_farkle import make_random_strategies

    df = simulate_many_games(n_games=1000, strategies=make_random_strategies(2, 42), seed=42, n_jobs=1)
    dt = time.perf_counter() - t0
    out["farkle_time_batch_players2_n1000_seed42"] = {"seconds": dt, "games_per_s": 1000 / dt, "rows": int(len(df))}

    dt, wins = _tournament_range((0, n_sh))
    games = n_sh * (len(strategies) // 2)
    out["tournament_loop_1_process"] = {"shuffles": n_sh, "games": games, "seconds": dt, "games_per_s": games / dt,
                                        "completed_games": int(wins)}

    procs = 8
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs) as pool:
        parts = pool.map(_tournament_range, [(p * n_sh, n_sh) for p in range(procs)])
    wall = time.perf_counter() - t0
    inner = max(p[0] for p in parts)
    out["tournament_loop_8_processes"] = {"shuffles": n_sh * procs, "games": games * procs, "seconds_play_max": inner,
                                          "seconds_wall_incl_spawn": wall, "games_per_s": games * procs / inner,
                                          "completed_games": int(sum(p[1] for p in parts))}
    dst = HERE.parent / "tests" / "golden" / "reference_cpu_timing.json"
    dst.write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
