"""``farkle time``: random strategies -> one game + N games timing (``src/farkle/simulation/time_farkle.py:23-131``)."""
from __future__ import annotations

import logging
import time

from .random import RandomPurpose, coordinate_rng
from .simulation import simulate_many_games, simulate_one_game
from .strategies import ThresholdStrategy, random_threshold_strategy

LOGGER = logging.getLogger(__name__)


def make_random_strategies(num_players: int, seed: int | None) -> list[ThresholdStrategy]:
    if seed is None:
        raise ValueError("make_random_strategies requires an explicit seed")
    return [random_threshold_strategy(coordinate_rng(RandomPurpose.STRATEGY, root_seed=seed, k=num_players, seat_index=i))
            for i in range(num_players)]


def measure_sim_times(*, n_games: int = 1000, players: int = 5, seed: int = 42, jobs: int = 1) -> dict:
    """Benchmark single-game and multi-game simulation; logs like the reference and returns the numbers."""
    LOGGER.info("Simulation timing start", extra={"stage": "simulation", "benchmark": "time_farkle", "players": players,
                                                   "seed": seed, "jobs": jobs, "n_games": n_games})
    strategies = make_random_strategies(players, seed)
    t0 = time.perf_counter()
    row = simulate_one_game(strategies=strategies, seed=seed)
    single_game_s = time.perf_counter() - t0
    LOGGER.info("Single game benchmark", extra={"stage": "simulation", "benchmark": "single_game", "players": players,
                                                "seed": seed, "elapsed_s": single_game_s, "winner": row["winner_seat"],
                                                "winning_score": row["winning_score"], "rounds": row["n_rounds"]})
    t0 = time.perf_counter()
    df = simulate_many_games(n_games=n_games, strategies=strategies, seed=seed, n_jobs=jobs)
    elapsed = time.perf_counter() - t0
    gps = (n_games / elapsed) if elapsed > 0 else 0.0
    # the reference reads df["winner"] here (time_farkle.py:103, a latent bug: rows carry "winner_seat")
    winners = df["winner_seat"].value_counts().to_dict() if len(df) else {}
    LOGGER.info("Batch benchmark", extra={"stage": "simulation", "benchmark": "batch", "players": players, "seed": seed,
                                          "jobs": jobs, "n_games": n_games, "elapsed_s": elapsed, "games_per_sec": gps,
                                          "winners": winners})
    return {"single_game_s": single_game_s, "elapsed_s": elapsed, "games_per_sec": gps, "winners": winners,
            "single_game_winner": row["winner_seat"]}


__all__ = ["measure_sim_times", "make_random_strategies"]
