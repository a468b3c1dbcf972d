"""Batch simulation helpers on the GPU engine.

Mirrors ``src/farkle/simulation/simulation.py``: ``PlayerRngCoordinates`` :333-358, ``_play_game`` :576-655,
``simulate_many_games`` :658-722, ``simulate_many_games_from_seeds`` :725-786, ``simulate_one_game`` :789-812.
Same names, argument meaning and error behaviour; the games themselves run in ``fk_play_games``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Iterable, Mapping, Sequence

import numpy as np

from .backend import COORD_DTYPE, FarkleHipError, FK_ERR_ROLL_LIMIT
from .engine import get_engine
from .random import RNG_SCHEME_VERSION, RandomPurpose, spawn_seeds
from .rows import row_to_dict, validate_simulation_row
from .strategies import (ThresholdStrategy, generate_strategy_grid, pack_strategies,  # noqa: F401 (re-export)
                         prepare_public_helper_strategies)

__all__ = ["PlayerRngCoordinates", "_play_game", "simulate_many_games", "simulate_many_games_from_seeds",
           "simulate_one_game", "generate_strategy_grid", "aggregate_metrics", "play_coordinate_games"]


@dataclass(frozen=True)
class PlayerRngCoordinates:
    """Complete semantic coordinates used to construct every seat stream."""

    purpose: RandomPurpose
    root_seed: int
    k: int
    shuffle_index: int = 0
    pair_id: int = 0
    order: int = 0
    game_index: int | None = None
    attempt_index: int | None = None

    def as_record(self) -> tuple:
        if self.game_index is not None and self.attempt_index is not None and int(self.game_index) != int(self.attempt_index):
            raise ValueError("game_index and attempt_index identify different coordinates")
        game = self.game_index if self.game_index is not None else (self.attempt_index or 0)
        return (int(RandomPurpose(int(self.purpose))), 0, int(self.root_seed), int(self.k), int(self.shuffle_index),
                int(self.pair_id), int(self.order), int(game), 0, 0)


def play_coordinate_games(coordinates: Sequence[PlayerRngCoordinates], strategies: Sequence[ThresholdStrategy],
                          seat_strategy: np.ndarray, *, target_score: int = 10_000, max_rounds: int = 200,
                          engine=None) -> np.ndarray:
    """Device rows for explicit (coordinate, seating) pairs; ``seat_strategy[g]`` indexes ``strategies``."""
    eng = engine or get_engine()
    k = np.asarray(seat_strategy).shape[-1]
    coords = np.zeros(len(coordinates), dtype=COORD_DTYPE)
    for i, c in enumerate(coordinates):
        if c.k != k:
            raise ValueError("Player RNG coordinate k does not match the number of seated strategies")
        coords[i] = c.as_record()
    try:
        return eng.play_games(coords, pack_strategies(strategies), seat_strategy, k, target_score, max_rounds)
    except FarkleHipError as exc:
        if exc.code == FK_ERR_ROLL_LIMIT:  # engine.py:242-243 raises RuntimeError
            raise RuntimeError(str(exc)) from exc
        raise


def _play_game(seed: int, strategies: Sequence[ThresholdStrategy], target_score: int = 10_000,
               provenance: Mapping[str, Any] | None = None, max_rounds: int = 200,
               player_rng_coordinates: PlayerRngCoordinates | None = None) -> Mapping[str, Any]:
    """Play a single game and return the flattened row mapping (one game = one kernel lane)."""
    k = len(strategies)
    coords = player_rng_coordinates or PlayerRngCoordinates(purpose=RandomPurpose.PLAYER, root_seed=seed, k=k)
    if coords.k != k:
        raise ValueError("Player RNG coordinate k does not match the number of seated strategies")
    rows = play_coordinate_games([coords], strategies, np.arange(k, dtype=np.int32)[None, :], target_score=target_score,
                                 max_rounds=max_rounds)
    ids = [(-1 - i) if s.strategy_id is None else int(s.strategy_id) for i, s in enumerate(strategies)]
    flat = row_to_dict(rows[0], k, ids, {
        "root_seed": seed, "k": k, "shuffle_index": None, "game_index": None, "deterministic_batch_id": None,
        "game_seed": seed, "rng_scheme_version": RNG_SCHEME_VERSION, "rng_purpose_namespace": int(RandomPurpose.INDEXED_SEED)})
    for i, s in enumerate(strategies):  # strategies without an id are reported by their string form (engine.py:500)
        if s.strategy_id is None:
            flat[f"P{i + 1}_strategy"] = str(s)
            if flat["winner_seat"] == f"P{i + 1}":
                flat["winner_strategy"] = str(s)
    if provenance is not None:
        flat.update(provenance)
    if all(s.strategy_id is not None for s in strategies):
        validate_simulation_row(flat)
    return flat


def _many(seeds: np.ndarray, coords: list[PlayerRngCoordinates], strategies, target_score, root_of, index_of):
    import pandas as pd

    resolved = prepare_public_helper_strategies(strategies)
    k = len(resolved)
    n = len(coords)
    if n == 0:
        return pd.DataFrame([])
    seat = np.tile(np.arange(k, dtype=np.int32), (n, 1))
    rows = play_coordinate_games(coords, resolved, seat, target_score=target_score)
    ids = [int(s.strategy_id) for s in resolved]
    out = []
    for i in range(n):
        flat = row_to_dict(rows[i], k, ids, {
            "root_seed": root_of(i), "k": k, "shuffle_index": None, "game_index": index_of(i), "deterministic_batch_id": None,
            "game_seed": int(seeds[i]), "rng_scheme_version": RNG_SCHEME_VERSION,
            "rng_purpose_namespace": int(RandomPurpose.INDEXED_SEED)})
        validate_simulation_row(flat)
        out.append(flat)
    return pd.DataFrame(out)


def simulate_many_games(*, n_games: int, strategies: Sequence[ThresholdStrategy], target_score: int = 10_000,
                        seed: int | None = None, n_jobs: int = 1):
    """Run many games and return one DataFrame row per game.  ``n_jobs`` is accepted for signature
    compatibility; all games run in one kernel launch."""
    del n_jobs
    if seed is None:
        raise ValueError("simulate_many_games requires an explicit seed")
    k = len(strategies)
    seeds = spawn_seeds(n_games, seed=seed)
    coords = [PlayerRngCoordinates(purpose=RandomPurpose.PLAYER, root_seed=seed, k=k, game_index=i) for i in range(n_games)]
    return _many(seeds, coords, strategies, target_score, lambda i: seed, lambda i: i)


def simulate_many_games_from_seeds(*, seeds: Iterable[int], strategies: Sequence[ThresholdStrategy],
                                   target_score: int = 10_000, n_jobs: int = 1, root_seed: int | None = None):
    del n_jobs
    seeds = [int(s) for s in seeds]
    k = len(strategies)
    coords = [PlayerRngCoordinates(purpose=RandomPurpose.PLAYER, root_seed=s if root_seed is None else root_seed, k=k,
                                   game_index=0 if root_seed is None else i) for i, s in enumerate(seeds)]
    return _many(np.asarray(seeds, dtype=np.uint64), coords, strategies, target_score,
                 lambda i: seeds[i] if root_seed is None else root_seed, lambda i: i)


def simulate_one_game(*, strategies: Sequence[ThresholdStrategy], target_score: int = 10_000, seed: int) -> Mapping[str, Any]:
    """Play one game with PLAYER-namespace streams rooted at ``seed``; returns the flat row mapping."""
    return _play_game(seed, strategies, target_score=target_score)


def aggregate_metrics(df) -> Mapping[str, Any]:
    return {"games": len(df), "avg_rounds": df["n_rounds"].mean(), "winner_freq": df["winner_seat"].value_counts().to_dict()}
