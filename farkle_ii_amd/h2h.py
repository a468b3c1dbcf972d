"""H2H block execution on the GPU engine.

Mirrors the attempt loop of ``_simulate_block_from_manifest`` (``src/farkle/analysis/h2h_schedule.py:1149-1243``),
``_block_progress`` (:1088-1146) and the ``BlockRunner`` hook (:1521, :1602).  Planning, power analysis and
publication stay in the reference.
"""
from __future__ import annotations

from typing import Any, Callable, Mapping

import numpy as np

from .backend import FarkleHipError, FK_ERR_ROLL_LIMIT
from .engine import get_engine
from .game_profile import GameProfile
from .strategies import FavorDiceOrScore, ThresholdStrategy, pack_strategies

_PROGRESS_KEYS = {"games_attempted", "games_completed", "games_safety_limit", "wins_seat1", "wins_seat2", "wins_a", "wins_b",
                  "replacement_attempt_count", "completion_status", "completion_game_rate", "safety_limit_game_rate",
                  "authenticated_attempt_index_start", "authenticated_attempt_index_stop_exclusive",
                  "attempt_coordinate_range_hash"}


def attempt_coordinate_range_hash(block: Mapping[str, Any], stop_exclusive: int) -> str:
    """SHA-256 of the contiguous semantic attempt-coordinate prefix (h2h_schedule.py:1072-1085)."""
    import hashlib
    import json

    payload = {"rng_scheme_version": int(block.get("rng_scheme_version", 2)),
               "purpose": int(block.get("rng_purpose_namespace", 203)), "root_seed": int(block["root_seed"]),
               "pair_id": int(block["pair_id"]), "order": int(block["order"]), "attempt_index_start": 0,
               "attempt_index_stop_exclusive": int(stop_exclusive)}
    return hashlib.sha256(json.dumps(payload, sort_keys=True, separators=(",", ":")).encode("utf-8")).hexdigest()


def block_progress(block: Mapping[str, Any], *, games_attempted: int, games_completed: int, games_safety_limit: int,
                   wins_seat1: int, wins_seat2: int) -> dict[str, Any]:
    target = int(block["n_completed_required"])
    max_attempts = int(block["max_attempts"])
    if games_completed >= target:
        status = "complete"
    elif games_attempted >= max_attempts:
        status = "unresolved_nonviable"
    else:
        status = "partial_resumable"
    out = {k: v for k, v in block.items() if not str(k).startswith("_") and k not in _PROGRESS_KEYS}
    order = int(block["order"])
    out.update({
        "wins_a": wins_seat1 if order == 0 else wins_seat2, "wins_b": wins_seat2 if order == 0 else wins_seat1,
        "games_attempted": games_attempted, "games_completed": games_completed, "games_safety_limit": games_safety_limit,
        "wins_seat1": wins_seat1, "wins_seat2": wins_seat2, "replacement_attempt_count": max(0, games_attempted - target),
        "completion_status": status,
        "completion_game_rate": games_completed / games_attempted if games_attempted else None,
        "safety_limit_game_rate": games_safety_limit / games_attempted if games_attempted else None,
        "authenticated_attempt_index_start": 0, "authenticated_attempt_index_stop_exclusive": games_attempted,
        "attempt_coordinate_range_hash": attempt_coordinate_range_hash(block, games_attempted),
    })
    return out


def simulate_block(block: Mapping[str, Any], strategy1: ThresholdStrategy, strategy2: ThresholdStrategy, chunk_games: int,
                   oracle_game_profile: GameProfile | None = None, engine=None) -> dict[str, Any]:
    """Advance one (pair, root, order) block by at most ``chunk_games`` attempts (in-order prefix semantics)."""
    eng = engine or get_engine()
    state = np.array([int(block.get(k, 0)) for k in ("games_attempted", "games_completed", "games_safety_limit",
                                                     "wins_seat1", "wins_seat2")], dtype=np.uint64)
    target_score, max_rounds, ov = 10_000, 200, None
    if oracle_game_profile is not None:
        target_score, max_rounds = oracle_game_profile.default_target_score, oracle_game_profile.default_max_rounds
        ov = oracle_game_profile.h2h_overrides()
    try:
        st = eng.h2h(pack_strategies([strategy1, strategy2]), int(block["root_seed"]), int(block["pair_id"]),
                     int(block["order"]), int(block["n_completed_required"]), int(block["max_attempts"]), int(chunk_games),
                     target_score=target_score, max_rounds=max_rounds, overrides=ov, state=state)
    except FarkleHipError as exc:
        if exc.code == FK_ERR_ROLL_LIMIT:
            raise RuntimeError(str(exc)) from exc
        raise
    a, c, s, w1, w2 = (int(v) for v in st)
    return block_progress(block, games_attempted=a, games_completed=c, games_safety_limit=s, wins_seat1=w1, wins_seat2=w2)


def strategy_from_manifest(value: Any, manifest) -> ThresholdStrategy:
    """Decode a canonical numeric strategy id through a strategy manifest frame (strategies.py:762-800)."""
    if isinstance(value, bool) or not (isinstance(value, (int, np.integer)) or (isinstance(value, str) and value.isdigit())):
        raise ValueError(f"Cannot parse nonnumeric strategy identifier: {value!r}")
    sid = int(value)
    match = manifest.loc[manifest["strategy_id"] == sid]
    if match.empty:
        raise KeyError(f"strategy_id {sid} missing from manifest/encoder")
    r = match.iloc[0]
    favor = r["favor_dice_or_score"]
    if not isinstance(favor, FavorDiceOrScore):
        favor = FavorDiceOrScore.SCORE if str(favor) == "score" else FavorDiceOrScore.DICE
    return ThresholdStrategy(int(r["score_threshold"]), int(r["dice_threshold"]), bool(r["smart_five"]), bool(r["smart_one"]),
                             bool(r["consider_score"]), bool(r["consider_dice"]), bool(r["require_both"]),
                             bool(r["auto_hot_dice"]), bool(r["run_up_score"]), favor, strategy_id=sid)


def gpu_block_runner(oracle_game_profile: GameProfile | None = None) -> Callable[[dict, Any, int], dict]:
    """A ``BlockRunner`` (h2h_schedule.py:1521): ``runner(block, strategy_manifest_path, attempt_count) -> block``."""
    cache: dict[str, Any] = {}

    def runner(block: dict, strategy_manifest_path, attempt_count: int) -> dict:
        import pandas as pd

        key = str(strategy_manifest_path)
        if key not in cache:
            cache[key] = pd.read_parquet(strategy_manifest_path)
        manifest = cache[key]
        return simulate_block(block, strategy_from_manifest(block["seat1_strategy"], manifest),
                              strategy_from_manifest(block["seat2_strategy"], manifest), attempt_count, oracle_game_profile)

    return runner
