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


_STATE_KEYS = ("games_attempted", "games_completed", "games_safety_limit", "wins_seat1", "wins_seat2")


def _block_state(block: Mapping[str, Any]) -> list[int]:
    return [int(block.get(k, 0)) for k in _STATE_KEYS]


def _is_terminal(block: Mapping[str, Any]) -> bool:
    a, c = int(block.get("games_attempted", 0)), int(block.get("games_completed", 0))
    return c >= int(block["n_completed_required"]) or a >= int(block["max_attempts"])


def _load_manifest(strategy_manifest):
    if hasattr(strategy_manifest, "loc"):  # already a frame
        return strategy_manifest
    import pandas as pd

    return pd.read_parquet(strategy_manifest)


def run_blocks(blocks, strategy_manifest, chunk_games: int | None = None, oracle_game_profile: GameProfile | None = None,
               engine=None, rank: int = 0, world: int = 1) -> list[dict[str, Any]]:
    """Many pending block dicts of a schedule advanced together: the schedule-level counterpart of the per-block
    ``BlockRunner`` call inside ``execute_h2h_schedule``'s serial loop (h2h_schedule.py:2038-2093; pool path :1957-1967).

    Every block (the dict ``_simulate_block_from_manifest`` takes, :1149-1243) is advanced by at most ``chunk_games``
    attempts from its recorded progress (``None``: to its terminal state), exactly as the serial loop would — the blocks of
    one root share each kernel launch (``fk_h2h_run_blocks``).  With ``world`` > 1 the blocks are dealt round-robin over the
    ranks (blocks are independent units: no data-path collective) and every rank gets every result back through one
    object gather.  Returns the blocks in input order in the shape ``_normalize_runner_result`` accepts (:1422-1468)."""
    eng = engine or get_engine()
    blocks = list(blocks)
    target_score, max_rounds, ov = 10_000, 200, None
    if oracle_game_profile is not None:
        target_score, max_rounds = oracle_game_profile.default_target_score, oracle_game_profile.default_max_rounds
        ov = oracle_game_profile.h2h_overrides()
    mine = [i for i in range(len(blocks)) if i % world == rank and not _is_terminal(blocks[i])]
    done: dict[int, list[int]] = {}
    if mine:
        manifest = _load_manifest(strategy_manifest)
        decoded: dict[Any, ThresholdStrategy] = {}

        def strat(value):
            if value not in decoded:
                decoded[value] = strategy_from_manifest(value, manifest)
            return decoded[value]

        by_root: dict[int, list[int]] = {}
        for i in mine:
            by_root.setdefault(int(blocks[i]["root_seed"]), []).append(i)
        for root, idx in by_root.items():
            seats = np.stack([pack_strategies([strat(blocks[i]["seat1_strategy"]), strat(blocks[i]["seat2_strategy"])]) for i in idx])
            states = np.array([_block_state(blocks[i]) for i in idx], dtype=np.uint64)
            max_att = np.array([int(blocks[i]["max_attempts"]) for i in idx], dtype=np.uint64)
            try:
                out = eng.h2h_blocks(seats, root, [int(blocks[i]["pair_id"]) for i in idx], [int(blocks[i]["order"]) for i in idx],
                                     np.array([int(blocks[i]["n_completed_required"]) for i in idx], dtype=np.uint64), max_att,
                                     chunk_games=int(max_att.max()) if chunk_games is None else int(chunk_games),
                                     target_score=target_score, max_rounds=max_rounds, overrides=ov, states=states)
            except FarkleHipError as exc:
                if exc.code == FK_ERR_ROLL_LIMIT:
                    raise RuntimeError(str(exc)) from exc
                raise
            for i, st in zip(idx, out):
                done[i] = [int(v) for v in st]
    if world > 1:
        from .distributed import gather_objects

        parts = gather_objects(gather_objects(done, dst=0), broadcast_from=0)
        done = {}
        for part in parts:
            done.update(part)
    results = []
    for i, block in enumerate(blocks):
        a, c, s_, w1, w2 = done.get(i, _block_state(block))
        results.append(block_progress(block, games_attempted=a, games_completed=c, games_safety_limit=s_, wins_seat1=w1, wins_seat2=w2))
    return results


class PrefetchingBlockRunner:
    """A ``BlockRunner`` (h2h_schedule.py:1521) for ``execute_h2h_schedule(cfg, block_runner=...)`` that plays the WHOLE
    schedule in shared launches.  The reference's custom-runner path is a serial loop — one call per block and chunk
    (:2038-2093) — so a per-block runner pays at least two kernel launches and a host round trip per call, 44 700 times in
    the production schedule.  This runner is constructed with the pending blocks of the schedule frame; the first call
    that asks for a block at its known progress advances EVERY pending block by one chunk (``run_blocks``) and the serial
    loop is then served from the cache.  Each served result equals what the per-block runner returns for the same
    ``(block, attempt_count)``: blocks are independent and a chunk is an in-order attempt prefix.  A block this runner
    does not know, or one asked for at another progress than it last returned, is played alone (``simulate_block``)."""

    def __init__(self, schedule_blocks, oracle_game_profile: GameProfile | None = None, engine=None, rank: int = 0, world: int = 1,
                 chunk_games: int | None = None):
        """``chunk_games``: the schedule's checkpoint attempt limit (``max_attempts_per_checkpoint`` of plan_h2h_chunk,
        :94-129).  When it is not given it is learned from the first call whose ``attempt_count`` is below the block's
        remaining attempts; until then only blocks that the asked-for count takes to their end are played ahead."""
        self._profile, self._engine, self._rank, self._world = oracle_game_profile, engine, rank, world
        self._chunk_games = None if chunk_games is None else int(chunk_games)
        self._frontier: dict[str, dict[str, Any]] = {str(b["block_id"]): dict(b) for b in schedule_blocks}
        self._cache: dict[tuple[str, int], dict[str, Any]] = {}
        self._manifests: dict[str, Any] = {}
        self.generations = 0  # shared launches groups issued so far (diagnostics / tests)
        self.single_block_calls = 0

    def _manifest(self, path):
        key = id(path) if hasattr(path, "loc") else str(path)
        if key not in self._manifests:
            self._manifests[key] = _load_manifest(path)
        return self._manifests[key]

    def __call__(self, block: dict, strategy_manifest_path, attempt_count: int) -> dict:
        bid, attempted = str(block["block_id"]), int(block.get("games_attempted", 0))
        key = (bid, attempted)
        known = self._frontier.get(bid)
        if key not in self._cache and known is not None and _block_state(known) == _block_state(block):
            # the chunk rule of plan_h2h_chunk (:94-129): attempt_count = min(max_attempts - attempted, checkpoint limit)
            remaining = int(block["max_attempts"]) - attempted
            if self._chunk_games is None and attempt_count < remaining:
                self._chunk_games = int(attempt_count)
            pending = [b for b in self._frontier.values() if not _is_terminal(b)]
            if self._chunk_games is None:  # limit still unknown (>= attempt_count): blocks that this count takes to their end
                pending = [b for b in pending if int(b["max_attempts"]) - int(b.get("games_attempted", 0)) <= attempt_count]
            results = run_blocks(pending, self._manifest(strategy_manifest_path), self._chunk_games, self._profile, self._engine,
                                 self._rank, self._world)
            self.generations += 1
            for before, after in zip(pending, results):
                self._cache[(str(before["block_id"]), int(before.get("games_attempted", 0)))] = after
                self._frontier[str(before["block_id"])] = after
        hit = self._cache.pop(key, None)
        if hit is not None and int(hit["games_attempted"]) - attempted <= int(attempt_count):
            return hit
        self.single_block_calls += 1
        manifest = self._manifest(strategy_manifest_path)
        return simulate_block(block, strategy_from_manifest(block["seat1_strategy"], manifest),
                              strategy_from_manifest(block["seat2_strategy"], manifest), attempt_count, self._profile, self._engine)


def gpu_block_runner(oracle_game_profile: GameProfile | None = None, engine=None) -> Callable[[dict, Any, int], dict]:
    """A ``BlockRunner`` (h2h_schedule.py:1521): ``runner(block, strategy_manifest_path, attempt_count) -> block``."""
    cache: dict[str, Any] = {}

    def runner(block: dict, strategy_manifest_path, attempt_count: int) -> dict:
        key = id(strategy_manifest_path) if hasattr(strategy_manifest_path, "loc") else str(strategy_manifest_path)
        if key not in cache:
            cache[key] = _load_manifest(strategy_manifest_path)
        manifest = cache[key]
        return simulate_block(block, strategy_from_manifest(block["seat1_strategy"], manifest),
                              strategy_from_manifest(block["seat2_strategy"], manifest), attempt_count, oracle_game_profile, engine)

    return runner
