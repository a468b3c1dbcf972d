"""Limit-only game settings (target score, max_rounds overrides) for deterministic workflow oracles.

Mirrors ``src/farkle/simulation/game_profile.py:24-191`` including the canonical identity hash that checkpoints and
manifests carry (``game_profile_sha256``); ``tournament_overrides`` / ``h2h_overrides`` flatten a profile into the
``fk_override[]`` records of ``include/farkle_hip.h``.
"""
from __future__ import annotations

import hashlib
import json
from dataclasses import asdict, dataclass

GAME_PROFILE_CONTRACT_VERSION = 1

import numpy as np


def _coord(value: int, name: str) -> None:
    if isinstance(value, bool) or not isinstance(value, int) or value < 0:
        raise ValueError(f"{name} must be a non-negative integer")


@dataclass(frozen=True, order=True)
class TournamentMaxRoundsOverride:
    root_seed: int
    k: int
    shuffle_index: int
    game_index: int
    max_rounds: int

    def __post_init__(self) -> None:
        for name in ("root_seed", "k", "shuffle_index", "game_index"):
            _coord(getattr(self, name), name)
        if self.k < 2:
            raise ValueError("k must be at least 2")
        _coord(self.max_rounds, "max_rounds")

    @property
    def coordinate(self) -> tuple[int, int, int, int]:
        return (self.root_seed, self.k, self.shuffle_index, self.game_index)


@dataclass(frozen=True, order=True)
class H2HMaxRoundsOverride:
    root_seed: int
    pair_id: int
    order: int
    attempt_index: int
    max_rounds: int

    def __post_init__(self) -> None:
        for name in ("root_seed", "pair_id", "order", "attempt_index"):
            _coord(getattr(self, name), name)
        if self.order not in (0, 1):
            raise ValueError("order must be 0 or 1")
        _coord(self.max_rounds, "max_rounds")

    @property
    def coordinate(self) -> tuple[int, int, int, int]:
        return (self.root_seed, self.pair_id, self.order, self.attempt_index)


@dataclass(frozen=True)
class GameLimits:
    target_score: int
    max_rounds: int


@dataclass(frozen=True)
class GameProfile:
    default_target_score: int = 10_000
    default_max_rounds: int = 200
    tournament_max_rounds_overrides: tuple[TournamentMaxRoundsOverride, ...] = ()
    h2h_max_rounds_overrides: tuple[H2HMaxRoundsOverride, ...] = ()

    def __post_init__(self) -> None:
        if isinstance(self.default_target_score, bool) or not isinstance(self.default_target_score, int) \
                or self.default_target_score <= 0:
            raise ValueError("default_target_score must be a positive integer")
        _coord(self.default_max_rounds, "max_rounds")
        for name in ("tournament_max_rounds_overrides", "h2h_max_rounds_overrides"):
            items = getattr(self, name)
            if not isinstance(items, tuple):
                raise TypeError(f"{name} must be a tuple")
            coords = [o.coordinate for o in items]
            if len(set(coords)) != len(coords):
                kind = "tournament" if name.startswith("tournament") else "H2H"
                raise ValueError(f"{kind} max-round overrides contain duplicate coordinates")

    def canonical_payload(self) -> dict:
        """Order-independent identity payload (game_profile.py:113-134)."""
        return {"game_profile_contract_version": GAME_PROFILE_CONTRACT_VERSION,
                "default_target_score": self.default_target_score, "default_max_rounds": self.default_max_rounds,
                "tournament_max_rounds_overrides": [asdict(o) for o in sorted(self.tournament_max_rounds_overrides,
                                                                              key=lambda o: o.coordinate)],
                "h2h_max_rounds_overrides": [asdict(o) for o in sorted(self.h2h_max_rounds_overrides,
                                                                       key=lambda o: o.coordinate)]}

    @property
    def sha256(self) -> str:
        """SHA-256 of the canonical JSON of the payload: sorted keys, compact separators
        (utils/authenticated_contract.py:100-114)."""
        text = json.dumps(self.canonical_payload(), sort_keys=True, separators=(",", ":"), ensure_ascii=False, allow_nan=False)
        return hashlib.sha256(text.encode("utf-8")).hexdigest()

    def tournament_limits(self, *, root_seed: int, k: int, shuffle_index: int, game_index: int) -> GameLimits:
        for o in self.tournament_max_rounds_overrides:
            if o.coordinate == (root_seed, k, shuffle_index, game_index):
                return GameLimits(self.default_target_score, o.max_rounds)
        return GameLimits(self.default_target_score, self.default_max_rounds)

    def h2h_limits(self, *, root_seed: int, pair_id: int, order: int, attempt_index: int) -> GameLimits:
        for o in self.h2h_max_rounds_overrides:
            if o.coordinate == (root_seed, pair_id, order, attempt_index):
                return GameLimits(self.default_target_score, o.max_rounds)
        return GameLimits(self.default_target_score, self.default_max_rounds)

    # ---- C-ABI records -------------------------------------------------------------------
    def tournament_overrides(self) -> np.ndarray:
        from .backend import make_overrides

        return make_overrides((o.root_seed, o.shuffle_index, o.game_index, o.k, o.max_rounds)
                              for o in self.tournament_max_rounds_overrides)

    def h2h_overrides(self) -> np.ndarray:
        from .backend import make_overrides

        return make_overrides((o.root_seed, o.pair_id, o.attempt_index, o.order, o.max_rounds)
                              for o in self.h2h_max_rounds_overrides)
