"""Versioned, coordinate-derived random streams — host side.

Mirrors ``src/farkle/utils/random.py`` (namespaces :18-37, ``coordinate_entropy`` :80-124,
``coordinate_rng`` :159-188, ``coordinate_seed`` :191-225, ``spawn_seeds`` :275-295).  Seat streams of
simulated games are built on the GPU from the same coordinates; the host only needs NumPy generators for
one-off draws (random strategies) and vectorised fingerprints (``game_seed`` / ``shuffle_seed`` row fields).
"""
from __future__ import annotations

from enum import IntEnum
from typing import Final

import numpy as np

RNG_SCHEME_VERSION: Final = 2
MAX_UINT32: Final = 2**32 - 1
MAX_UINT64: Final = 2**64 - 1


class RandomPurpose(IntEnum):
    """Permanent integer namespaces (random.py:18-37)."""

    INDEXED_SEED = 1
    PLAYER = 10
    STRATEGY = 11
    TOURNAMENT_SHUFFLE = 100
    SHUFFLE_PERMUTATION = 101
    TOURNAMENT_GAME = 102
    TOURNAMENT_PLAYER = 103
    H2H_PAIR = 200
    H2H_ORDER = 201
    H2H_GAME = 202
    H2H_PLAYER = 203
    TRUESKILL_DIAGNOSTIC = 300
    BOOTSTRAP = 400
    ROOT_STABILITY_BOOTSTRAP = 401
    TIE_BREAK = 500
    HGB = 600
    SEED_SELECTION = 700


def _uint64_words(value: int, *, name: str) -> tuple[int, int]:
    if isinstance(value, bool) or not 0 <= int(value) <= MAX_UINT64:
        raise ValueError(f"{name} must be an integer in [0, 2**64 - 1]")
    v = int(value)
    return v & MAX_UINT32, v >> 32


def _alias(primary, alias, primary_name: str, alias_name: str) -> int:
    if primary is not None and alias is not None and int(primary) != int(alias):
        raise ValueError(f"{primary_name} and {alias_name} identify different coordinates")
    value = primary if primary is not None else alias
    return 0 if value is None else int(value)


def coordinate_entropy(purpose, *, root_seed: int, k: int = 0, shuffle_index: int = 0, pair_index: int | None = None,
                       pair_id: int | None = None, order: int = 0, game_index: int | None = None,
                       attempt_index: int | None = None, seat_index: int = 0, replicate_index: int = 0) -> tuple[int, ...]:
    """The 18 uint32 SeedSequence entropy words of a semantic coordinate."""
    try:
        namespace = RandomPurpose(int(purpose))
    except (TypeError, ValueError) as exc:
        raise ValueError(f"unregistered RNG purpose namespace: {purpose!r}") from exc
    pair = _alias(pair_index, pair_id, "pair_index", "pair_id")
    game = _alias(game_index, attempt_index, "game_index", "attempt_index")
    words: list[int] = [RNG_SCHEME_VERSION, int(namespace)]
    for name, value in (("root_seed", root_seed), ("k", k), ("shuffle_index", shuffle_index), ("pair_id", pair),
                        ("order", order), ("game_index", game), ("seat_index", seat_index),
                        ("replicate_index", replicate_index)):
        words.extend(_uint64_words(value, name=name))
    return tuple(words)


def coordinate_seed_sequence(purpose, **coords) -> np.random.SeedSequence:
    return np.random.SeedSequence(coordinate_entropy(purpose, **coords))


def coordinate_rng(purpose, **coords) -> np.random.Generator:
    """Explicit PCG64DXSM generator for semantic coordinates (host-side draws only)."""
    return np.random.Generator(np.random.PCG64DXSM(coordinate_seed_sequence(purpose, **coords)))


def coordinate_seed(purpose, *, dtype=np.uint64, **coords) -> int:
    """Diagnostic fingerprint of a coordinate; never a root for a child stream."""
    return int(coordinate_seed_sequence(purpose, **coords).generate_state(1, dtype=dtype)[0])


# ---- vectorised SeedSequence fingerprints (many coordinates at once) ------------------------------------
_INIT_A, _MULT_A, _INIT_B, _MULT_B = 0x43B0D7E5, 0x931E8875, 0x8B51F9DD, 0x58F38DED
_MIX_L, _MIX_R = 0xCA01F9DD, 0x4973F715
_M32 = np.uint64(0xFFFFFFFF)


def _seedseq_pool(words: np.ndarray) -> np.ndarray:
    """words: uint64 array [n, 18] holding uint32 values -> pool [n, 4] (SeedSequence.mix_entropy)."""
    n = words.shape[0]
    hc = _INIT_A
    pool = np.zeros((n, 4), dtype=np.uint64)

    def hashmix(v: np.ndarray) -> np.ndarray:
        nonlocal hc
        v = v ^ np.uint64(hc)
        hc = (hc * _MULT_A) & 0xFFFFFFFF
        v = (v * np.uint64(hc)) & _M32
        return v ^ (v >> np.uint64(16))

    def mix(x: np.ndarray, y: np.ndarray) -> np.ndarray:
        r = (np.uint64(_MIX_L) * x - np.uint64(_MIX_R) * y) & _M32
        return r ^ (r >> np.uint64(16))

    for i in range(4):
        pool[:, i] = hashmix(words[:, i])
    for src in range(4):
        for dst in range(4):
            if src != dst:
                pool[:, dst] = mix(pool[:, dst], hashmix(pool[:, src]))
    for src in range(4, words.shape[1]):
        for dst in range(4):
            pool[:, dst] = mix(pool[:, dst], hashmix(words[:, src]))
    return pool


def coordinate_seeds(purpose, *, root_seed, k=0, shuffle_index=0, pair_id=0, order=0, game_index=0, seat_index=0,
                     replicate_index=0, dtype=np.uint64) -> np.ndarray:
    """Vectorised ``coordinate_seed``: any coordinate may be an array (broadcast)."""
    fields = [np.atleast_1d(np.asarray(v, dtype=np.uint64)) for v in
              (root_seed, k, shuffle_index, pair_id, order, game_index, seat_index, replicate_index)]
    n = max(len(f) for f in fields)
    words = np.zeros((n, 18), dtype=np.uint64)
    words[:, 0] = RNG_SCHEME_VERSION
    words[:, 1] = int(RandomPurpose(int(purpose)))
    for i, f in enumerate(fields):
        words[:, 2 + 2 * i] = f & _M32
        words[:, 3 + 2 * i] = f >> np.uint64(32)
    pool = _seedseq_pool(words)
    hc = _INIT_B
    out = []
    for i in range(2 if dtype is np.uint64 or dtype == np.uint64 else 1):
        v = pool[:, i % 4] ^ np.uint64(hc)
        hc = (hc * _MULT_B) & 0xFFFFFFFF
        v = (v * np.uint64(hc)) & _M32
        out.append(v ^ (v >> np.uint64(16)))
    if len(out) == 2:
        return out[0] | (out[1] << np.uint64(32))
    return out[0].astype(np.uint32)


def spawn_seeds(n: int, *, seed: int) -> np.ndarray:
    """Legacy external-boundary seeds, one uint32 fingerprint per index (random.py:275-295)."""
    if isinstance(n, bool) or n < 0:
        raise ValueError("n must be a non-negative integer")
    if n == 0:
        return np.zeros(0, dtype=np.uint32)
    return coordinate_seeds(RandomPurpose.INDEXED_SEED, root_seed=seed, game_index=np.arange(n, dtype=np.uint64), dtype=np.uint32)


def make_rng(seed: int) -> np.random.Generator:
    return coordinate_rng(RandomPurpose.INDEXED_SEED, root_seed=seed)


__all__ = ["MAX_UINT32", "RNG_SCHEME_VERSION", "RandomPurpose", "coordinate_entropy", "coordinate_rng", "coordinate_seed",
           "coordinate_seed_sequence", "coordinate_seeds", "make_rng", "spawn_seeds"]
