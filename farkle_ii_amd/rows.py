"""Device game rows -> the reference's row mappings / Arrow tables.

The kernels emit ``fk_row_hdr`` + k x ``fk_seat`` (4 + 28k bytes).  ``rows_to_dicts`` rebuilds the flat mapping
``_play_game`` returns (``src/farkle/simulation/simulation.py:628-655``) and ``rows_to_table`` the typed
outcome-schema-v2 Arrow table (``src/farkle/utils/schema_helpers.py:23-90``).
"""
from __future__ import annotations

from typing import Any, Mapping, Sequence

import numpy as np

OUTCOME_SCHEMA_VERSION = 2
TOURNAMENT_METHOD_VERSION = 2

_SEAT_FIELDS = (("score", "score"), ("farkles", "farkles"), ("rolls", "rolls"), ("n_turns", "n_turns"),
                ("highest_turn", "highest_turn"), ("smart_five_uses", "smart_five_uses"),
                ("n_smart_five_dice", "n_smart_five_dice"), ("smart_one_uses", "smart_one_uses"),
                ("n_smart_one_dice", "n_smart_one_dice"), ("hot_dice", "hot_dice"))


def raw_simulation_schema_for(n_players: int):
    """Arrow schema of persisted simulation rows (schema_helpers.py:79-90)."""
    import pyarrow as pa

    if n_players < 1:
        raise ValueError("n_players must be positive")
    nullable_strings = pa.list_(pa.field("item", pa.string(), nullable=True))
    base = [
        pa.field("root_seed", pa.int64(), nullable=False), pa.field("k", pa.int16(), nullable=False),
        pa.field("shuffle_index", pa.int64(), nullable=False), pa.field("game_index", pa.int32(), nullable=False),
        pa.field("deterministic_batch_id", pa.int32(), nullable=False), pa.field("shuffle_seed", pa.int64(), nullable=False),
        pa.field("termination_status", pa.string(), nullable=False), pa.field("hit_safety_limit", pa.bool_(), nullable=False),
        pa.field("outcome_schema_version", pa.int16(), nullable=False), pa.field("winner_seat", pa.string(), nullable=True),
        pa.field("winner_strategy", pa.int32(), nullable=True), pa.field("game_seed", pa.int64(), nullable=False),
        pa.field("rng_scheme_version", pa.int16(), nullable=False), pa.field("rng_purpose_namespace", pa.int32(), nullable=False),
        pa.field("seat_ranks", nullable_strings, nullable=False), pa.field("winning_score", pa.int32(), nullable=True),
        pa.field("victory_margin", pa.int32(), nullable=True), pa.field("n_rounds", pa.int16(), nullable=False),
    ]
    seat_template = {
        "score": (pa.int32(), False), "farkles": (pa.int16(), False), "rolls": (pa.int16(), False),
        "highest_turn": (pa.int16(), False), "strategy": (pa.int32(), False), "rank": (pa.int8(), True),
        "loss_margin": (pa.int32(), True), "smart_five_uses": (pa.int16(), False), "n_smart_five_dice": (pa.int16(), False),
        "smart_one_uses": (pa.int16(), False), "n_smart_one_dice": (pa.int16(), False), "hot_dice": (pa.int16(), False),
        "n_turns": (pa.int16(), False), "hit_max_rounds": (pa.bool_(), False),
    }
    seats = [pa.field(f"P{i}_{name}", dtype, nullable=nullable)
             for i in range(1, n_players + 1) for name, (dtype, nullable) in seat_template.items()]
    return pa.schema([*base, *seats])


def row_to_dict(row, k: int, strategy_ids: Sequence[int], provenance: Mapping[str, Any] | None = None) -> dict[str, Any]:
    """One device row -> the reference's flat row mapping."""
    completed = int(row["status"]) == 0
    seats = row["seats"]
    w = int(row["winner_seat"])
    scores = [int(seats[i]["score"]) for i in range(k)]
    flat: dict[str, Any] = {
        "termination_status": "completed" if completed else "safety_limit",
        "hit_safety_limit": not completed,
        "outcome_schema_version": OUTCOME_SCHEMA_VERSION,
        "winner_seat": f"P{w + 1}" if completed else None,
        "winner_strategy": int(strategy_ids[int(seats[w]["strategy"])]) if completed else None,
        "seat_ranks": ([f"P{i + 1}" for i in sorted(range(k), key=lambda j: int(seats[j]["rank"]))] if completed
                       else [None] * k),
        "winning_score": scores[w] if completed else None,
        "victory_margin": (scores[w] - (sorted(scores, reverse=True)[1] if k > 1 else 0)) if completed else None,
        "n_rounds": int(row["n_rounds"]),
    }
    if provenance:
        flat.update(provenance)
    for i in range(k):
        s = seats[i]
        p = f"P{i + 1}_"
        flat[p + "score"] = scores[i]
        flat[p + "farkles"] = int(s["farkles"])
        flat[p + "rolls"] = int(s["rolls"])
        flat[p + "n_turns"] = int(s["n_turns"])
        flat[p + "highest_turn"] = int(s["highest_turn"])
        flat[p + "strategy"] = int(strategy_ids[int(s["strategy"])])
        flat[p + "rank"] = int(s["rank"]) if completed else None
        flat[p + "loss_margin"] = (scores[w] - scores[i]) if completed else None
        flat[p + "smart_five_uses"] = int(s["smart_five_uses"])
        flat[p + "n_smart_five_dice"] = int(s["n_smart_five_dice"])
        flat[p + "smart_one_uses"] = int(s["smart_one_uses"])
        flat[p + "n_smart_one_dice"] = int(s["n_smart_one_dice"])
        flat[p + "hot_dice"] = int(s["hot_dice"])
        flat[p + "hit_max_rounds"] = bool(s["hit_max_rounds"])
    return flat


def validate_simulation_row(row: Mapping[str, Any]) -> None:
    """Closed outcome invariants of one flattened row (simulation.py:450-563)."""
    try:
        k = int(row["k"])
        status = row["termination_status"]
        if status not in ("completed", "safety_limit"):
            raise ValueError(status)
    except (KeyError, TypeError, ValueError) as exc:
        raise ValueError("Simulation row has invalid k or termination_status") from exc
    if k < 1:
        raise ValueError("Simulation row k must be positive")
    if row.get("outcome_schema_version") != OUTCOME_SCHEMA_VERSION:
        raise ValueError(f"Simulation row must use outcome_schema_version={OUTCOME_SCHEMA_VERSION}")
    seats = [f"P{i}" for i in range(1, k + 1)]
    strategies = [row.get(f"{s}_strategy") for s in seats]
    if any(v is None for v in strategies):
        raise ValueError("Simulation row missing seated strategy")
    if len(set(strategies)) != k:
        raise ValueError("Simulation row must seat distinct strategies")
    scores = [row.get(f"{s}_score") for s in seats]
    if any(isinstance(v, bool) or not isinstance(v, (int, np.integer)) for v in scores):
        raise ValueError("Simulation row scores must be integers")
    ranks = [row.get(f"{s}_rank") for s in seats]
    if status == "completed":
        order = sorted(range(k), key=lambda i: (-int(scores[i]), i))
        expected = [0] * k
        for r, i in enumerate(order, start=1):
            expected[i] = r
        if [None if r is None else int(r) for r in ranks] != expected:
            raise ValueError("Completed simulation row ranks are inconsistent with final scores")
        w = seats[order[0]]
        if row.get("winner_seat") != w or row.get("winner_strategy") != row.get(f"{w}_strategy"):
            raise ValueError("Completed simulation row must have exactly one winner matching its rank-1 seat")
        if row.get("winning_score") != max(scores):
            raise ValueError("Completed simulation row has inconsistent winning_score")
        if row.get("hit_safety_limit") is not False or any(row.get(f"{s}_hit_max_rounds") is not False for s in seats):
            raise ValueError("Completed simulation row cannot hit the safety limit")
        if list(row.get("seat_ranks") or []) != [seats[i] for i in order]:
            raise ValueError("Completed simulation row has inconsistent seat_ranks")
        return
    if row.get("hit_safety_limit") is not True or any(row.get(f"{s}_hit_max_rounds") is not True for s in seats):
        raise ValueError("Safety-limit simulation row must mark every seat at the safety limit")
    if any(row.get(name) is not None for name in ("winner_seat", "winner_strategy", "winning_score", "victory_margin")):
        raise ValueError("Safety-limit simulation row cannot claim a winner")
    if any(r is not None for r in ranks) or list(row.get("seat_ranks") or []) != [None] * k:
        raise ValueError("Safety-limit simulation row cannot assign participant ranks")


def simulation_rows_to_table(rows: Sequence[Mapping[str, Any]], n_players: int):
    """Validate and materialise row mappings with deliberate Arrow nullability (simulation.py:566-573)."""
    import pyarrow as pa

    for row in rows:
        validate_simulation_row(row)
        if int(row["k"]) != n_players:
            raise ValueError(f"Simulation row k={row['k']} does not match schema k={n_players}")
    return pa.Table.from_pylist(list(rows), schema=raw_simulation_schema_for(n_players))


def rows_to_table(rows: np.ndarray, k: int, strategy_ids: Sequence[int], *, root_seed: int, shuffle_index, game_index,
                  deterministic_batch_id, shuffle_seed, game_seed, rng_purpose_namespace: int):
    """Columnar (vectorised) conversion of device rows to the raw-simulation Arrow table.

    ``shuffle_index`` ... ``game_seed`` are per-row arrays (or scalars).  No per-row Python objects are built
    except the ``seat_ranks`` string lists, which Arrow needs as lists."""
    import pyarrow as pa

    n = len(rows)
    ids = np.asarray(strategy_ids, dtype=np.int32)
    completed = rows["status"] == 0
    seats = rows["seats"]
    scores = seats["score"].astype(np.int64)  # [n, k]
    w = np.where(completed, rows["winner_seat"].astype(np.int64), 0)
    ar = np.arange(n)
    win_score = scores[ar, w]
    second = np.sort(scores, axis=1)[:, -2] if k > 1 else np.zeros(n, dtype=np.int64)
    mask = ~completed

    def full(v, dtype):
        return np.broadcast_to(np.asarray(v, dtype=dtype), (n,))

    def nullable(values, dtype):
        return pa.array(values, type=dtype, mask=mask)

    # string columns without per-row Python objects: seat names are taken from a k-entry dictionary
    rank_order = np.argsort(seats["rank"].astype(np.int64), axis=1, kind="stable")
    names = pa.array([f"P{i + 1}" for i in range(k)], type=pa.string())
    flat_mask = np.repeat(mask, k)  # a safety-limit game lists k nulls (engine.py:485-489)
    ranked = names.take(pa.array(rank_order.ravel().astype(np.int32), mask=flat_mask))
    seat_ranks = pa.ListArray.from_arrays(pa.array(np.arange(0, n * k + 1, k, dtype=np.int32)), ranked,
                                          type=pa.list_(pa.field("item", pa.string(), nullable=True)))
    winner_seat = names.take(pa.array(w.astype(np.int32), mask=mask))
    status = pa.array(["completed", "safety_limit"], type=pa.string()).take(pa.array(mask.astype(np.int32)))
    cols: dict[str, Any] = {
        "root_seed": pa.array(full(root_seed, np.int64)), "k": pa.array(full(k, np.int16)),
        "shuffle_index": pa.array(full(shuffle_index, np.int64)), "game_index": pa.array(full(game_index, np.int32)),
        "deterministic_batch_id": pa.array(full(deterministic_batch_id, np.int32)),
        "shuffle_seed": pa.array(full(shuffle_seed, np.int64)),
        "termination_status": status,
        "hit_safety_limit": pa.array(mask), "outcome_schema_version": pa.array(full(OUTCOME_SCHEMA_VERSION, np.int16)),
        "winner_seat": winner_seat,
        "winner_strategy": nullable(ids[seats["strategy"][ar, w]], pa.int32()),
        "game_seed": pa.array(full(game_seed, np.int64)), "rng_scheme_version": pa.array(full(2, np.int16)),
        "rng_purpose_namespace": pa.array(full(rng_purpose_namespace, np.int32)),
        "seat_ranks": seat_ranks,
        "winning_score": nullable(win_score.astype(np.int32), pa.int32()),
        "victory_margin": nullable((win_score - (second if k > 1 else 0)).astype(np.int32), pa.int32()),
        "n_rounds": pa.array(rows["n_rounds"].astype(np.int16)),
    }
    for i in range(k):
        p = f"P{i + 1}_"
        s = seats[:, i]
        for src, dst in _SEAT_FIELDS:
            dtype = np.int32 if dst == "score" else np.int16
            if dst != "score" and int(s[src].max(initial=0)) > 32767:
                raise OverflowError(f"{p}{dst} exceeds the int16 range of the raw simulation schema")
            cols[p + dst] = pa.array(s[src].astype(dtype))
        cols[p + "strategy"] = pa.array(ids[s["strategy"]])
        cols[p + "rank"] = nullable(s["rank"].astype(np.int8), pa.int8())
        cols[p + "loss_margin"] = nullable((win_score - scores[:, i]).astype(np.int32), pa.int32())
        cols[p + "hit_max_rounds"] = pa.array(s["hit_max_rounds"].astype(bool))
    schema = raw_simulation_schema_for(k)
    return pa.Table.from_arrays([cols[f.name] for f in schema], schema=schema)
