"""Helpers shared by the oracle and HIP parity tests: load golden vectors, compare rows."""
from __future__ import annotations

import json
from functools import lru_cache
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"

SEAT_ORDER = ["score", "strategy", "farkles", "rolls", "n_turns", "highest_turn", "smart_five_uses",
              "n_smart_five_dice", "smart_one_uses", "n_smart_one_dice", "hot_dice", "rank", "hit_max_rounds"]

METRIC_LABELS = ("winning_score", "n_rounds", "winner_farkles", "winner_rolls", "winner_highest_turn",
                 "winner_smart_five_uses", "winner_n_smart_five_dice", "winner_smart_one_uses",
                 "winner_n_smart_one_dice", "winner_hot_dice", "winner_hit_max_rounds")


@lru_cache(maxsize=None)
def load(name: str):
    with open(GOLDEN / name) as fh:
        return json.load(fh)


def strategies_from_tuples(tuples, dtype) -> np.ndarray:
    out = np.zeros(len(tuples), dtype=dtype)
    for i, t in enumerate(tuples):
        vals = [int(v) for v in t]
        if vals[10] < 0:
            vals[10] = i
        out[i] = tuple(vals)
    return out


def row_as_compact(row, k: int, id_of) -> dict:
    """Convert one structured row (oracle or HIP layout) to the golden 'compact' dict.

    ``id_of(index)`` maps a strategy-table index to its strategy_id."""
    seats = []
    for i in range(k):
        s = row["seats"][i]
        vals = [int(s[name]) for name in SEAT_ORDER]
        vals[1] = int(id_of(vals[1]))
        seats.append(vals)
    return {"n_rounds": int(row["n_rounds"]), "status": int(row["status"]), "winner_seat": int(row["winner_seat"]),
            "seats": seats}


def assert_row_equal(actual: dict, golden: dict, ctx: str = "") -> None:
    for key in ("n_rounds", "status", "winner_seat", "seats"):
        assert actual[key] == golden[key], f"{ctx}: {key}: {actual[key]} != {golden[key]}"
    if golden["status"] == 0:
        w = golden["winner_seat"]
        scores = [s[0] for s in golden["seats"]]
        assert golden["winning_score"] == actual["seats"][w][0]
        assert golden["winner_strategy"] == actual["seats"][w][1]
        assert golden["victory_margin"] == scores[w] - (sorted(scores, reverse=True)[1] if len(scores) > 1 else 0)
        order = sorted(range(len(scores)), key=lambda i: actual["seats"][i][11])
        assert golden["seat_ranks"] == [f"P{i + 1}" for i in order]
    else:
        assert golden["winner_seat"] == -1 and all(s[11] == 0 for s in actual["seats"])


def tally_to_dicts(tally: np.ndarray, ids) -> dict:
    """[S][26] int64 -> the golden counter payload layout (zero entries dropped like Counter/defaultdict)."""
    out = {"wins": {}, "attempted": {}, "completed": {}, "safety": {}, "sums": {m: {} for m in METRIC_LABELS},
           "sq_sums": {m: {} for m in METRIC_LABELS}}
    for i, sid in enumerate(ids):
        key = str(int(sid))
        row = tally[i]
        for col, name in enumerate(("wins", "attempted", "completed", "safety")):
            if row[col]:
                out[name][key] = int(row[col])
        if row[0]:  # sums exist for every strategy that won at least once (defaultdict semantics)
            for j, m in enumerate(METRIC_LABELS):
                out["sums"][m][key] = int(row[4 + j])
                out["sq_sums"][m][key] = int(row[15 + j])
    return out


def assert_tally_matches(tally: np.ndarray, ids, golden: dict, ctx: str = "") -> None:
    mine = tally_to_dicts(tally, ids)
    for name in ("wins", "attempted", "completed", "safety"):
        assert mine[name] == golden[name], f"{ctx}: {name}"
    for m in METRIC_LABELS:
        assert mine["sums"][m] == golden["sums"][m], f"{ctx}: sums[{m}]"
        assert mine["sq_sums"][m] == golden["sq_sums"][m], f"{ctx}: sq_sums[{m}]"
    games = golden["games"]
    k_att = sum(golden["attempted"].values())
    assert int(tally[:, 1].sum()) == k_att and int(tally[:, 0].sum()) == games[1]
