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


def replay_binding_calls(engine, calls, strategy_dtype, override_dtype) -> int:
    """Replay the engine calls recorded while the binding ran INSIDE the reference (tests/golden/binding_vectors.json,
    oracle/gen_binding.py) on ``engine`` and compare every result with the recorded one.  Returns the number of calls."""
    import base64

    def table_of(rows):
        return np.array([tuple(int(v) for v in r) for r in rows], dtype=strategy_dtype)

    def overrides_of(rows):
        return np.array([tuple(int(v) for v in r) for r in rows], dtype=override_dtype) if rows else None

    for n, c in enumerate(calls):
        ctx = f"recorded call {n} ({c['method']})"
        if c["method"] == "tournament":
            res = engine.tournament(table_of(c["table"]), c["k"], c["root_seed"], c["shuffle_begin"], c["shuffle_end"],
                                    shuffles_per_batch=c["shuffles_per_batch"], target_score=c["target_score"], max_rounds=c["max_rounds"],
                                    overrides=overrides_of(c["overrides"]), want_rows=c["want_rows"])
            assert np.array_equal(np.asarray(res["tally"]), np.array(c["tally"], dtype=np.int64)), ctx
            if c["want_rows"]:
                assert res["rows"].tobytes() == base64.b64decode(c["rows_b64"]), ctx
        elif c["method"] == "h2h":
            seats = np.array([tuple(int(v) for v in r) for r in c["seats"]], dtype=strategy_dtype)
            state = None if c["state_in"] is None else np.array(c["state_in"], dtype=np.uint64)
            out = engine.h2h(seats, c["root_seed"], c["pair_id"], c["order"], c["target"], c["max_attempts"], c["chunk_games"],
                             target_score=c["target_score"], max_rounds=c["max_rounds"], overrides=overrides_of(c["overrides"]), state=state)
            assert [int(v) for v in out] == c["state_out"], ctx
        elif c["method"] == "h2h_blocks":
            seats = np.array([[tuple(int(v) for v in s) for s in pair] for pair in c["seats"]], dtype=strategy_dtype)
            states = None if c["states_in"] is None else np.array(c["states_in"], dtype=np.uint64)
            out = engine.h2h_blocks(seats, c["root_seed"], c["pair_ids"], c["orders"], np.array(c["target"], dtype=np.uint64),
                                    np.array(c["max_attempts"], dtype=np.uint64), chunk_games=c["chunk_games"],
                                    target_score=c["target_score"], max_rounds=c["max_rounds"], overrides=overrides_of(c["overrides"]),
                                    states=states)
            assert np.asarray(out).astype(np.int64).tolist() == c["states_out"], ctx
        else:
            raise AssertionError(f"unknown recorded method {c['method']}")
    return len(calls)
