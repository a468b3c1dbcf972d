"""Pin the CPU oracle (oracle/farkle_oracle.c) against the golden vectors.

The vectors come from (a) the reference's own test goldens and (b) the Python reference run in
the build container by oracle/gen_golden.py.  These tests need no GPU and no /root/reference.
"""
from __future__ import annotations

import numpy as np
import pytest

import golden_util as gu
import pyoracle as po


def _strats(tuples):
    return gu.strategies_from_tuples(tuples, po.STRATEGY_DTYPE)


# ------------------------------------------------------------------ RNG
def test_rng_streams_dice_perms_and_fingerprints():
    data = gu.load("rng_vectors.json")
    for case in data["cases"]:
        c = po.coord(case["purpose"], case["root_seed"], case["k"], case["shuffle_index"], case["pair_id"],
                     case["order"], case["game_index"], case["seat_index"])
        assert [int(v) for v in po.stream64(c, 8)] == case["raw64"]
        assert po.dice_stream(c, case["sizes"]).tolist() == case["dice"]
        assert po.coordinate_seed32(c) == case["seed32"]
        assert po.coordinate_seed64(c) == case["seed64"]
        for S, perm in case["perms"].items():
            assert po.permutation(c, int(S)).tolist() == perm
    big = data["perm5160"]
    c = po.coord(101, big["root_seed"], big["k"], big["shuffle_index"])
    assert po.permutation(c, 5160).tolist() == big["perm"]
    spawn = [po.coordinate_seed32(po.coord(1, 42, game_index=i)) for i in range(16)]
    assert spawn == data["spawn_seeds_42"]


def test_seedsequence_uint32_collision_kat():
    # tests/unit/utils/test_random_utils.py:73-81 pins the v1 fingerprint 2_963_478_802 for two coordinates;
    # under scheme v2 the same coordinates must give distinct TOURNAMENT_GAME fingerprints.
    a = po.coordinate_seed32(po.coord(102, 32, 2, 194, game_index=18))
    b = po.coordinate_seed32(po.coord(102, 32, 2, 4052, game_index=4))
    assert a != b


# ------------------------------------------------------------------ scoring
def test_score_table_923_entries():
    data = gu.load("scoring_vectors.json")
    assert len(data["table"]) == 923
    for c1, c2, c3, c4, c5, c6, score, used, sf, so in data["table"]:
        assert po.evaluate([c1, c2, c3, c4, c5, c6]) == (score, used, sf, so)


def test_reference_scoring_csv():
    # rows of the reference's tests/data/test_farkle_scores_data.csv (tests/unit/game/test_scoring.py:180-192)
    import json

    data = gu.load("scoring_vectors.json")
    assert len(data["csv_rows"]) >= 448  # the reference's test CSV + its three data/*.csv tables with explicit rolls
    for row in data["csv_rows"]:
        roll = json.loads(row["Dice_Roll"])
        counts = [roll.count(f) for f in range(1, 7)]
        score, used, sf, so = po.evaluate(counts)
        assert (score, used, len(roll) - used, sf, so) == (
            int(row["Score"]), int(row["Used_Dice"]), int(row["Reroll_Dice"]), int(row["Single_Fives"]),
            int(row["Single_Ones"])), row


def test_default_score_and_decide_cases():
    data = gu.load("scoring_vectors.json")
    for case in data["default_score"]:
        s = _strats([case["strategy"]])
        assert list(po.default_score(case["roll"], case["pre"], s)) == case["out"], case
    for case in data["decide"]:
        s = _strats([case["strategy"]])
        got = po.decide(s, case["turn_score"], case["dice_left"], case["has_scored"], case["final_round"],
                        case["score_to_beat"], case["running_total"])
        assert int(got) == case["out"], case


# ------------------------------------------------------------------ single games
def _table_for(grid_name, grids, builder_cache={}):
    if grid_name == "g64":
        return _strats(grids["g64"])
    raise KeyError(grid_name)


def test_game_rows_match_reference():
    from farkle_ii_amd.strategies import default_grid_tuples  # host grid builder (pure Python)

    data = gu.load("game_vectors.json")
    tables = {"g64": _strats(data["grids"]["g64"]), "default": _strats(default_grid_tuples())}
    assert len(tables["default"]) == data["grids"]["default_size"]
    for idx, tup in data["grids"]["default_sample"].items():
        assert [int(v) for v in tables["default"][int(idx)].tolist()] == tup
    n_safety = 0
    for g in data["games"]:
        table = tables[g["grid"]]
        c = po.coord(g["purpose"], g["root_seed"], g["k"], g["shuffle"], g["pair"], g["order"], g["game"])
        row = po.play_game(c, table, g["strategies"], g["target"], g["max_rounds"])[0]
        actual = gu.row_as_compact(row, g["k"], lambda i: table[i]["strategy_id"])
        gu.assert_row_equal(actual, g["row"], ctx=str(g["root_seed"]))
        n_safety += g["row"]["status"]
    assert n_safety >= 6  # the fixture includes safety-limit games


# ------------------------------------------------------------------ tournament
def test_tournament_shuffles_match_reference():
    data = gu.load("tournament_vectors.json")
    for case in data["cases"]:
        table = _strats(case["strategies"])
        ov = None
        if case["profile"] == "oracle":
            ov = po.make_overrides([(11, 0, 0, 2, 0)])
        res = po.tournament(table, case["k"], case["root_seed"], case["shuffle"], case["shuffle"] + 1,
                            target_score=case["target"], overrides=ov, want_rows=True, want_perms=True)
        assert res["perms"][0].tolist() == case["perm"]
        ids = table["strategy_id"]
        gu.assert_tally_matches(res["tally"][0], ids, case["tally"], ctx=case["name"])
        assert len(res["rows"]) == len(case["rows"])
        for row, gold in zip(res["rows"], case["rows"]):
            gu.assert_row_equal(gu.row_as_compact(row, case["k"], lambda i: ids[i]), gold, ctx=case["name"])


def test_reference_expected_rows():
    """EXPECTED_ROWS of tests/integration/test_raw_simulation_oracle.py:45-58, straight from the oracle."""
    data = gu.load("tournament_vectors.json")
    grid4 = _strats(gu.load("grid_vectors.json")["oracle4"])
    ov = po.make_overrides([(11, 0, 0, 2, 0)])
    for (root, k, shuffle, game), (seat_strats, status, winner_strategy, n_rounds, n_turns, scores) in data["EXPECTED_ROWS"]:
        res = po.tournament(grid4, k, root, shuffle, shuffle + 1, target_score=100, overrides=ov, want_rows=True)
        row = res["rows"][game]
        assert [int(row["seats"][i]["strategy"]) for i in range(k)] == seat_strats
        assert ("completed", "safety_limit")[int(row["status"])] == status
        w = int(row["winner_seat"])
        assert (None if w < 0 else int(row["seats"][w]["strategy"])) == winner_strategy
        assert int(row["n_rounds"]) == n_rounds
        assert sum(int(row["seats"][i]["n_turns"]) for i in range(k)) == n_turns
        assert [int(row["seats"][i]["score"]) for i in range(k)] == scores


def test_tournament_batches_and_threads_agree():
    table = _strats(gu.load("grid_vectors.json")["g64"])
    one = po.tournament(table, 2, 42, 0, 12, shuffles_per_batch=12)
    many = po.tournament(table, 2, 42, 0, 12, shuffles_per_batch=5, n_threads=4)
    assert many["tally"].shape[0] == 3
    assert np.array_equal(one["tally"][0], many["tally"].sum(axis=0))
    t = one["tally"][0]
    assert np.array_equal(t[:, 1], t[:, 2] + t[:, 3]) and t[:, 1].sum() == 12 * 64
    assert t[:, 0].sum() * 2 == t[:, 2].sum()


# ------------------------------------------------------------------ H2H
def test_h2h_blocks_match_reference_and_goldens():
    data = gu.load("h2h_vectors.json")
    grid4 = _strats(data["oracle4"])
    ov = po.make_overrides([(11, 0, 0, 0, 0), (11, 1, 0, 0, 0), (11, 1, 1, 0, 0)])  # (root, pair, attempt, order, max_rounds)
    expected = {tuple(k): v for k, v in data["EXPECTED_H2H_BLOCKS"]}
    for b in data["blocks"]:
        seats = grid4[[b["seat1_strategy"], b["seat2_strategy"]]]
        st = po.h2h_block(seats, b["root_seed"], b["pair_id"], b["order"], 1, 2, 5000, target_score=100, overrides=ov)
        attempted, completed, safety, w1, w2 = (int(v) for v in st)
        wins_a, wins_b = (w1, w2) if b["order"] == 0 else (w2, w1)
        status = "complete" if completed >= 1 else ("unresolved_nonviable" if attempted >= 2 else "partial_resumable")
        assert [attempted, completed, safety, w1, w2, wins_a, wins_b, max(0, attempted - 1), status] == b["out"]
        assert [attempted, completed, safety, wins_a, wins_b, max(0, attempted - 1), status] == \
            expected[(b["pair_id"], b["root_seed"], b["order"])]
    g64 = _strats(gu.load("grid_vectors.json")["g64"])
    for b in data["g64_blocks"]:
        seats = g64[[b["seat1_strategy"], b["seat2_strategy"]]]
        state = None
        for step in b["trace"]:
            state = po.h2h_block(seats, b["root_seed"], b["pair_id"], b["order"], b["n_completed_required"],
                                 b["max_attempts"], b["chunk"], state=state)
            assert [int(v) for v in state] == step


# ------------------------------------------------------------------ farkle time path
def test_time_path_random_strategies_and_games():
    data = gu.load("time_path_vectors.json")
    for case in data["random_strategies"]:
        for seat, tup in enumerate(case["strategies"]):
            got = po.random_strategy(case["seed"], case["players"], seat)[0].tolist()
            assert [int(v) for v in got] == tup
    kat = data["kat_counts"]
    assert kat["winner_seat_counts"] == kat["expected"]
    for block in [kat] + data["many_games"]:
        table = _strats(block["strategies"])
        k = len(table)
        n = block["n_games"]
        coords = np.concatenate([po.coord(10, block["seed"], k, game_index=i) for i in range(n)])
        rows = po.play_games(coords, table, np.tile(np.arange(k, dtype=np.int32), n), k,
                             target_score=block.get("target", 10_000))
        for i, (row, gold) in enumerate(zip(rows, block["rows"])):
            gu.assert_row_equal(gu.row_as_compact(row, k, lambda j: table[j]["strategy_id"]), gold, ctx=f"game {i}")
            if "game_seeds" in block:
                assert po.coordinate_seed32(po.coord(1, block["seed"], game_index=i)) == block["game_seeds"][i]
    counts = {}
    table = _strats(kat["strategies"])
    coords = np.concatenate([po.coord(10, 123, 3, game_index=i) for i in range(10)])
    rows = po.play_games(coords, table, np.tile(np.arange(3, dtype=np.int32), 10), 3, target_score=5000)
    for row in rows:
        key = f"P{int(row['winner_seat']) + 1}"
        counts[key] = counts.get(key, 0) + 1
    assert counts == kat["expected"]  # tests/unit/simulation/test_simulation.py:184-199


# ------------------------------------------------------------------ scripted-dice engine cases
def test_scripted_engine_cases():
    """Hand cases in the spirit of tests/unit/game/test_engine.py (scripted dice through the RNG seam)."""
    s = _strats([[300, 2, 0, 0, 1, 1, 0, 0, 0, 1, 0], [300, 2, 0, 0, 1, 1, 0, 1, 0, 1, 1]])
    # seat 0: [1,1,1,2,3,4] = 300 (3 dice left > 2 but score >= 300 -> OR semantics: continue needs both unmet) -> bank? has_scored False & <500 -> must roll
    faces = [1, 1, 1, 2, 3, 4,  5, 5, 2,  2, 3, 4, 6, 6, 2]
    row = po.play_game_scripted(faces + [2, 3, 4, 6, 6, 3] * 50, s, [0, 1], target_score=10_000, max_rounds=1)[0]
    assert int(row["status"]) == 1 and int(row["n_rounds"]) == 1 and int(row["winner_seat"]) == -1
    assert int(row["seats"][0]["rolls"]) >= 2
    # hot dice: six scoring dice with auto_hot_dice rolls again without consulting decide
    hot = _strats([[50, 6, 0, 0, 1, 1, 0, 1, 0, 1, 0]])
    faces = [1, 1, 1, 5, 5, 5,  2, 3, 4, 6, 6, 2]
    row = po.play_game_scripted(faces, hot, [0], target_score=10_000, max_rounds=1)[0]
    assert int(row["seats"][0]["hot_dice"]) == 1 and int(row["seats"][0]["farkles"]) == 1
    assert int(row["seats"][0]["score"]) == 0 and int(row["seats"][0]["rolls"]) == 2


def test_max_rounds_zero_is_safety_limit():
    table = _strats(gu.load("grid_vectors.json")["g64"])
    row = po.play_game(po.coord(103, 1, 2), table, [0, 1], max_rounds=0)[0]
    assert int(row["status"]) == 1 and int(row["n_rounds"]) == 0
    assert all(int(row["seats"][i]["n_turns"]) == 0 and int(row["seats"][i]["hit_max_rounds"]) == 1 for i in range(2))


def test_oracle_argument_errors():
    table = _strats(gu.load("grid_vectors.json")["g64"])
    with pytest.raises(po.OracleError):
        po.tournament(table, 3, 0, 0, 1)  # 64 % 3 != 0 (run_tournament.py:274)


def test_should_continue_cases():
    data = gu.load("scoring_vectors.json")
    for case in data["should_continue"]:
        s = _strats([case["strategy"]])
        got = po.should_continue(s, case["turn_score"], case["dice_left"], case["has_scored"], case["final_round"],
                                 case["score_to_beat"], case["player_score"])
        assert int(got) == case["out"], case


def test_random_table_games_match_reference():
    """Games on random legal strategy tables (k = 1..8, extreme thresholds, odd limits) from the Python reference."""
    for g in gu.load("fuzz_vectors.json")["games"]:
        table = _strats(g["strategies"])
        c = po.coord(103, g["root_seed"], g["k"], g["shuffle"], game_index=g["game"])
        row = po.play_game(c, table, list(range(g["k"])), g["target"], g["max_rounds"])[0]
        gu.assert_row_equal(gu.row_as_compact(row, g["k"], lambda i: table[i]["strategy_id"]), g["row"], ctx=str(g["root_seed"]))
