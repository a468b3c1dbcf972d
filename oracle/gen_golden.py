"""TEST INFRASTRUCTURE ONLY — generate tests/golden/*.json by running the upstream Python
reference (imported from /root/reference in the BUILD CONTAINER; see oracle/ref_import.py).

The fixtures are data only: inputs (coordinates, strategies, rolls) and the outputs the
reference produced for them, plus constants transcribed from the reference's own test
goldens (EXPECTED_ROWS, EXPECTED_H2H_BLOCKS, the deterministic-counts KAT).  Nothing here
travels to the GPU box except the generated JSON.

    python oracle/gen_golden.py            # regenerate everything (about a minute)
"""
from __future__ import annotations

import csv
import json
import sys
from pathlib import Path

import numpy as np


class _Enc(json.JSONEncoder):
    def default(self, o):
        if isinstance(o, np.integer):
            return int(o)
        if isinstance(o, np.bool_):
            return bool(o)
        return super().default(o)

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import ref_import  # noqa: E402

ref_import.import_reference()

from farkle.game.scoring import SCORE_TABLE, default_score  # noqa: E402
from farkle.simulation import run_tournament as rt  # noqa: E402
from farkle.simulation.game_profile import (  # noqa: E402
    GameProfile,
    H2HMaxRoundsOverride,
    TournamentMaxRoundsOverride,
)
from farkle.simulation.simulation import (  # noqa: E402
    PlayerRngCoordinates,
    _play_game,
    generate_strategy_grid,
    simulate_many_games,
)
from farkle.simulation.strategies import (  # noqa: E402
    FavorDiceOrScore,
    ThresholdStrategy,
    build_strategy_manifest,
)
from farkle.simulation.time_farkle import make_random_strategies  # noqa: E402
from farkle.utils import random as ur  # noqa: E402
from farkle.utils.random import RandomPurpose  # noqa: E402

OUT = HERE.parent / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)


def _dump(obj, fh):
    json.dump(obj, fh, cls=_Enc, separators=(",", ":"))
    fh.close()


def strat_tuple(s: ThresholdStrategy) -> list[int]:
    return [
        int(s.score_threshold), int(s.dice_threshold), int(s.smart_five), int(s.smart_one),
        int(s.consider_score), int(s.consider_dice), int(s.require_both), int(s.auto_hot_dice),
        int(s.run_up_score), int(s.favor_dice_or_score is FavorDiceOrScore.SCORE),
        -1 if s.strategy_id is None else int(s.strategy_id),
    ]


def row_to_compact(row: dict, k: int) -> dict:
    """Reduce a reference row dict to the integer fields the engine owns."""
    seats = []
    for i in range(1, k + 1):
        p = f"P{i}_"
        seats.append(
            [row[p + "score"], row[p + "strategy"], row[p + "farkles"], row[p + "rolls"], row[p + "n_turns"],
             row[p + "highest_turn"], row[p + "smart_five_uses"], row[p + "n_smart_five_dice"],
             row[p + "smart_one_uses"], row[p + "n_smart_one_dice"], row[p + "hot_dice"],
             0 if row[p + "rank"] is None else int(row[p + "rank"]), int(bool(row[p + "hit_max_rounds"]))]
        )
    return {
        "n_rounds": int(row["n_rounds"]),
        "status": 0 if row["termination_status"] == "completed" else 1,
        "winner_seat": -1 if row["winner_seat"] is None else int(row["winner_seat"][1:]) - 1,
        "winner_strategy": row["winner_strategy"],
        "winning_score": row["winning_score"],
        "victory_margin": row["victory_margin"],
        "seat_ranks": row["seat_ranks"],
        "game_seed": int(row["game_seed"]),
        "seats": [[int(v) for v in s] for s in seats],
    }


def grid(**kw) -> list[ThresholdStrategy]:
    strategies, _ = generate_strategy_grid(**kw)
    return strategies


def grid64():
    return grid(score_thresholds=[250, 300, 350, 400], dice_thresholds=[0, 1, 2, 3], smart_five_opts=[True],
                smart_one_opts=[True], consider_score_opts=[True], consider_dice_opts=[True],
                auto_hot_dice_opts=[True], run_up_score_opts=[True], include_stop_at=False,
                include_stop_at_heuristic=False)


def grid_oracle4():
    # tests/helpers/raw_simulation_oracle.py:108-118
    return grid(score_thresholds=[500], dice_thresholds=[2], smart_five_opts=[False], smart_one_opts=[False],
                consider_score_opts=[True], consider_dice_opts=[True], auto_hot_dice_opts=[False, True],
                run_up_score_opts=[False], include_stop_at=False, include_stop_at_heuristic=False)


def grid_default():
    return grid()


# ---------------------------------------------------------------------------
def gen_rng():
    rs = np.random.default_rng(20261004)
    cases = []
    for _ in range(24):
        purpose = int(rs.choice([1, 10, 11, 101, 102, 103, 202, 203]))
        kw = dict(root_seed=int(rs.integers(0, 2**62)) * 4 + int(rs.integers(0, 4)), k=int(rs.integers(0, 13)),
                  shuffle_index=int(rs.integers(0, 2**40)), pair_id=int(rs.integers(0, 5000)),
                  order=int(rs.integers(0, 2)), game_index=int(rs.integers(0, 2**33)), seat_index=int(rs.integers(0, 12)))
        g = ur.coordinate_rng(purpose, **kw)
        raw = [int(x) for x in g.bit_generator.random_raw(8)]
        sizes = [int(x) for x in rs.integers(1, 7, size=40)]
        g = ur.coordinate_rng(purpose, **kw)
        dice = [int(v) for n in sizes for v in g.integers(1, 7, size=n)]
        perms = {}
        for S in (4, 64, 80):
            g = ur.coordinate_rng(purpose, **kw)
            perms[str(S)] = [int(v) for v in g.permutation(S)]
        cases.append({"purpose": purpose, **kw, "raw64": raw, "sizes": sizes, "dice": dice, "perms": perms,
                      "seed32": ur.coordinate_seed(purpose, dtype=np.uint32, **kw),
                      "seed64": ur.coordinate_seed(purpose, dtype=np.uint64, **kw)})
    # one full-grid permutation and the spawn_seeds path
    g = ur.coordinate_rng(RandomPurpose.SHUFFLE_PERMUTATION, root_seed=0, k=4, shuffle_index=7)
    big = [int(v) for v in g.permutation(5160)]
    spawn = [int(v) for v in ur.spawn_seeds(16, seed=42)]
    _dump({"cases": cases, "perm5160": {"root_seed": 0, "k": 4, "shuffle_index": 7, "perm": big},
               "spawn_seeds_42": spawn}, open(OUT / "rng_vectors.json", "w"))


def scoring_csv_rows():
    """Data fixtures the reference holds for the scorer: its test CSV (tests/data/test_farkle_scores_data.csv, used by
    tests/unit/game/test_scoring.py:76) and the three data/*.csv tables that list explicit rolls."""
    rows = []
    for path in ("tests/data/test_farkle_scores_data.csv", "data/farkle_all_scoring_combos.csv",
                 "data/farkle_scores_data.csv", "data/farkle_missing_patterns_1.csv"):
        for row in csv.DictReader(open(f"/root/reference/{path}")):
            rows.append({"source": path, **{k: row[k] for k in ("Score", "Number_of_Dice", "Dice_Roll", "Used_Dice", "Reroll_Dice",
                                                               "Single_Fives", "Single_Ones")}})
    return rows


def gen_scoring_csv():
    """Refresh only the CSV part of scoring_vectors.json."""
    path = OUT / "scoring_vectors.json"
    data = json.load(open(path))
    data["csv_rows"] = scoring_csv_rows()
    _dump(data, open(path, "w"))


def gen_scoring():
    table = [[int(x) for x in (*key, v[0], v[1], v[3], v[4])] for key, v in SCORE_TABLE.items()]
    assert len(table) == 923
    rows = scoring_csv_rows()
    rs = np.random.default_rng(7)
    grids = grid_default()
    cases = []
    for _ in range(2500):
        n = int(rs.integers(1, 7))
        roll = [int(v) for v in rs.integers(1, 7, size=n)]
        if rs.random() < 0.35:  # bias toward ones/fives so discards trigger
            roll = [int(rs.choice([1, 5, 5, 1, 2, 3, 4, 6])) for _ in range(n)]
        s = grids[int(rs.integers(0, len(grids)))]
        pre = int(rs.integers(0, 30)) * 50
        res = default_score(roll, turn_score_pre=pre, smart_five=s.smart_five, smart_one=s.smart_one,
                            consider_score=s.consider_score, consider_dice=s.consider_dice,
                            require_both=s.require_both, score_threshold=s.score_threshold,
                            dice_threshold=s.dice_threshold, favor_dice_or_score=s.favor_dice_or_score,
                            return_discards=True)
        cases.append({"roll": roll, "pre": pre, "strategy": strat_tuple(s), "out": [int(v) for v in res]})
    dec = []
    for _ in range(1500):
        s = grids[int(rs.integers(0, len(grids)))]
        a = dict(turn_score=int(rs.integers(0, 40)) * 50, dice_left=int(rs.integers(1, 7)),
                 has_scored=bool(rs.integers(0, 2)), final_round=bool(rs.integers(0, 2)),
                 score_to_beat=int(rs.integers(90, 110)) * 100, running_total=int(rs.integers(90, 110)) * 100)
        dec.append({"strategy": strat_tuple(s), **{k: int(v) for k, v in a.items()},
                    "out": int(s.decide(score_needed=0, **a))})
    # FarklePlayer._should_continue (engine.py:156-205) through a real player object
    from farkle.game.engine import FarklePlayer

    cont = []
    for _ in range(1500):
        s = grids[int(rs.integers(0, len(grids)))]
        pl = FarklePlayer(name="P1", strategy=s, rng=None)
        pl.score = int(rs.integers(0, 120)) * 100
        pl.has_scored = bool(rs.integers(0, 2))
        a = dict(turn_score=int(rs.integers(1, 40)) * 50, dice_left=int(rs.integers(1, 7)),
                 final_round=bool(rs.integers(0, 2)), score_to_beat=int(rs.integers(95, 115)) * 100)
        cont.append({"strategy": strat_tuple(s), "player_score": pl.score, "has_scored": int(pl.has_scored),
                     **{k: int(v) for k, v in a.items()}, "out": int(pl._should_continue(target_score=10_000, **a))})
    _dump({"table": table, "csv_rows": rows, "default_score": cases, "decide": dec, "should_continue": cont},
              open(OUT / "scoring_vectors.json", "w"))


def play(strats, purpose, root, k, shuffle=0, pair=0, order=0, game=0, target=10_000, max_rounds=200):
    coords = PlayerRngCoordinates(purpose=purpose, root_seed=root, k=k, shuffle_index=shuffle, pair_id=pair,
                                  order=order, game_index=game)
    return _play_game(0, strats, target_score=target, max_rounds=max_rounds, player_rng_coordinates=coords)


def gen_games():
    rs = np.random.default_rng(99)
    out = {"grids": {}, "games": []}
    grids = {"g64": grid64(), "default": grid_default()}
    out["grids"]["g64"] = [strat_tuple(s) for s in grids["g64"]]
    # the default grid is regenerated by the build's own grid builder; store only its size and a digest sample
    out["grids"]["default_size"] = len(grids["default"])
    out["grids"]["default_sample"] = {str(i): strat_tuple(grids["default"][i]) for i in range(0, 5160, 129)}
    plan = [("g64", 2, 60, {}), ("g64", 4, 25, {}), ("g64", 8, 10, {}), ("default", 2, 60, {}), ("default", 3, 20, {}),
            ("default", 4, 30, {}), ("default", 5, 12, {}), ("default", 6, 12, {}), ("default", 8, 10, {}),
            ("default", 12, 6, {}), ("g64", 2, 20, {"target": 1500}), ("default", 4, 10, {"max_rounds": 3}),
            ("g64", 2, 6, {"max_rounds": 0}), ("default", 2, 10, {"target": 100, "max_rounds": 1})]
    for gname, k, n, extra in plan:
        g = grids[gname]
        for _ in range(n):
            idx = [int(v) for v in rs.choice(len(g), size=k, replace=False)]
            purpose = int(rs.choice([103, 203, 10])) if k == 2 else int(rs.choice([103, 10]))
            root = int(rs.integers(0, 2**63))
            kw = dict(shuffle=int(rs.integers(0, 10**6)), game=int(rs.integers(0, 3000)))
            if purpose == 203:
                kw = dict(pair=int(rs.integers(0, 1000)), order=int(rs.integers(0, 2)), game=int(rs.integers(0, 10**6)))
            if purpose == 10:
                kw = dict(game=int(rs.integers(0, 10**6)))
            row = play([g[i] for i in idx], purpose, root, k, target=extra.get("target", 10_000),
                       max_rounds=extra.get("max_rounds", 200), **kw)
            out["games"].append({"grid": gname, "strategies": idx, "k": k, "purpose": purpose, "root_seed": root,
                                 **{"shuffle": 0, "pair": 0, "order": 0, "game": 0, **kw},
                                 "target": extra.get("target", 10_000), "max_rounds": extra.get("max_rounds", 200),
                                 "row": row_to_compact(row, k)})
    # force some never-bank pairings (safety-limit games) on the 64-grid
    g = grids["g64"]
    never = [i for i, s in enumerate(g) if s.dice_threshold == 0 and s.require_both is False]
    for j in range(6):
        idx = [never[j], never[j + 1]]
        root = 1000 + j
        row = play([g[i] for i in idx], 103, root, 2, shuffle=j, game=j)
        out["games"].append({"grid": "g64", "strategies": idx, "k": 2, "purpose": 103, "root_seed": root,
                             "shuffle": j, "pair": 0, "order": 0, "game": j, "target": 10_000, "max_rounds": 200,
                             "row": row_to_compact(row, 2)})
    _dump(out, open(OUT / "game_vectors.json", "w"))


def counter_payload(wins, sums, sqs):
    return {"wins": {str(k): int(v) for k, v in wins.items()},
            "attempted": {str(k): int(v) for k, v in wins.attempted_exposures.items()},
            "completed": {str(k): int(v) for k, v in wins.completed_exposures.items()},
            "safety": {str(k): int(v) for k, v in wins.safety_limit_exposures.items()},
            "games": [wins.games_attempted, wins.games_completed, wins.games_safety_limit],
            "sums": {m: {str(k): int(v) for k, v in d.items()} for m, d in sums.items()},
            "sq_sums": {m: {str(k): int(v) for k, v in d.items()} for m, d in sqs.items()}}


def gen_tournament():
    out = {"cases": []}
    for name, strategies, k, root, shuffles, profile in [
        ("g64_k2", grid64(), 2, 42, [0, 1, 2, 312499], None),
        ("g64_k4", grid64(), 4, 42, [0, 5], None),
        ("g64_k8", grid64(), 8, 7, [3], None),
        ("oracle4_k2_r11", grid_oracle4(), 2, 11, [0, 1], "oracle"),
        ("oracle4_k4_r11", grid_oracle4(), 4, 11, [0, 1], "oracle"),
        ("oracle4_k2_r22", grid_oracle4(), 2, 22, [0, 1], "oracle"),
        ("oracle4_k4_r22", grid_oracle4(), 4, 22, [0, 1], "oracle"),
    ]:
        gp = None
        if profile == "oracle":  # tests/helpers/raw_simulation_oracle.py:57-78
            gp = GameProfile(default_target_score=100, default_max_rounds=200,
                             tournament_max_rounds_overrides=(TournamentMaxRoundsOverride(11, 2, 0, 0, 0),))
        cfg = rt.TournamentConfig(n_players=k, n_strategies=len(strategies))
        rt._init_worker(strategies, cfg, gp)
        for sh in shuffles:
            task = rt.ShuffleTask(root_seed=root, k=k, shuffle_index=sh, shuffle_seed=0, deterministic_batch_id=0)
            wins, sums, sqs, rows = rt._play_one_shuffle(task, collect_rows=True)
            perm = ur.coordinate_rng(RandomPurpose.SHUFFLE_PERMUTATION, root_seed=root, k=k, shuffle_index=sh).permutation(len(strategies))
            out["cases"].append({"name": name, "k": k, "root_seed": root, "shuffle": sh,
                                 "target": 100 if gp else 10_000, "profile": profile,
                                 "strategies": [strat_tuple(s) for s in strategies], "perm": [int(v) for v in perm],
                                 "tally": counter_payload(wins, sums, sqs),
                                 "rows": [row_to_compact(r, k) for r in rows]})
    # the reference's own EXPECTED_ROWS (tests/integration/test_raw_simulation_oracle.py:45-58):
    # (root, k, shuffle, game) -> (seat strategies, status, winner_strategy, n_rounds, sum n_turns, scores)
    out["EXPECTED_ROWS"] = [
        [[11, 2, 0, 0], [[0, 2], "safety_limit", None, 0, 0, [0, 0]]],
        [[11, 2, 0, 1], [[1, 3], "completed", 1, 2, 5, [1950, 1100]]],
        [[11, 2, 1, 0], [[2, 1], "completed", 2, 2, 4, [500, 0]]],
        [[11, 2, 1, 1], [[0, 3], "completed", 0, 1, 2, [600, 0]]],
        [[11, 4, 0, 0], [[0, 1, 2, 3], "completed", 2, 1, 4, [700, 0, 800, 0]]],
        [[11, 4, 1, 0], [[3, 2, 1, 0], "completed", 3, 1, 5, [3050, 2900, 0, 0]]],
        [[22, 2, 0, 0], [[3, 0], "completed", 3, 1, 2, [600, 0]]],
        [[22, 2, 0, 1], [[1, 2], "completed", 2, 1, 2, [500, 1100]]],
        [[22, 2, 1, 0], [[2, 0], "completed", 2, 1, 2, [950, 0]]],
        [[22, 2, 1, 1], [[3, 1], "completed", 3, 1, 3, [750, 550]]],
        [[22, 4, 0, 0], [[1, 2, 0, 3], "completed", 2, 1, 5, [0, 700, 0, 0]]],
        [[22, 4, 1, 0], [[0, 2, 1, 3], "completed", 1, 1, 4, [700, 0, 1100, 0]]],
    ]
    _dump(out, open(OUT / "tournament_vectors.json", "w"))


def gen_h2h():
    from farkle.analysis import h2h_schedule as h2h

    strategies = grid_oracle4()
    manifest = build_strategy_manifest(strategies)
    profile = GameProfile(default_target_score=100, default_max_rounds=200,
                          h2h_max_rounds_overrides=(H2HMaxRoundsOverride(11, 0, 0, 0, 0), H2HMaxRoundsOverride(11, 1, 0, 0, 0),
                                                    H2HMaxRoundsOverride(11, 1, 0, 1, 0)))
    pairs = {0: (0, 1), 1: (0, 3), 2: (1, 3)}
    blocks = []
    for pair_id, (a, b) in pairs.items():
        for root in (11, 22):
            for order in (0, 1):
                s1, s2 = (a, b) if order == 0 else (b, a)
                block = {"pair_id": pair_id, "root_seed": root, "order": order, "seat1_strategy": s1, "seat2_strategy": s2,
                         "n_completed_required": 1, "max_attempts": 2,
                         "rng_scheme_version": 2, "rng_purpose_namespace": 203}
                res = h2h._simulate_block_from_manifest(dict(block), manifest, 5000, profile)
                block["range_hash"] = res["attempt_coordinate_range_hash"]
                blocks.append({**block, "out": [res["games_attempted"], res["games_completed"], res["games_safety_limit"],
                                                res["wins_seat1"], res["wins_seat2"], res["wins_a"], res["wins_b"],
                                                res["replacement_attempt_count"], res["completion_status"]]})
    # bigger blocks on the 64-grid, default limits, incl. a chunked (resumed) block
    g = grid64()
    m64 = build_strategy_manifest(g)
    big = []
    for pair_id, (a, b), root, order, target, max_att, chunk in [(5, (3, 40), 42, 0, 40, 60, 5000), (5, (3, 40), 42, 1, 40, 60, 5000),
                                                                  (9, (0, 8), 7, 0, 30, 33, 5000), (2, (17, 63), 1, 1, 25, 50, 7)]:
        s1, s2 = (a, b) if order == 0 else (b, a)
        block = {"pair_id": pair_id, "root_seed": root, "order": order, "seat1_strategy": s1, "seat2_strategy": s2,
                 "n_completed_required": target, "max_attempts": max_att,
                 "rng_scheme_version": 2, "rng_purpose_namespace": 203}
        trace = []
        cur = dict(block)
        for _ in range(50):
            cur = h2h._simulate_block_from_manifest(dict(cur), m64, chunk, None)
            trace.append([cur["games_attempted"], cur["games_completed"], cur["games_safety_limit"], cur["wins_seat1"], cur["wins_seat2"]])
            if cur["completion_status"] != "partial_resumable":
                break
        big.append({**block, "chunk": chunk, "trace": trace, "status": cur["completion_status"]})
    # EXPECTED_H2H_BLOCKS from tests/helpers/tournament_analysis_oracle.py:65-78
    expected = [[[0, 11, 0], [2, 1, 1, 1, 0, 1, "complete"]], [[0, 11, 1], [1, 1, 0, 0, 1, 0, "complete"]],
                [[0, 22, 0], [1, 1, 0, 1, 0, 0, "complete"]], [[0, 22, 1], [1, 1, 0, 0, 1, 0, "complete"]],
                [[1, 11, 0], [2, 0, 2, 0, 0, 1, "unresolved_nonviable"]], [[1, 11, 1], [1, 1, 0, 0, 1, 0, "complete"]],
                [[1, 22, 0], [1, 1, 0, 0, 1, 0, "complete"]], [[1, 22, 1], [1, 1, 0, 0, 1, 0, "complete"]],
                [[2, 11, 0], [1, 1, 0, 1, 0, 0, "complete"]], [[2, 11, 1], [1, 1, 0, 0, 1, 0, "complete"]],
                [[2, 22, 0], [1, 1, 0, 0, 1, 0, "complete"]], [[2, 22, 1], [1, 1, 0, 0, 1, 0, "complete"]]]
    _dump({"oracle4": [strat_tuple(s) for s in strategies], "blocks": blocks, "g64_blocks": big,
               "EXPECTED_H2H_BLOCKS": expected}, open(OUT / "h2h_vectors.json", "w"))


def gen_time_path():
    out = {"random_strategies": [], "many_games": []}
    for players, seed in [(2, 42), (5, 42), (3, 7), (8, 123456789)]:
        out["random_strategies"].append({"players": players, "seed": seed,
                                         "strategies": [strat_tuple(s) for s in make_random_strategies(players, seed)]})
    # tests/unit/simulation/test_simulation.py:184-199 deterministic-counts KAT
    strategies = [ThresholdStrategy(score_threshold=0, dice_threshold=6), ThresholdStrategy(score_threshold=500, dice_threshold=3),
                  ThresholdStrategy(score_threshold=1000, dice_threshold=2)]
    df = simulate_many_games(n_games=10, strategies=strategies, target_score=5000, seed=123, n_jobs=1)
    out["kat_counts"] = {"strategies": [strat_tuple(s) for s in strategies], "target": 5000, "seed": 123, "n_games": 10,
                         "expected": {"P2": 6, "P1": 2, "P3": 2},
                         "winner_seat_counts": {str(k): int(v) for k, v in df["winner_seat"].value_counts().items()},
                         "rows": [row_to_compact(r, 3) for r in df.to_dict(orient="records")]}
    for players, seed, n in [(2, 42, 40), (5, 42, 12)]:
        strategies = make_random_strategies(players, seed)
        df = simulate_many_games(n_games=n, strategies=strategies, seed=seed, n_jobs=1)
        out["many_games"].append({"players": players, "seed": seed, "n_games": n,
                                  "strategies": [strat_tuple(s) for s in strategies],
                                  "game_seeds": [int(v) for v in df["game_seed"]],
                                  "rows": [row_to_compact(r, players) for r in df.to_dict(orient="records")]})
    _dump(out, open(OUT / "time_path_vectors.json", "w"))


def gen_grids():
    """Grid enumeration order (strategy_id = position): sizes + full tuples for the small grids."""
    import yaml

    fast = yaml.safe_load(open("/root/reference/configs/fast_config.yaml"))["sim"]
    keys = ["score_thresholds", "dice_thresholds", "smart_five_opts", "smart_one_opts", "consider_score_opts",
            "consider_dice_opts", "auto_hot_dice_opts", "run_up_score_opts"]
    fast_kw = {k: fast[k] for k in keys if k in fast}
    fast_grid = grid(**fast_kw, include_stop_at=bool(fast.get("include_stop_at", False)),
                     include_stop_at_heuristic=bool(fast.get("include_stop_at_heuristic", False)))
    d = grid_default()
    import hashlib

    digest = hashlib.sha256(json.dumps([strat_tuple(s) for s in d]).encode()).hexdigest()
    from farkle.utils.schema_helpers import raw_simulation_schema_for

    schemas = {str(k): [[f.name, str(f.type), bool(f.nullable)] for f in raw_simulation_schema_for(k)] for k in (2, 4)}
    _dump({"raw_schema": schemas, "fast_kwargs": fast_kw, "fast": [strat_tuple(s) for s in fast_grid], "g64": [strat_tuple(s) for s in grid64()],
               "oracle4": [strat_tuple(s) for s in grid_oracle4()], "default_size": len(d), "default_sha256": digest,
               "default_head": [strat_tuple(s) for s in d[:40]], "default_tail": [strat_tuple(s) for s in d[-40:]]},
              open(OUT / "grid_vectors.json", "w"))


def gen_fuzz():
    """Games on random legal strategy tables (extreme thresholds, every flag combination, k = 1..8, odd limits)."""
    rs = np.random.default_rng(77)
    games = []
    for _ in range(160):
        k = int(rs.choice([1, 2, 2, 3, 4, 5, 6, 8]))
        strats = []
        for i in range(k):
            sf = bool(rs.integers(0, 2))
            so = bool(rs.integers(0, 2)) if sf else False
            cs, cd = bool(rs.integers(0, 2)), bool(rs.integers(0, 2))
            rb = bool(rs.integers(0, 2)) if (cs and cd) else False
            strats.append(ThresholdStrategy(int(rs.choice([0, 50, 199, 250, 300, 500, 1000, 1350, 10000])), int(rs.integers(-1, 7)),
                                            sf, so, cs, cd, rb, bool(rs.integers(0, 2)), bool(rs.integers(0, 2)),
                                            FavorDiceOrScore.SCORE if rs.integers(0, 2) else FavorDiceOrScore.DICE, strategy_id=i))
        root, sh, gi = int(rs.integers(0, 2**63)), int(rs.integers(0, 10**6)), int(rs.integers(0, 3000))
        tgt, mr = int(rs.choice([100, 500, 2000, 10000, 20000])), int(rs.choice([0, 1, 3, 50, 200, 300]))
        row = play(strats, 103, root, k, shuffle=sh, game=gi, target=tgt, max_rounds=mr)
        games.append({"strategies": [strat_tuple(s) for s in strats], "k": k, "root_seed": root, "shuffle": sh, "game": gi,
                      "target": tgt, "max_rounds": mr, "row": row_to_compact(row, k)})
    _dump({"games": games}, open(OUT / "fuzz_vectors.json", "w"))


def gen_runner():
    """Workload plans and config path resolution of the reference's run surface."""
    from farkle.config import AppConfig, IOConfig, SimConfig
    from farkle.simulation.workload_planner import plan_tournament_workload

    plans = []
    for kw in [dict(root_seed=42, k=2, strategy_count=80, resolution_delta=0.03),
               dict(root_seed=0, k=4, strategy_count=5160, resolution_delta=0.03, batch_count=100, min_shuffles_per_batch=30),
               dict(root_seed=11, k=2, strategy_count=4, resolution_delta=0.9, batch_count=2, min_shuffles_per_batch=1),
               dict(root_seed=5, k=5, strategy_count=80, resolution_delta=0.1, confidence=0.9, batch_count=10, min_shuffles_per_batch=3, shuffle_cap=20),
               dict(root_seed=5, k=8, strategy_count=64, resolution_delta=0.05, projected_games_per_second=1e6)]:
        plans.append({"kwargs": kw, "plan": plan_tournament_workload(**kw).to_dict()})
    paths = []
    for prefix, seed, row_dir, n in [("results", 0, "rows", 5), ("results_x_seed_7", 7, "rows", 2), ("/abs/out", 3, "{n}p_rows", 4),
                                     ("res", 1, None, 2), ("res", 1, "sub/rows", 6), ("res", 9, "/abs/rows_{p}", 2)]:
        cfg = AppConfig(io=IOConfig(results_dir_prefix=Path(prefix)), sim=SimConfig(seed=seed, row_dir=None if row_dir is None else Path(row_dir)))
        rd = cfg.simulation_row_dir(n)
        paths.append({"prefix": prefix, "seed": seed, "row_dir": row_dir, "n": n, "results_root": str(cfg.results_root),
                      "n_dir": str(cfg.n_dir(n)), "row": None if rd is None else str(rd), "checkpoint": str(cfg.checkpoint_path(n)),
                      "manifest": str(cfg.strategy_manifest_root_path())})
    _dump({"plans": plans, "paths": paths}, open(OUT / "runner_vectors.json", "w"))


def gen_all_player():
    """The reference's unconditional all-player batch metrics (analysis/all_player_metrics.py:_iter_batch_tables -> _update_
    exposure_columns -> _finish_row, :257-470) over raw rows the reference itself simulated: rows of `_play_one_shuffle` for a
    few deterministic batches are written with the reference's raw row schema and fed to its own accumulation.  Cases include
    a never-banking table at a small round limit (safety-limit exposures: null rank / loss_margin, hit_max_rounds)."""
    import shutil
    import tempfile

    import pyarrow as pa
    import pyarrow.parquet as pq
    from farkle.analysis import all_player_metrics as apm
    from farkle.utils.schema_helpers import raw_simulation_schema_for

    class _Guard:
        def check_before_schedule(self, force: bool = False) -> None:
            return None

    never = [ThresholdStrategy(300, 0, True, True, True, True, True, True, True, s.favor_dice_or_score) for s in grid64()[:8]]
    cases = []
    tmp = Path(tempfile.mkdtemp(prefix="fk_apm_"))
    try:
        for name, strategies, k, root, n_sh, spb, max_rounds in [
            ("g64_k2", grid64(), 2, 42, 4, 2, 200), ("g64_k4", grid64(), 4, 7, 3, 2, 200), ("g64_k8", grid64(), 8, 3, 2, 2, 200),
            ("never8_k4_mr6", never, 4, 5, 4, 2, 6), ("g64_k2_mr12", grid64(), 2, 9, 3, 3, 12),
        ]:
            gp = GameProfile(default_target_score=10_000, default_max_rounds=max_rounds)
            cfg = rt.TournamentConfig(n_players=k, n_strategies=len(strategies))
            rt._init_worker(strategies, cfg, gp)
            rows = []
            for sh in range(n_sh):
                task = rt.ShuffleTask(root_seed=root, k=k, shuffle_index=sh, shuffle_seed=0, deterministic_batch_id=sh // spb)
                rows.extend(rt._play_one_shuffle(task, collect_rows=True)[3])
            schema = raw_simulation_schema_for(k)
            path = tmp / f"{name}.parquet"
            pq.write_table(pa.Table.from_pylist([{f.name: r.get(f.name) for f in schema} for r in rows], schema=schema), path)
            out_rows = []
            for table in apm._iter_batch_tables(path, k, max_batch_bytes=1 << 30, max_batch_rows=1 << 20, memory_guard=_Guard()):
                out_rows.extend(table.to_pylist())
            cases.append({"name": name, "k": k, "root_seed": root, "n_shuffles": n_sh, "shuffles_per_batch": spb, "max_rounds": max_rounds,
                          "strategies": [strat_tuple(s) for s in strategies], "columns": apm.all_player_batch_schema().names,
                          "batch_rows": [[r[c] for c in apm.all_player_batch_schema().names] for r in out_rows]})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    _dump({"cases": cases}, open(OUT / "all_player_vectors.json", "w"))


def gen_rng_lags():
    """The strategy-family rows of the reference's RNG diagnostics, by the reference's OWN code over rows it simulated: rows of
    ``_play_one_shuffle`` -> ``simulation_rows_to_table`` -> ``_extract_batch_arrays`` -> ``_observation_records``
    (analysis/rng_diagnostics.py:1092-1150, 1870-1905), the strategy records sorted by ``_observation_sort_order`` (:1964) and pushed
    through ``_OnlineMetric`` / ``_rows_for_online_group`` exactly as ``_write_stats_partition`` does (:2162-2206).  Frozen: the rows,
    the metric states' six sums per lag, and the value series per strategy (so that range merges can be checked at every cut)."""
    from farkle.analysis import rng_diagnostics as rd
    from farkle.simulation.simulation import simulation_rows_to_table

    lags = (1, 2, 5)
    out = {"lags": list(lags), "cases": []}
    strategies = grid(score_thresholds=[300, 500], dice_thresholds=[1, 2], smart_five_opts=[False, True], smart_one_opts=[False],
                      consider_score_opts=[True], consider_dice_opts=[True], auto_hot_dice_opts=[True], run_up_score_opts=[False])
    for k, root, n_sh, target, overrides in ((2, 7, 40, 1500, ((7, 2, 3, 1, 2), (7, 2, 17, 0, 0))), (4, 9, 24, 1000, ())):
        gp = GameProfile(default_target_score=target, default_max_rounds=200,
                         tournament_max_rounds_overrides=tuple(TournamentMaxRoundsOverride(*o) for o in overrides))
        cfg = rt.TournamentConfig(n_players=k, num_shuffles=n_sh, n_strategies=len(strategies))
        rt._init_worker(strategies, cfg, gp)
        rows = []
        for sh in range(n_sh):
            seed = ur.coordinate_seed(RandomPurpose.TOURNAMENT_SHUFFLE, root_seed=root, k=k, shuffle_index=sh, dtype=np.uint32)
            task = rt.ShuffleTask(root_seed=root, k=k, shuffle_index=sh, shuffle_seed=int(seed), deterministic_batch_id=sh // 8)
            rows.extend(rt._play_one_shuffle(task, collect_rows=True)[3])
        table = simulation_rows_to_table(rows, k)
        names = table.schema.names
        records = []
        for batch in table.to_batches(max_chunksize=37):  # several batches, as the stage streams them
            arrays = rd._extract_batch_arrays(batch, winner_col=rd._winner_column(set(names)),
                                              strat_cols=rd._seat_strategy_columns(None, names), expected_root_seed=root)
            records.append(rd._observation_records(arrays))
        records = np.concatenate(records)
        records = records[records["group_type"] == rd._GROUP_STRATEGY]
        records = records[rd._observation_sort_order(records)]
        dtype = records.dtype
        stats_rows, states, series = [], {}, {}
        current, rounds, wins = None, None, None

        def flush():
            stats_rows.extend(rd._rows_for_online_group(current, lags=lags, rounds=rounds, wins=wins))
            states[str(current[2])] = {name: {"n_obs": int(m.n_obs), "pair_count": m.pair_count.tolist(), "sum_x": m.sum_x.tolist(),
                                              "sum_y": m.sum_y.tolist(), "sum_x2": m.sum_x2.tolist(), "sum_y2": m.sum_y2.tolist(),
                                              "sum_xy": m.sum_xy.tolist()} for name, m in (("win_indicator", wins), ("n_rounds", rounds))}

        for record in records:
            identity = rd._group_identity(record, dtype)
            if identity != current:
                if current is not None:
                    flush()
                current, rounds, wins = identity, rd._OnlineMetric(lags), rd._OnlineMetric(lags)
            rounds.push(float(record["n_rounds"]))
            wins.push(float(record["win_indicator"]))
            series.setdefault(str(identity[2]), []).append([int(record["shuffle_index"]), int(record["n_rounds"]), int(record["win_indicator"])])
        flush()
        out["cases"].append({"k": k, "root_seed": root, "n_shuffles": n_sh, "target_score": target, "max_rounds": 200,
                             "overrides": [list(o) for o in overrides], "strategies": [strat_tuple(s) for s in strategies],
                             "rows": stats_rows, "states": states, "series": series,
                             "safety_limit_games": sum(1 for r in rows if r["termination_status"] != "completed")})
    _dump(out, open(OUT / "rng_lag_vectors.json", "w"))


def gen_sidecars():
    """The reference's simulation sidecar payloads (`_simulation_output_sidecar`, runner.py:338-376, canonical JSON) for every
    operation it publishes, and — the acceptance check — sidecars written by THIS engine's farkle_ii_amd/sidecars.py run through
    the reference's own `load_artifact_sidecar` + `validate_artifact_sidecar` (artifact_contract.py:607-668): the fixture is only
    written when the reference accepts them and rejects a tampered artifact."""
    import shutil
    import tempfile

    from farkle.config import AppConfig
    from farkle.simulation import runner
    from farkle.utils import artifact_contract as ac

    sys.path.insert(0, str(HERE.parent))
    from farkle_ii_amd import sidecars as mine
    from farkle_ii_amd.config import AppConfig as MyConfig

    cfg = AppConfig()
    sources = [Path("results_seed_0/strategy_manifest.parquet"), Path("results_seed_0/2_players/simulation_workload_plan.json")]
    payloads = {}
    for kind, operation in mine.OPERATIONS.items():
        sc = runner._simulation_output_sidecar(cfg, Path(f"out/{kind}.bin"), n_players=2, operation=operation,
                                               sources=() if kind == "strategy_manifest" else sources,
                                               support_counts=[2, 4] if kind == "strategy_manifest" else None)
        payloads[kind] = json.loads(ac._canonical_json(sc))
    tmp = Path(tempfile.mkdtemp(prefix="fk_sidecar_"))
    accepted = []
    try:
        for kind, operation in mine.OPERATIONS.items():
            art = tmp / f"{kind}.parquet"
            art.write_bytes(f"artifact bytes of {kind}".encode() * 7)
            template = mine.simulation_output_sidecar(MyConfig(), art, n_players=2, operation=operation,
                                                      sources=() if kind == "strategy_manifest" else sources,
                                                      support_counts=[2, 4] if kind == "strategy_manifest" else None)
            mine.write_sidecar(art, template)
            meta = ac.validate_artifact_sidecar(art, expected={"producer": "simulation", "operation": operation, "scope": "diagnostics"})
            assert meta.artifact_sha256 == mine.sha256_file(art)
            art.write_bytes(art.read_bytes() + b"!")
            try:
                ac.validate_artifact_sidecar(art)
            except ac.ArtifactContractError:
                accepted.append(kind)
            else:
                raise AssertionError("the reference accepted a tampered artifact")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    assert accepted == list(mine.OPERATIONS)
    _dump({"reference_payloads": payloads, "reference_validator_accepts_this_engines_sidecars": accepted,
           "engine_specific_fields": ["artifact_contract_version", "config_hash", "code_revision", "artifact_name", "artifact_sha256",
                                      "artifact_size_bytes"]}, open(OUT / "sidecar_vectors.json", "w"))


def gen_wilson():
    """worst_case_wilson_width / minimum_shuffles_for_resolution of the reference, bit patterns (float.hex): every sample size
    below 20 000 at which a re-associated form of the radicand (z^2 / n / (4 n) instead of z^2 / (4 n^2)) rounds differently,
    a spread of other sizes, and the resolution searches they feed."""
    import math

    from farkle.simulation.workload_planner import minimum_shuffles_for_resolution, worst_case_wilson_width
    from scipy.stats import norm

    def reassociated(n, confidence):
        z = float(norm.ppf(0.5 + confidence / 2.0))
        q = z * z / n
        best = 0.0
        for successes in {n // 2, n - n // 2}:
            p_hat = successes / n
            best = max(best, 2.0 * z * math.sqrt(p_hat * (1.0 - p_hat) / n + q / (4.0 * n)) / (1.0 + q))
        return best

    widths = []
    for confidence in (0.8, 0.9, 0.95, 0.99, 0.999):
        for n in range(1, 20001):
            ref = worst_case_wilson_width(n, confidence=confidence)
            if n <= 40 or n % 997 == 0 or ref != reassociated(n, confidence):
                widths.append([confidence, n, ref.hex()])
    searches = [[delta, confidence, minimum_shuffles_for_resolution(delta, confidence=confidence)]
                for confidence in (0.9, 0.95, 0.99) for delta in (0.5, 0.2, 0.1, 0.05, 0.03, 0.02, 0.01, 0.004, 0.0123)]
    _dump({"widths": widths, "searches": searches}, open(OUT / "wilson_vectors.json", "w"))


ARTIFACT_CONFIG = {
    # the reference's tiny oracle config (tests/helpers/raw_simulation_oracle.py:80-190) with a finer screening
    # resolution (more shuffles) and three deterministic batches; artifact_contract v3 sidecars are out of scope
    "sim": {"n_players_list": [2, 4], "seed": 11, "seed_list": [11], "n_jobs": 1, "expanded_metrics": True, "row_dir": "rows",
            "metric_chunk_dir": "metric_chunks", "desired_sec_per_chunk": 1, "ckpt_every_sec": 1, "score_thresholds": [500],
            "dice_thresholds": [2], "smart_five_opts": [False], "smart_one_opts": [False], "consider_score_opts": [True],
            "consider_dice_opts": [True], "auto_hot_dice_opts": [False, True], "run_up_score_opts": [False],
            "include_stop_at": False, "include_stop_at_heuristic": False},
    "screening": {"resolution_delta": 0.4, "interval_confidence": 0.95},
    "batching": {"target_batches": 3, "min_shuffles_per_batch": 2},
}


def _jsonable(obj):
    """Checkpoint payloads use int keys and Counter/defaultdict values: JSON wants string keys."""
    if isinstance(obj, dict):
        return {str(k): _jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_jsonable(v) for v in obj]
    return obj


def gen_artifacts():
    """Run the reference's ``run_single_n`` (runner.py:1326) on a tiny config and freeze every artifact it writes:
    parquet schemas + records, manifest records, workload plan, checkpoint payload.  The authenticated v3 sidecars
    are not produced here (artifact-contract version 2: the plain write path).  The stage-done stamp IS: ``write_stage_done``
    (utils/stage_completion.py:391-512) takes the code identity from ``cfg._code_identity`` when one is supplied
    (``_code_identity_payload`` :190-211), so the reference writes its own ``simulation.done.json`` without a Git checkout; it is
    frozen with the per-run path prefix replaced by ``<root>``."""
    import dataclasses
    import pickle
    import shutil
    import tempfile

    import pyarrow.parquet as pq
    import yaml
    from farkle.config import load_app_config
    from farkle.simulation import runner
    from farkle.utils.authenticated_contract import CodeIdentityError

    tmp = Path(tempfile.mkdtemp(prefix="fk_artifacts_"))
    try:
        payload = dict(ARTIFACT_CONFIG)
        payload["io"] = {"results_dir_prefix": str(tmp / "out"), "analysis_subdir": "analysis"}
        cfg_path = tmp / "tiny.yaml"
        cfg_path.write_text(yaml.safe_dump(payload))
        cfg = load_app_config(cfg_path, seed_list_len=1)
        # plain (pre-v3) write path: same bytes minus the authenticated sidecars
        cfg.artifact_contract = dataclasses.replace(cfg.artifact_contract, artifact_contract_version=2)
        cfg._code_identity = {"state": "supplied_by_caller", "commit": None, "dirty_fingerprint_sha256": None,
                              "revision": "reference imported from /root/reference (not a Git checkout)"}
        gp = GameProfile(default_target_score=100, default_max_rounds=200,
                         tournament_max_rounds_overrides=(TournamentMaxRoundsOverride(11, 2, 0, 0, 0),))
        cfg._game_profile_sha256 = gp.sha256  # what the run context binds when a profile is in use (orchestration/run_contexts.py)
        out = {"config": ARTIFACT_CONFIG, "game_profile": {"target": 100, "max_rounds": 200, "overrides": [[11, 2, 0, 0, 0]]},
               "runs": {}}
        volatile = {"ts", "pid"}
        for k in (2, 4):
            try:
                runner.run_single_n(cfg, k, oracle_game_profile=gp)
                stamp = "written"
            except CodeIdentityError:
                stamp = "needs the reference's Git identity (not produced here)"
            root = cfg.results_root
            n_dir = root / f"{k}_players"
            run = {"done_stamp": stamp, "files": sorted(str(f.relative_to(root)) for f in root.rglob("*")
                                                        if f.is_file() and (f.parent == root or n_dir in f.parents)),
                   "parquet": {}, "jsonl": {}}
            for f in sorted(root.rglob("*.parquet")):
                if f.parent == root or n_dir in f.parents:
                    t = pq.read_table(f)
                    run["parquet"][str(f.relative_to(root))] = {
                        "schema": [[fld.name, str(fld.type)] for fld in t.schema], "records": t.to_pylist()}
            for f in sorted(n_dir.rglob("*.jsonl")):
                run["jsonl"][str(f.relative_to(root))] = [
                    {kk: vv for kk, vv in json.loads(line).items() if kk not in volatile} for line in f.read_text().splitlines()]
            run["workload_plan"] = json.loads((n_dir / "simulation_workload_plan.json").read_text())
            done_file = n_dir / "simulation.done.json"
            if done_file.exists():  # the reference's own stamp; absolute paths -> <root>/...
                run["stage_done"] = json.loads(done_file.read_text().replace(str(root), "<root>"))
            ck = pickle.loads((n_dir / f"{k}p_checkpoint.pkl").read_bytes())
            run["checkpoint"] = {"win_totals": _jsonable(dict(ck["win_totals"])), "outcome_counts": _jsonable(ck["outcome_counts"]),
                                 "metric_sums": _jsonable({m: dict(v) for m, v in ck["metric_sums"].items()}),
                                 "metric_square_sums": _jsonable({m: dict(v) for m, v in ck["metric_square_sums"].items()}),
                                 "meta": _jsonable(ck["meta"])}
            out["runs"][str(k)] = run
        _dump(out, open(OUT / "artifact_vectors.json", "w"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def gen_resume():
    """A genuine MID-RUN checkpoint of the reference: its own ``run_single_n`` (plain v2 write path, no row shards / metric
    chunks, so the pickle is the only recovery authority) is interrupted right after its parent loop has written the
    checkpoint that owns the first process block.  Frozen: the pickle's bytes (base64 — data the reference wrote, its
    ``win_totals`` pickled through ``farkle.simulation.run_tournament._restore_outcome_counter``) and, from an
    uninterrupted run of the same configuration, the final checkpoint payload the resumed run must reproduce."""
    import base64
    import dataclasses
    import pickle
    import shutil
    import tempfile

    import yaml
    from farkle.config import load_app_config
    from farkle.simulation import run_tournament as rt
    from farkle.simulation import runner
    from farkle.utils.authenticated_contract import CodeIdentityError

    class _Interrupt(Exception):
        pass

    out = {"config": None, "runs": {}}
    for k in (2, 4):
        tmp = Path(tempfile.mkdtemp(prefix="fk_resume_"))
        try:
            payload = {key: dict(val) for key, val in ARTIFACT_CONFIG.items()}
            payload["sim"].update({"row_dir": None, "metric_chunk_dir": None, "ckpt_every_sec": 0, "n_players_list": [k]})
            payload["io"] = {"results_dir_prefix": str(tmp / "out"), "analysis_subdir": "analysis"}
            out["config"] = {key: val for key, val in payload.items() if key != "io"}
            cfg_path = tmp / "tiny.yaml"
            cfg_path.write_text(yaml.safe_dump(payload))

            def load():
                cfg = load_app_config(cfg_path, seed_list_len=1)
                cfg.artifact_contract = dataclasses.replace(cfg.artifact_contract, artifact_contract_version=2)
                return cfg

            cfg = load()
            ckpt = cfg.results_root / f"{k}_players" / f"{k}p_checkpoint.pkl"
            original = rt._save_checkpoint
            calls = {"n": 0}

            def interrupting(path, *args, **kwargs):
                original(path, *args, **kwargs)
                calls["n"] += 1
                raise _Interrupt()  # the process dies right after the first periodic checkpoint

            rt._save_checkpoint = interrupting
            try:
                runner.run_single_n(cfg, k)
                raise AssertionError("the reference run was not interrupted")
            except _Interrupt:
                pass
            finally:
                rt._save_checkpoint = original
            partial_bytes = ckpt.read_bytes()
            partial = pickle.loads(partial_bytes)
            shutil.rmtree(cfg.results_root, ignore_errors=True)
            try:
                runner.run_single_n(load(), k)
            except CodeIdentityError:
                pass  # everything but the authenticated stage stamp has been written
            full = pickle.loads(ckpt.read_bytes())
            out["runs"][str(k)] = {
                "partial_checkpoint_pickle_b64": base64.b64encode(partial_bytes).decode("ascii"),
                "partial_meta": _jsonable({key: partial["meta"][key] for key in ("completed_shuffle_indices",
                                          "completed_process_block_indices", "num_shuffles", "deterministic_batch_size")}),
                "partial_games_attempted": partial["outcome_counts"]["games_attempted"],
                "final": {"win_totals": _jsonable(dict(full["win_totals"])), "outcome_counts": _jsonable(full["outcome_counts"]),
                          "metric_sums": _jsonable({m: dict(v) for m, v in full["metric_sums"].items()}),
                          "metric_square_sums": _jsonable({m: dict(v) for m, v in full["metric_square_sums"].items()}),
                          "meta": _jsonable(full["meta"])}}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    _dump(out, open(OUT / "resume_vectors.json", "w"))


if __name__ == "__main__":
    if len(sys.argv) > 1:  # e.g. `python oracle/gen_golden.py gen_artifacts`
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    gen_artifacts()
    gen_resume()
    gen_runner()
    gen_wilson()
    gen_all_player()
    gen_rng_lags()
    gen_sidecars()
    gen_fuzz()
    gen_rng()
    gen_scoring()
    gen_games()
    gen_tournament()
    gen_h2h()
    gen_time_path()
    gen_grids()
    for p in sorted(OUT.glob("*.json")):
        print(p.name, p.stat().st_size)
