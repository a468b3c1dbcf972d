"""GPU tests of the host-side mirror of the reference's operator surface: these read like the reference's
own tests (tests/unit/simulation/test_simulation.py, test_run_tournament.py, tests/unit/analysis/test_h2h_schedule.py)
but every game runs in the HIP kernels."""
from __future__ import annotations

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


def _strategies(tuples):
    from farkle_ii_amd.strategies import FavorDiceOrScore, ThresholdStrategy

    out = []
    for t in tuples:
        out.append(ThresholdStrategy(t[0], t[1], bool(t[2]), bool(t[3]), bool(t[4]), bool(t[5]), bool(t[6]), bool(t[7]), bool(t[8]),
                                     FavorDiceOrScore.SCORE if t[9] else FavorDiceOrScore.DICE, strategy_id=None if t[10] < 0 else t[10]))
    return out


def _compact(row: dict, k: int) -> dict:
    seats = [[row[f"P{i}_{n}"] for n in ("score", "strategy", "farkles", "rolls", "n_turns", "highest_turn", "smart_five_uses",
                                          "n_smart_five_dice", "smart_one_uses", "n_smart_one_dice", "hot_dice")]
             + [0 if row[f"P{i}_rank"] is None else row[f"P{i}_rank"], int(row[f"P{i}_hit_max_rounds"])] for i in range(1, k + 1)]
    return {"n_rounds": row["n_rounds"], "status": 0 if row["termination_status"] == "completed" else 1,
            "winner_seat": -1 if row["winner_seat"] is None else int(row["winner_seat"][1:]) - 1, "seats": seats,
            "winner_strategy": row["winner_strategy"], "winning_score": row["winning_score"], "victory_margin": row["victory_margin"],
            "seat_ranks": row["seat_ranks"], "game_seed": row["game_seed"]}


def test_simulate_many_games_deterministic_counts():
    # tests/unit/simulation/test_simulation.py:184-199
    from farkle_ii_amd.simulation import simulate_many_games
    from farkle_ii_amd.strategies import ThresholdStrategy

    strategies = [ThresholdStrategy(score_threshold=0, dice_threshold=6), ThresholdStrategy(score_threshold=500, dice_threshold=3),
                  ThresholdStrategy(score_threshold=1000, dice_threshold=2)]
    df = simulate_many_games(n_games=10, strategies=strategies, target_score=5000, seed=123, n_jobs=1)
    assert df["winner_seat"].value_counts().to_dict() == {"P2": 6, "P1": 2, "P3": 2}
    df2 = simulate_many_games(n_games=10, strategies=strategies, target_score=5000, seed=123, n_jobs=4)
    assert df.equals(df2)
    with pytest.raises(ValueError, match="explicit seed"):
        simulate_many_games(n_games=1, strategies=strategies)


def test_simulate_many_games_rows_match_reference():
    from farkle_ii_amd.simulation import simulate_many_games

    data = gu.load("time_path_vectors.json")
    for block in data["many_games"]:
        strategies = _strategies(block["strategies"])
        df = simulate_many_games(n_games=block["n_games"], strategies=strategies, seed=block["seed"])
        assert [int(v) for v in df["game_seed"]] == block["game_seeds"]
        for got, gold in zip(df.to_dict(orient="records"), block["rows"]):
            c = _compact(got, block["players"])
            for key in gold:
                assert c[key] == gold[key], key


def test_play_game_single_rows_and_errors():
    from farkle_ii_amd.random import RandomPurpose
    from farkle_ii_amd.simulation import PlayerRngCoordinates, _play_game

    data = gu.load("game_vectors.json")
    g64 = _strategies(data["grids"]["g64"])
    for g in [x for x in data["games"] if x["grid"] == "g64"][:25]:
        coords = PlayerRngCoordinates(purpose=RandomPurpose(g["purpose"]), root_seed=g["root_seed"], k=g["k"],
                                      shuffle_index=g["shuffle"], pair_id=g["pair"], order=g["order"], game_index=g["game"])
        row = _play_game(0, [g64[i] for i in g["strategies"]], target_score=g["target"], max_rounds=g["max_rounds"],
                         player_rng_coordinates=coords)
        c = _compact(row, g["k"])
        for key in ("n_rounds", "status", "winner_seat", "seats", "winner_strategy", "winning_score", "victory_margin", "seat_ranks"):
            assert c[key] == g["row"][key], key
    with pytest.raises(ValueError, match="coordinate k"):
        _play_game(0, g64[:3], player_rng_coordinates=PlayerRngCoordinates(purpose=RandomPurpose.PLAYER, root_seed=1, k=2))


def test_play_one_shuffle_matches_reference():
    # the reference's _play_one_shuffle (run_tournament.py:301-393): wins, sums, square sums, rows with provenance
    from farkle_ii_amd import tournament as rt
    from farkle_ii_amd.game_profile import GameProfile, TournamentMaxRoundsOverride

    data = gu.load("tournament_vectors.json")
    for case in data["cases"]:
        strategies = _strategies(case["strategies"])
        profile = None
        if case["profile"] == "oracle":
            profile = GameProfile(default_target_score=100, tournament_max_rounds_overrides=(TournamentMaxRoundsOverride(11, 2, 0, 0, 0),))
        rt._init_worker(strategies, rt.TournamentConfig(n_players=case["k"], n_strategies=len(strategies)), profile)
        task = rt.ShuffleTask(case["root_seed"], case["k"], case["shuffle"], 0, 0)
        wins, sums, sqs, rows = rt._play_one_shuffle(task, collect_rows=True)
        gold = case["tally"]
        assert {str(k): v for k, v in wins.items()} == gold["wins"]
        assert {str(k): v for k, v in wins.safety_limit_exposures.items()} == gold["safety"]
        assert [wins.games_attempted, wins.games_completed, wins.games_safety_limit] == gold["games"]
        for m in rt.METRIC_LABELS:
            assert {str(k): int(v) for k, v in sums[m].items()} == gold["sums"][m]
            assert {str(k): int(v) for k, v in sqs[m].items()} == gold["sq_sums"][m]
        assert len(rows) == len(case["rows"])
        for got, g in zip(rows, case["rows"]):
            c = _compact(got, case["k"])
            for key in g:
                assert c[key] == g[key], (case["name"], key)
            assert got["shuffle_index"] == case["shuffle"] and got["rng_purpose_namespace"] == 102
        assert rt._play_shuffle(task) == wins


def test_run_chunk_and_row_shards(tmp_path):
    import json

    import pyarrow.parquet as pq

    from farkle_ii_amd import tournament as rt
    from farkle_ii_amd.rows import raw_simulation_schema_for, simulation_rows_to_table

    strategies = _strategies(gu.load("grid_vectors.json")["g64"])
    rt._init_worker(strategies, rt.TournamentConfig(n_players=2, n_strategies=64, deterministic_batch_size=4))
    tasks = rt.shuffle_tasks(42, 2, 0, 10, 4)
    total = rt._run_chunk(tasks)
    per = rt.OutcomeCounter()
    for t in tasks:
        per.absorb(rt._play_shuffle(t))
    assert total == per and total.attempted_exposures == per.attempted_exposures and total.games_attempted == 320
    wins, sums, sqs = rt._run_chunk_metrics(tasks[2:6], collect_rows=True, row_dir=tmp_path / "rows")
    assert wins.games_attempted == 4 * 32
    records = [json.loads(line) for line in open(tmp_path / "rows" / "manifest.jsonl")]
    assert [r["shuffle_index"] for r in records] == [2, 3, 4, 5] and records[0]["path"] == "rows_42_2p_000000000002.parquet"
    table = pq.read_table(tmp_path / "rows" / records[0]["path"])
    assert table.schema == raw_simulation_schema_for(2) and table.num_rows == 32
    _, _, _, rows = rt._play_one_shuffle(tasks[2], collect_rows=True)
    assert table.to_pylist() == simulation_rows_to_table(rows, 2).to_pylist()
    # non-contiguous task lists are split into contiguous launches
    odd = [tasks[0], tasks[2], tasks[3], tasks[7]]
    got = rt._run_chunk(odd)
    exp = rt.OutcomeCounter()
    for t in odd:
        exp.absorb(rt._play_shuffle(t))
    assert got == exp


def test_h2h_block_runner_matches_goldens(tmp_path):
    import pandas as pd

    from farkle_ii_amd.game_profile import GameProfile, H2HMaxRoundsOverride
    from farkle_ii_amd.h2h import gpu_block_runner

    data = gu.load("h2h_vectors.json")
    strategies = _strategies(data["oracle4"])
    manifest = pd.DataFrame([{"strategy_id": s.strategy_id, "score_threshold": s.score_threshold, "dice_threshold": s.dice_threshold,
                              "smart_five": s.smart_five, "smart_one": s.smart_one, "consider_score": s.consider_score,
                              "consider_dice": s.consider_dice, "require_both": s.require_both, "auto_hot_dice": s.auto_hot_dice,
                              "run_up_score": s.run_up_score, "favor_dice_or_score": s.favor_dice_or_score.value} for s in strategies])
    path = tmp_path / "strategy_manifest.parquet"
    manifest.to_parquet(path)
    profile = GameProfile(default_target_score=100, h2h_max_rounds_overrides=(H2HMaxRoundsOverride(11, 0, 0, 0, 0),
                                                                              H2HMaxRoundsOverride(11, 1, 0, 0, 0),
                                                                              H2HMaxRoundsOverride(11, 1, 0, 1, 0)))
    runner = gpu_block_runner(profile)
    expected = {tuple(k): v for k, v in data["EXPECTED_H2H_BLOCKS"]}
    for b in data["blocks"]:
        block = {k: b[k] for k in ("pair_id", "root_seed", "order", "seat1_strategy", "seat2_strategy", "n_completed_required", "max_attempts")}
        out = runner(block, path, 5000)
        got = [out["games_attempted"], out["games_completed"], out["games_safety_limit"], out["wins_a"], out["wins_b"],
               out["replacement_attempt_count"], out["completion_status"]]
        assert got == expected[(b["pair_id"], b["root_seed"], b["order"])]
        assert out["games_attempted"] == out["games_completed"] + out["games_safety_limit"]
        assert out["wins_seat1"] + out["wins_seat2"] == out["games_completed"]


def test_farkle_time_path():
    from farkle_ii_amd.time_farkle import measure_sim_times

    out = measure_sim_times(n_games=200, players=2, seed=42, jobs=1)
    assert sum(out["winners"].values()) <= 200 and out["games_per_sec"] > 0


def test_farkle_time_cli(capsys):
    from farkle_ii_amd.cli import main

    main(["time", "--players", "2", "--n-games", "1000", "--seed", "42"])
    assert "1000 games, 2 players" in capsys.readouterr().out


def _h2h_gpu_worker(rank: int, world: int, port: int, out_path: str) -> None:
    import os
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np
    import torch.distributed as dist

    from farkle_ii_amd.backend import STRATEGY_DTYPE, Engine
    from farkle_ii_amd.distributed import h2h_block_distributed

    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = Engine(0)  # both ranks share the one GPU of the test box; on a node each rank opens its own device
    seats = np.zeros(2, dtype=STRATEGY_DTYPE)
    seats[0] = (300, 2, 1, 1, 1, 1, 0, 1, 1, 1, 0)
    seats[1] = (450, 1, 1, 0, 1, 1, 1, 0, 1, 0, 1)
    st = np.zeros(5, dtype=np.uint64)
    for _ in range(3):  # three chunks of a 200 000-game block, the last one holds the cut
        st = h2h_block_distributed(eng.h2h, seats, 42, 5, 0, 200_000, 300_000, 90_000, state=st)
    if rank == 0:
        np.save(out_path, st.astype(np.int64))
    eng.close()
    dist.destroy_process_group()


def test_h2h_block_two_ranks_matches_single_engine(tmp_path):
    """Two gloo ranks cut one H2H block's attempt range in two and reproduce the single-engine prefix exactly."""
    import os

    import torch.multiprocessing as mp

    from farkle_ii_amd.backend import STRATEGY_DTYPE
    from farkle_ii_amd.engine import get_engine

    out = str(tmp_path / "h2h.npy")
    mp.spawn(_h2h_gpu_worker, args=(2, 33500 + os.getpid() % 2000, out), nprocs=2, join=True)
    seats = np.zeros(2, dtype=STRATEGY_DTYPE)
    seats[0] = (300, 2, 1, 1, 1, 1, 0, 1, 1, 1, 0)
    seats[1] = (450, 1, 1, 0, 1, 1, 1, 0, 1, 0, 1)
    eng = get_engine()
    st = np.zeros(5, dtype=np.uint64)
    for _ in range(3):
        st = eng.h2h(seats, 42, 5, 0, 200_000, 300_000, 90_000, state=st)
    got = np.load(out)
    assert np.array_equal(got, st.astype(np.int64)) and got[1] == 200_000 and got[0] >= got[1]


def test_prefetching_block_runner_equals_the_per_block_runner_on_10000_blocks():
    """The schedule-level runner (h2h.PrefetchingBlockRunner -> h2h.run_blocks -> fk_h2h_run_blocks) serves the reference's
    serial custom-runner loop (h2h_schedule.py:2038-2093) block by block with exactly what gpu_block_runner returns."""
    from h2h_schedule_util import make_schedule, manifest_frame, serial_schedule_loop

    from farkle_ii_amd.h2h import PrefetchingBlockRunner, gpu_block_runner, run_blocks
    from farkle_ii_amd.strategies import STRATEGY_DTYPE

    table = gu.strategies_from_tuples(gu.load("grid_vectors.json")["g64"], STRATEGY_DTYPE)
    table["strategy_id"] = np.arange(len(table))
    manifest = manifest_frame(table)
    blocks = make_schedule(table, 10_000, seed=11, target_range=(20, 120))
    want, calls = serial_schedule_loop(blocks, gpu_block_runner(), manifest, 64)
    runner = PrefetchingBlockRunner(blocks)
    got, calls2 = serial_schedule_loop(blocks, runner, manifest, 64)
    assert calls == calls2 and got == want
    assert runner.single_block_calls == 0 and runner.generations <= 8  # (the checkpoint limit is learned from the calls)
    told = PrefetchingBlockRunner(blocks, chunk_games=64)
    got, _ = serial_schedule_loop(blocks, told, manifest, 64)
    assert got == want and told.single_block_calls == 0 and told.generations <= 3  # 180 attempts at most = three chunks
    assert run_blocks(blocks, manifest, None) == want  # and in one call, to the terminal states


def test_reference_binding_calls_replay_on_the_hip_engine():
    """The engine calls recorded while the binding ran INSIDE the reference (its own run_single_n / execute_h2h_schedule writing
    artifacts equal to an unpatched run, oracle/gen_binding.py) — same arguments, same results from the HIP kernels."""
    from farkle_ii_amd.backend import OVERRIDE_DTYPE
    from farkle_ii_amd.engine import get_engine
    from farkle_ii_amd.strategies import STRATEGY_DTYPE

    doc = gu.load("binding_vectors.json")
    eng = get_engine()
    n = gu.replay_binding_calls(eng, doc["tournament"]["calls"], STRATEGY_DTYPE, OVERRIDE_DTYPE)
    for variant in ("tournament_metric_chunks_no_rows", "tournament_counts_only"):  # the per-chunk service (round 6): one tally per batch
        n += gu.replay_binding_calls(eng, doc[variant]["calls"], STRATEGY_DTYPE, OVERRIDE_DTYPE)
    for mode in ("block_runner", "prefetching_block_runner"):
        n += gu.replay_binding_calls(eng, doc["h2h"]["calls"][mode], STRATEGY_DTYPE, OVERRIDE_DTYPE)
    assert n == 32


def test_reference_binding_on_a_stand_in_module_runs_the_hip_engine():
    """TournamentBinding with the process's HIP engine (engine=None), on the stand-in module of tests/test_binding_cpu.py."""
    import pyoracle as po
    from test_binding_cpu import _stand_in_module

    from farkle_ii_amd import reference_binding as rb
    from farkle_ii_amd import tournament as tn
    from farkle_ii_amd.strategies import generate_strategy_grid, pack_strategies

    strategies, _ = generate_strategy_grid(score_thresholds=[300, 500], dice_thresholds=[1, 2], smart_five_opts=[False, True],
                                           smart_one_opts=[False], consider_score_opts=[True], consider_dice_opts=[True],
                                           auto_hot_dice_opts=[True], run_up_score_opts=[False])
    rt = _stand_in_module(strategies, 4)
    tasks = tn.shuffle_tasks(5, 4, 0, 7, 7)
    with rb.TournamentBinding(rt) as binding:
        wins, sums, sqs = rt._run_chunk_metrics(tasks, collect_rows=False)
    assert binding.launches == 1
    ref = po.tournament(pack_strategies(strategies).view(po.STRATEGY_DTYPE), 4, 5, 0, 7)
    want, want_sums, want_sqs = tn.tally_to_counters(ref["tally"][0], [int(s.strategy_id) for s in strategies], 4)
    assert dict(wins) == dict(want) and wins.outcome_payload() == want.outcome_payload()
    assert {m: dict(v) for m, v in sums.items()} == {m: dict(v) for m, v in want_sums.items()}
    assert {m: dict(v) for m, v in sqs.items()} == {m: dict(v) for m, v in want_sqs.items()}


def test_integration_md_ctypes_blocks_run():
    """INTEGRATION.md §1-§4c as ONE program on the MI355X: the ctypes stub a reference maintainer would add, checked against the oracle."""
    import re
    from pathlib import Path

    import pyoracle as po

    from farkle_ii_amd.backend import LIB_PATH
    from farkle_ii_amd.strategies import generate_strategy_grid, pack_strategies

    root = Path(__file__).resolve().parent.parent
    blocks = re.findall(r"<!-- ctypes:([a-z0-9_]+) -->\s*```python\n(.*?)```", (root / "INTEGRATION.md").read_text(encoding="utf-8"), flags=re.S)
    ns: dict = {"LIB_PATH": str(LIB_PATH)}
    for name, code in blocks:
        exec(compile(code, f"INTEGRATION.md[ctypes:{name}]", "exec"), ns)
    try:
        strategies, _ = generate_strategy_grid(score_thresholds=[250, 300, 350, 400], dice_thresholds=[0, 1, 2, 3], smart_five_opts=[True],
                                               smart_one_opts=[True], consider_score_opts=[True], consider_dice_opts=[True],
                                               auto_hot_dice_opts=[True], run_up_score_opts=[True])
        table = ns["pack"](strategies)
        assert table.tobytes() == pack_strategies(strategies).tobytes()
        otable = table.view(po.STRATEGY_DTYPE)
        tally, rows = ns["tournament_chunk"](table, 4, 42, 3, 11, shuffles_per_batch=3, collect_rows=True)
        ref = po.tournament(otable, 4, 42, 3, 11, shuffles_per_batch=3, want_rows=True)
        assert np.array_equal(tally, ref["tally"]) and rows.tobytes() == ref["rows"].tobytes()
        seat_strategy = np.tile(np.array([5, 40, 17], dtype=np.int32), (50, 1))
        got = ns["play_games"](table, seat_strategy, 3, 123, 50, target_score=5000)
        coords = np.zeros(50, po.COORD_DTYPE)
        coords["purpose"], coords["root_seed"], coords["k"], coords["game_index"] = 10, 123, 3, np.arange(50)
        assert got.tobytes() == po.play_games(coords, otable, seat_strategy, 3, target_score=5000).tobytes()
        block = {"root_seed": 42, "pair_id": 5, "order": 0, "n_completed_required": 200, "max_attempts": 400}
        out = ns["h2h_block"](table[[3, 40]], block, 10**6)
        want = po.h2h_block(otable[[3, 40]], 42, 5, 0, 200, 400, 10**6)
        assert [out[key] for key in ns["STATE_KEYS"]] == [int(v) for v in want]
        pending = [dict(block, pair_id=5 + i, order=i & 1) for i in range(3)]
        pairs = [table[[3, 40]], table[[0, 2]], table[[17, 9]]]
        outs = ns["h2h_blocks"](pairs, pending, 42, 300)
        for i, o in enumerate(outs):
            w = po.h2h_block(pairs[i].view(po.STRATEGY_DTYPE), 42, 5 + i, i & 1, 200, 400, 300)
            assert [o[key] for key in ns["STATE_KEYS"]] == [int(v) for v in w], i
        from oracle_engine_stub import seat_ratio_sums_from_rows, seat_stats_from_rows

        t2, stats, ratios = ns["all_player_accumulators"](table, 4, 42, 3, 11, 3)
        assert np.array_equal(t2, ref["tally"])
        assert np.array_equal(stats, seat_stats_from_rows(ref["rows"], 4, len(table), len(table) // 4, 3))
        assert ratios.tobytes() == seat_ratio_sums_from_rows(ref["rows"], 4, len(table), len(table) // 4, 3).tobytes()
    finally:
        ns["lib"].fk_destroy.restype = None
        ns["lib"].fk_destroy.argtypes = [ns["C"].c_void_p]
        ns["lib"].fk_destroy(ns["ctx"])


def test_baseline_config0_fast_config_file_through_farkle_run_on_the_hip_engine(tmp_path):
    """BASELINE configs[0] as a FILE: configs/fast_config.yaml with the YAML overlay {sim: {n_players_list: [2], seed_list: [42]}}
    (lists do not go through the reference's --set) through `farkle run` on the HIP engine; the checkpoint's totals against the
    oracle over every game of the plan."""
    import pickle
    from pathlib import Path

    import pyoracle as po
    import yaml

    from farkle_ii_amd import checkpoint as ckpt
    from farkle_ii_amd import engine as eng_mod
    from farkle_ii_amd import runner
    from farkle_ii_amd import tournament as tn
    from farkle_ii_amd.cli import main
    from farkle_ii_amd.config import load_app_config
    from farkle_ii_amd.strategies import pack_strategies

    root = Path(__file__).resolve().parent.parent
    base = yaml.safe_load((root / "configs" / "fast_config.yaml").read_text())
    base["sim"].update({"n_players_list": [2], "seed_list": [42]})          # the overlay
    base["io"] = {"results_dir_prefix": str(tmp_path / "fast")}
    cfg_path = tmp_path / "fast_config_overlaid.yaml"
    cfg_path.write_text(yaml.safe_dump(base))
    eng_mod.set_engine(None)
    main(["--config", str(cfg_path), "run", "--metrics"])
    cfg = load_app_config(cfg_path, seed_list_len=1)
    payload = pickle.loads(cfg.checkpoint_path(2).read_bytes())
    n_sh = payload["meta"]["num_shuffles"]
    strategies, grid = runner._resolve_strategies(cfg, None)
    assert grid == 80 and n_sh * 40 >= 100_000 and payload["meta"]["complete"]
    got = ckpt.payload_to_tally(payload, [int(s.strategy_id) for s in strategies], tn.METRIC_LABELS)
    want = po.tournament(pack_strategies(strategies).view(po.STRATEGY_DTYPE), 2, 42, 0, n_sh, n_threads=8)["tally"][0]
    assert np.array_equal(got, want)
    assert eng_mod.get_engine().device_info()["arch"].startswith("gfx")  # the HIP engine played them
