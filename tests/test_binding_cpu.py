"""The binding that ran INSIDE the reference (oracle/gen_binding.py, build container): its fixture on the CPU.

`tests/golden/binding_vectors.json` holds the engine calls recorded while the reference's own `runner.run_single_n` and
`execute_h2h_schedule` ran with `farkle_ii_amd.reference_binding` installed (artifacts equal to the unpatched reference run —
asserted by the generator before it writes the fixture).  Here: the fixture's claims, the replay of the calls on the oracle
stub, the INTEGRATION.md snippets being the ones that were executed, and the binding class itself on a stand-in module."""
from __future__ import annotations

import hashlib
import re
import sys
import types
from collections import defaultdict
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(Path(__file__).resolve().parent))

import golden_util as gu  # noqa: E402
from oracle_engine_stub import Engine as StubEngine  # noqa: E402

from farkle_ii_amd import reference_binding as rb  # noqa: E402
from farkle_ii_amd import tournament as tn  # noqa: E402
from farkle_ii_amd.backend import OVERRIDE_DTYPE  # noqa: E402
from farkle_ii_amd.strategies import STRATEGY_DTYPE, generate_strategy_grid  # noqa: E402


def test_fixture_records_a_v3_authenticated_run_of_the_reference():
    doc = gu.load("binding_vectors.json")
    t = doc["tournament"]
    assert t["artifact_contract_version"] == 3
    assert sorted(t["stage_stamps"]) == ["2_players/simulation.done.json", "4_players/simulation.done.json"]
    assert t["sidecar_count"] >= 50 and len(t["calls"]) == 6
    for rel, ck in t["checkpoints"].items():  # the pickles hold the REFERENCE's counter type, filled by the binding
        assert ck["win_totals_type"] == "farkle.simulation.run_tournament.OutcomeCounter", rel
    for stamp in t["stage_done"].values():  # the v3 stamp (release_identity.write_v3_stage_completion), written by the reference
        assert stamp["state"] == "complete_valid" and stamp["lifecycle_contract_version"] == 1
        assert len(stamp["stage_identity_sha256"]) == 64 and stamp["outputs"]
    h = doc["h2h"]
    assert h["runs"]["block_runner"] == {"engine_calls": 12, "blocks_written_by_the_reference": 12}
    assert h["runs"]["prefetching_block_runner"]["engine_calls"] == 2  # one shared launch per root
    got = {(int(r["pair_id"]), int(r["root_seed"]), int(r["order"])): [int(r["games_attempted"]), int(r["games_completed"]),
           int(r["games_safety_limit"]), int(r["wins_a"]), int(r["wins_b"]), int(r["replacement_attempt_count"]), r["completion_status"]]
           for r in h["order_counts"]}
    assert got == {tuple(k): v for k, v in h["expected_blocks"]}  # EXPECTED_H2H_BLOCKS, tournament_analysis_oracle.py:65-78


def test_recorded_calls_replay_on_the_oracle():
    doc = gu.load("binding_vectors.json")
    eng = StubEngine()
    n = gu.replay_binding_calls(eng, doc["tournament"]["calls"], STRATEGY_DTYPE, OVERRIDE_DTYPE)
    for variant in ("tournament_metric_chunks_no_rows", "tournament_counts_only"):  # served per CHUNK: one tally per deterministic batch
        calls = doc[variant]["calls"]
        assert all(c["shuffles_per_batch"] == c["shuffle_end"] - c["shuffle_begin"] and not c["want_rows"] for c in calls)
        n += gu.replay_binding_calls(eng, calls, STRATEGY_DTYPE, OVERRIDE_DTYPE)
    for mode in ("block_runner", "prefetching_block_runner"):
        n += gu.replay_binding_calls(eng, doc["h2h"]["calls"][mode], STRATEGY_DTYPE, OVERRIDE_DTYPE)
    assert n == 6 + 6 + 6 + 12 + 2


def test_integration_md_snippets_are_the_ones_that_ran():
    text = (ROOT / "INTEGRATION.md").read_text(encoding="utf-8")
    found = dict(re.findall(r"<!-- binding:([a-z0-9_]+) -->\s*```python\n(.*?)```", text, flags=re.S))
    want = gu.load("binding_vectors.json")["integration_snippets_sha256"]
    assert sorted(found) == sorted(want) == ["h2h_block_runner", "h2h_prefetching_block_runner", "tournament"]
    for name, code in found.items():
        assert hashlib.sha256(code.encode("utf-8")).hexdigest() == want[name], \
            f"INTEGRATION.md snippet {name} changed: re-run oracle/gen_binding.py in the build container"
    assert "overrides_from" not in text and "counters_from" not in text  # round 3's pseudo-code is gone


def _stand_in_module(strategies, k):
    """What TournamentBinding needs of `farkle.simulation.run_tournament`: the state, the coercion, the counter class and
    chunk bodies that look their shuffle functions up in the module namespace at call time."""
    rt = types.SimpleNamespace()
    rt.OutcomeCounter = tn.OutcomeCounter
    rt._coerce_shuffle_task = tn._coerce_shuffle_task
    rt._STATE = types.SimpleNamespace(strats=list(strategies), cfg=tn.TournamentConfig(n_players=k, n_strategies=len(strategies)),
                                      game_profile=None)

    def _run_chunk(tasks):
        total = rt.OutcomeCounter()
        for t in tasks:
            total.absorb(rt._play_shuffle(t))
        return total

    def _run_chunk_metrics(tasks, *, collect_rows=False, row_dir=None, seen_rows=None):
        wins_total = rt.OutcomeCounter()
        sums = {m: defaultdict(float) for m in tn.METRIC_LABELS}
        sqs = {m: defaultdict(float) for m in tn.METRIC_LABELS}
        for t in tasks:
            wins, s, q, rows = rt._play_one_shuffle(t, collect_rows=collect_rows)
            wins_total.absorb(wins)
            for m in tn.METRIC_LABELS:
                for key, v in s[m].items():
                    sums[m][key] += v
                for key, v in q[m].items():
                    sqs[m][key] += v
            if seen_rows is not None:
                seen_rows.extend(rows)
        return wins_total, sums, sqs

    rt._run_chunk, rt._run_chunk_metrics = _run_chunk, _run_chunk_metrics
    rt.progress = []
    rt.report_worker_progress = lambda name, *, event_id, counters: rt.progress.append((name, event_id, dict(counters)))
    rt._play_shuffle = rt._play_one_shuffle = lambda *a, **kw: (_ for _ in ()).throw(AssertionError("unpatched shuffle function"))
    return rt


def test_tournament_binding_serves_a_chunk_from_one_launch_and_restores_the_module():
    strategies, _ = generate_strategy_grid(score_thresholds=[300, 500], dice_thresholds=[1, 2], smart_five_opts=[False, True],
                                           smart_one_opts=[False], consider_score_opts=[True], consider_dice_opts=[True],
                                           auto_hot_dice_opts=[True], run_up_score_opts=[False])
    k = 2
    rt = _stand_in_module(strategies, k)
    originals = {name: getattr(rt, name) for name in rb.TournamentBinding._NAMES}
    tasks = tn.shuffle_tasks(7, k, 3, 9, 3)
    eng = StubEngine()
    binding = rb.TournamentBinding(rt, engine=eng)
    rows: list = []
    with binding:
        # row shards to write: the reference's own chunk body runs, every shuffle's rows served from ONE launch
        wins, sums, sqs = rt._run_chunk_metrics(tasks, collect_rows=True, row_dir=Path("rows"), seen_rows=rows)
        assert binding.launches == 1 and rt.progress == []  # (the stand-in body reports nothing; the reference's reports per shuffle)
        # nothing to write: the chunk is ONE tally, converted once — no per-shuffle counters, one progress event with the summed counters
        wins2, sums2, sqs2 = rt._run_chunk_metrics(tasks)
        plain = rt._run_chunk(tasks)
        assert binding.launches == 3 and not binding._served
        gps = len(strategies) // k
        assert rt.progress == [("simulation_shuffle_complete", "simulation:7:2:3-8", {"worker_completed_shuffles": 6, "worker_completed_games": 6 * gps})] * 2
        single = rt._play_shuffle(tasks[2])  # outside a chunk: played alone
    assert binding.launches == 4
    assert dict(wins2) == dict(wins) and wins2.outcome_payload() == wins.outcome_payload()
    assert {m: dict(v) for m, v in sums2.items()} == {m: dict(v) for m, v in sums.items()}
    assert {m: dict(v) for m, v in sqs2.items()} == {m: dict(v) for m, v in sqs.items()}
    assert {name: getattr(rt, name) for name in originals} == originals
    # the same numbers as this package's own chunk function on the same engine
    from farkle_ii_amd import engine as engine_holder

    engine_holder.set_engine(eng)
    try:
        tn._init_worker(strategies, rt._STATE.cfg)
        want_wins, want_sums, want_sqs = tn._run_chunk_metrics(tasks)
        want_single = tn._play_shuffle(tasks[2])
    finally:
        engine_holder.set_engine(None)
    assert dict(wins) == dict(want_wins) == dict(plain) and wins.outcome_payload() == want_wins.outcome_payload()
    assert {m: dict(v) for m, v in sums.items()} == {m: dict(v) for m, v in want_sums.items()}
    assert {m: dict(v) for m, v in sqs.items()} == {m: dict(v) for m, v in want_sqs.items()}
    assert dict(single) == dict(want_single)
    assert len(rows) == len(tasks) * (len(strategies) // k) and rows[0]["shuffle_index"] == 3 and rows[-1]["game_index"] == len(strategies) // k - 1


def test_binding_refuses_a_worker_pool_and_a_forked_caller():
    """The binding exists in the process that installed it.  While installed, `rt.parallel.process_map(..., n_jobs != 1)` raises
    (spawned workers would re-import the module and play on the reference's CPU path: run_tournament.py:1576-1586), n_jobs = 1
    goes through to the real function, a patched callable reached from another process (fork) raises, and uninstalling puts the
    real module back.  The same refusal inside the REFERENCE's run_single_n is recorded in the fixture (oracle/gen_binding.py)."""
    import os

    import pytest

    strategies, _ = generate_strategy_grid(score_thresholds=[300], dice_thresholds=[2], smart_five_opts=[False], smart_one_opts=[False],
                                           consider_score_opts=[True], consider_dice_opts=[True], auto_hot_dice_opts=[True, False],
                                           run_up_score_opts=[False])
    rt = _stand_in_module(strategies, 2)
    calls = []
    real_parallel = types.SimpleNamespace(
        normalize_n_jobs=lambda n, default=1: default if n is None else (os.cpu_count() if int(n) <= 0 else int(n)),
        process_map=lambda fn, items, *, n_jobs=None, **kw: calls.append((n_jobs, sorted(kw))) or [fn(i) for i in items],
        other_attribute="kept")
    rt.parallel = real_parallel
    binding = rb.TournamentBinding(rt, engine=StubEngine())
    with binding:
        assert rt.parallel is not real_parallel and rt.parallel.other_attribute == "kept"
        for bad in (2, 8, 0, -1):  # 0 / -1: "all cores" in the reference's normalisation
            if bad <= 0 and (os.cpu_count() or 1) == 1:
                continue
            with pytest.raises(rb.BindingWorkerPoolError, match="n_jobs"):
                rt.parallel.process_map(len, [[1]], n_jobs=bad, initializer=None)
        assert rt.parallel.process_map(len, [[1, 2], [3]], n_jobs=1, window=4) == [2, 1]
        assert rt.parallel.process_map(len, [[1]], n_jobs=None) == [1]
        assert calls == [(1, ["window"]), (None, [])]
        tasks = tn.shuffle_tasks(7, 2, 0, 2, 2)
        binding._pid = os.getpid() + 1  # as a forked worker would see it
        for call in (lambda: rt._run_chunk(tasks), lambda: rt._run_chunk_metrics(tasks), lambda: rt._play_shuffle(tasks[0]),
                     lambda: rt._play_one_shuffle(tasks[0])):
            with pytest.raises(rb.BindingWorkerPoolError, match="forked worker"):
                call()
        binding._pid = os.getpid()
        assert dict(rt._run_chunk(tasks))  # the installing process still plays
    assert rt.parallel is real_parallel
    msg = gu.load("binding_vectors.json")["tournament"]["n_jobs_2_refused"]
    assert "n_jobs = 2" in msg and "sim.n_jobs = 1" in msg  # raised inside the reference's own run_single_n (sim.n_jobs = 2)


def test_coerce_game_profile_keeps_the_identity_hash():
    from farkle_ii_amd.game_profile import GameProfile, H2HMaxRoundsOverride, TournamentMaxRoundsOverride

    ours = GameProfile(100, 200, (TournamentMaxRoundsOverride(11, 2, 0, 0, 0),), (H2HMaxRoundsOverride(11, 1, 0, 1, 0),))
    foreign = types.SimpleNamespace(default_target_score=100, default_max_rounds=200,
                                    tournament_max_rounds_overrides=(types.SimpleNamespace(root_seed=11, k=2, shuffle_index=0, game_index=0, max_rounds=0),),
                                    h2h_max_rounds_overrides=(types.SimpleNamespace(root_seed=11, pair_id=1, order=0, attempt_index=1, max_rounds=0),))
    assert rb.coerce_game_profile(foreign) == ours and rb.coerce_game_profile(foreign).sha256 == ours.sha256
    assert rb.coerce_game_profile(None) is None and rb.coerce_game_profile(ours) is ours
    # the profile of the fixture run: same sha256 as the reference computed for its own GameProfile object
    h = gu.load("binding_vectors.json")["h2h"]
    fixture = GameProfile(100, 200, (TournamentMaxRoundsOverride(11, 2, 0, 0, 0),),
                          tuple(H2HMaxRoundsOverride(*o) for o in h["game_profile"]["h2h_overrides"]))
    assert fixture.sha256 == h["schedule"][0]["game_profile_sha256"]


def _ctypes_blocks():
    text = (ROOT / "INTEGRATION.md").read_text(encoding="utf-8")
    return re.findall(r"<!-- ctypes:([a-z0-9_]+) -->\s*```python\n(.*?)```", text, flags=re.S)


def test_integration_md_ctypes_blocks_compile_and_use_declared_entry_points():
    blocks = _ctypes_blocks()
    assert [name for name, _ in blocks] == ["load", "tournament", "play_games", "h2h", "h2h_blocks", "all_player"]
    header = (ROOT / "include" / "farkle_hip.h").read_text()
    declared = set(re.findall(r"\b(fk_[a-z0-9_]+)\s*\(", header))
    for name, code in blocks:
        compile(code, f"INTEGRATION.md[ctypes:{name}]", "exec")
        for fn in re.findall(r"lib\.(fk_[a-z0-9_]+)", code):
            assert fn in declared, f"{name}: {fn} is not declared in include/farkle_hip.h"
    # the numpy record layouts of the load block are the header's structs (same sizes as the product binding's)
    ns: dict = {"LIB_PATH": None}
    load = dict(blocks)["load"]
    layout_only = load[load.index("STRATEGY = "):load.index("def ptr")]
    import numpy as _np

    ns["np"] = _np
    exec(compile(layout_only, "INTEGRATION.md[ctypes:load/layouts]", "exec"), ns)
    from farkle_ii_amd import backend as be

    assert ns["STRATEGY"] == be.STRATEGY_DTYPE and ns["SEAT"] == be.SEAT_DTYPE and ns["OVERRIDE"] == be.OVERRIDE_DTYPE
    assert ns["COORD"] == be.COORD_DTYPE and ns["H2H_BLOCK"] == be.H2H_BLOCK_DTYPE and ns["row_dtype"](3) == be.row_dtype(3)
