"""TEST INFRASTRUCTURE ONLY — run this engine's binding INSIDE the reference, in the build container, and freeze the proof.

The reference is imported from /root/reference (oracle/ref_import.py); the engine behind the binding is the CPU oracle stub
(tests/oracle_engine_stub.py — there is no GPU in this container), wrapped in a recorder.  Three things happen:

(a) TOURNAMENT.  The reference's own ``runner.run_single_n`` (simulation/runner.py:1326 -> ``run_tournament.run_tournament``,
    run_tournament.py:1050) runs twice on the same tiny configuration, artifact-contract version 3, with the code identity
    handed in through ``cfg._code_identity`` (``stage_completion._code_identity_payload`` :190-211,
    ``release_identity.publish_staged_v3_from_metadata`` :683 take a supplied identity; /root/reference is not a Git checkout):
    once unpatched, once with ``farkle_ii_amd.reference_binding.TournamentBinding`` installed on
    ``farkle.simulation.run_tournament``.  Every artifact the reference writes — row shards, manifests, metric chunks,
    checkpoint pickle + parquet summaries, workload plan, v3 sidecars, the authenticated ``simulation.done.json`` — must be
    equal between the two runs (volatile fields: timestamps, pids, and digests of files that contain them).
(b) H2H.  ``plan_h2h_schedule`` + ``execute_h2h_schedule`` (analysis/h2h_schedule.py:632, 1597) on the candidate family
    {0, 1, 3} of the tiny oracle configuration (the schedule behind ``EXPECTED_H2H_BLOCKS``,
    tests/helpers/tournament_analysis_oracle.py:65-78): once with the reference's default runner, once with
    ``reference_binding.block_runner`` and once with ``reference_binding.prefetching_block_runner``.  The block parquets the
    REFERENCE writes (``_write_block`` :1471, after ``_normalize_runner_result`` :1422) and ``root_order_counts.parquet`` must
    be equal, and equal to EXPECTED_H2H_BLOCKS.
(c) The code that does (a) and (b) is not written here: it is the fenced blocks of INTEGRATION.md marked
    ``<!-- binding:NAME -->``, extracted and executed — the document's snippets are the code that ran.

Output: tests/golden/binding_vectors.json = the recorded engine calls (arguments -> results) of the patched runs, replayed on
the oracle stub by tests/test_binding_cpu.py and on the HIP engine by tests/test_host_gpu.py (-m gpu), plus the sha256 of the
executed snippets.  Only data travels: nothing of the reference's source.

    python oracle/gen_binding.py
"""
from __future__ import annotations

import base64
import hashlib
import json
import pickle
import re
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
for p in (HERE, ROOT, ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import ref_import  # noqa: E402

ref_import.import_reference()

import pandas as pd  # noqa: E402
import pyarrow as pa  # noqa: E402
import pyarrow.parquet as pq  # noqa: E402
import yaml  # noqa: E402
from farkle.analysis.h2h_schedule import execute_h2h_schedule, plan_h2h_schedule  # noqa: E402
from farkle.config import AppConfig, ArtifactScope, IOConfig, SimConfig, load_app_config  # noqa: E402
from farkle.simulation import run_tournament as rt  # noqa: E402
from farkle.simulation import runner  # noqa: E402
from farkle.simulation.game_profile import GameProfile, H2HMaxRoundsOverride, TournamentMaxRoundsOverride  # noqa: E402
from farkle.simulation.simulation import generate_strategy_grid  # noqa: E402
from farkle.simulation.strategies import build_strategy_manifest  # noqa: E402
from farkle.utils.artifact_contract import make_artifact_sidecar  # noqa: E402
from farkle.utils.artifacts import write_json_artifact_atomic, write_parquet_artifact_atomic  # noqa: E402
from farkle.utils.authenticated_contract import CodeIdentity  # noqa: E402

from oracle_engine_stub import Engine as StubEngine  # noqa: E402

OUT = ROOT / "tests" / "golden" / "binding_vectors.json"
INTEGRATION = ROOT / "INTEGRATION.md"

# the reference's tiny oracle configuration (tests/helpers/raw_simulation_oracle.py:80-190), simulation part
TINY_SIM = {"n_players_list": [2, 4], "seed": 11, "seed_list": [11], "n_jobs": 1, "expanded_metrics": True, "row_dir": "rows",
            "metric_chunk_dir": "metric_chunks", "desired_sec_per_chunk": 1, "ckpt_every_sec": 1, "score_thresholds": [500],
            "dice_thresholds": [2], "smart_five_opts": [False], "smart_one_opts": [False], "consider_score_opts": [True],
            "consider_dice_opts": [True], "auto_hot_dice_opts": [False, True], "run_up_score_opts": [False],
            "include_stop_at": False, "include_stop_at_heuristic": False}
TINY_CONFIG = {"sim": TINY_SIM, "screening": {"resolution_delta": 0.4, "interval_confidence": 0.95},
               "batching": {"target_batches": 3, "min_shuffles_per_batch": 2}}
EXPECTED_H2H_BLOCKS = {  # tests/helpers/tournament_analysis_oracle.py:65-78 (pair, root, order) -> attempted, completed, safety, wins_a, wins_b, replacements, status
    (0, 11, 0): (2, 1, 1, 1, 0, 1, "complete"), (0, 11, 1): (1, 1, 0, 0, 1, 0, "complete"),
    (0, 22, 0): (1, 1, 0, 1, 0, 0, "complete"), (0, 22, 1): (1, 1, 0, 0, 1, 0, "complete"),
    (1, 11, 0): (2, 0, 2, 0, 0, 1, "unresolved_nonviable"), (1, 11, 1): (1, 1, 0, 0, 1, 0, "complete"),
    (1, 22, 0): (1, 1, 0, 0, 1, 0, "complete"), (1, 22, 1): (1, 1, 0, 0, 1, 0, "complete"),
    (2, 11, 0): (1, 1, 0, 1, 0, 0, "complete"), (2, 11, 1): (1, 1, 0, 0, 1, 0, "complete"),
    (2, 22, 0): (1, 1, 0, 0, 1, 0, "complete"), (2, 22, 1): (1, 1, 0, 0, 1, 0, "complete"),
}


def fixture_code_identity() -> CodeIdentity:
    """The identity handed to the reference's writers in place of a Git checkout (its API takes a supplied identity)."""
    tag = b"farkle_ii_amd binding fixture (oracle/gen_binding.py)"
    return CodeIdentity(commit=hashlib.sha1(tag).hexdigest(), policy="development_dirty", state="development_dirty",
                        dirty_fingerprint_sha256=hashlib.sha256(tag).hexdigest())


def oracle_profile() -> GameProfile:
    """tests/helpers/raw_simulation_oracle.py:59-78."""
    return GameProfile(default_target_score=100, default_max_rounds=200,
                       tournament_max_rounds_overrides=(TournamentMaxRoundsOverride(11, 2, 0, 0, 0),),
                       h2h_max_rounds_overrides=(H2HMaxRoundsOverride(11, 0, 0, 0, 0), H2HMaxRoundsOverride(11, 1, 0, 0, 0),
                                                 H2HMaxRoundsOverride(11, 1, 0, 1, 0)))


class RecordingEngine:
    """Engine-shaped wrapper: forwards to the wrapped engine and keeps (arguments -> results) of every call."""

    def __init__(self, inner):
        self.inner = inner
        self.calls: list[dict] = []

    def __getattr__(self, name):
        return getattr(self.inner, name)

    def tournament(self, table, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch=None, target_score=10_000,
                   max_rounds=200, overrides=None, want_rows=False, **kw):
        res = self.inner.tournament(table, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch=shuffles_per_batch,
                                    target_score=target_score, max_rounds=max_rounds, overrides=overrides, want_rows=want_rows, **kw)
        self.calls.append({"method": "tournament", "table": np.asarray(table).tolist(), "k": int(k), "root_seed": int(root_seed),
                           "shuffle_begin": int(shuffle_begin), "shuffle_end": int(shuffle_end),
                           "shuffles_per_batch": shuffles_per_batch, "target_score": int(target_score), "max_rounds": int(max_rounds),
                           "overrides": [] if overrides is None else np.asarray(overrides).tolist(), "want_rows": bool(want_rows),
                           "tally": np.asarray(res["tally"]).tolist(),
                           "rows_b64": base64.b64encode(res["rows"].tobytes()).decode("ascii") if want_rows else None})
        return res

    def h2h(self, seats, root_seed, pair_id, order, target, max_attempts, chunk_games, target_score=10_000, max_rounds=200,
            overrides=None, state=None):
        out = self.inner.h2h(seats, root_seed, pair_id, order, target, max_attempts, chunk_games, target_score=target_score,
                             max_rounds=max_rounds, overrides=overrides, state=state)
        self.calls.append({"method": "h2h", "seats": np.asarray(seats).tolist(), "root_seed": int(root_seed), "pair_id": int(pair_id),
                           "order": int(order), "target": int(target), "max_attempts": int(max_attempts), "chunk_games": int(chunk_games),
                           "target_score": int(target_score), "max_rounds": int(max_rounds),
                           "overrides": [] if overrides is None else np.asarray(overrides).tolist(),
                           "state_in": None if state is None else [int(v) for v in state], "state_out": [int(v) for v in out]})
        return out

    def h2h_blocks(self, seats, root_seed, pair_ids, orders, target, max_attempts, chunk_games=None, target_score=10_000,
                   max_rounds=200, overrides=None, states=None):
        out = self.inner.h2h_blocks(seats, root_seed, pair_ids, orders, target, max_attempts, chunk_games=chunk_games,
                                    target_score=target_score, max_rounds=max_rounds, overrides=overrides, states=states)
        self.calls.append({"method": "h2h_blocks", "seats": np.asarray(seats).tolist(), "root_seed": int(root_seed),
                           "pair_ids": [int(v) for v in pair_ids], "orders": [int(v) for v in orders],
                           "target": np.asarray(target).astype(np.int64).tolist(), "max_attempts": np.asarray(max_attempts).astype(np.int64).tolist(),
                           "chunk_games": None if chunk_games is None else int(chunk_games), "target_score": int(target_score),
                           "max_rounds": int(max_rounds), "overrides": [] if overrides is None else np.asarray(overrides).tolist(),
                           "states_in": None if states is None else np.asarray(states).astype(np.int64).tolist(),
                           "states_out": np.asarray(out).astype(np.int64).tolist()})
        return out


# ---- INTEGRATION.md snippets ---------------------------------------------------------------------------------------------

def integration_snippets() -> dict[str, str]:
    """Fenced python blocks of INTEGRATION.md that follow a ``<!-- binding:NAME -->`` marker."""
    text = INTEGRATION.read_text(encoding="utf-8")
    found = dict(re.findall(r"<!-- binding:([a-z0-9_]+) -->\s*```python\n(.*?)```", text, flags=re.S))
    if not found:
        raise SystemExit("INTEGRATION.md has no <!-- binding:NAME --> snippets")
    return found


def run_snippet(name: str, namespace: dict) -> dict:
    code = integration_snippets()[name]
    exec(compile(code, f"INTEGRATION.md[{name}]", "exec"), namespace)  # noqa: S102 - our own document
    return namespace


# ---- (a) tournament ---------------------------------------------------------------------------------------------------------

VOLATILE_KEYS = {"ts", "pid", "created_at", "completed_at", "written_at", "elapsed_seconds", "wall_seconds"}


def _scrub(obj, volatile_values: set[str]):
    if isinstance(obj, dict):
        return {k: _scrub(v, volatile_values) for k, v in obj.items() if k not in VOLATILE_KEYS}
    if isinstance(obj, list):
        return [_scrub(v, volatile_values) for v in obj]
    if isinstance(obj, str) and obj in volatile_values:
        return "<digest of a file with volatile fields>"
    return obj


def snapshot_tree(root: Path, pickle_bytes_are_volatile: bool = False) -> dict:
    """Everything under ``root`` in comparable form.  Files whose bytes contain pids / timestamps (JSON-lines manifests) change
    their SHA-256 from run to run; every digest that is the SHA-256 of such a file is replaced by a marker in the JSON files that
    mention it (sidecars, the completion stamp)."""
    files = sorted(p for p in root.rglob("*") if p.is_file())
    digests = {p: hashlib.sha256(p.read_bytes()).hexdigest() for p in files}
    snap: dict = {"files": [str(p.relative_to(root)) for p in files], "parquet": {}, "jsonl": {}, "json": {}, "pickle": {}}
    for p in files:
        rel = str(p.relative_to(root))
        if p.suffix == ".parquet":
            t = pq.read_table(p)
            snap["parquet"][rel] = {"schema": [[f.name, str(f.type)] for f in t.schema], "records": t.to_pylist()}
        elif p.suffix == ".jsonl":
            snap["jsonl"][rel] = [json.loads(line) for line in p.read_text().splitlines()]
        elif p.suffix == ".json":
            snap["json"][rel] = json.loads(p.read_text())
        elif p.suffix == ".pkl":
            ck = pickle.loads(p.read_bytes())
            snap["pickle"][rel] = {"win_totals": {str(k): v for k, v in dict(ck["win_totals"]).items()},
                                   "win_totals_type": type(ck["win_totals"]).__module__ + "." + type(ck["win_totals"]).__name__,
                                   "outcome_counts": json.loads(json.dumps(ck["outcome_counts"], default=lambda o: dict(o), sort_keys=True)),
                                   "metric_sums": {m: {str(k): v for k, v in d.items()} for m, d in (ck.get("metric_sums") or {}).items()},
                                   "metric_square_sums": {m: {str(k): v for k, v in d.items()} for m, d in (ck.get("metric_square_sums") or {}).items()},
                                   "meta": json.loads(json.dumps(ck.get("meta"), default=str, sort_keys=True))}
    # which files are volatile: any JSON / JSONL carrying a volatile key, then transitively any JSON mentioning their digest
    def has_volatile(o):
        if isinstance(o, dict):
            return any(k in VOLATILE_KEYS for k in o) or any(has_volatile(v) for v in o.values())
        if isinstance(o, list):
            return any(has_volatile(v) for v in o)
        return False
    volatile_files = {root / rel for rel, recs in snap["jsonl"].items() if has_volatile(recs)}
    volatile_files |= {root / rel for rel, doc in snap["json"].items() if has_volatile(doc)}
    volatile_digests = {digests[p] for p in volatile_files}
    if pickle_bytes_are_volatile:
        # counts-only checkpoints pickle the merged OutcomeCounter as it is: its dictionaries list strategies in FIRST-SEEN order — the order
        # seats and outcomes came up in the reference's per-shuffle loop, table order when a chunk arrives as one tally.  Same mappings
        # (compared by content below), different bytes: the pickles' digests, their sidecars and the stamps naming those are volatile here.
        volatile_files |= {p for p in files if p.suffix == ".pkl"}
        volatile_digests |= {digests[p] for p in files if p.suffix == ".pkl"}
    for _ in range(6):  # sidecars of volatile files, stamps naming those sidecars, ...
        grew = False
        for rel, doc in snap["json"].items():
            p = root / rel
            if p in volatile_files:
                continue
            text = json.dumps(doc)
            if any(d in text for d in volatile_digests):
                volatile_files.add(p)
                volatile_digests.add(digests[p])
                grew = True
        if not grew:
            break
    # pickles and parquets: compared by content above; their digests are volatile when the pickle embeds a pid/timestamp —
    # treat every .pkl digest as volatile (pickle bytes depend on the counter CLASS's module path only; checked by content)
    volatile_digests |= {digests[p] for p in files if p.suffix == ".pkl"}
    snap["jsonl"] = {rel: _scrub(v, volatile_digests) for rel, v in snap["jsonl"].items()}
    snap["json"] = {rel: _scrub(v, volatile_digests) for rel, v in snap["json"].items()}
    if pickle_bytes_are_volatile:  # digests COMPUTED OVER a volatile digest (contract digests of the pickle's sidecar, the stamp's identity)
        derived = {"sidecar_contract_sha256", "stage_identity_sha256", "sidecar_sha256", "content_sha256"}

        def drop(o):
            if isinstance(o, dict):
                return {k: drop(v) for k, v in o.items() if k not in derived}
            return [drop(v) for v in o] if isinstance(o, list) else o

        for rel in list(snap["json"]):
            if root / rel in volatile_files:
                snap["json"][rel] = drop(snap["json"][rel])
    snap["stable_sha256"] = {str(p.relative_to(root)): digests[p] for p in files
                             if p not in volatile_files and p.suffix not in (".pkl",) and digests[p] not in volatile_digests}
    return snap


def diff_snapshots(a: dict, b: dict) -> list[str]:
    out = []
    if a["files"] != b["files"]:
        out.append(f"file lists differ: only unpatched {sorted(set(a['files']) - set(b['files']))}, only patched {sorted(set(b['files']) - set(a['files']))}")
    for kind in ("parquet", "jsonl", "json", "pickle", "stable_sha256"):
        for rel in sorted(set(a[kind]) | set(b[kind])):
            if a[kind].get(rel) != b[kind].get(rel):
                out.append(f"{kind}: {rel} differs")
    return out


def gen_tournament_per_chunk(tmp: Path, name: str, sim_override: dict, pickle_bytes_are_volatile: bool = False) -> dict:
    """The per-CHUNK service of the binding (``_run_chunk`` and ``_run_chunk_metrics`` without row shards: one launch and one tally per
    deterministic batch, no per-shuffle objects): the reference's ``run_single_n`` unpatched vs patched on the tiny configuration with
    ``sim_override`` (rows off; metrics on or off) — every artifact equal, and the recorded engine calls for the replay."""
    gp = oracle_profile()
    payload = {key: dict(val) for key, val in TINY_CONFIG.items()}
    payload["sim"].update(sim_override)
    payload["io"] = {"results_dir_prefix": str(tmp / name), "analysis_subdir": "analysis"}
    cfg_path = tmp / f"{name}.yaml"
    cfg_path.write_text(yaml.safe_dump(payload))
    snaps, launches = {}, 0
    recorder = RecordingEngine(StubEngine())
    for mode in ("unpatched", "patched"):
        cfg = load_app_config(cfg_path, seed_list_len=1)
        cfg._code_identity = fixture_code_identity()
        shutil.rmtree(cfg.results_root, ignore_errors=True)
        for k in (2, 4):
            if mode == "unpatched":
                runner.run_single_n(cfg, k, oracle_game_profile=gp)
            else:
                ns = run_snippet("tournament", {"rt": rt, "runner": runner, "cfg": cfg, "k": k, "game_profile": gp, "engine": recorder})
                launches += ns["binding"].launches
        snaps[mode] = snapshot_tree(cfg.results_root, pickle_bytes_are_volatile)
        shutil.rmtree(cfg.results_root, ignore_errors=True)
    problems = diff_snapshots(snaps["unpatched"], snaps["patched"])
    if problems:
        for line in problems:
            print("  ", line)
        raise SystemExit(f"tournament binding ({name}): the patched run's artifacts differ from the unpatched run's")
    per_batch = all(c["shuffles_per_batch"] == c["shuffle_end"] - c["shuffle_begin"] and not c["want_rows"] for c in recorder.calls)
    assert per_batch, f"{name}: the binding did not serve whole chunks from one tally"
    print(f"tournament ({name}): {len(snaps['patched']['files'])} artifacts equal (unpatched vs binding), {launches} engine launches = "
          f"{len(recorder.calls)} chunks, one tally each")
    return {"sim_override": sim_override, "files": snaps["patched"]["files"], "calls": recorder.calls}


def gen_tournament(tmp: Path) -> dict:
    gp = oracle_profile()
    payload = {key: dict(val) for key, val in TINY_CONFIG.items()}
    payload["io"] = {"results_dir_prefix": str(tmp / "out"), "analysis_subdir": "analysis"}
    cfg_path = tmp / "tiny.yaml"
    cfg_path.write_text(yaml.safe_dump(payload))

    def load():
        cfg = load_app_config(cfg_path, seed_list_len=1)
        cfg._code_identity = fixture_code_identity()   # artifact-contract version 3 stays on
        return cfg

    snaps = {}
    recorder = RecordingEngine(StubEngine())
    for mode in ("unpatched", "patched"):
        cfg = load()
        shutil.rmtree(cfg.results_root, ignore_errors=True)
        launches = 0
        for k in (2, 4):
            if mode == "unpatched":
                runner.run_single_n(cfg, k, oracle_game_profile=gp)
            else:
                ns = run_snippet("tournament", {"rt": rt, "runner": runner, "cfg": cfg, "k": k, "game_profile": gp, "engine": recorder})
                launches += ns["binding"].launches
        snaps[mode] = snapshot_tree(cfg.results_root)
        if mode == "patched":
            snaps["launches"] = launches
        assert rt._play_one_shuffle.__module__ == "farkle.simulation.run_tournament", "binding still installed"
    # The binding lives in the calling process: a run that asks for a worker pool must FAIL LOUDLY, not produce CPU-played
    # artifacts (round-4 review).  The reference's own run_single_n, sim.n_jobs = 2, binding installed:
    from farkle_ii_amd.reference_binding import BindingWorkerPoolError, TournamentBinding

    cfg = load()
    cfg.sim.n_jobs = 2
    shutil.rmtree(cfg.results_root, ignore_errors=True)
    guard_message = None
    try:
        with TournamentBinding(rt, engine=recorder):
            runner.run_single_n(cfg, 2, oracle_game_profile=gp)
    except BindingWorkerPoolError as exc:
        guard_message = str(exc)
    assert guard_message and "n_jobs" in guard_message, "sim.n_jobs = 2 under the binding did not raise BindingWorkerPoolError"
    assert rt.parallel.__class__.__name__ == "module", "the process_map guard is still installed"
    written = [p for p in cfg.results_root.rglob("rows_*.parquet")] if cfg.results_root.exists() else []
    assert not written, f"a refused run left row shards behind: {written[:3]}"
    print(f"tournament: sim.n_jobs = 2 under the binding refused ({guard_message[:70]}...)")
    shutil.rmtree(cfg.results_root, ignore_errors=True)
    problems = diff_snapshots(snaps["unpatched"], snaps["patched"])
    if problems:
        for line in problems:
            print("  ", line)
        raise SystemExit("tournament binding: the patched run's artifacts differ from the unpatched run's")
    stamp = [rel for rel in snaps["patched"]["files"] if rel.endswith("simulation.done.json")]
    sidecars = [rel for rel in snaps["patched"]["files"] if rel.endswith(".sidecar.json")]
    pickles = snaps["patched"]["pickle"]
    assert all(v["win_totals_type"] == "farkle.simulation.run_tournament.OutcomeCounter" for v in pickles.values()), pickles
    print(f"tournament: {len(snaps['patched']['files'])} artifacts equal (unpatched vs binding), {len(sidecars)} v3 sidecars, "
          f"{len(stamp)} authenticated stage stamps, {snaps['launches']} engine launches, {len(recorder.calls)} recorded calls")
    done = {rel: doc for rel, doc in snaps["patched"]["json"].items() if rel.endswith("simulation.done.json")}
    return {"config": TINY_CONFIG, "game_profile": {"target": 100, "max_rounds": 200, "tournament_overrides": [[11, 2, 0, 0, 0]]},
            "artifact_contract_version": 3, "files": snaps["patched"]["files"], "sidecar_count": len(sidecars),
            "stage_stamps": stamp, "stage_done": done, "checkpoints": pickles, "calls": recorder.calls,
            "n_jobs_2_refused": guard_message}


# ---- (b) H2H --------------------------------------------------------------------------------------------------------------

def h2h_config(tmp: Path, gp: GameProfile) -> AppConfig:
    cfg = AppConfig(io=IOConfig(results_dir_prefix=tmp / "results"), sim=SimConfig(seed=11, seed_list=[11, 22], n_players_list=[2, 4]))
    h = cfg.head2head   # the head2head block of the tiny oracle configuration (raw_simulation_oracle.py:122-136)
    h.n_jobs, h.family_alpha, h.target_power, h.practical_delta = 1, 0.5, 0.1, 0.2
    h.sensitivity_deltas, h.candidate_cap, h.total_game_cap = (0.2, 0.04), 3, 24
    cfg.screening.practical_delta_by_k = {2: 0.2, 4: 0.2}
    cfg.screening.delta_across_k = 0.2
    cfg._game_profile_sha256 = gp.sha256
    cfg._code_identity = fixture_code_identity()
    return cfg


def write_frozen_family(cfg: AppConfig, strategies=(0, 1, 3)) -> None:
    """The two inputs ``plan_h2h_schedule`` authenticates: family membership + manifest (what ``freeze_h2h_candidate_family`` publishes)."""
    family_hash = "a" * 64
    roots = list(cfg.sim.seed_list)
    membership = pd.DataFrame({"strategy": list(strategies), "final_family": [True] * len(strategies), "family_hash": [family_hash] * len(strategies)})
    membership["strategy"] = pd.array(membership["strategy"].tolist(), dtype="Int32")
    manifest = {"family_hash": family_hash, "candidates": list(strategies), "candidate_count": len(strategies), "root_seeds": roots,
                "single_root_execution": len(roots) == 1}
    common = dict(producer="gen_binding", scope=ArtifactScope.H2H_2P, source_scope=ArtifactScope.CROSS_SEED, operation="candidate_family_freeze",
                  player_counts=[2], required_player_counts=[2], missing_cell_policy="fail", seed_scope="both_roots_combined")
    p = cfg.h2h_candidate_family_path()
    write_parquet_artifact_atomic(pa.Table.from_pandas(membership, preserve_index=False), p,
                                  sidecar=make_artifact_sidecar(cfg, p, consistency_columns=membership.columns.tolist(), **common))
    mp = cfg.h2h_candidate_family_manifest_path()
    write_json_artifact_atomic(manifest, mp, sidecar=make_artifact_sidecar(cfg, mp, consistency_columns=list(manifest), **common))


def gen_h2h(tmp: Path) -> dict:
    gp = oracle_profile()
    strategies, _ = generate_strategy_grid(**{key: TINY_SIM[key] for key in (
        "score_thresholds", "dice_thresholds", "smart_five_opts", "smart_one_opts", "consider_score_opts", "consider_dice_opts",
        "auto_hot_dice_opts", "run_up_score_opts", "include_stop_at", "include_stop_at_heuristic")})
    manifest = build_strategy_manifest(strategies)
    tables, out = {}, {"runs": {}}
    recorders = {}
    for mode in ("reference_runner", "block_runner", "prefetching_block_runner"):
        base = tmp / f"h2h_{mode}"
        cfg = h2h_config(base, gp)
        write_frozen_family(cfg)
        plan = plan_h2h_schedule(cfg)
        mpath = cfg.strategy_manifest_root_path()
        mpath.parent.mkdir(parents=True, exist_ok=True)
        manifest.to_parquet(mpath)
        schedule = pq.read_table(plan.block_manifest).to_pylist()
        rec = RecordingEngine(StubEngine())
        recorders[mode] = rec
        if mode == "reference_runner":
            result = execute_h2h_schedule(cfg, n_jobs=1, oracle_game_profile=gp)
        else:
            ns = run_snippet("h2h_" + mode, {"execute_h2h_schedule": execute_h2h_schedule, "cfg": cfg, "game_profile": gp, "engine": rec,
                                             "schedule": schedule})
            result = ns["result"]
        counts = pq.read_table(result.order_counts).to_pylist()
        blocks = [pq.read_table(p).to_pylist()[0] for p in result.block_paths]
        tables[mode] = {"counts": counts, "blocks": blocks}
        got = {(int(r["pair_id"]), int(r["root_seed"]), int(r["order"])): (
            int(r["games_attempted"]), int(r["games_completed"]), int(r["games_safety_limit"]), int(r["wins_a"]), int(r["wins_b"]),
            int(r["replacement_attempt_count"]), str(r["completion_status"])) for r in counts}
        assert got == EXPECTED_H2H_BLOCKS, (mode, got)
        out["runs"][mode] = {"engine_calls": len(rec.calls), "blocks_written_by_the_reference": len(blocks)}
    for mode in ("block_runner", "prefetching_block_runner"):
        assert tables[mode] == tables["reference_runner"], f"{mode}: block parquets / root_order_counts differ from the reference runner's"
    print("h2h: 12 block parquets + root_order_counts equal (reference runner vs block_runner vs prefetching_block_runner), "
          f"EXPECTED_H2H_BLOCKS reproduced; engine calls {out['runs']}")
    out["schedule"] = [{key: (val if not isinstance(val, float) or np.isfinite(val) else None) for key, val in rec_.items()} for rec_ in schedule]
    out["expected_blocks"] = [[list(key), list(val)] for key, val in sorted(EXPECTED_H2H_BLOCKS.items())]
    out["strategy_manifest"] = json.loads(manifest.to_json(orient="records"))
    out["game_profile"] = {"target": 100, "max_rounds": 200, "h2h_overrides": [[11, 0, 0, 0, 0], [11, 1, 0, 0, 0], [11, 1, 0, 1, 0]]}
    out["calls"] = {mode: recorders[mode].calls for mode in ("block_runner", "prefetching_block_runner")}
    out["order_counts"] = tables["reference_runner"]["counts"]
    return out


def main() -> None:
    tmp = Path(tempfile.mkdtemp(prefix="fk_binding_"))
    try:
        snippets = integration_snippets()
        doc = {"generated_by": "oracle/gen_binding.py (reference imported in the build container; engine = CPU oracle stub)",
               "integration_snippets_sha256": {name: hashlib.sha256(code.encode("utf-8")).hexdigest() for name, code in sorted(snippets.items())},
               "tournament": gen_tournament(tmp),
               "tournament_metric_chunks_no_rows": gen_tournament_per_chunk(tmp, "chunks", {"row_dir": None}),
               "tournament_counts_only": gen_tournament_per_chunk(tmp, "counts", {"row_dir": None, "metric_chunk_dir": None, "expanded_metrics": False},
                                                                  pickle_bytes_are_volatile=True),
               "h2h": gen_h2h(tmp)}
        OUT.write_text(json.dumps(doc, sort_keys=True, default=lambda o: int(o) if isinstance(o, np.integer) else str(o)))
        print(OUT.name, OUT.stat().st_size)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
