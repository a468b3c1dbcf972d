"""AppConfig front-end of the tournament: grid -> workload plan -> GPU batches -> artifacts.

Mirrors ``src/farkle/simulation/runner.py`` (``run_single_n`` :1326-1752, ``run_multi`` :1757-1816) and the parent
loop of ``src/farkle/simulation/run_tournament.py:1050-1839`` for the v2 (no-sidecar) artifact set:

    <results_root>/strategy_manifest.parquet
    <results_root>/<k>_players/simulation_workload_plan.json
    <results_root>/<k>_players/<k>p_checkpoint.pkl        {win_totals, outcome_counts, metric_sums, metric_square_sums, meta}
    <results_root>/<k>_players/<k>p_checkpoint.parquet    per-strategy summary (runner.py:1612-1648)
    <results_root>/<k>_players/<k>p_metrics.parquet       expanded metrics (runner.py:1651-1712), with sim.expanded_metrics
    <row_dir>/rows_<root>_<k>p_<shuffle:012d>.parquet + manifest.jsonl        (with sim.row_dir)
    <metric_chunk_dir>/metrics_<batch:06d>.parquet + metrics_manifest.jsonl   (with sim.metric_chunk_dir)
    <all_player_batch_dir>/all_player_batch_<batch:06d>.parquet + all_player_manifest.jsonl   (with sim.all_player_batch_dir:
                                                            the all-player batch metrics of analysis/all_player_metrics.py
                                                            from device accumulators, no rows; see all_player.py)
    <results_root>/<k>_players/simulation.done.json       plain completion marker (v3 sidecars are out of scope)

Resume ownership is the deterministic batch, as in the reference (``completed_process_block_indices``).  With
``torch.distributed`` initialised (one process per GPU) whole batches are partitioned over the ranks and the per-batch
tallies are SUM-reduced to rank 0, which writes the artifacts.
"""
from __future__ import annotations

import json
import logging
import os
import sys
import time
from pathlib import Path
from typing import Any, Callable, Mapping, Sequence

import numpy as np

from . import checkpoint as ckpt
from . import random as urandom
from . import tournament as rt
from .config import AppConfig
from .distributed import barrier, gather_objects, reduce_tally, shard_shuffle_range
from .engine import get_engine
from .game_profile import GameProfile
from .all_player import all_player_batch_table
from .backend import COL_ATTEMPTED, COL_COMPLETED, COL_SAFETY, COL_SQ_SUMS, COL_SUMS, COL_WINS, SEAT_RATIO_COLS, SEAT_STAT_COLS
from .rows import OUTCOME_SCHEMA_VERSION, TOURNAMENT_METHOD_VERSION, raw_simulation_schema_for
from .strategies import (STRATEGY_TUPLE_FIELDS, FavorDiceOrScore, ThresholdStrategy, generate_strategy_grid,
                         prepare_public_helper_strategies, strategy_tuple)
from .workload_planner import TournamentWorkloadPlan, WorkloadCapExceeded, plan_tournament_workload, write_workload_plan

LOGGER = logging.getLogger(__name__)
MAX_GAMES_PER_LAUNCH = 200_000_000  # checkpoint cadence on the GPU: a launch group is at most this many games
ROWS_ASYNC = os.environ.get("FK_ROWS_ASYNC", "1") != "0"  # (A/B switch: the images' last copy awaited by the shard job / by the engine call)
ROWS_PIPELINE = os.environ.get("FK_ROWS_PIPELINE", "0") != "0"  # (A/B switch: a launch group's engine part on the launcher thread / in line)
ROWS_SLOTS = max(2, int(os.environ.get("FK_ROWS_SLOTS", "3")))  # rows mode: page-locked image buffers per engine = launch groups played ahead of the shard writer + 1
ROWS_GROUP_BYTES = int(os.environ.get("FK_ROWS_GROUP_MB", "256")) << 20  # rows mode: a launch group's column images (one of two page-locked buffers; group i is written while i + 1 plays)
ROW_WRITER_THREADS = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))  # row-shard writer PROCESSES (rows mode)


_TRACE: list | None = [] if os.environ.get("FK_RUN_TRACE") else None  # diagnostics: (time, thread, label) of a run's phases, dumped to stderr


def _trace(label: str) -> None:
    if _TRACE is not None:
        import threading

        _TRACE.append((time.perf_counter(), threading.current_thread().name, label))


def _rank_world() -> tuple[int, int]:
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def _atomic_write_bytes(path: Path, content: bytes) -> None:
    path.parent.mkdir(parents=True, exist_ok=True)
    tmp = path.with_name(path.name + ".tmp")
    tmp.write_bytes(content)
    os.replace(tmp, path)


def _write_parquet_atomic(table, path: Path, **writer_options) -> None:
    import pyarrow.parquet as pq

    path.parent.mkdir(parents=True, exist_ok=True)
    tmp = path.with_name(path.name + ".tmp")
    pq.write_table(table, tmp, **writer_options)
    os.replace(tmp, path)


# Metric chunk files: 11 x S rows of eight columns per deterministic batch.  Dictionary pages for the one string column only and no
# column statistics (nothing reads them: the reference's reducer scans whole chunks, run_tournament.py:905-922) halve the encoding —
# 15.1 -> 7.5 ms per 56 760-row chunk, same file size; the table content is what the artifact contract pins, not the page layout.
_CHUNK_WRITER_OPTIONS = {"use_dictionary": ["metric"], "write_statistics": False}


_HELPER = None


def _helper_thread():
    """One thread that encodes the metric manifest's per-shuffle lists (625 000 integers per 10^7 two-player games) while
    the engine call of the same launch group is in flight (ctypes drops the GIL for its duration).  The per-batch parquet
    files are NOT written on threads: Arrow's writer holds the GIL for most of a 704-row file (measured: four threads
    1.6 ms per file against 1.1 ms serial)."""
    global _HELPER
    if _HELPER is None:
        from concurrent.futures import ThreadPoolExecutor

        _HELPER = ThreadPoolExecutor(max_workers=1, thread_name_prefix="fk-manifest")
    return _HELPER


_SHARD_THREAD = None


def _write_group_shards(eng, rows_event, write):
    """One launch group's shard job (on the shard thread): wait for the group's column images to be on the host — ``rows_event`` names the
    engine's completion event of an ``async_rows`` call — then frame and publish the files (``write``: the prepared library call)."""
    _trace("shard job begins")
    if rows_event is not None:
        eng.rows_wait(rows_event)
    _trace("shard job: images here")
    try:
        return write()
    finally:
        _trace("shard job ends")


def _shard_thread():
    """The thread that hands a launch group's column images to the library's shard writer (its own host threads do the work; the call
    drops the GIL) while the main thread is inside the next group's engine call."""
    global _SHARD_THREAD
    if _SHARD_THREAD is None:
        from concurrent.futures import ThreadPoolExecutor

        _SHARD_THREAD = ThreadPoolExecutor(max_workers=1, thread_name_prefix="fk-shards")
    return _SHARD_THREAD


_LAUNCHER = None


def _launcher_thread():
    """The thread the engine part of a launch group runs on in rows mode (run_tournament: `pipelined`)."""
    global _LAUNCHER
    if _LAUNCHER is None:
        from concurrent.futures import ThreadPoolExecutor

        _LAUNCHER = ThreadPoolExecutor(max_workers=1, thread_name_prefix="fk-launch")
    return _LAUNCHER


_PIN_THREADS = None


def _pin_threads():
    """The thread that makes an engine's page-locked column-image buffers while the run's first launches are prepared and played, instead
    of in front of them."""
    global _PIN_THREADS
    if _PIN_THREADS is None:
        from concurrent.futures import ThreadPoolExecutor

        # (one thread: three buffers populated at once took 89 ms each — the kernel serialises a process's page faults — against ~20 ms
        # one after the other, and the first launch only needs the first)
        _PIN_THREADS = ThreadPoolExecutor(max_workers=1, thread_name_prefix="fk-pin")
    return _PIN_THREADS


_PUBLISHER = None


def _publisher_thread():
    """The thread run_multi publishes one player count's results on while the next one plays (not the helper thread: a publishing tail
    joins work it queued there)."""
    global _PUBLISHER
    if _PUBLISHER is None:
        from concurrent.futures import ThreadPoolExecutor

        _PUBLISHER = ThreadPoolExecutor(max_workers=1, thread_name_prefix="fk-publish")
    return _PUBLISHER


def _shuffle_list_fragments(seeds: np.ndarray, first: int, spb: int, bounds: Sequence[tuple[int, int]]) -> list[tuple[str, str]]:
    """JSON text of ``shuffle_indices`` / ``shuffle_seeds`` for every (first_shuffle, last_shuffle) batch: ``str`` of a list of
    ints is exactly what ``json.dumps`` writes for it, at 60 % of the time."""
    del spb
    return [(str(list(range(lo, hi))), str(seeds[lo - first:hi - first].tolist())) for lo, hi in bounds]


def _resolve_strategies(cfg: AppConfig, strategies: list[ThresholdStrategy] | None) -> tuple[list[ThresholdStrategy], int]:
    if strategies is None:
        strategies, _ = generate_strategy_grid(
            score_thresholds=cfg.sim.score_thresholds, dice_thresholds=cfg.sim.dice_thresholds,
            smart_five_opts=cfg.sim.smart_five_opts, smart_one_opts=cfg.sim.smart_one_opts,
            consider_score_opts=cfg.sim.consider_score_opts, consider_dice_opts=cfg.sim.consider_dice_opts,
            auto_hot_dice_opts=cfg.sim.auto_hot_dice_opts, run_up_score_opts=cfg.sim.run_up_score_opts,
            include_stop_at=cfg.sim.include_stop_at, include_stop_at_heuristic=cfg.sim.include_stop_at_heuristic)
    strategies = prepare_public_helper_strategies(strategies)
    LOGGER.info("Strategy grid prepared: %d strategies", len(strategies))
    return strategies, len(strategies)


def build_strategy_manifest(strategies: Sequence[ThresholdStrategy]):
    """Manifest frame mapping strategy ids to attributes (strategies.py:725-748)."""
    import pandas as pd

    seen: set[int] = set()
    tuples, id_col, str_col = [], [], []
    for s in strategies:  # first occurrence of every id, columns assembled once (a dict per strategy + DataFrame(list of dicts): 25 ms per 5 160)
        if s.strategy_id is None or int(s.strategy_id) in seen:
            continue
        seen.add(int(s.strategy_id))
        tuples.append(strategy_tuple(s))
        id_col.append(int(s.strategy_id))
        str_col.append(str(s))
    columns = {name: list(col) for name, col in zip(STRATEGY_TUPLE_FIELDS, zip(*tuples))} if tuples else {}
    if "favor_dice_or_score" in columns:
        columns["favor_dice_or_score"] = [v.value if isinstance(v, FavorDiceOrScore) else v for v in columns["favor_dice_or_score"]]
    if tuples:
        columns["strategy_id"] = id_col
        columns["strategy_str"] = str_col
    frame = pd.DataFrame(columns)
    if not frame.empty:
        frame["strategy_id"] = frame["strategy_id"].astype("Int32")
        frame = frame.sort_values("strategy_id", kind="mergesort").reset_index(drop=True)
    return frame


def _publish_shard_v3(table, path: Path, template: Mapping[str, Any], **writer_options) -> dict[str, Any]:
    """Write one Parquet shard and its contract-v3 sidecar; returns the identity fields its manifest record carries."""
    import hashlib

    import pyarrow as pa
    import pyarrow.parquet as pq

    from .contract_v3 import fill_shard_template, sidecar_path

    sink = pa.BufferOutputStream()
    pq.write_table(table, sink, **writer_options)
    blob = sink.getvalue().to_pybytes()
    digest = hashlib.sha256(blob).hexdigest()
    text, side_sha = fill_shard_template(template, path.name, len(blob), digest)
    sidecar_path(path).unlink(missing_ok=True)  # never new bytes under an old sidecar
    _atomic_write_bytes(path, blob)
    _atomic_write_bytes(sidecar_path(path), text)
    return {"byte_length": len(blob), "data_sha256": digest, "sidecar_sha256": side_sha,
            "schema_fingerprint_sha256": template["schema_fingerprint_sha256"]}


class _SweepShared:
    """What every player count of one sweep derives from the SAME strategy list (run_multi): the prepared strategies, their manifest
    frame / Arrow table / sha256, the packed table for the C-ABI and which manifest files have been checked against it — 60 % of the host
    time of the reference's eight-count production sweep when each count rebuilt them (tools/profile_run_host_null.py)."""

    def __init__(self, strategies: list[ThresholdStrategy], background: bool = False):
        self.strategies = strategies
        self.packed = rt.pack_strategies(strategies)  # (the engine call needs it; the rest is for the artifacts)
        self.verified: set[str] = set()  # manifest files known to equal `table`
        self._derived: tuple | None = None
        # background: the manifest frame / sha256 / Arrow table are made on the helper thread (2 ms for 64 strategies, 70 ms for 5 160)
        # while the caller goes on to the first engine call; whoever reads them first waits for that job
        self._job = _helper_thread().submit(self._derive) if background else None

    def _derive(self) -> tuple:
        import hashlib

        import pyarrow as pa

        manifest = build_strategy_manifest(self.strategies)
        sha = hashlib.sha256(manifest.to_csv(index=False).encode("utf-8")).hexdigest()  # runner.py:790-800
        return manifest, sha, pa.Table.from_pandas(manifest, preserve_index=False)

    def _get(self) -> tuple:
        if self._derived is None:
            import threading

            # Called FROM the helper thread (a job queued there that reads the manifest) the pending job cannot be waited for: it sits
            # behind the caller in the same single-worker queue — derive inline instead of deadlocking (round-5 advisor finding).
            on_helper = threading.current_thread().name.startswith("fk-manifest")
            self._derived = self._job.result() if self._job is not None and not (on_helper and not self._job.done()) else self._derive()
        return self._derived

    @property
    def manifest(self):
        return self._get()[0]

    @property
    def manifest_sha(self) -> str:
        return self._get()[1]

    @property
    def table(self):
        return self._get()[2]


def _plan_workload_from_config(cfg: AppConfig, n_strategies: int, n_players: int) -> TournamentWorkloadPlan:
    return plan_tournament_workload(
        root_seed=cfg.sim.seed, k=n_players, strategy_count=n_strategies, resolution_delta=cfg.screening.resolution_delta,
        confidence=cfg.screening.interval_confidence, batch_count=cfg.batching.target_batches,
        min_shuffles_per_batch=cfg.batching.min_shuffles_per_batch, shuffle_cap=cfg.screening.max_shuffles_per_root_k,
        projected_games_per_second=cfg.screening.projected_games_per_second)


def _filter_player_counts(player_counts: Sequence[int], grid_size: int) -> tuple[list[int], list[int]]:
    valid = [n for n in player_counts if n > 0 and grid_size % n == 0]
    invalid = [n for n in player_counts if n not in valid]
    if invalid:
        LOGGER.warning("Dropping incompatible player counts: %s", invalid)
    return valid, invalid


def simulation_done_path(cfg: AppConfig, n_players: int) -> Path:
    return cfg.n_dir(n_players) / "simulation.done.json"


def simulation_is_complete(cfg: AppConfig, n_players: int, plan: TournamentWorkloadPlan | None = None) -> bool:
    path = simulation_done_path(cfg, n_players)
    if not path.exists() or not cfg.checkpoint_path(n_players).exists():
        return False
    try:
        meta = json.loads(path.read_text(encoding="utf-8"))  # the contract sits at the top level (stage_completion.py:505-512)
    except (OSError, json.JSONDecodeError):
        return False
    if isinstance(meta, dict) and "stage_identity_sha256" in meta and "outputs" in meta and "status" not in meta:
        # an authenticated (contract v3) completion: valid iff the stage identity recomputed from the outputs' sidecars under the CURRENT
        # configuration and code identity is the stamp's (runner.py:274-317 of the reference); the workload plan is one of those outputs
        from .contract_v3 import ContractError, SimulationContract

        if cfg._code_identity is None:
            raise ContractError(f"{path} is an authenticated contract-v3 completion: pass the code identity it was written under "
                                "(--code-identity / --reference-checkout) to validate it, or --force to replace the run")
        if plan is not None:
            from .workload_planner import workload_plan_bytes

            plan_path = cfg.n_dir(n_players) / "simulation_workload_plan.json"
            if not plan_path.exists() or plan_path.read_bytes() != workload_plan_bytes(plan):
                return False
        return SimulationContract(cfg, cfg._code_identity, game_profile_sha256=cfg._game_profile_sha256,
                                  run_lineage_sha256=cfg._run_lineage_sha256).is_complete(path)
    if not isinstance(meta, dict) or meta.get("status") != "success" or meta.get("completion_state") != "complete_valid":
        return False
    if plan is not None and (meta.get("num_shuffles") != plan.required_shuffles or meta.get("n_strategies") != plan.strategy_count
                             or meta.get("shuffles_per_batch") != plan.shuffles_per_batch):
        return False
    return meta.get("root_seed") == cfg.sim.seed and meta.get("k") == n_players


_CHUNK_KEYS: dict = {}  # (ids, seated mask) -> the seated rows in chunk order + the two key columns (the same for every batch of a run)


def _metric_chunk_table(batch_tally: np.ndarray, ids: Sequence[int], k: int):
    """Rows of one ``metrics_<idx>.parquet`` (run_tournament.py:1603-1642): for every metric label, one row per strategy that
    was seated in the batch, strategies in the order of their decimal strings (the reference sorts with ``key=str``).
    Columns are built as arrays (a 5 160-strategy grid has 56 760 rows per chunk); the ``metric`` / ``strategy`` key columns
    depend only on which strategies were seated — every one, in a full shuffle — and are built once per run."""
    import pyarrow as pa

    del k
    t = np.asarray(batch_tally, dtype=np.int64)
    mask = (t[:, 1] > 0) | (t[:, 0] > 0)
    key = (id(ids), len(ids), mask.tobytes())
    cached = _CHUNK_KEYS.get(key)
    if cached is None or cached[0] is not ids:
        sid = np.asarray([int(x) for x in ids], dtype=np.int64)
        seated = np.flatnonzero(mask)
        seated = seated[np.argsort(sid[seated].astype(str), kind="stable")]
        labels = pa.array(list(rt.METRIC_LABELS), type=pa.string())
        metric = labels.take(pa.array(np.repeat(np.arange(len(labels), dtype=np.int32), len(seated))))
        strategy = pa.array(np.tile(sid[seated], len(labels)), type=pa.int64())
        _CHUNK_KEYS.clear()
        cached = _CHUNK_KEYS[key] = (ids, seated, metric, strategy)
    _, seated, metric, strategy = cached
    m = len(rt.METRIC_LABELS)
    rows = t[seated]
    outcome = np.tile(rows[:, :4].T, (1, m))  # wins / attempted / completed / safety-limit exposures repeat on every metric's rows
    return pa.table({
        "metric": metric,
        "strategy": strategy,
        "sum": pa.array(rows[:, 4:4 + m].T.reshape(-1).astype(np.float64)),
        "square_sum": pa.array(rows[:, 15:15 + m].T.reshape(-1).astype(np.float64)),
        "wins": pa.array(outcome[0], type=pa.int64()),
        "attempted_exposures": pa.array(outcome[1], type=pa.int64()),
        "completed_exposures": pa.array(outcome[2], type=pa.int64()),
        "safety_limit_exposures": pa.array(outcome[3], type=pa.int64()),
    })


def _shuffle_seeds(eng, root_seed: int, k: int, first: int, last: int) -> np.ndarray:
    """ns-100 uint32 fingerprints of shuffles [first, last) (the manifests' ``shuffle_seeds``): on the device when the engine
    offers ``fk_coordinate_seeds`` (312 500 of them are 0.4 s of NumPy hashing, microseconds of GPU)."""
    idx = np.arange(first, last, dtype=np.uint64)
    if hasattr(eng, "coordinate_seeds") and len(idx):
        from .backend import make_coords

        return eng.coordinate_seeds(make_coords(int(urandom.RandomPurpose.TOURNAMENT_SHUFFLE), root_seed, k, idx), want32=True)[0]
    return urandom.coordinate_seeds(urandom.RandomPurpose.TOURNAMENT_SHUFFLE, root_seed=root_seed, k=k, shuffle_index=idx, dtype=np.uint32)


class _Sidecars:
    """Sidecar writers of one (root, k) run — or a no-op when they are off.  ``artifact_contract.artifact_contract_version`` (default 3,
    as in the reference) picks the contract: 3 = the authenticated sidecars / sealed manifests / completion of contract_v3.py, signed
    with the code identity the caller supplied; 2 = the structural sidecars of sidecars.py (reference: runner.py:1470-1505)."""

    _V3_SOURCES = {"strategy_manifest": 0, "workload_plan": 1}  # how many of (strategy manifest, workload plan) an artifact kind names

    def __init__(self, cfg: AppConfig, n_players: int, sources: Sequence[Path], enabled: bool):
        self.cfg, self.k, self.sources, self.enabled = cfg, n_players, list(sources), enabled
        self.v3 = None
        self._templates: dict = {}
        if enabled and cfg.artifact_contract_version == 3:
            from .contract_v3 import ContractError, SimulationContract

            if cfg._code_identity is None:
                raise ContractError(
                    "artifact-contract version 3 signs every sidecar with the code identity of the checkout that will ingest the results "
                    "(the reference's analyze ingest recomputes it): pass --code-identity COMMIT[:DIRTY_SHA256] or --reference-checkout PATH, "
                    "or ask for the structural contract with --set artifact_contract.artifact_contract_version=2")
            self.v3 = SimulationContract(cfg, cfg._code_identity, game_profile_sha256=cfg._game_profile_sha256,
                                         run_lineage_sha256=cfg._run_lineage_sha256)
        elif enabled and cfg.artifact_contract_version != 2:
            raise ValueError(f"artifact_contract.artifact_contract_version must be 2 or 3, got {cfg.artifact_contract_version}")

    def _v3_sources(self, kind: str) -> list[Path]:
        return self.sources[:self._V3_SOURCES.get(kind, 2)]

    def template(self, kind: str, path: Path, *, sources: Sequence[Path] | None = None, support_counts: Sequence[int] | None = None,
                 schema=None):
        """What a shard writer needs to publish the sidecar of a shard in ``path``'s directory (picklable)."""
        if not self.enabled:
            return None
        if self.v3 is not None:
            key = (kind, str(Path(path).parent))
            if key not in self._templates:
                self._templates[key] = self.v3.shard_template(kind, Path(path).parent, schema() if callable(schema) else schema,
                                                              n_players=self.k, sources=self._v3_sources(kind))
            return self._templates[key]
        from .sidecars import OPERATIONS, simulation_output_sidecar

        return simulation_output_sidecar(self.cfg, path, n_players=self.k, operation=OPERATIONS[kind],
                                         sources=self.sources if sources is None else sources, support_counts=support_counts)

    def write(self, kind: str, path: Path, *, sources: Sequence[Path] | None = None, support_counts: Sequence[int] | None = None) -> None:
        if not self.enabled or not Path(path).exists():
            return
        if self.v3 is not None:  # (the sources of a v3 simulation output are fixed by its kind, runner.py:1438-1505, 1649, 1708)
            self.v3.write_sidecar(path, kind, n_players=self.k, sources=self._v3_sources(kind), support_counts=support_counts)
            return
        from .sidecars import write_sidecar

        write_sidecar(path, self.template(kind, path, sources=sources, support_counts=support_counts))


def _stored_schema(schema, **writer_options):
    """The Arrow schema a reader gets back from a Parquet file of ``schema`` written with ``writer_options`` — what the reference
    fingerprints (``parquet_schema_identity``, authenticated_contract.py:293-300)."""
    import pyarrow as pa
    import pyarrow.parquet as pq

    sink = pa.BufferOutputStream()
    pq.write_table(schema.empty_table(), sink, **writer_options)
    return pq.read_schema(pa.BufferReader(sink.getvalue()))


def _read_manifest(path: Path) -> list[dict]:
    if not path.exists():
        return []
    lines = [line for line in (raw.strip() for raw in path.read_text(encoding="utf-8").splitlines()) if line]
    try:  # one call of the C decoder for the file (a 4 300-shard manifest: 8 ms instead of 22)
        out = json.loads("[" + ",".join(lines) + "]")
        if len(out) == len(lines) and all(isinstance(record, dict) for record in out):
            return out
    except json.JSONDecodeError:
        pass
    out = []
    for line in lines:
        try:
            out.append(json.loads(line))
        except json.JSONDecodeError:
            continue  # a torn last line of an interrupted append
    return out


def _rewrite_manifest(path: Path, records: Sequence[Mapping[str, Any]]) -> None:
    _atomic_write_bytes(path, "".join(json.dumps(r, sort_keys=True) + "\n" for r in records).encode("utf-8"))


def _prune_manifest(path: Path, owned: set[int], batch_of) -> int:
    """Keep one record per unit, and only units whose deterministic batch the checkpoint owns: a run that died between
    appending manifest lines and writing the checkpoint that owns them replays that launch group, and its records must
    not be there twice (the reference replays every manifest record on recovery, run_tournament.py:820-941)."""
    records = _read_manifest(path)
    kept, seen = [], set()
    for r in records:
        unit, batch = batch_of(r)
        if batch in owned and unit not in seen:
            seen.add(unit)
            kept.append(r)
    if path.exists() and len(kept) != len(records):
        _rewrite_manifest(path, kept)
    return len(records) - len(kept)


def _recover_from_metric_chunks(metric_chunk_dir: Path, ids: Sequence[int], owned: set[int]) -> tuple[np.ndarray, set[int]] | None:
    """Aggregates and completed batches rebuilt from the metric chunk files a manifest lists — the reference's recovery
    authority when the periodic checkpoint is missing or behind (``_load_metric_chunk_aggregates``,
    run_tournament.py:871-941).  The manifest is read first; chunk files are only opened when it lists a batch the
    checkpoint does not own (None otherwise).  A listed file that is missing is an error, as in the reference (:887)."""
    import pyarrow.parquet as pq

    records, seen = [], set()
    for rec in _read_manifest(metric_chunk_dir / "metrics_manifest.jsonl"):
        batch = int(rec["chunk_index"]) - 1
        if batch not in seen:
            seen.add(batch)
            records.append((batch, metric_chunk_dir / str(rec["path"])))
    if not (seen - owned):
        return None
    sid = np.asarray([int(x) for x in ids], dtype=np.int64)
    order = np.argsort(sid, kind="stable")
    labels = list(rt.METRIC_LABELS)
    tally = np.zeros((len(ids), 26), dtype=np.int64)
    for batch, path in records:
        if not path.exists():
            raise FileNotFoundError(f"metric chunk manifest lists a missing file: {path}")
        t = pq.read_table(path)
        strat = np.asarray(t.column("strategy").to_numpy(zero_copy_only=False), dtype=np.int64)
        pos = np.searchsorted(sid[order], strat)
        if (pos >= len(sid)).any() or (sid[order][np.minimum(pos, len(sid) - 1)] != strat).any():
            raise ValueError(f"{path} names a strategy that is not in the configured grid")
        row = order[pos]
        metric = np.asarray([labels.index(m) for m in t.column("metric").to_pylist()], dtype=np.int64)
        col = lambda name: np.asarray(t.column(name).to_numpy(zero_copy_only=False)).astype(np.int64)  # noqa: E731
        np.add.at(tally, (row, 4 + metric), col("sum"))
        np.add.at(tally, (row, 15 + metric), col("square_sum"))
        first = metric == 0  # the outcome columns repeat on every metric's row: count them once (run_tournament.py:905-922)
        for c, name in enumerate(("wins", "attempted_exposures", "completed_exposures", "safety_limit_exposures")):
            np.add.at(tally[:, c], row[first], col(name)[first])
    return tally, seen


def _check_ownership(total: np.ndarray, done_batches: set[int], spb: int, required_shuffles: int, what: str) -> None:
    """Every strategy is seated exactly once per shuffle, so the attempted exposures of each strategy must equal the number
    of shuffles the owned batches hold: totals that include a batch the run will replay (or miss one it will skip) would
    otherwise be counted twice (or never) without any other check noticing."""
    owned = sum(min((b + 1) * spb, required_shuffles) - b * spb for b in done_batches)
    att = total[:, 1]
    if not ((att == owned).all() and np.array_equal(att, total[:, 2] + total[:, 3])):
        raise ValueError(
            f"{what}: the recovered totals cover {int(att.min())}..{int(att.max())} shuffles per strategy but the batches it owns hold "
            f"{owned}; its shuffle list and block list disagree (a checkpoint written after a partial artifact recovery). "
            "Resume needs metric chunk files to rebuild from, or use --force")


def run_tournament(*, cfg: AppConfig, n_players: int, strategies: list[ThresholdStrategy], plan: TournamentWorkloadPlan,
                   checkpoint_path: Path, collect_metrics: bool, row_dir: Path | None, metric_chunk_dir: Path | None,
                   resume: bool, checkpoint_metadata: "Mapping[str, Any] | Callable[[], Mapping[str, Any]]", oracle_game_profile: GameProfile | None = None,
                   all_player_dir: Path | None = None, sidecars: "_Sidecars | None" = None,
                   rng_lags: Sequence[int] | None = None, defer_final_checkpoint: bool = False, packed_table: np.ndarray | None = None,
                   defer_tail: list | None = None) -> dict:
    """Play every deterministic batch not yet owned by the checkpoint and persist the aggregates.  ``defer_final_checkpoint``: the final
    checkpoint's file write may still be in flight on return — the caller joins ``result["checkpoint_written"]`` before reading the file.  ``rng_lags``: also accumulate the lag
    sufficient statistics of the RNG diagnostics' strategy family over the WHOLE shuffle range (``fk_tournament_run_lags``; launch
    groups and ranks are contiguous ranges that merge in order, rng_lags.LagSummary) — returned as ``result["lag_summary"]``."""
    rank, world = _rank_world()
    _trace(f"{n_players}p run_tournament")
    eng = get_engine()
    _trace("engine here")
    k = n_players
    sidecars = sidecars or _Sidecars(cfg, n_players, (), False)
    S = len(strategies)
    ids = [int(s.strategy_id) for s in strategies]
    spb = plan.shuffles_per_batch
    n_batches = plan.batch_count
    meta_cache: list[dict] = []

    def get_meta() -> dict:
        """The checkpoint's contract metadata (built on first use: ``checkpoint_metadata`` may be a callable whose value — the strategy
        manifest's sha256 — is still being computed on the helper thread while the first launch plays)."""
        if not meta_cache:
            extra = checkpoint_metadata() if callable(checkpoint_metadata) else checkpoint_metadata
            meta_cache.append({
                "n_players": k, "num_shuffles": plan.required_shuffles, "global_seed": cfg.sim.seed, "n_strategies": S,
                "rng_scheme_version": urandom.RNG_SCHEME_VERSION, "outcome_schema_version": OUTCOME_SCHEMA_VERSION,
                "tournament_method_version": TOURNAMENT_METHOD_VERSION, "rng_bit_generator": "PCG64DXSM",
                "coordinate_contract_version": 1, "shuffle_purpose_namespace": int(urandom.RandomPurpose.TOURNAMENT_SHUFFLE),
                "shuffle_permutation_purpose_namespace": int(urandom.RandomPurpose.SHUFFLE_PERMUTATION),
                "game_purpose_namespace": int(urandom.RandomPurpose.TOURNAMENT_GAME),
                "player_purpose_namespace": int(urandom.RandomPurpose.TOURNAMENT_PLAYER), "deterministic_batch_size": spb,
                **({"game_profile_sha256": oracle_game_profile.sha256} if oracle_game_profile is not None else {}),  # run_tournament.py:1156
                **dict(extra),
                "workload_plan_version": plan.plan_version, "screening_resolution_delta": plan.resolution_delta,
                "screening_interval_confidence": plan.confidence, "batch_count": plan.batch_count, "shuffles_per_batch": spb,
                "batch_construction": plan.batch_construction,
            })
        return meta_cache[0]

    total = np.zeros((S, 26), dtype=np.int64)
    done_batches: set[int] = set()
    row_manifest = (row_dir / "manifest.jsonl") if row_dir is not None else None
    metrics_manifest = (metric_chunk_dir / "metrics_manifest.jsonl") if metric_chunk_dir is not None else None
    all_player_manifest = (all_player_dir / "all_player_manifest.jsonl") if all_player_dir is not None else None
    resume_error: str | None = None
    if resume and rank == 0:
        try:
            if checkpoint_path.exists():
                payload = ckpt.load_checkpoint(checkpoint_path)  # written by this engine or by the reference
                old = payload.get("meta", {})
                stale = [key for key in ("n_players", "num_shuffles", "global_seed", "n_strategies", "deterministic_batch_size",
                                         "strategy_manifest_sha", "rng_scheme_version")
                         if key in old and old.get(key) != get_meta().get(key)]
                if stale:
                    raise ValueError(f"checkpoint {checkpoint_path} was written under a different contract: {stale}; use --force")
                done_batches = set(int(b) - 1 for b in old.get("completed_process_block_indices", []))  # recorded 1-based
                if done_batches:
                    total = ckpt.payload_to_tally(payload, ids, rt.METRIC_LABELS)
                    if collect_metrics and not payload.get("metric_sums"):
                        raise ValueError(f"checkpoint {checkpoint_path} holds no metric sums but this run collects metrics; use --force")
            if metric_chunk_dir is not None:
                try:
                    _check_ownership(total, done_batches, spb, plan.required_shuffles, "checkpoint")
                    consistent = True
                except ValueError:
                    consistent = False  # totals and block list disagree: rebuild from the chunk files if they can
                recovered = _recover_from_metric_chunks(metric_chunk_dir, ids, done_batches if consistent else set())
                if recovered is not None:  # the chunk files are ahead of the pickle (or it is inconsistent): they are the authority
                    LOGGER.info("Recovered %d batches from metric chunks (checkpoint owned %d)", len(recovered[1]), len(done_batches))
                    total, done_batches = recovered
            # the totals must be exactly the owned batches (the reference trusts the block list, run_tournament.py:1289-1330;
            # its checkpoints after an artifact recovery can list shuffles of blocks they do not list, and blocks without
            # their shuffles)
            _check_ownership(total, done_batches, spb, plan.required_shuffles, f"checkpoint {checkpoint_path}")
            # manifests hold exactly what the recovered state owns (no record of a replayed group survives twice)
            if metrics_manifest is not None:
                _prune_manifest(metrics_manifest, done_batches, lambda r: (int(r["chunk_index"]), int(r["chunk_index"]) - 1))
            if row_manifest is not None:
                _prune_manifest(row_manifest, done_batches, lambda r: (int(r["shuffle_index"]), int(r["shuffle_index"]) // spb))
            if all_player_manifest is not None:
                _prune_manifest(all_player_manifest, done_batches, lambda r: (int(r["deterministic_batch_id"]), int(r["deterministic_batch_id"])))
            LOGGER.info("Resuming: %d of %d batches already complete", len(done_batches), n_batches)
        except (ValueError, FileNotFoundError, KeyError) as exc:
            if world == 1:
                raise
            resume_error = f"{type(exc).__name__}: {exc}"
    if world > 1:  # every rank plans against rank 0's recovered state — or raises rank 0's recovery error with it
        resume_error, shared = gather_objects((resume_error, sorted(done_batches)), broadcast_from=0)
        if resume_error is not None:
            raise ValueError(f"resume failed on rank 0: {resume_error}")
        done_batches = set(shared)
    pending = [b for b in range(n_batches) if b not in done_batches]
    shard_identities: dict[str, tuple[int, str]] = {}
    lag_total = None
    if rng_lags:
        if done_batches:  # a strategy's series runs over every shuffle of the root: a partial replay cannot rebuild it
            raise ValueError("lag statistics need the whole shuffle range of the run; this checkpoint already owns batches: use --force")
        if row_dir is not None or all_player_dir is not None:
            raise ValueError("--rng-lag-sums runs without rows / all-player batches (one post-pass per launch): run them separately")
    target = oracle_game_profile.default_target_score if oracle_game_profile else 10_000
    max_rounds = oracle_game_profile.default_max_rounds if oracle_game_profile else 200
    ov = oracle_game_profile.tournament_overrides() if oracle_game_profile else None
    table = packed_table if packed_table is not None else rt.pack_strategies(strategies)
    gps = S // k
    group_batches = max(1, MAX_GAMES_PER_LAUNCH // max(spb * gps, 1))
    want_rows = row_dir is not None
    # rows as per-shuffle column images (fk_tournament_run_columns) whenever the engine offers them: the shards are framed by the library's
    # own Parquet writer on host threads instead of Arrow in writer processes (tournament.write_row_shards_from_columns).  Launch groups
    # are then cut by image bytes, and group i's shards are written while group i + 1 plays (two page-locked buffers per engine).
    columns_mode = want_rows and getattr(eng, "columns_with_seeds", False) and k <= 64 and not rng_lags and all_player_dir is None
    checkpoint_batches = group_batches  # an intermediate checkpoint at least every this many batches (or sim.ckpt_every_sec)
    if columns_mode:
        from .backend import row_columns_bytes

        image_bytes = row_columns_bytes(k, gps)
        group_batches = max(1, min(group_batches, ROWS_GROUP_BYTES // max(spb * image_bytes, 1)))
        if pending and hasattr(eng, "pinned_empty") and getattr(eng, "_pinned_columns", None) is None:
            # the engine's page-locked image buffers, made beside what follows, one after the other on their own thread (fk_host_alloc
            # populates the pages outside the HIP runtime: ~20 ms per 256 MB, no launch waits meanwhile)
            size = max(group_batches * spb * image_bytes, min(ROWS_GROUP_BYTES, plan.required_shuffles * image_bytes))
            eng._pinned_columns = {"slots": [None] * ROWS_SLOTS, "jobs": [None] * ROWS_SLOTS, "turn": 0,
                                   "allocating": [_pin_threads().submit(eng.pinned_empty, size, np.uint8) for _ in range(ROWS_SLOTS)]}
        if pending:  # what every shard job of this player count shares, made while the first buffer is being page-locked (the first
            from .parquet_template import shard_footer_template  # use of Arrow's writer in a process takes 30 ms)

            shard_footer_template(k)
            sidecars.template("row_shard", row_dir / "rows_template.parquet",
                              schema=lambda: _stored_schema(raw_simulation_schema_for(k), **rt.SHARD_WRITER_OPTIONS))
    pinned_rows = None
    t_start = time.perf_counter()
    games_done = 0

    # The final checkpoint lists every shuffle index of the run (the reference's format: 312 500 Python ints on BASELINE config 2 — 5 ms
    # to build, as long as everything else the host does after the last launch): built on the helper thread under the first engine call.
    every_shuffle = (_helper_thread().submit(lambda: list(range(plan.required_shuffles))) if rank == 0 and pending else None)
    checkpoint_written: list = []  # the final checkpoint's write, in flight on the helper thread (joined before anything reads the file)

    def save(final: bool) -> None:
        wins, sums, sqs = rt.tally_to_counters(total, ids, k, dense=metric_chunk_dir is not None)
        completed = sorted(done_batches)
        # process blocks are numbered from 1 (run_tournament.py:1576-1586); one block = one deterministic batch here
        if every_shuffle is not None and len(completed) == n_batches:
            shuffle_list: list[int] = every_shuffle.result()
        else:
            shuffle_list = []
            run_first = run_last = None  # runs of consecutive batches become ONE range
            for b in completed + [None]:
                if run_last is not None and b is not None and b == run_last + 1:
                    run_last = b
                    continue
                if run_first is not None:
                    shuffle_list.extend(range(run_first * spb, min((run_last + 1) * spb, plan.required_shuffles)))
                run_first = run_last = b
        ck_meta = {**get_meta(), "completed_shuffle_indices": shuffle_list,
                   "completed_process_block_indices": [b + 1 for b in completed], "complete": final}
        content = ckpt.dump_checkpoint(wins, sums if collect_metrics else None, sqs if collect_metrics else None, ck_meta)
        if final and defer_final_checkpoint and not (sidecars is not None and sidecars.enabled):
            # 1.4 MB for config 2: the write goes to the helper thread (file I/O drops the GIL) while the caller builds the summary tables;
            # `checkpoint_written` is joined before the completion stamp — or anything else — reads the file
            checkpoint_written.append(_helper_thread().submit(_atomic_write_bytes, checkpoint_path, content))
        else:
            _atomic_write_bytes(checkpoint_path, content)

    n_groups = 0
    from collections import deque

    # the launch groups whose shards are being written while the next ones play, oldest first: one fewer than the engine has image buffers
    # (three buffers: the engine may be two groups ahead of the writer — a group's engine and writer times differ by player count and by
    # group, and with a single group in flight each waited for the other in turn: 0.15 s of a 0.74-s production sweep)
    in_flight: deque = deque()
    max_in_flight = 1
    batches_since_save, last_save = 0, time.perf_counter()

    def finish(b0, b1, lo, hi, j, local, local_stats, local_ratios, row_records, fragments, res, shard_job, per_batch) -> None:
        """What follows a launch group's engine call: its shards' manifest lines, the reduction over ranks, chunk files, checkpoint."""
        nonlocal total, games_done, lag_total, batches_since_save, last_save
        if shard_job is not None:
            row_records = shard_job.result()
            if callable(row_records):  # the per-shard manifest lines: built here, not on the shard thread (its host threads would idle)
                row_records = row_records()
        if rng_lags:  # this group's ranges in rank order (contiguous whole batches per rank), appended to the run's summary
            from .rng_lags import LagSummary

            part = LagSummary.from_engine(res, rng_lags) if hi > lo else None
            parts = gather_objects(part, dst=0) if world > 1 else [part]
            if rank == 0:
                for piece in parts:
                    if piece is not None:
                        lag_total = piece if lag_total is None else lag_total.merge(piece)
        group = reduce_tally(local, dst=0)
        group_stats = reduce_tally(local_stats, dst=0) if local_stats is not None else None  # integer sums, like the tally
        # the float64 sums: a deterministic batch is played whole by ONE rank (the others hold +0.0 = all-zero bits), so the int64 SUM of
        # the bit patterns hands rank 0 every batch's sums bit for bit — no floating-point addition across ranks
        group_ratios = (reduce_tally(np.ascontiguousarray(local_ratios).view(np.int64), dst=0).view(np.float64)
                        if local_ratios is not None else None)
        if want_rows and world > 1:
            gathered = gather_objects(row_records, dst=0)
            row_records = [r for part in (gathered or []) for r in part]
        if rank == 0:
            if want_rows:
                row_records.sort()
                rt.append_manifest_lines(row_manifest, [rec[1] for rec in row_records])
                shard_prefix = f"{os.fspath(row_dir)}{os.sep}rows_{cfg.sim.seed}_{k}p_"
                for sh, _, size, sha in row_records:  # the shards' byte identities, for the completion stamp
                    shard_identities[f"{shard_prefix}{sh:012d}.parquet"] = (int(size), sha)
            chunk_lines, all_player_records = [], []
            lists = fragments.result() if fragments is not None else None
            for n, b in enumerate(range(b0, b1)):
                if all_player_dir is not None:
                    ap = all_player_batch_table(group_stats[n], ids, cfg.sim.seed, k, b, group_ratios[n])
                    name = f"all_player_batch_{b + 1:06d}.parquet"
                    _write_parquet_atomic(ap, all_player_dir / name)
                    all_player_records.append({"path": name, "rows": ap.num_rows, "root_seed": cfg.sim.seed, "n_players": k,
                                               "deterministic_batch_id": b, "outcome_schema_version": OUTCOME_SCHEMA_VERSION,
                                               "absent_columns": [],
                                               "float_sum_order": "per strategy, exposures in ascending (shuffle, game, seat) order, sequential "
                                                                  "float64 (the order np.add.at visits curated rows, all_player_metrics.py:174-177)"})
                if metric_chunk_dir is not None:
                    chunk = _metric_chunk_table(group[n], ids, k)
                    name = f"metrics_{b + 1:06d}.parquet"  # chunk / process-block indices count from 1 (run_tournament.py:1603-1642)
                    chunk_identity = None
                    if sidecars.v3 is not None:
                        # encoded in memory: the chunk's bytes give its sidecar (a template around size / SHA-256 / name) and the four
                        # identity fields a sealed manifest needs of every shard (runner.py:570-577 of the reference)
                        chunk_identity = _publish_shard_v3(chunk, metric_chunk_dir / name, sidecars.template(
                            "metric_chunk", metric_chunk_dir / name, schema=lambda: _stored_schema(chunk.schema, **_CHUNK_WRITER_OPTIONS)),
                            **_CHUNK_WRITER_OPTIONS)
                    else:
                        _write_parquet_atomic(chunk, metric_chunk_dir / name, **_CHUNK_WRITER_OPTIONS)
                        sidecars.write("metric_chunk", metric_chunk_dir / name)
                    first_sh, last_sh = b * spb, min((b + 1) * spb, plan.required_shuffles)
                    record = {"path": name, "rows": chunk.num_rows, "chunk_index": b + 1, "process_block_index": b + 1,
                              "root_seed": cfg.sim.seed, "n_players": k, "deterministic_batch_id": b,
                              "shuffle_index_start": first_sh, "shuffle_index_end": last_sh - 1, "shuffle_count": last_sh - first_sh,
                              "shuffle_indices": "@indices@", "shuffle_seeds": "@seeds@",  # (spliced in below)
                              "rng_scheme_version": urandom.RNG_SCHEME_VERSION,
                              "rng_purpose_namespace": int(urandom.RandomPurpose.TOURNAMENT_SHUFFLE),
                              "outcome_schema_version": OUTCOME_SCHEMA_VERSION,
                              "tournament_method_version": TOURNAMENT_METHOD_VERSION}
                    if oracle_game_profile is not None:  # run_tournament.py:1668
                        record["game_profile_sha256"] = oracle_game_profile.sha256
                    if chunk_identity is not None:
                        record.update(chunk_identity)
                    indices, seeds = lists[n]
                    chunk_lines.append(json.dumps(record, sort_keys=True).replace('"@indices@"', indices).replace('"@seeds@"', seeds))
                if per_batch:
                    total += group[n]
                done_batches.add(b)
            if chunk_lines:
                rt.append_manifest_lines(metrics_manifest, chunk_lines)
            if all_player_records:
                rt.append_manifest_records(all_player_manifest, all_player_records)
            if not per_batch:
                total += group[0]
            games_done += (min(b1 * spb, plan.required_shuffles) - b0 * spb) * gps
            batches_since_save += b1 - b0
            if j + 1 < len(pending) and (batches_since_save >= checkpoint_batches or
                                         time.perf_counter() - last_save >= max(float(cfg.sim.ckpt_every_sec), 0.0)):
                save(final=False)  # the checkpoint that owns the manifest lines appended so far (the last group's is the final one below)
                batches_since_save, last_save = 0, time.perf_counter()
            LOGGER.info("Batches %d..%d done: %.3g games/s so far", b0, b1 - 1,
                        games_done / max(time.perf_counter() - t_start, 1e-9))
    # A launch group passes three steps: the prelude (its shuffle range, buffers, image slot), the ENGINE PART (every device call, in
    # order: hint, the games, the shuffle and game fingerprints) and what follows (shard job, manifest lines, checkpoint: `post`).  One
    # process in rows mode runs the engine part on its own thread, so that group g + 1 plays while this thread lays out group g's shard
    # job and finishes earlier groups: 5 ms of interpreter work per group that used to stand between two 7-ms launches (production
    # sweep, 41 groups).  The engine is only ever entered from one thread at a time: this one between joining a group's engine part and
    # submitting the next, the launcher in between.
    pipelined = columns_mode and world == 1 and hasattr(eng, "rows_wait") and ROWS_PIPELINE

    def engine_part(g: dict) -> dict:
        lo, hi = g["lo"], g["hi"]
        if g["hint"] is not None:
            eng.hint_next(*g["hint"], need_state=want_rows)
        batch_arg = spb if g["per_batch"] else hi - lo
        _trace(f"{k}p group {g['index']}: engine call")
        if g["use_columns"]:
            # (the shard job was laid out in the prelude, around the buffers this call fills)
            res = eng.tournament_columns(table, k, cfg.sim.seed, lo, hi, ids, shuffles_per_batch=batch_arg, target_score=target,
                                         max_rounds=max_rounds, overrides=ov, columns_out=g["pinned_rows"], shuffle_seeds_out=g["shuffle_seeds32"],
                                         game_seeds_out=g["game_seeds"], **({"async_rows": True} if g["async_rows"] else {}))
            _trace(f"{k}p group {g['index']}: engine returned")
            g["tasks"].shuffle_seed[...] = g["shuffle_seeds32"]  # (int64 in the shard records)
            shard_job = _shard_thread().submit(_write_group_shards, eng, res.get("rows_event") if g["async_rows"] else None, g["write"])
            if g["slot"] is not None:
                eng._pinned_columns["jobs"][g["slot"]] = shard_job
            return {"res": res, "shard_job": shard_job}
        elif rng_lags:
            res = eng.tournament_lags(table, k, cfg.sim.seed, lo, hi, rng_lags, shuffles_per_batch=batch_arg, target_score=target,
                                      max_rounds=max_rounds, overrides=ov)
        else:
            res = eng.tournament(table, k, cfg.sim.seed, lo, hi, shuffles_per_batch=batch_arg, target_score=target, max_rounds=max_rounds,
                                 overrides=ov, want_rows=want_rows, **g["extra"])
        _trace(f"{k}p group {g['index']}: engine returned")
        played = {"res": res, "shuffle_seeds": None, "seeds102": None, "shard_job": None}
        if want_rows:
            played["shuffle_seeds"] = _shuffle_seeds(eng, cfg.sim.seed, k, lo, hi)
            if hasattr(eng, "game_seeds"):  # the rows' game_seed column, hashed on the device
                played["seeds102"] = eng.game_seeds(int(urandom.RandomPurpose.TOURNAMENT_GAME), cfg.sim.seed, k, lo, hi, gps)
            _trace("fingerprints here")
        return played

    def post(g: dict, played: dict | None) -> None:
        nonlocal n_groups
        b0, b1, lo, hi, per_batch = g["b0"], g["b1"], g["lo"], g["hi"], g["per_batch"]
        local, local_stats, local_ratios, row_records = g["local"], g["local_stats"], g["local_ratios"], g["row_records"]
        shard_job = res = None
        if played is not None:
            res = played["res"]
            first = lo // spb - b0 if per_batch else 0
            local[first:first + len(res["tally"])] = res["tally"]
            if local_stats is not None:
                local_stats[first:first + len(res["seat_stats"])] = res["seat_stats"]
                local_ratios[first:first + len(res["seat_ratio_sums"])] = res["seat_ratio_sums"]
            if played["shard_job"] is not None:  # (laid out in the prelude, submitted by the engine part)
                shard_job = played["shard_job"]
            elif want_rows:  # AoS rows (more than 64 seats, lag or all-player runs, an engine without column images): Arrow in writer processes
                sh_index = np.arange(lo, hi, dtype=np.int64)  # the ShuffleTask identities (run_tournament.py:97-105), as arrays
                tasks = rt.ShuffleRange(cfg.sim.seed, k, sh_index, played["shuffle_seeds"], sh_index // spb)
                sha = oracle_game_profile.sha256 if oracle_game_profile else None
                seeds102 = played["seeds102"]
                shard_sidecar = sidecars.template("row_shard", row_dir / "rows_template.parquet",
                                                  schema=lambda: _stored_schema(raw_simulation_schema_for(k), **rt.SHARD_WRITER_OPTIONS))
                row_records.extend(rt.write_row_shards(row_dir, tasks, res["rows"], ids, sha, threads=ROW_WRITER_THREADS,
                                                       game_seeds=seeds102, as_lines=True, sidecar=shard_sidecar))
        group_args = dict(b0=b0, b1=b1, lo=lo, hi=hi, j=g["j"], local=local, local_stats=local_stats, local_ratios=local_ratios,
                          row_records=row_records, fragments=g["fragments"], res=res, shard_job=shard_job, per_batch=per_batch)
        n_groups += 1
        if shard_job is not None:
            in_flight.append(group_args)
        while len(in_flight) > (max_in_flight if shard_job is not None else 0):  # earlier groups: their shards were written while later ones played
            finish(**in_flight.popleft())
        if shard_job is None:
            finish(**group_args)

    i = 0
    _trace("launch loop begins")
    awaiting_post: tuple | None = None  # (pipelined) the group whose engine part has returned and whose `post` runs beside the next one's
    try:
      while i < len(pending):
          # contiguous run of pending batches, capped by the launch-group size
          j = i
          while j + 1 < len(pending) and pending[j + 1] == pending[j] + 1 and (j + 1 - i) < group_batches:
              j += 1
          b0, b1 = pending[i], pending[j] + 1
          lo, hi = shard_shuffle_range(b0 * spb, min(b1 * spb, plan.required_shuffles), rank, world, batch_size=spb)
          # Per-batch tallies are only needed for the metric chunk files; without them the group is one tally, which the
          # engine keeps in LDS when the table is small.
          per_batch = metric_chunk_dir is not None or all_player_dir is not None
          g: dict[str, Any] = dict(
              index=n_groups + (1 if awaiting_post is not None else 0), b0=b0, b1=b1, lo=lo, hi=hi, j=j, per_batch=per_batch,
              local=np.zeros((b1 - b0 if per_batch else 1, S, 26), dtype=np.int64),
              local_stats=np.zeros((b1 - b0, S, SEAT_STAT_COLS), dtype=np.int64) if all_player_dir is not None else None,
              local_ratios=np.zeros((b1 - b0, S, SEAT_RATIO_COLS), dtype=np.float64) if all_player_dir is not None else None,
              row_records=[],  # (shuffle index, manifest line, shard bytes, shard sha256)
              fragments=None, hint=None, use_columns=False, async_rows=False, pinned_rows=None, slot=None, extra={}, write=None, tasks=None,
              shuffle_seeds32=None, game_seeds=None)
          if metric_chunk_dir is not None and rank == 0:
              # the shuffle fingerprints of the whole group in one vectorised pass (on the device), and their JSON text on the
              # helper thread while the group plays
              g_first, g_last = b0 * spb, min(b1 * spb, plan.required_shuffles)
              group_seeds = _shuffle_seeds(eng, cfg.sim.seed, k, g_first, g_last)
              bounds = [(b * spb, min((b + 1) * spb, plan.required_shuffles)) for b in range(b0, b1)]
              g["fragments"] = _helper_thread().submit(_shuffle_list_fragments, group_seeds, g_first, spb, bounds)
          if hi > lo:
              if j + 1 < len(pending) and hasattr(eng, "hint_next"):
                  # the launch group after this one (same table, k, root): prepared in the drain tail of this group's kernel
                  j2 = j + 1
                  while j2 + 1 < len(pending) and pending[j2 + 1] == pending[j2] + 1 and (j2 + 1 - (j + 1)) < group_batches:
                      j2 += 1
                  lo2, hi2 = shard_shuffle_range(pending[j + 1] * spb, min((pending[j2] + 1) * spb, plan.required_shuffles), rank, world,
                                                 batch_size=spb)
                  if hi2 > lo2:
                      g["hint"] = (lo2, hi2)
              if all_player_dir is not None:
                  g["extra"]["want_seat_stats"] = True
              g["use_columns"] = columns_mode
              if columns_mode:
                  need = (hi - lo) * image_bytes
                  if hasattr(eng, "pinned_empty"):
                      # page-locked image buffers per ENGINE (kept for its life, grown when a launch group needs more: page-locking a gigabyte
                      # takes ~0.25 s, and a sweep over eight player counts that asked for one per count spent 2.4 of its 3.5 s there);
                      # the ring turns over the ENGINE's launch groups (not this call's): the last groups of a player count may still be
                      # written — their tail deferred to the publisher thread, run_multi — while the next count's first groups play
                      pin = getattr(eng, "_pinned_columns", None)
                      if pin is None:
                          pin = eng._pinned_columns = {"slots": [None] * ROWS_SLOTS, "jobs": [None] * ROWS_SLOTS, "turn": 0}
                      for _ in pin["slots"]:  # the next buffer of the ring that exists (one still being page-locked is passed over)
                          slot = pin["turn"] % len(pin["slots"])
                          pin["turn"] += 1
                          waiting = pin.get("allocating", [None] * len(pin["slots"]))[slot]
                          if pin["slots"][slot] is not None or waiting is None or waiting.done() or pin["turn"] <= 1:
                              break  # (waiting is None: no allocation was started ahead — made below)
                      max_in_flight = len(pin["slots"]) - 1
                      if pin["jobs"][slot] is not None:  # the shard job that read this buffer last
                          try:
                              pin["jobs"][slot].result()
                          except Exception:  # noqa: BLE001 - its own group raises it when it is finished
                              pass
                          pin["jobs"][slot] = None
                      slots = pin["slots"]
                      if slots[slot] is None and pin.get("allocating") and pin["allocating"][slot] is not None:
                          slots[slot], pin["allocating"][slot] = pin["allocating"][slot].result(), None
                      if slots[slot] is None or len(slots[slot]) < need:
                          slots[slot] = None
                          slots[slot] = eng.pinned_empty(max(need, min(ROWS_GROUP_BYTES, plan.required_shuffles * image_bytes)), np.uint8)
                      g["pinned_rows"], g["slot"] = slots[slot], slot
                      _trace(f"image buffer {slot} here")
              elif want_rows and hasattr(eng, "pinned_empty"):
                  # rows land in a page-locked buffer (one per run, sized for the largest launch group): one DMA per chunk at PCIe
                  # rate, under the next chunk's game kernel
                  need = (hi - lo) * gps
                  if pinned_rows is None or len(pinned_rows) < need:
                      from .backend import row_dtype

                      pinned_rows = None
                      pinned_rows = eng.pinned_empty(need, row_dtype(k))
                  g["extra"]["rows_out"] = pinned_rows
              # the images' last copy to the host is awaited by the shard job, not here: the next launch group's games run beside it
              g["async_rows"] = columns_mode and g["pinned_rows"] is not None and hasattr(eng, "rows_wait") and ROWS_ASYNC
              if columns_mode:
                  # The shard job of the group, laid out BEFORE it plays: around the image buffer and two fingerprint arrays the engine
                  # call fills (fk_tournament_run_columns_seeds), so that the engine part hands it to the shard thread the moment the call
                  # returns — between a launch and its shards' first byte there is then no interpreter work that a publishing tail
                  # holding the lock could delay (measured at every change of player count: 10 + 3 ms, the writer's threads idle).
                  n_sh = hi - lo
                  sh_index = np.arange(lo, hi, dtype=np.int64)  # the ShuffleTask identities (run_tournament.py:97-105), as arrays
                  g["shuffle_seeds32"], g["game_seeds"] = np.empty(n_sh, dtype=np.uint32), np.empty(n_sh * gps, dtype=np.uint32)
                  g["tasks"] = rt.ShuffleRange(cfg.sim.seed, k, sh_index, np.zeros(n_sh, dtype=np.int64), sh_index // spb)
                  if g["pinned_rows"] is None:  # (an engine without page-locked buffers: a plain one per group)
                      g["pinned_rows"] = np.empty(n_sh * image_bytes, dtype=np.uint8)
                  images = g["pinned_rows"].reshape(-1)[:n_sh * image_bytes].reshape(n_sh, image_bytes)
                  shard_sidecar = sidecars.template("row_shard", row_dir / "rows_template.parquet",
                                                    schema=lambda: _stored_schema(raw_simulation_schema_for(k), **rt.SHARD_WRITER_OPTIONS))
                  g["write"] = rt.write_row_shards_from_columns(row_dir, g["tasks"], images, g["game_seeds"],
                                                                oracle_game_profile.sha256 if oracle_game_profile else None,
                                                                threads=ROW_WRITER_THREADS, sidecar=shard_sidecar, deferred_lines=True, deferred_write=True)
                  _trace("shard job laid out")
              if pipelined:
                  engine_call = _launcher_thread().submit(engine_part, g)
                  try:
                      if awaiting_post is not None:
                          previous, awaiting_post = awaiting_post, None
                          post(*previous)
                  except BaseException:
                      try:
                          engine_call.result()  # the engine is not left running; the error of `post` is the one raised
                      except BaseException:  # noqa: BLE001
                          pass
                      raise
                  awaiting_post = (g, engine_call.result())
              else:
                  post(g, engine_part(g))
          else:
              post(g, None)
          i = j + 1
      if awaiting_post is not None:
          previous, awaiting_post = awaiting_post, None
          post(*previous)
      if not (defer_tail is not None and world == 1):
          while in_flight:
              finish(**in_flight.popleft())
    finally:
        if sys.exc_info()[0] is not None:
            for group in in_flight:  # an error above: no writer thread may outlive the call
                try:
                    if group["shard_job"] is not None:
                        group["shard_job"].result()
                except Exception:  # noqa: BLE001 - the first error is the one that propagates
                    pass
    result = {"shard_identities": shard_identities, "checkpoint_written": checkpoint_written}
    last_groups, in_flight = list(in_flight), deque()

    def complete() -> dict:
        """The run's tail: the last launch groups' shards (still being written when the tail is deferred), the final checkpoint, the
        manifests' sidecars."""
        for group in last_groups:
            finish(**group)
        if rank == 0:
            save(final=True)
            sidecars.write("checkpoint", checkpoint_path)
            for manifest in (row_manifest, metrics_manifest):  # the manifests are final now: their sidecars bind the complete files
                if manifest is not None and sidecars.v3 is not None:
                    if manifest.exists():  # sealed: canonical lines in coordinate order + the coordinate-sorted root over the shards' identities
                        sidecars.v3.publish_manifest(manifest, n_players=k)
                elif manifest is not None:
                    sidecars.write("shard_manifest", manifest)
        barrier()
        result.update(tally=total, games=games_done, seconds=time.perf_counter() - t_start, lag_summary=lag_total)
        return result

    if last_groups:  # (only with defer_tail, one process): the caller runs it — run_multi on its publisher thread, under the next count
        defer_tail.append(complete)
        return result
    return complete()


def run_single_n(cfg: AppConfig, n: int, strategies: list[ThresholdStrategy] | None = None, *, force: bool = False,
                 oracle_game_profile: GameProfile | None = None, _shared: "_SweepShared | None" = None, _defer_publish: list | None = None) -> int:
    """Run a Farkle tournament for a single player count ``n``; returns the number of games of the plan.  ``_defer_publish`` (run_multi):
    a list that receives the run's publishing tail — summary tables, completion stamp — as a callable instead of having it executed here."""
    import pyarrow as pa

    rank, _ = _rank_world()
    _trace(f"{n}p run_single_n")
    if _shared is not None:
        strategies, grid_size = _shared.strategies, len(_shared.strategies)
    else:
        strategies, grid_size = _resolve_strategies(cfg, strategies)
    plan = _plan_workload_from_config(cfg, grid_size, n)
    n_dir = cfg.n_dir(n)
    n_dir.mkdir(parents=True, exist_ok=True)
    if not force and simulation_is_complete(cfg, n, plan):
        # a completed run that is now asked for the lag statistics it was not run with: the series needs every shuffle of the root, so
        # nothing can be added to published outputs — say so instead of returning without the files (round-4 advisor)
        if cfg.sim.rng_lag_sums and not (cfg.rng_lag_sums_path(n).exists() and cfg.rng_lag_stats_path(n).exists()):
            raise ValueError(f"{n}p is already complete without {cfg.rng_lag_sums_path(n).name}: --rng-lag-sums needs every shuffle of the "
                             "run; use --force to replay it with the lag statistics")
        LOGGER.info("Simulation already complete; preserving published outputs for %sp", n)
        return plan.required_games
    plan_path = n_dir / "simulation_workload_plan.json"
    row_dir = cfg.simulation_row_dir(n)
    metric_chunk_dir = cfg.metric_chunk_dir(n)
    all_player_dir = cfg.all_player_batch_dir(n)
    if plan.cap_exceeded:
        if rank == 0:
            write_workload_plan(plan_path, plan)
        raise WorkloadCapExceeded(plan)
    ckpt_path = cfg.checkpoint_path(n)
    # (a single process derives the manifest frame / sha256 / Arrow table on the helper thread, under the first engine call)
    shared = _shared if _shared is not None else _SweepShared(strategies, background=rank == 0 and _rank_world()[1] == 1)
    write_manifest = False
    if rank == 0:
        if force:
            for path in (ckpt_path, n_dir / f"{n}p_checkpoint.parquet", cfg.metrics_path(n), simulation_done_path(cfg, n),
                         cfg.rng_lag_sums_path(n), cfg.rng_lag_stats_path(n)):
                path.unlink(missing_ok=True)
                path.with_name(path.name + ".sidecar.json").unlink(missing_ok=True)
            for d in (row_dir, metric_chunk_dir, all_player_dir):
                if d is not None and d.exists():
                    for f in d.iterdir():
                        if f.suffix in {".parquet", ".jsonl", ".tmp", ".json"}:  # (.json: the sidecars of a previous run)
                            f.unlink()
        manifest_path = cfg.strategy_manifest_root_path()
        if manifest_path.exists():
            if str(manifest_path) not in shared.verified:  # (the sweep's other player counts share the root's manifest: checked once)
                import pyarrow.parquet as pq

                if not pq.read_table(manifest_path).equals(shared.table):
                    raise ValueError(f"Strategy manifest at {manifest_path} does not match the configured grid")
                shared.verified.add(str(manifest_path))
        else:
            write_manifest = True
    _trace(f"{n}p checked")
    sidecars = _Sidecars(cfg, n, [cfg.strategy_manifest_root_path(), plan_path], bool(cfg.sim.sidecars))

    def publish_inputs() -> None:
        """The run's two input artifacts (strategy manifest, workload plan) and their sidecars.  Nothing the engine needs: a single
        process writes them on the helper thread while the first launch plays (joined before anything reads them back)."""
        if write_manifest:
            _write_parquet_atomic(shared.table, cfg.strategy_manifest_root_path())
            shared.verified.add(str(cfg.strategy_manifest_root_path()))
        write_workload_plan(plan_path, plan)
        sidecars.write("strategy_manifest", cfg.strategy_manifest_root_path(), sources=(),
                       support_counts=sorted({int(v) for v in cfg.sim.n_players_list}))
        sidecars.write("workload_plan", plan_path, sources=[cfg.strategy_manifest_root_path()])

    published = None
    if rank == 0:
        if _rank_world()[1] == 1 and not cfg.sim.sidecars:  # (sidecars of later artifacts name these files as sources: keep the order)
            published = _helper_thread().submit(publish_inputs)
        else:
            publish_inputs()
    for d in (row_dir, metric_chunk_dir, all_player_dir):
        if d is not None:
            d.mkdir(parents=True, exist_ok=True)
    barrier()  # rank 0's --force cleanup and manifest write are complete before any rank plays or writes a shard
    _trace(f"{n}p inputs published")
    run_tail: list = []  # rows mode under run_multi: the last launch group's shard writing + final checkpoint, deferred with the publishing tail
    try:
        result = run_tournament(cfg=cfg, n_players=n, strategies=strategies, plan=plan, checkpoint_path=ckpt_path,
                            collect_metrics=cfg.sim.expanded_metrics, row_dir=row_dir, metric_chunk_dir=metric_chunk_dir,
                            resume=not force, checkpoint_metadata=lambda: {"strategy_manifest_sha": shared.manifest_sha},
                            oracle_game_profile=oracle_game_profile, all_player_dir=all_player_dir, sidecars=sidecars,
                            rng_lags=cfg.rng_diagnostic_lags() if cfg.sim.rng_lag_sums else None, defer_final_checkpoint=True,
                            packed_table=shared.packed, defer_tail=run_tail if _defer_publish is not None else None)
    finally:
        if published is not None:
            published.result()  # the inputs are on disk (or their error is raised) before the summaries and the stamp name them
    if rank != 0:
        return plan.required_games

    def publish() -> None:
        _trace(f"{n}p tail begins")
        for tail in run_tail:
            tail()  # (fills `result`)
        _trace(f"{n}p tail: run complete")
        _publish_results(cfg, n, strategies, plan, result, grid_size, ckpt_path, n_dir, sidecars, oracle_game_profile)
        _trace(f"{n}p tail ends")

    if _defer_publish is not None:
        _defer_publish.append(publish)
    else:
        publish()
    return plan.required_games


def _publish_results(cfg: AppConfig, n: int, strategies: list[ThresholdStrategy], plan: TournamentWorkloadPlan, result: dict, grid_size: int,
                     ckpt_path: Path, n_dir: Path, sidecars: "_Sidecars", oracle_game_profile: GameProfile | None) -> None:
    """What follows the last launch of a player count on rank 0: lag tables, summary / metrics parquets, the completion stamp."""
    import pyarrow as pa

    ids = [int(s.strategy_id) for s in strategies]
    if cfg.sim.rng_lag_sums and result.get("lag_summary") is not None:
        # the strategy family of the reference's RNG diagnostics without rows: the sufficient statistics, and the stats rows
        # (_rows_for_online_group, analysis/rng_diagnostics.py:2110-2160) computed from them
        from .rng_lags import lag_sums_table, lag_stats_table

        _write_parquet_atomic(lag_sums_table(result["lag_summary"], ids, cfg.sim.seed, n), cfg.rng_lag_sums_path(n))
        _write_parquet_atomic(lag_stats_table(result["lag_summary"], ids, n), cfg.rng_lag_stats_path(n))
    # (A) summary parquet, (B) expanded metrics parquet — column order, types and values as in runner.py:1612-1712, built column by
    # column from the tally (the per-strategy dict loop cost 25 us per strategy and table: 140 ms of a 330-ms config-3 run).  Rows in
    # the reference's order: strategies sorted by the STRING of their id, those without an attempted exposure left out.
    tally = np.asarray(result["tally"], dtype=np.int64)
    ids_arr = np.asarray(ids, dtype=np.int64)
    order = np.array(sorted(range(len(ids)), key=lambda i: str(ids[i])), dtype=np.int64)
    order = order[tally[order, COL_ATTEMPTED] > 0]
    t = tally[order]
    strat_col = ids_arr[order]
    w_i, attempted, completed, safety = t[:, COL_WINS], t[:, COL_ATTEMPTED], t[:, COL_COMPLETED], t[:, COL_SAFETY]
    w_f = w_i.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        rate = w_f / attempted  # (int / int in Python = the float64 quotient of the two integers, as here: both below 2^53)
        rate_completed = np.where(completed > 0, w_f / np.where(completed > 0, completed, 1), np.nan)
        safety_rate = safety.astype(np.float64) / attempted
        has_win = w_i > 0
        w_safe = np.where(has_win, w_f, 1.0)
        label_sum = {label: t[:, COL_SUMS + j].astype(np.float64) for j, label in enumerate(rt.METRIC_LABELS)}
        label_sq = {label: t[:, COL_SQ_SUMS + j].astype(np.float64) for j, label in enumerate(rt.METRIC_LABELS)}
        label_mean = {label: np.where(has_win, label_sum[label] / w_safe, 0.0) for label in rt.METRIC_LABELS}
    summary_cols: dict[str, Any] = {"strategy": strat_col, "wins": w_f, "attempted_exposures": attempted, "completed_exposures": completed,
                                    "safety_limit_exposures": safety, "losses": attempted - w_i, "win_rate_per_attempt": rate,
                                    "win_rate": rate, "win_rate_given_completion": rate_completed, "safety_limit_exposure_rate": safety_rate}
    metrics_cols: dict[str, Any] = {}
    if cfg.sim.expanded_metrics:
        for label in rt.METRIC_LABELS:
            summary_cols[f"mean_{label}"] = label_mean[label]
        metrics_cols = {"strategy": strat_col, "wins": w_i, "total_games_strat": attempted, "attempted_exposures": attempted,
                        "completed_exposures": completed, "safety_limit_exposures": safety, "losses": attempted - w_i,
                        "win_rate_per_attempt": rate, "win_rate": rate, "win_rate_given_completion": rate_completed,
                        "safety_limit_exposure_rate": safety_rate}
        for label in rt.METRIC_LABELS:
            mean = label_mean[label]
            # mean ** 2 stays Python's float power (the reference's expression, runner.py:1690): C pow(x, 2) is not guaranteed to round
            # like x * x, and the column is compared bit for bit with the reference's file
            mean_sq = np.array([m ** 2 for m in mean.tolist()], dtype=np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                var = np.where(has_win, np.maximum(label_sq[label] / w_safe - mean_sq, 0.0), 0.0)
            metrics_cols[f"sum_{label}"] = label_sum[label]
            metrics_cols[f"sq_sum_{label}"] = label_sq[label]
            metrics_cols[f"mean_{label}"] = mean
            metrics_cols[f"var_{label}"] = var
        metrics_cols["expected_score"] = label_sum["winning_score"] / attempted
    summary = len(order) > 0
    metrics_rows = summary and bool(metrics_cols)
    if summary:
        _write_parquet_atomic(pa.table(summary_cols), n_dir / f"{n}p_checkpoint.parquet")
        sidecars.write("checkpoint_summary", n_dir / f"{n}p_checkpoint.parquet", sources=[ckpt_path])
    if metrics_rows:
        _write_parquet_atomic(pa.table(metrics_cols), cfg.metrics_path(n))
        sidecars.write("metrics_summary", cfg.metrics_path(n), sources=[ckpt_path])
    # the completion stamp, in the reference's shape (write_simulation_done, simulation/runner.py:685-743 -> write_stage_done,
    # utils/stage_completion.py:391-512): the simulation contract at the TOP level, where ingest reads it (ingest.py:191-215)
    from . import stage_completion as sc
    from .sidecars import config_hash, engine_code_revision

    profile_sha = oracle_game_profile.sha256 if oracle_game_profile is not None else None
    metadata: dict[str, Any] = {
        "n_players": n, "seed": cfg.sim.seed, "root_seed": cfg.sim.seed, "k": n, "num_shuffles": plan.required_shuffles,
        "shuffle_index_start": 0, "shuffle_index_end": plan.required_shuffles - 1, "deterministic_batch_count": plan.batch_count,
        "shuffles_per_batch": plan.shuffles_per_batch, "rng_scheme_version": urandom.RNG_SCHEME_VERSION,
        "rng_purpose_namespace": int(urandom.RandomPurpose.TOURNAMENT_SHUFFLE), "outcome_schema_version": OUTCOME_SCHEMA_VERSION,
        "tournament_method_version": TOURNAMENT_METHOD_VERSION, "n_strategies": grid_size}
    if profile_sha is not None:
        metadata["game_profile_sha256"] = profile_sha
    for pending_write in result.get("checkpoint_written", ()):
        pending_write.result()  # the final checkpoint is on disk (or its error is raised) before the stamp names and hashes it
    done_path = simulation_done_path(cfg, n)
    outputs: list[Path] = [ckpt_path, n_dir / "simulation_workload_plan.json"]  # the order of simulation/runner.py:1714-1727
    for extra in (n_dir / f"{n}p_checkpoint.parquet", cfg.metrics_path(n), cfg.strategy_manifest_root_path(),
                  cfg.simulation_row_dir(n), cfg.metric_chunk_dir(n), cfg.all_player_batch_dir(n),
                  cfg.rng_lag_sums_path(n) if cfg.sim.rng_lag_sums else None, cfg.rng_lag_stats_path(n) if cfg.sim.rng_lag_sums else None):
        if extra is not None and Path(extra).exists():
            outputs.append(Path(extra))
    if sidecars.v3 is not None:
        # the authenticated completion (write_simulation_done -> write_v3_stage_completion, release_identity.py:1227-1284): the reference's
        # output inventory only — every entry has a contract-v3 sidecar, shard directories stand as their sealed manifests; this engine's
        # extra outputs (all-player batches, lag statistics) are not part of the stage the reference's consumers authenticate
        reference_outputs = [p for p in outputs if p not in (cfg.all_player_batch_dir(n), cfg.rng_lag_sums_path(n), cfg.rng_lag_stats_path(n))]
        sidecars.v3.write_completion(done_path, sc.completion_output_files(reference_outputs, done_path))
        return
    engine_config_sha = config_hash(cfg)
    sc.write_stage_done(
        done_path, inputs=[cfg.strategy_manifest_root_path(), n_dir / "simulation_workload_plan.json"],
        outputs=sc.completion_output_files(outputs, done_path), stage="simulation", config_sha=engine_config_sha,
        stage_config_sha=sc.simulation_stage_config_sha(engine_config_sha, cfg.sim.seed, n, profile_sha),
        cache_key_version=sc.SIMULATION_CACHE_KEY_VERSION, freshness_key=sc.freshness_key(cfg, profile_sha),
        code_identity={"state": "supplied_by_caller", "commit": None, "dirty_fingerprint_sha256": None,
                       "revision": engine_code_revision(), "engine": "farkle_ii_amd/hip-gfx950"},
        metadata=metadata, known_identities=None if cfg.sim.sidecars else result.get("shard_identities"))


def run_multi(cfg: AppConfig, player_counts: Sequence[int] | None = None, *, force: bool = False,
              oracle_game_profile: GameProfile | None = None) -> dict[int, int]:
    """Run tournaments for several player counts (sequential over k, as in the reference)."""
    counts = list(player_counts) if player_counts is not None else list(cfg.sim.n_players_list)
    strategies, grid_size = _resolve_strategies(cfg, None)
    valid, _ = _filter_player_counts(counts, grid_size)
    results: dict[int, int] = {}
    # A player count's publishing tail (summary tables, completion stamp: 10 - 20 ms of Python on the 5 160-strategy grid; in rows mode
    # also the count's last launch groups, whose shards are still being written) runs on its own thread under the NEXT player counts'
    # engine calls (ctypes drops the GIL for their duration).  The tails run in order on that one thread; the launching thread joins a
    # tail — and raises its error — once it is done or two later ones are queued behind it, and all of them before the sweep returns
    # (it used to join the previous tail before queueing the next: 15 - 20 ms without a launch at every change of player count, the
    # shard writer idle meanwhile).  Ranked runs publish in line (their barriers order the ranks).
    overlap = _rank_world()[1] == 1
    shared = _SweepShared(strategies, background=overlap)  # manifest, packed table, ... once for the sweep (every count plays the same grid)
    from collections import deque

    tails: deque = deque()
    # The launching thread shares the interpreter with the publishing tail and the shard thread: with the default 5-ms switch interval
    # every return from an engine call or a file operation could wait that long for the lock while a tail ran Python (measured on the
    # production sweep with contract-v3 sidecars: 118 ms between two launches, the shard writer's threads idle for 0.3 of its 1.1 s).
    switch_interval = sys.getswitchinterval()
    if overlap:
        sys.setswitchinterval(min(switch_interval, 2e-4))
    try:
        for n in valid:
            tail: list = []
            results[n] = run_single_n(cfg, n, strategies=strategies, force=force, oracle_game_profile=oracle_game_profile, _shared=shared,
                                      _defer_publish=tail if overlap else None)
            while tails and (tails[0].done() or len(tails) > 2):
                tails.popleft().result()
            if tail:
                tails.append(_publisher_thread().submit(tail[0]))
    finally:
        sys.setswitchinterval(switch_interval)
        first_error = None
        while tails:  # every tail has run (or failed) before the sweep returns; the first error is the one raised
            try:
                tails.popleft().result()
            except BaseException as exc:  # noqa: BLE001
                first_error = first_error or exc
        if first_error is not None and sys.exc_info()[0] is None:
            raise first_error
        if _TRACE:
            t0 = _TRACE[0][0]
            for t, thread, label in _TRACE:
                print(f"[fk trace] {(t - t0) * 1e3:9.2f} ms  {thread:24s} {label}", file=sys.stderr)
            _TRACE.clear()
    return results


def write_active_config(cfg: AppConfig, dest_dir: Path) -> Path:
    """Persist the resolved simulation config next to the results (orchestration/seed_utils.py:86)."""
    import dataclasses

    import yaml

    def plain(obj):
        if dataclasses.is_dataclass(obj):
            return {f.name: plain(getattr(obj, f.name)) for f in dataclasses.fields(obj) if not f.name.startswith("_")}
        if isinstance(obj, Path):
            return str(obj)
        if isinstance(obj, (list, tuple)):
            return [plain(v) for v in obj]
        if isinstance(obj, dict):
            return {str(k): plain(v) for k, v in obj.items()}
        return obj

    data = plain(cfg)
    data.update(data.pop("opaque", {}))
    dest_dir = Path(dest_dir)
    dest_dir.mkdir(parents=True, exist_ok=True)
    path = dest_dir / "active_config.yaml"
    path.write_text(yaml.dump(data, Dumper=getattr(yaml, "CSafeDumper", yaml.SafeDumper), sort_keys=True), encoding="utf-8")
    return path
