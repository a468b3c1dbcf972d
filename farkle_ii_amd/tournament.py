"""Tournament shuffle/chunk functions on the GPU engine.

Mirrors ``src/farkle/simulation/run_tournament.py``: ``TournamentConfig`` :78-94, ``ShuffleTask`` :97-105,
``METRIC_LABELS`` :109-121, ``OutcomeCounter`` :165-245, ``_init_worker`` :265-277, ``_play_one_shuffle`` :301-393,
``_play_shuffle`` :396-400, ``_run_chunk`` :403-457, ``_run_chunk_metrics`` :473-585.  A chunk (contiguous shuffles)
is ONE kernel launch; the returned objects have the reference's types so its parent loop can absorb them.
"""
from __future__ import annotations

from collections import Counter, defaultdict
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Mapping, Sequence, Tuple

import json

import numpy as np

from . import random as urandom
from .backend import (COL_ATTEMPTED, COL_COMPLETED, COL_SAFETY, COL_SQ_SUMS, COL_SUMS, COL_WINS, FarkleHipError,
                      FK_ERR_ROLL_LIMIT)
from .engine import get_engine
from .game_profile import GameProfile
from .rows import OUTCOME_SCHEMA_VERSION, TOURNAMENT_METHOD_VERSION, row_to_dict, rows_to_table
from .strategies import ThresholdStrategy, pack_strategies, prepare_public_helper_strategies

NUM_SHUFFLES = 5_907
DESIRED_SEC_PER_CHUNK = 10
CKPT_EVERY_SEC = 30

METRIC_LABELS: Tuple[str, ...] = (
    "winning_score", "n_rounds", "winner_farkles", "winner_rolls", "winner_highest_turn", "winner_smart_five_uses",
    "winner_n_smart_five_dice", "winner_smart_one_uses", "winner_n_smart_one_dice", "winner_hot_dice",
    "winner_hit_max_rounds",
)


@dataclass
class TournamentConfig:
    n_players: int = 5
    num_shuffles: int = NUM_SHUFFLES
    desired_sec_per_chunk: int = DESIRED_SEC_PER_CHUNK
    ckpt_every_sec: int = CKPT_EVERY_SEC
    n_strategies: int = 7_140
    mp_start_method: str | None = None
    deterministic_batch_size: int = 30

    @property
    def games_per_shuffle(self) -> int:
        return self.n_strategies // self.n_players


@dataclass(frozen=True)
class ShuffleTask:
    """Stable coordinate identity for one complete tournament shuffle."""

    root_seed: int
    k: int
    shuffle_index: int
    shuffle_seed: int
    deterministic_batch_id: int


_EXPOSURE_COUNTERS = ("attempted_exposures", "completed_exposures", "safety_limit_exposures")
_GAME_TOTALS = ("games_attempted", "games_completed", "games_safety_limit")


class OutcomeCounter(Counter):
    """Win counter that also carries the per-strategy exposure counters and the three game totals
    (the pickled interface type of run_tournament.py:165-245; ``absorb`` merges, ``outcome_payload`` serialises)."""

    def __init__(self, *args: Any, **kwargs: Any) -> None:
        for name in _EXPOSURE_COUNTERS:
            setattr(self, name, Counter())
        for name in _GAME_TOTALS:
            setattr(self, name, 0)
        super().__init__(*args, **kwargs)

    def absorb(self, other: Counter) -> None:
        super().update(other)
        if isinstance(other, OutcomeCounter):
            for name in _EXPOSURE_COUNTERS:
                getattr(self, name).update(getattr(other, name))
            for name in _GAME_TOTALS:
                setattr(self, name, getattr(self, name) + getattr(other, name))
            return
        # a plain win counter: every win is one completed game, every winner one completed exposure
        completed = int(sum(other.values()))
        self.attempted_exposures.update(other)
        self.completed_exposures.update(other)
        self.games_attempted += completed
        self.games_completed += completed

    def outcome_payload(self) -> dict[str, Any]:
        return {**{name: getattr(self, name) for name in _GAME_TOTALS},
                **{name: dict(getattr(self, name)) for name in _EXPOSURE_COUNTERS}}

    def __reduce__(self):
        return (_restore_outcome_counter, (dict(self), self.outcome_payload()))


def _restore_outcome_counter(counts: dict, outcome_counts: dict) -> OutcomeCounter:
    restored = OutcomeCounter(counts)
    for name in _EXPOSURE_COUNTERS:
        getattr(restored, name).update(outcome_counts.get(name, {}))
    for name in _GAME_TOTALS:
        setattr(restored, name, int(outcome_counts.get(name, 0)))
    return restored


@dataclass
class WorkerState:
    strats: list[ThresholdStrategy]
    cfg: TournamentConfig
    game_profile: GameProfile | None = None
    table: np.ndarray = field(default=None, repr=False)  # type: ignore[assignment]


_STATE: WorkerState | None = None


def _init_worker(strategies: Sequence[ThresholdStrategy], config: TournamentConfig, game_profile: GameProfile | None = None,
                 progress_endpoint: Any = None) -> None:
    """Initialise per-process state (strategy table uploaded lazily with each launch)."""
    del progress_endpoint
    global _STATE
    if len(strategies) % config.n_players != 0:
        raise ValueError(f"n_players must divide {len(strategies):,}")
    resolved = prepare_public_helper_strategies(strategies)
    _STATE = WorkerState(resolved, config, game_profile, pack_strategies(resolved))


def _coerce_shuffle_task(task: ShuffleTask | int) -> ShuffleTask:
    if isinstance(task, ShuffleTask):
        return task
    k = _STATE.cfg.n_players if _STATE is not None else 0
    return ShuffleTask(root_seed=int(task), k=k, shuffle_index=0, shuffle_seed=int(task), deterministic_batch_id=0)


def tally_to_counters(tally: np.ndarray, ids: Sequence[int], k: int, *, dense: bool = False, counter_cls=None):
    """``int64[S][26]`` -> (OutcomeCounter, sums, square sums) with the reference's dict shapes.

    ``dense``: every strategy that was seated gets explicit zero entries, which is what the reference's totals look
    like when they are rebuilt from metric chunk files (``_reduce_metric_chunk_payloads``, run_tournament.py:905-922:
    each chunk row is added with ``+=``, zeros included); worker-side counters (``dense=False``) only hold what was
    incremented.  ``counter_cls``: the counter class to build (``reference_binding`` passes the reference's own
    ``OutcomeCounter`` so that its ``absorb`` / pickling see their own type)."""
    wins = (counter_cls or OutcomeCounter)()
    sums: Dict[str, Dict[int, float]] = {m: defaultdict(float) for m in METRIC_LABELS}
    sqs: Dict[str, Dict[int, float]] = {m: defaultdict(float) for m in METRIC_LABELS}
    # column-wise (a Python loop over the strategies cost 2 us per cell: 10 ms per 5 160-strategy tally, the binding route's ceiling);
    # every dict is filled in ascending table order, as the row loop did
    tally = np.asarray(tally)
    ids_arr = np.asarray(ids, dtype=np.int64)
    att, comp, safe, won = (tally[:, c] for c in (COL_ATTEMPTED, COL_COMPLETED, COL_SAFETY, COL_WINS))
    seated = (att != 0) if dense else np.zeros(len(ids_arr), dtype=bool)

    def fill(target, mask, column) -> None:
        idx = np.flatnonzero(mask)
        if len(idx):
            target.update(dict(zip(ids_arr[idx].tolist(), column[idx].tolist())))  # (a Counter: a mapping ADDS counts, zeros included)

    fill(wins.attempted_exposures, att != 0, att)
    fill(wins.completed_exposures, (comp != 0) | seated, comp)
    fill(wins.safety_limit_exposures, (safe != 0) | seated, safe)
    winners = np.flatnonzero((won != 0) | seated)
    if len(winners):
        wid = ids_arr[winners].tolist()
        Counter.update(wins, dict(zip(wid, won[winners].tolist())))
        block = tally[winners].astype(np.float64)
        for j, label in enumerate(METRIC_LABELS):
            sums[label].update(zip(wid, block[:, COL_SUMS + j].tolist()))
            sqs[label].update(zip(wid, block[:, COL_SQ_SUMS + j].tolist()))
    wins.games_attempted = int(tally[:, COL_ATTEMPTED].sum()) // k
    wins.games_completed = int(tally[:, COL_COMPLETED].sum()) // k
    wins.games_safety_limit = int(tally[:, COL_SAFETY].sum()) // k
    return wins, sums, sqs


def _limits(state: WorkerState) -> tuple[int, int, np.ndarray | None]:
    gp = state.game_profile
    if gp is None:
        return 10_000, 200, None
    return gp.default_target_score, gp.default_max_rounds, gp.tournament_overrides()


def _launch(tasks: Sequence[ShuffleTask], *, want_rows: bool, per_shuffle: bool = False):
    """Run the games of ``tasks`` (any shuffle indices of one (root, k) cell), grouped into contiguous launches."""
    state = _STATE
    if state is None:
        raise RuntimeError("tournament worker state is not initialised (_init_worker)")
    eng = get_engine()
    target, max_rounds, ov = _limits(state)
    out = []
    i = 0
    while i < len(tasks):
        j = i
        while (j + 1 < len(tasks) and tasks[j + 1].shuffle_index == tasks[j].shuffle_index + 1
               and tasks[j + 1].root_seed == tasks[i].root_seed and tasks[j + 1].k == tasks[i].k):
            j += 1
        first, last = tasks[i], tasks[j]
        try:
            res = eng.tournament(state.table, first.k, first.root_seed, first.shuffle_index, last.shuffle_index + 1,
                                 shuffles_per_batch=1 if per_shuffle else None, target_score=target, max_rounds=max_rounds,
                                 overrides=ov, want_rows=want_rows)
        except FarkleHipError as exc:
            if exc.code == FK_ERR_ROLL_LIMIT:
                raise RuntimeError(str(exc)) from exc
            raise
        out.append((tasks[i:j + 1], res))
        i = j + 1
    return out


def _shuffle_rows(task: ShuffleTask, rows: np.ndarray, ids: Sequence[int]) -> List[Dict[str, Any]]:
    k = task.k
    gps = len(rows)
    game_seeds = urandom.coordinate_seeds(urandom.RandomPurpose.TOURNAMENT_GAME, root_seed=task.root_seed, k=k,
                                          shuffle_index=task.shuffle_index, game_index=np.arange(gps, dtype=np.uint64),
                                          dtype=np.uint32)
    out = []
    for g in range(gps):
        out.append(row_to_dict(rows[g], k, ids, {
            "root_seed": task.root_seed, "k": k, "shuffle_index": task.shuffle_index, "game_index": g,
            "deterministic_batch_id": task.deterministic_batch_id, "shuffle_seed": task.shuffle_seed,
            "game_seed": int(game_seeds[g]), "rng_scheme_version": urandom.RNG_SCHEME_VERSION,
            "rng_purpose_namespace": int(urandom.RandomPurpose.TOURNAMENT_GAME)}))
    return out


def _play_one_shuffle(task: ShuffleTask | int, *, collect_rows: bool = False):
    """Play all games for one shuffle and aggregate the results: (wins, sums, sq_sums, rows)."""
    work = _coerce_shuffle_task(task)
    state = _STATE
    (_, res), = _launch([work], want_rows=collect_rows)
    ids = [int(s.strategy_id) for s in state.strats]  # type: ignore[union-attr]
    wins, sums, sqs = tally_to_counters(res["tally"][0], ids, work.k)
    rows = _shuffle_rows(work, res["rows"], ids) if collect_rows else []
    return wins, sums, sqs, rows


def _play_shuffle(task: ShuffleTask | int) -> Counter:
    wins, _, _, _ = _play_one_shuffle(task, collect_rows=False)
    return wins


def _run_chunk(shuffle_tasks: Sequence[ShuffleTask | int]) -> Counter:
    """Play a batch of shuffles and tally wins (one launch per contiguous shuffle range)."""
    tasks = [_coerce_shuffle_task(t) for t in shuffle_tasks]
    state = _STATE
    total = OutcomeCounter()
    ids = [int(s.strategy_id) for s in state.strats]  # type: ignore[union-attr]
    for group, res in _launch(tasks, want_rows=False):
        wins, _, _ = tally_to_counters(res["tally"].sum(axis=0), ids, group[0].k)
        total.absorb(wins)
    return total


def _run_chunk_metrics(shuffle_tasks: Sequence[ShuffleTask | int], *, collect_rows: bool = False,
                       row_dir: Path | None = None, manifest_path: Path | None = None, row_sidecar: Any = None):
    """Play shuffles and accumulate metrics: (wins, sums, square_sums).  With ``collect_rows`` and ``row_dir``
    one ``rows_{root}_{k}p_{shuffle:012d}.parquet`` shard per shuffle is written and recorded in ``manifest.jsonl``
    (run_tournament.py:530-558).  ``row_sidecar``: a sidecar template (``sidecars.simulation_output_sidecar``, the
    counterpart of the reference's ``_simulation_output_sidecar`` object, run_tournament.py:557): every shard then gets its
    ``<shard>.sidecar.json`` bound to the shard's bytes (contract version 2; see sidecars.py for what is not built)."""
    tasks = [_coerce_shuffle_task(t) for t in shuffle_tasks]
    state = _STATE
    ids = [int(s.strategy_id) for s in state.strats]  # type: ignore[union-attr]
    wins_total = OutcomeCounter()
    sums_total: Dict[str, Dict[int, float]] = {m: defaultdict(float) for m in METRIC_LABELS}
    sq_total: Dict[str, Dict[int, float]] = {m: defaultdict(float) for m in METRIC_LABELS}
    for group, res in _launch(tasks, want_rows=collect_rows):
        wins, sums, sqs = tally_to_counters(res["tally"].sum(axis=0), ids, group[0].k)
        wins_total.absorb(wins)
        for label in METRIC_LABELS:
            for key, v in sums[label].items():
                sums_total[label][key] += v
            for key, v in sqs[label].items():
                sq_total[label][key] += v
        if row_dir is not None and collect_rows:
            gps = len(res["rows"]) // len(group)
            for n, task in enumerate(group):
                shard = write_row_shard(Path(row_dir), manifest_path, task, res["rows"][n * gps:(n + 1) * gps], ids,
                                        game_profile_sha256=state.game_profile.sha256 if state.game_profile else None)  # :549-553
                if row_sidecar is not None:
                    from .sidecars import write_sidecar

                    write_sidecar(shard, row_sidecar)
    return wins_total, sums_total, sq_total


def append_manifest_lines(manifest: Path, lines: Sequence[str]) -> None:
    """Append already-encoded JSON lines to a manifest (one write call, flushed)."""
    if not lines:
        return
    manifest.parent.mkdir(parents=True, exist_ok=True)
    with open(manifest, "ab") as fh:
        fh.write("".join(line + "\n" for line in lines).encode("utf-8"))
        fh.flush()


def append_manifest_records(manifest: Path, records: Sequence[Mapping[str, Any]]) -> None:
    """Append JSON lines to a manifest (one write call, flushed)."""
    import json

    append_manifest_lines(manifest, [json.dumps(r, sort_keys=True) for r in records])


def write_row_shard(row_dir: Path, manifest_path: Path | None, task: ShuffleTask, rows: np.ndarray, ids: Sequence[int],
                    game_profile_sha256: str | None = None, *, append_manifest: bool = True, return_record: bool = False):
    """One shuffle's rows -> parquet shard + manifest record (run_tournament.py:530-558).  With ``append_manifest=False``
    the record is only returned (multi-rank runs gather the records and let rank 0 append them)."""
    import json
    import os

    import pyarrow.parquet as pq

    gps = len(rows)
    game_seeds = urandom.coordinate_seeds(urandom.RandomPurpose.TOURNAMENT_GAME, root_seed=task.root_seed, k=task.k,
                                          shuffle_index=task.shuffle_index, game_index=np.arange(gps, dtype=np.uint64),
                                          dtype=np.uint32)
    table = rows_to_table(rows, task.k, ids, root_seed=task.root_seed, shuffle_index=task.shuffle_index,
                          game_index=np.arange(gps), deterministic_batch_id=task.deterministic_batch_id,
                          shuffle_seed=task.shuffle_seed, game_seed=game_seeds.astype(np.int64),
                          rng_purpose_namespace=int(urandom.RandomPurpose.TOURNAMENT_GAME))
    row_dir.mkdir(parents=True, exist_ok=True)
    out = row_dir / f"rows_{task.root_seed}_{task.k}p_{task.shuffle_index:012d}.parquet"
    tmp = out.with_suffix(".parquet.tmp")
    pq.write_table(table, tmp)
    os.replace(tmp, out)
    manifest = Path(manifest_path) if manifest_path else row_dir / "manifest.jsonl"
    record = {"path": out.name, "rows": gps, "root_seed": task.root_seed, "n_players": task.k,
              "shuffle_index": task.shuffle_index, "shuffle_seed": task.shuffle_seed,
              "deterministic_batch_id": task.deterministic_batch_id, "rng_scheme_version": urandom.RNG_SCHEME_VERSION,
              "rng_purpose_namespace": int(urandom.RandomPurpose.TOURNAMENT_SHUFFLE),
              "outcome_schema_version": OUTCOME_SCHEMA_VERSION, "tournament_method_version": TOURNAMENT_METHOD_VERSION,
              "pid": os.getpid()}
    if game_profile_sha256 is not None:  # run_tournament.py:549-553
        record["game_profile_sha256"] = game_profile_sha256
    if append_manifest:
        append_manifest_records(manifest, [record])
    return (out, record) if return_record else out


SHARD_WRITER_OPTIONS = {"write_statistics": False, "use_dictionary": False}  # how a row shard is encoded (_write_shard_group)


def _shard_record(name: str, gps: int, root_seed: int, k: int, shuffle_index: int, shuffle_seed: int, batch_id: int, pid: int,
                  game_profile_sha256: str | None) -> dict:
    record = {"path": name, "rows": gps, "root_seed": root_seed, "n_players": k, "shuffle_index": shuffle_index,
              "shuffle_seed": shuffle_seed, "deterministic_batch_id": batch_id, "rng_scheme_version": urandom.RNG_SCHEME_VERSION,
              "rng_purpose_namespace": int(urandom.RandomPurpose.TOURNAMENT_SHUFFLE),
              "outcome_schema_version": OUTCOME_SCHEMA_VERSION, "tournament_method_version": TOURNAMENT_METHOD_VERSION, "pid": pid}
    if game_profile_sha256 is not None:  # run_tournament.py:549-553
        record["game_profile_sha256"] = game_profile_sha256
    return record


def _write_shard_group(row_dir: str, k: int, ids: np.ndarray, gps: int, root_seed: int, rows: np.ndarray, shuffle_index: np.ndarray,
                       shuffle_seed: np.ndarray, batch_id: np.ndarray, game_seeds: np.ndarray | None,
                       game_profile_sha256: str | None, sidecar: Mapping[str, Any] | None = None, as_lines: bool = False,
                       atomic: bool = True) -> list:
    """Shards of a run of shuffles (rows = their games, shuffle-major): one vectorised Arrow conversion, then one parquet
    file per shuffle — a zero-copy slice of that table (run_tournament.py:530-558).  Runs in a writer process or inline.
    Returns the manifest records — or, with ``as_lines``, ``(shuffle_index, JSON line, bytes, sha256)`` tuples, so that the encoding of one
    line per shuffle happens in the writer processes too."""
    import hashlib
    import json
    import os

    import pyarrow as pa
    import pyarrow.parquet as pq

    n_sh = len(shuffle_index)
    sh = np.repeat(shuffle_index.astype(np.uint64), gps)
    gi = np.tile(np.arange(gps, dtype=np.uint64), n_sh)
    if game_seeds is None:
        game_seeds = urandom.coordinate_seeds(urandom.RandomPurpose.TOURNAMENT_GAME, root_seed=root_seed, k=k, shuffle_index=sh,
                                              game_index=gi, dtype=np.uint32)
    table = rows_to_table(rows, k, ids, root_seed=root_seed, shuffle_index=sh.astype(np.int64), game_index=gi.astype(np.int32),
                          deterministic_batch_id=np.repeat(batch_id.astype(np.int32), gps),
                          shuffle_seed=np.repeat(shuffle_seed.astype(np.int64), gps),
                          game_seed=np.asarray(game_seeds).reshape(-1).astype(np.int64),
                          rng_purpose_namespace=int(urandom.RandomPurpose.TOURNAMENT_GAME))
    pid = os.getpid()
    records = []
    for i in range(n_sh):
        name = f"rows_{root_seed}_{k}p_{int(shuffle_index[i]):012d}.parquet"
        out = os.path.join(row_dir, name)
        # a 32-row file: column statistics and dictionary pages are a third of its encoding time and nobody prunes on them
        # encoded in memory, so that the writer can hand the shard's byte identity (size, SHA-256) back with its manifest line: the
        # completion stamp (stage_completion.py) then does not read 51 200 files again
        sink = pa.BufferOutputStream()
        pq.write_table(table.slice(i * gps, gps), sink, **SHARD_WRITER_OPTIONS)
        blob = sink.getvalue()
        fd = os.open(out + ".tmp" if atomic else out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
            os.write(fd, blob)
        finally:
            os.close(fd)
        if atomic:
            os.replace(out + ".tmp", out)
        # (not atomic: one directory operation per shard instead of two, see write_row_shards)
        record = _shard_record(name, gps, root_seed, k, int(shuffle_index[i]), int(shuffle_seed[i]), int(batch_id[i]), pid,
                               game_profile_sha256)
        digest = None
        if sidecar is not None and "body" in sidecar:
            # artifact-contract version 3 (contract_v3.SimulationContract.shard_template): the sidecar is constant text around the shard's
            # size, SHA-256 and name; its manifest record carries the shard's byte / sidecar / schema identity (what a sealed manifest's
            # root is computed from, simulation/runner.py:570-590 of the reference) and no process id
            from .contract_v3 import arrow_schema_identity, fill_shard_template

            if i == 0 and arrow_schema_identity(pq.read_schema(pa.BufferReader(blob)))["fingerprint_sha256"] != sidecar["schema_fingerprint_sha256"]:
                raise ValueError(f"row shard {name}: the stored Arrow schema is not the one its sidecar template was built for")
            digest = hashlib.sha256(blob).hexdigest()
            text, side_sha = fill_shard_template(sidecar, name, blob.size, digest)
            fd = os.open(out + ".sidecar.json", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            try:
                os.write(fd, text)
            finally:
                os.close(fd)
            del record["pid"]
            record.update(byte_length=blob.size, data_sha256=digest, sidecar_sha256=side_sha,
                          schema_fingerprint_sha256=sidecar["schema_fingerprint_sha256"])
        elif sidecar is not None:
            from .sidecars import write_sidecar

            write_sidecar(out, sidecar)
        if as_lines:  # (shuffle, manifest line, shard size, shard SHA-256; without a sidecar the stamp needs nothing else of the file)
            records.append((int(shuffle_index[i]), json.dumps(record, sort_keys=True, separators=(",", ":")) if digest else json.dumps(record, sort_keys=True),
                            blob.size, digest or hashlib.sha256(blob).hexdigest()))
        else:
            records.append(record)
    return records


def _shard_worker_init() -> None:
    import pyarrow as pa

    pa.set_cpu_count(1)  # 32-row files: Arrow's per-column pool only adds contention (3.1 ms per shard with it, 0.9 ms without)
    pa.set_io_thread_count(1)


_SHARD_POOL = None


def _shard_pool(workers: int):
    """The run's row-shard writer processes (shard_writer.py), started on first use."""
    global _SHARD_POOL
    if _SHARD_POOL is None or _SHARD_POOL[1] != workers:
        import atexit

        from .shard_writer import ShardWriters

        if _SHARD_POOL is not None:
            _SHARD_POOL[0].close()
        pool = ShardWriters(workers)
        atexit.register(pool.close)
        _SHARD_POOL = (pool, workers)
    return _SHARD_POOL[0]


@dataclass(frozen=True)
class ShuffleRange:
    """The ShuffleTask identities of many shuffles of one (root, k) cell as arrays (``write_row_shards`` takes either)."""

    root_seed: int
    k: int
    shuffle_index: np.ndarray
    shuffle_seed: np.ndarray
    deterministic_batch_id: np.ndarray

    def __len__(self) -> int:
        return len(self.shuffle_index)


def write_row_shards(row_dir: Path, tasks: "Sequence[ShuffleTask] | ShuffleRange", rows: np.ndarray, ids: Sequence[int],
                     game_profile_sha256: str | None = None, *, threads: int = 1, group: int = 64,
                     game_seeds: np.ndarray | None = None, sidecar: Mapping[str, Any] | None = None, as_lines: bool = False,
                     atomic: bool = True) -> list:
    """Row shards of many shuffles of one (root, k) cell: the same files and manifest records as ``write_row_shard`` per
    shuffle (run_tournament.py:530-558).  The shuffles are cut into runs of ``group``; each run is converted to Arrow once and
    written shard by shard, by ``threads`` writer processes (``threads`` <= 1: inline).  ``game_seeds``: the ns-102
    fingerprints ``[n_shuffles][gps]`` when the caller has them (``Engine.game_seeds``), else they are hashed here.
    Returns the manifest records in task order (``as_lines``: ``(shuffle_index, JSON line, shard bytes, shard sha256)`` tuples); the caller appends them."""
    if len(tasks) == 0:
        return []
    if isinstance(tasks, ShuffleRange):
        k, root = tasks.k, tasks.root_seed
        sh = np.asarray(tasks.shuffle_index, dtype=np.int64)
        seeds = np.asarray(tasks.shuffle_seed, dtype=np.int64)
        batch = np.asarray(tasks.deterministic_batch_id, dtype=np.int32)
    else:
        k, root = tasks[0].k, tasks[0].root_seed
        sh = np.array([t.shuffle_index for t in tasks], dtype=np.int64)
        seeds = np.array([t.shuffle_seed for t in tasks], dtype=np.int64)
        batch = np.array([t.deterministic_batch_id for t in tasks], dtype=np.int32)
    gps = len(rows) // len(tasks)
    row_dir.mkdir(parents=True, exist_ok=True)
    ids = np.asarray(ids, dtype=np.int32)
    gs = None if game_seeds is None else np.asarray(game_seeds).reshape(len(tasks), gps)
    jobs = []
    for g0 in range(0, len(tasks), group):
        g1 = min(g0 + group, len(tasks))
        jobs.append((str(row_dir), k, ids, gps, root, rows[g0 * gps:g1 * gps], sh[g0:g1], seeds[g0:g1], batch[g0:g1],
                     None if gs is None else gs[g0:g1], game_profile_sha256, sidecar, as_lines, atomic))
    if threads > 1 and len(jobs) > 1:
        parts = _shard_pool(threads).map(jobs)
    else:
        parts = [_write_shard_group(*job) for job in jobs]
    return [r for part in parts for r in part]


def write_row_shards_from_columns(row_dir: Path, tasks: ShuffleRange, columns: np.ndarray, game_seeds: np.ndarray,
                                  game_profile_sha256: str | None = None, *, threads: int = 1, sidecar: Mapping[str, Any] | None = None,
                                  atomic: bool = True, deferred_lines: bool = False, deferred_write: bool = False):
    """The row shards of ``tasks`` from per-shuffle COLUMN IMAGES (``Engine.tournament_columns``): the files are framed by the library's
    own Parquet writer on ``threads`` host threads (``fk_write_row_shards``, csrc/fk_shard_writer.h) — the same tables and Arrow schema as
    ``write_row_shards`` writes through Arrow (run_tournament.py:530-558), without the AoS -> Arrow conversion and Arrow's per-column
    encoder set-up.  ``sidecar``: a contract-v3 shard template (the library writes each shard's sidecar too).  Returns
    ``(shuffle_index, manifest line, shard bytes, shard sha256)`` tuples in task order, as ``write_row_shards(as_lines=True)`` does —
    ``deferred_lines``: a zero-argument callable that builds them (the files are written on return; a caller that runs this on a writer
    thread keeps the per-shard Python off it: ~4 ms per 1 000 shards during which its host threads would idle); ``deferred_write``: a
    zero-argument callable that WRITES (and returns what this function would have) — the job is checked and laid out here, ``columns``
    may be filled in between."""
    import os

    from .backend import prepare_row_shards_native

    n = len(tasks)
    if n == 0:
        empty = (lambda: []) if deferred_lines else []
        return (lambda: empty) if deferred_write else empty
    row_dir.mkdir(parents=True, exist_ok=True)
    k, root = tasks.k, tasks.root_seed
    gps = np.asarray(game_seeds).size // n
    v3 = sidecar is not None and "body" in sidecar
    job = prepare_row_shards_native(row_dir, k, root, columns, tasks.shuffle_index, tasks.shuffle_seed, tasks.deterministic_batch_id, game_seeds,
                                    int(urandom.RandomPurpose.TOURNAMENT_GAME), threads=threads, atomic=atomic, sidecar=sidecar if v3 else None)

    def write():
        res = job()

        def lines() -> list:
            return _shard_manifest_lines(res, tasks, n, gps, root, k, game_profile_sha256, sidecar, v3, row_dir)

        return lines if deferred_lines else lines()

    return write if deferred_write else write()


def _shard_manifest_lines(res, tasks: ShuffleRange, n: int, gps: int, root: int, k: int, game_profile_sha256: str | None,
                          sidecar: Mapping[str, Any] | None, v3: bool, row_dir: Path) -> list:
    """The per-shard records of ``write_row_shards_from_columns`` from the native writer's result arrays."""
    import os

    sizes = res["byte_length"].tolist()
    digests = [d.tobytes().hex() for d in res["sha256"]]
    sh, seeds, batch = (np.asarray(a).tolist() for a in (tasks.shuffle_index, tasks.shuffle_seed, tasks.deterministic_batch_id))
    # one manifest line per shard: the record of _shard_record, JSON text assembled around the values that change
    probe = _shard_record("@path@", gps, root, k, -1, -2, -3, os.getpid(), game_profile_sha256)
    if v3:
        del probe["pid"]
        probe.update(byte_length=-4, data_sha256="@data@", sidecar_sha256="@side@", schema_fingerprint_sha256=sidecar["schema_fingerprint_sha256"])
        text = json.dumps(probe, sort_keys=True, separators=(",", ":"))
        side = [d.tobytes().hex() for d in res["sidecar_sha256"]]
    else:
        text = json.dumps(probe, sort_keys=True)
        side = [None] * n
    marks = {'"@path@"': None, "-1": None, "-2": None, "-3": None}
    if v3:
        marks.update({"-4": None, '"@data@"': None, '"@side@"': None})
    pieces, order, rest = [], [], text
    found = sorted((text.index(m), m) for m in marks)
    if any(text.count(m) != 1 for m in marks):
        raise ValueError("manifest record template: a marker value occurs twice")
    cursor = 0
    for pos, m in found:
        pieces.append(text[cursor:pos])
        order.append(m)
        cursor = pos + len(m)
    pieces.append(text[cursor:])
    # column-wise: one list of texts per marker, one join per line (a dict per shard cost 4 us: 140 ms over a production sweep's 34 400 shards)
    columns = {'"@path@"': [f'"rows_{root}_{k}p_{v:012d}.parquet"' for v in sh], "-1": list(map(str, sh)), "-2": list(map(str, seeds)),
               "-3": list(map(str, batch)), "-4": list(map(str, sizes)), '"@data@"': [f'"{d}"' for d in digests],
               '"@side@"': [f'"{d}"' for d in side]}
    from itertools import repeat

    parts: list = []
    for piece, m in zip(pieces, order):
        parts.extend((repeat(piece, n), columns[m]))
    parts.append(repeat(pieces[-1], n))
    out = list(zip(sh, map("".join, zip(*parts)), sizes, digests))
    if not v3 and sidecar is not None:  # the structural (contract-2) sidecars are written here, after the files
        from .sidecars import write_sidecar

        for i in range(n):
            write_sidecar(row_dir / f"rows_{root}_{k}p_{sh[i]:012d}.parquet", sidecar)
    return out


def shuffle_tasks(root_seed: int, k: int, shuffle_begin: int, shuffle_end: int, deterministic_batch_size: int) -> list[ShuffleTask]:
    """Stable ShuffleTask identities of a shuffle range (shuffle_seed = ns-100 uint32 fingerprint)."""
    idx = np.arange(shuffle_begin, shuffle_end, dtype=np.uint64)
    seeds = urandom.coordinate_seeds(urandom.RandomPurpose.TOURNAMENT_SHUFFLE, root_seed=root_seed, k=k, shuffle_index=idx,
                                     dtype=np.uint32) if len(idx) else []
    return [ShuffleTask(root_seed, k, int(i), int(s), int(i) // deterministic_batch_size) for i, s in zip(idx, seeds)]


def _measure_throughput(sample_strategies: Sequence[ThresholdStrategy], sample_games: int = 2_000, seed: int = 0,
                        game_profile: GameProfile | None = None) -> float:
    """Quick benchmark returning games processed per second (run_tournament.py:593-619)."""
    import time

    from .simulation import PlayerRngCoordinates, play_coordinate_games

    k = len(sample_strategies)
    coords = [PlayerRngCoordinates(purpose=urandom.RandomPurpose.TOURNAMENT_PLAYER, root_seed=seed, k=k, game_index=i)
              for i in range(sample_games)]
    target = game_profile.default_target_score if game_profile else 10_000
    t0 = time.perf_counter()
    play_coordinate_games(coords, prepare_public_helper_strategies(sample_strategies),
                          np.tile(np.arange(k, dtype=np.int32), (sample_games, 1)), target_score=target)
    return sample_games / max(time.perf_counter() - t0, 1e-9)
