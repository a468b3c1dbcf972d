"""The engine behind the reference's OWN seams — the drop-in a reference maintainer installs.

The reference (Isaac-McPadden/Farkle_II) is pure Python; its seams for this path are in-process callables:

* tournament: ``farkle.simulation.run_tournament`` looks ``_play_one_shuffle`` / ``_play_shuffle`` / ``_run_chunk`` /
  ``_run_chunk_metrics`` up in its module globals at call time (``run_tournament.py:301-585``; the reference's own tests
  swap ``_play_shuffle`` there, ``tests/unit/simulation/test_run_tournament.py:45-60``).  :class:`TournamentBinding`
  replaces those four names; everything above them — ``run_tournament.run_tournament`` (:1050), ``runner.run_single_n``,
  checkpoints, row shards (``run_streaming_shard``), metric chunks, sidecars of whatever contract version the run is
  configured for, ``simulation.done.json`` — stays the reference's code, fed with objects of the reference's own types.
* H2H: ``execute_h2h_schedule(cfg, block_runner=...)`` (``analysis/h2h_schedule.py:1597``) takes a ``BlockRunner``
  (:1521); :func:`block_runner` / :func:`prefetching_block_runner` build one.  Every result goes through the reference's
  ``_normalize_runner_result`` (:1422) and is published by its ``_write_block`` (:1471).

This module never imports the reference: the caller hands over the module object it has already imported.  It has no
CPU path either — without an engine (``farkle_ii_amd.engine.get_engine``, i.e. a MI355X) every call raises.

    from farkle.simulation import run_tournament as rt, runner
    from farkle_ii_amd.reference_binding import TournamentBinding
    with TournamentBinding(rt):
        runner.run_single_n(cfg, k)          # sim.n_jobs = 1: the GPU is the worker pool

The binding lives in the CALLING process.  ``run_tournament`` hands its chunks to ``parallel.process_map(..., n_jobs=resolved_n_jobs)``
(``run_tournament.py:1576-1586``): with more than one job, spawned workers would re-import the module and silently play on the
reference's CPU path, forked ones would inherit a parent that may hold a HIP context.  Both are refused LOUDLY (the reference
enforces the same for custom H2H runners, ``analysis/h2h_schedule.py:1894-1895``): while the binding is installed,
``rt.parallel`` is a guard whose ``process_map`` raises :class:`BindingWorkerPoolError` for ``n_jobs != 1``, and every patched
callable raises it when called from another process than the one that installed it.
"""
from __future__ import annotations

import os
from collections import defaultdict
from typing import Any, Callable, Dict, Mapping, Sequence

import numpy as np

from . import h2h as _h2h
from .backend import FarkleHipError, FK_ERR_ROLL_LIMIT
from .engine import get_engine
from .game_profile import GameProfile, H2HMaxRoundsOverride, TournamentMaxRoundsOverride
from .strategies import STRATEGY_DTYPE
from .tournament import METRIC_LABELS, _shuffle_rows, tally_to_counters


class BindingWorkerPoolError(RuntimeError):
    """The reference tried to play on a process pool while the engine binding was installed (sim.n_jobs must be 1)."""


class _ParallelGuard:
    """Stands in for ``farkle.utils.parallel`` inside ``run_tournament``'s namespace while the binding is installed: everything is
    the real module's, except that ``process_map`` refuses a worker pool (the patched callables exist in THIS process only)."""

    def __init__(self, real: Any):
        self._real = real

    def __getattr__(self, name: str) -> Any:
        return getattr(self._real, name)

    def process_map(self, *args, **kwargs):  # (utils/parallel.py:843: process_map(fn, items, *, n_jobs=None, initializer=None, ...))
        n_jobs = kwargs.get("n_jobs")
        normalize = getattr(self._real, "normalize_n_jobs", None)
        resolved = normalize(n_jobs, default=1) if normalize is not None else (1 if n_jobs is None else int(n_jobs))
        if int(resolved) != 1:
            raise BindingWorkerPoolError(
                f"TournamentBinding is installed and the run asks for n_jobs = {n_jobs!r} (resolved {resolved}): worker processes would not "
                "see the binding and would play on the reference's CPU path.  Set sim.n_jobs = 1 — the GPU is the worker pool.")
        return self._real.process_map(*args, **kwargs)


def coerce_game_profile(profile: Any) -> GameProfile | None:
    """Any object with the field names of ``simulation/game_profile.py:24-191`` (the reference's ``GameProfile`` is one)
    -> this package's :class:`GameProfile` (same canonical payload, so the same ``sha256``)."""
    if profile is None or isinstance(profile, GameProfile):
        return profile
    return GameProfile(
        default_target_score=int(profile.default_target_score), default_max_rounds=int(profile.default_max_rounds),
        tournament_max_rounds_overrides=tuple(
            TournamentMaxRoundsOverride(int(o.root_seed), int(o.k), int(o.shuffle_index), int(o.game_index), int(o.max_rounds))
            for o in profile.tournament_max_rounds_overrides),
        h2h_max_rounds_overrides=tuple(
            H2HMaxRoundsOverride(int(o.root_seed), int(o.pair_id), int(o.order), int(o.attempt_index), int(o.max_rounds))
            for o in profile.h2h_max_rounds_overrides))


def pack_reference_strategies(strategies: Sequence[Any]) -> np.ndarray:
    """``ThresholdStrategy`` objects of the reference (``simulation/strategies.py:165-194``) -> ``fk_strategy[S]``.
    ``favor_dice_or_score`` is the reference's enum: its member name is compared, not its identity."""
    out = np.zeros(len(strategies), dtype=STRATEGY_DTYPE)
    for i, s in enumerate(strategies):
        favor = s.favor_dice_or_score
        favor_score = str(getattr(favor, "name", favor)).upper().endswith("SCORE")
        sid = s.strategy_id if getattr(s, "strategy_id", None) is not None else i
        out[i] = (int(s.score_threshold), int(s.dice_threshold), int(s.smart_five), int(s.smart_one), int(s.consider_score),
                  int(s.consider_dice), int(s.require_both), int(s.auto_hot_dice), int(s.run_up_score), int(favor_score), int(sid))
    return out


class TournamentBinding:
    """Installs the engine behind ``farkle.simulation.run_tournament``'s four shuffle / chunk callables.

    ``rt`` is the imported reference module.  A chunk (the contiguous shuffles of one deterministic batch,
    ``run_tournament.py:974``) is ONE ``fk_tournament_run`` launch with a tally per shuffle; the reference's own
    ``_run_chunk`` / ``_run_chunk_metrics`` bodies then run unchanged — their per-shuffle ``_play_shuffle`` /
    ``_play_one_shuffle`` calls are served from that launch — so the shard writer, the manifest records and the sidecars
    are the reference's.  A shuffle asked for outside a chunk is played alone.  The returned objects are the reference's
    types: its ``OutcomeCounter`` (``rt.OutcomeCounter``), ``defaultdict(float)`` sums keyed by strategy id, row dicts in
    the shape ``_play_game`` returns (``simulation.py:576-655``)."""

    _NAMES = ("_play_one_shuffle", "_play_shuffle", "_run_chunk", "_run_chunk_metrics")

    def __init__(self, rt: Any, engine: Any = None):
        self.rt = rt
        self._engine = engine
        self._orig: dict[str, Callable] = {}
        self._served: dict[tuple, tuple] = {}   # (root, k, shuffle, rows?) -> (wins, sums, sqs, rows) of a chunk launch
        self._table_of: tuple[tuple[int, int], np.ndarray, list[int]] | None = None
        self._pid: int | None = None
        self._real_parallel: Any = None
        self.launches = 0

    # ---- install / uninstall -------------------------------------------------------------------------------------
    def install(self) -> "TournamentBinding":
        if self._orig:
            return self
        for name in self._NAMES:
            self._orig[name] = getattr(self.rt, name)
            setattr(self.rt, name, getattr(self, name))
        self._pid = os.getpid()
        real = getattr(self.rt, "parallel", None)  # run_tournament.py calls parallel.process_map (:1576)
        if real is not None and not isinstance(real, _ParallelGuard):
            self._real_parallel = real
            self.rt.parallel = _ParallelGuard(real)
        return self

    def uninstall(self) -> None:
        for name, fn in self._orig.items():
            setattr(self.rt, name, fn)
        if self._real_parallel is not None:
            self.rt.parallel = self._real_parallel
            self._real_parallel = None
        self._orig.clear()
        self._served.clear()

    def _check_process(self) -> None:
        if self._pid is not None and os.getpid() != self._pid:
            raise BindingWorkerPoolError(
                f"TournamentBinding was installed in process {self._pid} and is being called in process {os.getpid()}: a forked worker "
                "inherited it.  Set sim.n_jobs = 1 — the GPU is the worker pool.")

    __enter__ = install

    def __exit__(self, *exc) -> None:
        self.uninstall()

    # ---- state of the reference's worker (run_tournament.py:253-277) ----------------------------------------------
    def _state(self):
        state = self.rt._STATE
        if state is None:
            raise RuntimeError("farkle.simulation.run_tournament._STATE is not initialised (_init_worker)")
        key = (id(state.strats), len(state.strats))
        if self._table_of is None or self._table_of[0] != key:
            table = pack_reference_strategies(state.strats)
            self._table_of = (key, table, [int(v) for v in table["strategy_id"]])
        return state, self._table_of[1], self._table_of[2]

    def _launch(self, tasks: Sequence[Any], want_rows: bool) -> None:
        """One launch per contiguous shuffle range of ``tasks``; the per-shuffle results go to the serving table."""
        state, table, ids = self._state()
        profile = coerce_game_profile(state.game_profile)
        target, max_rounds, ov = 10_000, 200, None
        if profile is not None:
            target, max_rounds, ov = profile.default_target_score, profile.default_max_rounds, profile.tournament_overrides()
        eng = self._engine or get_engine()
        i = 0
        while i < len(tasks):
            j = i
            while (j + 1 < len(tasks) and tasks[j + 1].shuffle_index == tasks[j].shuffle_index + 1
                   and tasks[j + 1].root_seed == tasks[i].root_seed and tasks[j + 1].k == tasks[i].k):
                j += 1
            first, last = tasks[i], tasks[j]
            try:
                res = eng.tournament(table, int(first.k), int(first.root_seed), int(first.shuffle_index), int(last.shuffle_index) + 1,
                                     shuffles_per_batch=1, target_score=target, max_rounds=max_rounds, overrides=ov,
                                     want_rows=want_rows)
            except FarkleHipError as exc:
                if exc.code == FK_ERR_ROLL_LIMIT:  # engine.py:242-243 raises RuntimeError for a 1000-roll turn
                    raise RuntimeError(str(exc)) from exc
                raise
            self.launches += 1
            gps = len(table) // int(first.k)
            for n, task in enumerate(tasks[i:j + 1]):
                wins, sums, sqs = tally_to_counters(res["tally"][n], ids, int(task.k), counter_cls=self.rt.OutcomeCounter)
                rows = _shuffle_rows(task, res["rows"][n * gps:(n + 1) * gps], ids) if want_rows else []
                self._served[(int(task.root_seed), int(task.k), int(task.shuffle_index), bool(want_rows))] = (wins, sums, sqs, rows)
            i = j + 1

    # ---- the four callables (same names, arguments and return shapes as the reference's) ---------------------------
    def _play_one_shuffle(self, task, *, collect_rows: bool = False):
        self._check_process()
        work = self.rt._coerce_shuffle_task(task)
        key = (int(work.root_seed), int(work.k), int(work.shuffle_index), bool(collect_rows))
        if key not in self._served:
            self._launch([work], bool(collect_rows))
        return self._served.pop(key)

    def _play_shuffle(self, task):
        wins, _, _, _ = self._play_one_shuffle(task, collect_rows=False)
        return wins

    def _chunk_tally(self, tasks: Sequence[Any]) -> tuple[np.ndarray, list[int], int]:
        """The chunk as ONE tally: a launch per contiguous shuffle range of ``tasks`` (a deterministic batch is one range) with a single
        batch each, summed.  -> (int64 [S][26], strategy ids, games attempted)."""
        state, table, ids = self._state()
        profile = coerce_game_profile(state.game_profile)
        target, max_rounds, ov = 10_000, 200, None
        if profile is not None:
            target, max_rounds, ov = profile.default_target_score, profile.default_max_rounds, profile.tournament_overrides()
        eng = self._engine or get_engine()
        total = np.zeros((len(table), 26), dtype=np.int64)
        games = 0
        i = 0
        while i < len(tasks):
            j = i
            while (j + 1 < len(tasks) and tasks[j + 1].shuffle_index == tasks[j].shuffle_index + 1
                   and tasks[j + 1].root_seed == tasks[i].root_seed and tasks[j + 1].k == tasks[i].k):
                j += 1
            first, last = tasks[i], tasks[j]
            try:
                res = eng.tournament(table, int(first.k), int(first.root_seed), int(first.shuffle_index), int(last.shuffle_index) + 1,
                                     shuffles_per_batch=j - i + 1, target_score=target, max_rounds=max_rounds, overrides=ov)
            except FarkleHipError as exc:
                if exc.code == FK_ERR_ROLL_LIMIT:  # engine.py:242-243 raises RuntimeError for a 1000-roll turn
                    raise RuntimeError(str(exc)) from exc
                raise
            self.launches += 1
            total += res["tally"][0]
            games += (j - i + 1) * (len(table) // int(first.k))
            i = j + 1
        return total, ids, games

    def _report_chunk(self, tasks: Sequence[Any], games: int) -> None:
        """The progress the reference reports shuffle by shuffle (``report_worker_progress``, run_tournament.py:443-455, 571-583), for the
        whole chunk in one event: the same counters, summed."""
        report = getattr(self.rt, "report_worker_progress", None)
        if report is None or not tasks:
            return
        first, last = tasks[0], tasks[-1]
        report("simulation_shuffle_complete", event_id=f"simulation:{first.root_seed}:{first.k}:{first.shuffle_index}-{last.shuffle_index}",
               counters={"worker_completed_shuffles": len(tasks), "worker_completed_games": int(games)})

    def _run_chunk(self, shuffle_tasks):
        """``_run_chunk`` (run_tournament.py:403-457) served per CHUNK: one launch, one tally, one ``OutcomeCounter`` — the reference's body
        would absorb one counter per shuffle into the same sums (0.17 - 5 ms of Python per shuffle, round-5 measurement)."""
        self._check_process()
        tasks = [self.rt._coerce_shuffle_task(t) for t in shuffle_tasks]
        if not tasks:
            return self.rt.OutcomeCounter()
        tally, ids, games = self._chunk_tally(tasks)
        wins, _, _ = tally_to_counters(tally, ids, int(tasks[0].k), counter_cls=self.rt.OutcomeCounter)
        self._report_chunk(tasks, games)
        return wins

    def _run_chunk_metrics(self, shuffle_tasks, *, collect_rows: bool = False, row_dir=None, **kwargs):
        """``_run_chunk_metrics`` (run_tournament.py:473-585).  Without row shards to write it returns per-chunk sums and writes nothing:
        served from ONE tally.  With ``collect_rows`` and a ``row_dir`` the reference's own body runs — its shard writer, manifest records
        and sidecars — with every shuffle's rows served from one launch."""
        self._check_process()
        tasks = [self.rt._coerce_shuffle_task(t) for t in shuffle_tasks]
        if collect_rows and row_dir is not None:
            self._launch(tasks, True)
            try:
                return self._orig["_run_chunk_metrics"](tasks, collect_rows=collect_rows, row_dir=row_dir, **kwargs)
            finally:
                self._served.clear()
        if not tasks:
            return self._orig["_run_chunk_metrics"](tasks, collect_rows=collect_rows, row_dir=row_dir, **kwargs)
        tally, ids, games = self._chunk_tally(tasks)
        out = chunk_counters(self.rt, tally, ids, int(tasks[0].k))
        self._report_chunk(tasks, games)
        return out


def chunk_counters(rt: Any, tally: np.ndarray, strategy_ids: Sequence[int], k: int):
    """``int64[S][26]`` of one chunk -> the ``(wins, sums, square_sums)`` triple ``rt._run_chunk_metrics`` returns, built
    from the reference's own ``OutcomeCounter`` class (for callers that bind ``fk_tournament_run`` per chunk themselves)."""
    wins, sums, sqs = tally_to_counters(tally, strategy_ids, k, counter_cls=rt.OutcomeCounter)
    full: Dict[str, Dict[int, float]] = {m: defaultdict(float, sums[m]) for m in METRIC_LABELS}
    return wins, full, {m: defaultdict(float, sqs[m]) for m in METRIC_LABELS}


# ---- H2H: BlockRunner objects for execute_h2h_schedule(cfg, block_runner=...) -------------------------------------------

def block_runner(oracle_game_profile: Any = None, engine: Any = None) -> Callable[[dict, Any, int], dict]:
    """One block per call (``BlockRunner``, h2h_schedule.py:1521): ``runner(block, strategy_manifest_path, attempt_count)``."""
    return _h2h.gpu_block_runner(coerce_game_profile(oracle_game_profile), engine=engine)


def prefetching_block_runner(schedule_blocks: Sequence[Mapping[str, Any]], oracle_game_profile: Any = None, engine: Any = None,
                             chunk_games: int | None = None) -> "_h2h.PrefetchingBlockRunner":
    """The whole schedule in shared launches, served to the reference's serial loop (h2h_schedule.py:2038-2093) block by
    block.  ``schedule_blocks``: the records of the block manifest (``cfg.h2h_block_manifest_path()``) as dicts."""
    return _h2h.PrefetchingBlockRunner([dict(b) for b in schedule_blocks], coerce_game_profile(oracle_game_profile), engine,
                                       chunk_games=chunk_games)


__all__ = ["TournamentBinding", "BindingWorkerPoolError", "block_runner", "prefetching_block_runner", "chunk_counters", "coerce_game_profile",
           "pack_reference_strategies"]
