"""Unconditional all-player batch metrics without rows.

The reference derives its per-(root, k, deterministic batch, strategy) all-player metrics from curated game rows
(``src/farkle/analysis/all_player_metrics.py``: schema ``all_player_batch_schema`` :101-119, per-seat columns
``_seat_exposure_columns`` :257-340, row ``_finish_row`` :371-425).  Every column of that table except two pairs is a sum
of integers over the seat exposures of a batch; the engine produces those sums on the device
(``fk_tournament_run_stats``, 31 int64 accumulators per batch and strategy, ``backend.SEAT_STAT_NAMES``) and this module
lays them out in the reference's column order and types, so that the metrics stage can read the table directly instead of
re-deriving it from rows.

The four float64 sums that are NOT sums of integers — ``raw_turn_return_game_weighted_exact_{sum,square_sum}`` (score / n_turns per
exposure) and ``raw_turn_return_round_proxy_{sum,square_sum}`` (score / n_rounds) — come from the device too since round 5
(``fk_tournament_run_all_player``): the reference adds them with an unbuffered ``np.add.at`` in source-row order (:174-177), i.e. ONE
sequential float64 sum per strategy over its exposures in (shuffle, game, seat) order; a strategy sits once per shuffle, so one
thread per (batch, strategy) runs exactly that sequence.  CONTRACT of those four columns and of the four fields derived from them
(``turn_return_game_weighted_exact``, ``turn_return_round_proxy``, ``round_proxy_gap``, ``round_proxy_relative_gap``): bit-identical to
the reference's ``_iter_batch_tables`` whenever its curated rows are in (shuffle, game) order — what the fixture
``tests/golden/all_player_vectors.json`` (the reference's own accumulation over rows it simulated) pins; rows in another order
re-associate the sums, and then agreement is to rounding (a few ulp of the sum), as it is between two runs of the reference over
differently ordered row files.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np

from .backend import SEAT_STAT_NAMES

BEHAVIOR_SUFFIXES = ("rank", "loss_margin", "rolls", "farkles", "highest_turn", "hot_dice", "smart_five_uses",
                     "n_smart_five_dice", "smart_one_uses", "n_smart_one_dice")  # all_player_metrics.py:32-43
CORE_COUNT_FIELDS = ("raw_player_game_exposures", "raw_completed_player_game_exposures", "raw_safety_limit_player_game_exposures",
                     "raw_wins", "raw_losses", "raw_turn_round_mismatch_count", "raw_max_round_abort_exposures")
CORE_SUM_FIELDS = ("raw_final_score_sum", "raw_final_score_square_sum", "raw_n_turns_sum", "raw_n_turns_square_sum",
                   "raw_turn_return_game_weighted_exact_sum", "raw_turn_return_game_weighted_exact_square_sum",
                   "raw_turn_return_round_proxy_sum", "raw_turn_return_round_proxy_square_sum", "raw_turn_minus_rounds_sum",
                   "raw_turn_minus_rounds_square_sum")
DERIVED_FIELDS = ("turn_return_turn_weighted", "turn_return_game_weighted_exact", "turn_return_round_proxy", "round_proxy_gap",
                  "round_proxy_relative_gap", "turn_round_mismatch_prevalence", "win_rate_per_attempt", "win_rate_given_completion",
                  "safety_limit_exposure_rate")
ROW_ORDER_FLOAT_FIELDS = ("raw_turn_return_game_weighted_exact_sum", "raw_turn_return_game_weighted_exact_square_sum",
                          "raw_turn_return_round_proxy_sum", "raw_turn_return_round_proxy_square_sum",
                          "turn_return_game_weighted_exact", "turn_return_round_proxy", "round_proxy_gap", "round_proxy_relative_gap")
_COL = {name: i for i, name in enumerate(SEAT_STAT_NAMES)}


def all_player_batch_schema():
    """The reference's column order, types and nullability (all_player_metrics.py:101-119)."""
    import pyarrow as pa

    fields = [pa.field("root_seed", pa.int64(), nullable=False), pa.field("k", pa.int16(), nullable=False),
              pa.field("deterministic_batch_id", pa.int32(), nullable=False), pa.field("strategy", pa.int32(), nullable=False)]
    fields += [pa.field(name, pa.int64(), nullable=False) for name in CORE_COUNT_FIELDS]
    fields += [pa.field(name, pa.float64(), nullable=False) for name in CORE_SUM_FIELDS]
    for suffix in BEHAVIOR_SUFFIXES:
        fields += [pa.field(f"raw_{suffix}_observations", pa.int64(), nullable=False),
                   pa.field(f"raw_{suffix}_sum", pa.float64(), nullable=False),
                   pa.field(f"raw_{suffix}_square_sum", pa.float64(), nullable=False)]
    fields += [pa.field(name, pa.float64()) for name in DERIVED_FIELDS]
    return pa.schema(fields)


def _ratio(num: np.ndarray, den: np.ndarray):
    """num / den as float64 where den != 0, else None — the reference divides Python numbers (``x / n if n else None``)."""
    out = np.full(len(num), np.nan)
    ok = den != 0
    out[ok] = num[ok].astype(np.float64) / den[ok].astype(np.float64)
    return [None if not o else float(v) for v, o in zip(out, ok)]


def all_player_batch_columns(seat_stats: np.ndarray, strategy_ids: Sequence[int], root_seed: int, k: int, batch_id: int,
                             seat_ratio_sums: np.ndarray | None = None) -> dict:
    """One deterministic batch: ``seat_stats`` is ``[S][31]`` (``SEAT_STAT_NAMES``), ``seat_ratio_sums`` ``[S][4]`` float64
    (``fk_tournament_run_all_player``); rows in ascending strategy id, as the reference flushes them (:440-455).  Strategies
    without an exposure in the batch are left out, as there."""
    st = np.asarray(seat_stats, dtype=np.int64)
    ids = np.asarray(strategy_ids, dtype=np.int64)
    if seat_ratio_sums is None:
        raise ValueError("seat_ratio_sums is required: the all-player table has no nullable accumulator column (all_player_metrics.py:101-119)")
    ratios = np.asarray(seat_ratio_sums, dtype=np.float64)
    keep = np.flatnonzero(st[:, _COL["exposures"]] > 0)
    keep = keep[np.argsort(ids[keep], kind="stable")]
    st, ids, ratios = st[keep], ids[keep], ratios[keep]
    n = len(ids)
    exposures, completed, safety, wins = (st[:, _COL[c]] for c in ("exposures", "completed_exposures", "safety_limit_exposures", "wins"))
    if not (np.array_equal(exposures, completed + safety) and (wins <= completed).all()):
        raise ValueError("attempted exposures must equal completed plus safety-limit exposures")  # _finish_row :384-388
    cols: dict = {"root_seed": np.full(n, root_seed, dtype=np.int64), "k": np.full(n, k, dtype=np.int16),
                  "deterministic_batch_id": np.full(n, batch_id, dtype=np.int32), "strategy": ids.astype(np.int32),
                  "raw_player_game_exposures": exposures, "raw_completed_player_game_exposures": completed,
                  "raw_safety_limit_player_game_exposures": safety, "raw_wins": wins, "raw_losses": exposures - wins,
                  "raw_turn_round_mismatch_count": st[:, _COL["turn_round_mismatch_count"]],
                  # every seat of a safety-limit game carries hit_max_rounds (engine.py:485-489): the count is the safety exposures
                  "raw_max_round_abort_exposures": safety}
    f64 = lambda name: st[:, _COL[name]].astype(np.float64)  # noqa: E731  (integer sums, exact in float64 below 2^53)
    cols.update({"raw_final_score_sum": f64("final_score_sum"), "raw_final_score_square_sum": f64("final_score_square_sum"),
                 "raw_n_turns_sum": f64("n_turns_sum"), "raw_n_turns_square_sum": f64("n_turns_square_sum"),
                 "raw_turn_minus_rounds_sum": f64("turn_minus_rounds_sum"),
                 "raw_turn_minus_rounds_square_sum": f64("turn_minus_rounds_square_sum")})
    cols.update({"raw_turn_return_game_weighted_exact_sum": ratios[:, 0], "raw_turn_return_game_weighted_exact_square_sum": ratios[:, 1],
                 "raw_turn_return_round_proxy_sum": ratios[:, 2], "raw_turn_return_round_proxy_square_sum": ratios[:, 3]})
    for suffix in BEHAVIOR_SUFFIXES:
        # rank and loss_margin are null on safety-limit rows (simulation.py:628-655): observed on completed exposures only
        cols[f"raw_{suffix}_observations"] = completed if suffix in ("rank", "loss_margin") else exposures
        cols[f"raw_{suffix}_sum"] = f64(f"{suffix}_sum")
        cols[f"raw_{suffix}_square_sum"] = f64(f"{suffix}_square_sum")
    cols["turn_return_turn_weighted"] = _ratio(st[:, _COL["final_score_sum"]], st[:, _COL["n_turns_sum"]])
    # _finish_row :393-399 on Python floats: sum / exposures, the gap, the gap relative to the exact return (None when that is 0)
    exact = [float(v) / int(e) if e else None for v, e in zip(ratios[:, 0].tolist(), exposures.tolist())]
    proxy = [float(v) / int(e) if e else None for v, e in zip(ratios[:, 2].tolist(), exposures.tolist())]
    gap = [p - x if p is not None and x is not None else None for p, x in zip(proxy, exact)]
    cols["turn_return_game_weighted_exact"], cols["turn_return_round_proxy"], cols["round_proxy_gap"] = exact, proxy, gap
    cols["round_proxy_relative_gap"] = [g / x if g is not None and x else None for g, x in zip(gap, exact)]
    cols["turn_round_mismatch_prevalence"] = _ratio(st[:, _COL["turn_round_mismatch_count"]], exposures)
    cols["win_rate_per_attempt"] = _ratio(wins, exposures)
    cols["win_rate_given_completion"] = _ratio(wins, completed)
    cols["safety_limit_exposure_rate"] = _ratio(safety, exposures)
    return cols


def all_player_batch_table(seat_stats: np.ndarray, strategy_ids: Sequence[int], root_seed: int, k: int, batch_id: int,
                           seat_ratio_sums: np.ndarray | None = None):
    import pyarrow as pa

    schema = all_player_batch_schema()
    cols = all_player_batch_columns(seat_stats, strategy_ids, root_seed, k, batch_id, seat_ratio_sums)
    return pa.Table.from_pydict({name: cols[name] for name in schema.names}, schema=schema)
