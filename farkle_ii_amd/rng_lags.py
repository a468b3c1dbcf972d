"""Lag autocorrelation rows of the reference's RNG diagnostics, strategy family, from device sufficient statistics.

``analysis/rng_diagnostics.py`` orders every strategy's observations by ``(root_seed, k, shuffle_index, game_index, seat_index)``
(``_SEAT_COORDINATE_COLUMNS`` :89-90, ``_observation_sort_fields`` :1950-1961) and pushes ``win_indicator`` and ``n_rounds`` through
an ``_OnlineMetric`` (:2031-2076): per lag the six sums ``pair_count, sum x, sum y, sum x^2, sum y^2, sum xy`` (x = the earlier
observation).  A strategy is seated exactly once per shuffle, so the series is indexed by the shuffle and the sums are what
``fk_tournament_run_lags`` returns — exact integers where the reference adds the same integers in float64.

:class:`LagSummary` is the summary of one contiguous shuffle range; ``a.merge(b)`` (``b`` directly follows ``a``) adds the pairs
that straddle the cut from ``a``'s last and ``b``'s first ``max(lags)`` observations — launch groups and ranks (contiguous whole
batches each) combine this way, in range order.  :func:`strategy_lag_rows` turns a summary into the rows ``_rows_for_online_group``
writes (:2110-2160) with ``_OnlineMetric.result``'s arithmetic (:2066-2076).

The MATCHUP family of the same module (one group per sorted seat tuple, O(games) groups) has no pre-aggregation; it keeps reading
rows (``farkle run`` rows mode)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Sequence

import numpy as np

from .backend import LAG_COLS

WIN_BIT = 0x8000
EXPECTED_NOTE_STRATEGY_ORDER = "root_seed,k,shuffle_index,game_index,seat_index"  # ",".join(_SEAT_COORDINATE_COLUMNS)


def _pair_sums(earlier: np.ndarray, later: np.ndarray) -> np.ndarray:
    """``[S][LAG_COLS]`` of the pairs (earlier[i], later[i]), i over axis 0; values are ``n_rounds | won << 15``."""
    a, b = (earlier & 0x7FFF).astype(np.int64), (later & 0x7FFF).astype(np.int64)
    wa, wb = (earlier >> 15).astype(np.int64), (later >> 15).astype(np.int64)
    cols = [np.full(a.shape[1], a.shape[0], dtype=np.int64), wa.sum(0), wb.sum(0), wa.sum(0), wb.sum(0), (wa & wb).sum(0),
            a.sum(0), b.sum(0), (a * a).sum(0), (b * b).sum(0), (a * b).sum(0)]
    return np.stack(cols, axis=1)


@dataclass
class LagSummary:
    """Sufficient statistics of one contiguous shuffle range ``[begin, begin + n)`` of a (root, k) cell."""

    lags: tuple[int, ...]
    n: int                      # observations per strategy = shuffles in the range
    sums: np.ndarray            # int64 [S][n_lags][LAG_COLS]
    head: np.ndarray            # uint16 [min(max lag, n)][S]: the first observations
    tail: np.ndarray            # uint16 [min(max lag, n)][S]: the last observations

    @classmethod
    def from_engine(cls, result: dict, lags: Sequence[int]) -> "LagSummary":
        return cls(tuple(int(v) for v in lags), int(result["n_shuffles"]), np.asarray(result["lag_sums"], dtype=np.int64).copy(),
                   np.asarray(result["lag_head"], dtype=np.uint16).copy(), np.asarray(result["lag_tail"], dtype=np.uint16).copy())

    @classmethod
    def from_series(cls, values: np.ndarray, lags: Sequence[int]) -> "LagSummary":
        """From the whole ``[n][S]`` value matrix (host statement of the rule; tests, CPU engines)."""
        values = np.asarray(values, dtype=np.uint16)
        lags = tuple(int(v) for v in lags)
        n, S = values.shape
        sums = np.zeros((S, len(lags), LAG_COLS), dtype=np.int64)
        for i, lag in enumerate(lags):
            if n > lag:
                sums[:, i, :] = _pair_sums(values[:-lag], values[lag:])
        m = min(max(lags), n)
        return cls(lags, n, sums, values[:m].copy(), values[n - m:].copy())

    def merge(self, other: "LagSummary") -> "LagSummary":
        """``other`` covers the shuffles that directly follow this range."""
        if self.lags != other.lags or self.sums.shape != other.sums.shape:
            raise ValueError("lag summaries of different lag lists / strategy tables cannot be merged")
        if other.n == 0:
            return self
        if self.n == 0:
            return other
        sums = self.sums + other.sums
        L = max(self.lags)
        seam = np.concatenate([self.tail, other.head])          # the observations around the cut, in order
        cut = len(self.tail)
        for i, lag in enumerate(self.lags):                      # pairs (t - lag, t) with t - lag before the cut and t after it
            later = np.arange(cut, min(cut + lag, len(seam)))
            later = later[later - lag >= 0]
            if len(later):
                sums[:, i, :] += _pair_sums(seam[later - lag], seam[later])
        n = self.n + other.n
        m = min(L, n)
        head = np.concatenate([self.head, other.head])[:m] if len(self.head) < m else self.head[:m]
        tail = np.concatenate([self.tail, other.tail])[-m:] if len(other.tail) < m else other.tail[-m:]
        return LagSummary(self.lags, n, sums, head, tail)


def autocorr(pairs: int, sx: float, sy: float, sxx: float, syy: float, sxy: float) -> tuple[float | None, str]:
    """``_OnlineMetric.result`` (rng_diagnostics.py:2066-2076), same operation order, float64."""
    if pairs < 2:
        return None, "insufficient_pairs"
    count = float(pairs)
    numerator = count * sxy - sx * sy
    den_x = count * sxx - sx ** 2
    den_y = count * syy - sy ** 2
    if den_x <= 0.0 or den_y <= 0.0:
        return None, "zero_variance"
    return float(numerator / (den_x * den_y) ** 0.5), "estimated"


def strategy_lag_rows(summary: LagSummary, strategy_ids: Sequence[int], k: int, note: str = "") -> list[dict[str, Any]]:
    """The ``summary_level == "strategy"`` rows of ``rng_diagnostics`` (``_rows_for_online_group`` :2110-2160), strategies in
    table order, per strategy ``win_indicator`` then ``n_rounds``, each over the lags."""
    rows: list[dict[str, Any]] = []
    for s, sid in enumerate(strategy_ids):
        for metric, base in (("win_indicator", 1), ("n_rounds", 6)):
            for i, lag in enumerate(summary.lags):
                v = summary.sums[s, i]
                pairs = int(v[0])
                ac, status = autocorr(pairs, *(float(x) for x in v[base:base + 5]))
                half = 1.96 / pairs ** 0.5 if pairs > 0 else None
                rows.append({"summary_level": "strategy", "strategy": int(sid), "matchup_id": None, "matchup": None,
                             "participant_strategy_ids": None, "n_players": int(k), "observations": int(summary.n), "lagged_pairs": pairs,
                             "lag": int(lag), "metric": metric, "autocorr": ac, "estimability_status": status,
                             "zero_centered_descriptive_reference_band_lower": -half if half is not None else None,
                             "zero_centered_descriptive_reference_band_upper": half,
                             "sequence_order": EXPECTED_NOTE_STRATEGY_ORDER, "note": note})
    return rows


def lag_sums_table(summary: LagSummary, strategy_ids: Sequence[int], root_seed: int, k: int):
    """The sufficient statistics as an Arrow table (``farkle run --rng-lag-sums``): one row per (strategy, lag)."""
    import pyarrow as pa

    S, nl = len(strategy_ids), len(summary.lags)
    flat = summary.sums.reshape(S * nl, LAG_COLS)
    names = ["lagged_pairs", "win_sum_x", "win_sum_y", "win_sum_x2", "win_sum_y2", "win_sum_xy",
             "n_rounds_sum_x", "n_rounds_sum_y", "n_rounds_sum_x2", "n_rounds_sum_y2", "n_rounds_sum_xy"]
    cols = {"root_seed": pa.array(np.full(S * nl, root_seed, dtype=np.int64)), "n_players": pa.array(np.full(S * nl, k, dtype=np.int16)),
            "strategy": pa.array(np.repeat(np.asarray(strategy_ids, dtype=np.int32), nl)),
            "lag": pa.array(np.tile(np.asarray(summary.lags, dtype=np.int32), S)),
            "observations": pa.array(np.full(S * nl, summary.n, dtype=np.int64))}
    for j, name in enumerate(names):
        cols[name] = pa.array(flat[:, j])
    return pa.table(cols)


STATS_NOTE = ("Zero-centered approximate descriptive reference band only; values inside or outside the band do not establish or "
              "refute independence")  # _EXPECTED_NOTE, rng_diagnostics.py:80-83


def lag_stats_table(summary: LagSummary, strategy_ids: Sequence[int], k: int):
    """The strategy-level rows in the reference's ``_stats_schema`` (rng_diagnostics.py:2079-2098)."""
    import pyarrow as pa

    schema = pa.schema([
        pa.field("summary_level", pa.string(), nullable=False), pa.field("strategy", pa.int32()), pa.field("matchup_id", pa.uint64()),
        pa.field("matchup", pa.string()), pa.field("participant_strategy_ids", pa.list_(pa.int32())),
        pa.field("n_players", pa.int16(), nullable=False), pa.field("observations", pa.int64(), nullable=False),
        pa.field("lagged_pairs", pa.int64(), nullable=False), pa.field("lag", pa.int32(), nullable=False),
        pa.field("metric", pa.string(), nullable=False), pa.field("autocorr", pa.float64()),
        pa.field("estimability_status", pa.string(), nullable=False),
        pa.field("zero_centered_descriptive_reference_band_lower", pa.float64()),
        pa.field("zero_centered_descriptive_reference_band_upper", pa.float64()),
        pa.field("sequence_order", pa.string(), nullable=False), pa.field("note", pa.string(), nullable=False)])
    return pa.Table.from_pylist(strategy_lag_rows(summary, strategy_ids, k, note=STATS_NOTE), schema=schema)
