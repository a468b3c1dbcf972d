"""Lag sufficient statistics of the RNG diagnostics' strategy family (second half) on the CPU: the host statement of
the rule and the range merge against `tests/golden/rng_lag_vectors.json` — rows and metric states produced by the reference's OWN
`_extract_batch_arrays` / `_observation_records` / `_OnlineMetric` / `_rows_for_online_group` over rows it simulated
(`oracle/gen_golden.py:gen_rng_lags`)."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import golden_util as gu  # noqa: E402
from oracle_engine_stub import Engine as StubEngine  # noqa: E402

from farkle_ii_amd.backend import make_overrides  # noqa: E402
from farkle_ii_amd.rng_lags import LagSummary, strategy_lag_rows  # noqa: E402
from farkle_ii_amd.strategies import STRATEGY_DTYPE  # noqa: E402

NOTE = ("Zero-centered approximate descriptive reference band only; values inside or outside the band do not establish or "
        "refute independence")


def case_inputs(case):
    table = gu.strategies_from_tuples(case["strategies"], STRATEGY_DTYPE)
    ov = make_overrides((o[0], o[2], o[3], o[1], o[4]) for o in case["overrides"]) if case["overrides"] else None
    return table, ov


def check_against_reference(case, lags, result):
    """Engine result of the whole range == the reference's metric states and rows."""
    summary = LagSummary.from_engine(result, lags)
    ids = [int(v) for v in gu.strategies_from_tuples(case["strategies"], STRATEGY_DTYPE)["strategy_id"]]
    for s, sid in enumerate(ids):
        st = case["states"][str(sid)]
        for metric, base in (("win_indicator", 1), ("n_rounds", 6)):
            m = st[metric]
            assert m["n_obs"] == summary.n
            got = summary.sums[s][:, [0, base, base + 1, base + 2, base + 3, base + 4]].T.tolist()
            want = [m["pair_count"], m["sum_x"], m["sum_y"], m["sum_x2"], m["sum_y2"], m["sum_xy"]]
            assert got == [[int(v) for v in row] for row in want], (sid, metric)
    rows = strategy_lag_rows(summary, ids, case["k"], note=NOTE)
    want_rows = sorted(case["rows"], key=lambda r: (r["strategy"], r["metric"] != "win_indicator", r["lag"]))
    assert rows == want_rows  # floats included: the same float64 operations on the same integers


@pytest.mark.parametrize("index", [0, 1])
def test_stub_engine_lag_statistics_equal_the_reference_rows(index):
    doc = gu.load("rng_lag_vectors.json")
    case, lags = doc["cases"][index], doc["lags"]
    table, ov = case_inputs(case)
    res = StubEngine().tournament_lags(table, case["k"], case["root_seed"], 0, case["n_shuffles"], lags,
                                       target_score=case["target_score"], max_rounds=case["max_rounds"], overrides=ov)
    check_against_reference(case, lags, res)
    # the value series themselves (head / tail are its ends)
    ids = [int(v) for v in table["strategy_id"]]
    series = np.array([[(r | (w << 15)) for _, r, w in case["series"][str(sid)]] for sid in ids], dtype=np.uint16).T
    assert np.array_equal(res["lag_head"], series[:max(lags)]) and np.array_equal(res["lag_tail"], series[-max(lags):])
    assert [row[0] for row in case["series"][str(ids[0])]] == list(range(case["n_shuffles"]))  # one observation per shuffle, in order


def test_ranges_merge_to_the_whole_at_every_cut():
    doc = gu.load("rng_lag_vectors.json")
    case, lags = doc["cases"][0], doc["lags"]
    ids = [int(t[10]) if t[10] >= 0 else i for i, t in enumerate(case["strategies"])]
    series = np.array([[(r | (w << 15)) for _, r, w in case["series"][str(sid)]] for sid in ids], dtype=np.uint16).T
    whole = LagSummary.from_series(series, lags)
    n = len(series)
    for cut in range(0, n + 1):
        a = LagSummary.from_series(series[:cut], lags) if cut else LagSummary(tuple(lags), 0, np.zeros_like(whole.sums), series[:0], series[:0])
        b = LagSummary.from_series(series[cut:], lags) if cut < n else LagSummary(tuple(lags), 0, np.zeros_like(whole.sums), series[:0], series[:0])
        m = a.merge(b)
        assert m.n == n and np.array_equal(m.sums, whole.sums), cut
        assert np.array_equal(m.head, whole.head) and np.array_equal(m.tail, whole.tail), cut
    # three-way, with pieces shorter than the largest lag
    for c1, c2 in ((1, 3), (2, 4), (4, 7), (17, 18), (n - 3, n - 1)):
        parts = [LagSummary.from_series(p, lags) for p in (series[:c1], series[c1:c2], series[c2:])]
        m = parts[0].merge(parts[1]).merge(parts[2])
        assert np.array_equal(m.sums, whole.sums) and np.array_equal(m.tail, whole.tail) and np.array_equal(m.head, whole.head), (c1, c2)


def test_stub_engine_split_calls_merge_to_one_call():
    doc = gu.load("rng_lag_vectors.json")
    case, lags = doc["cases"][1], doc["lags"]
    table, ov = case_inputs(case)
    eng = StubEngine()
    kw = dict(target_score=case["target_score"], max_rounds=case["max_rounds"], overrides=ov)
    whole = LagSummary.from_engine(eng.tournament_lags(table, case["k"], case["root_seed"], 0, 24, lags, **kw), lags)
    merged = None
    for b, e in ((0, 3), (3, 4), (4, 15), (15, 24)):
        part = LagSummary.from_engine(eng.tournament_lags(table, case["k"], case["root_seed"], b, e, lags, **kw), lags)
        merged = part if merged is None else merged.merge(part)
    assert np.array_equal(merged.sums, whole.sums) and merged.n == whole.n
