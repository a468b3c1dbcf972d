"""fk_tournament_run_lags on the MI355X: the reference's own RNG-diagnostics rows (strategy family), the oracle-backed statement of
the rule on larger tables, chunk carries shorter than the largest lag, and range merges."""
from __future__ import annotations

import numpy as np
import pytest

import golden_util as gu
from test_rng_lags_cpu import case_inputs, check_against_reference

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from farkle_ii_amd.engine import get_engine

    return get_engine()


@pytest.mark.parametrize("index", [0, 1])
def test_hip_lag_statistics_equal_the_reference_rows(eng, index):
    doc = gu.load("rng_lag_vectors.json")
    case, lags = doc["cases"][index], doc["lags"]
    table, ov = case_inputs(case)
    res = eng.tournament_lags(table, case["k"], case["root_seed"], 0, case["n_shuffles"], lags, target_score=case["target_score"],
                              max_rounds=case["max_rounds"], overrides=ov)
    check_against_reference(case, lags, res)


def _same(a: dict, b: dict) -> None:
    for key in ("tally", "lag_sums", "lag_head", "lag_tail"):
        assert np.array_equal(a[key], b[key]), key


def test_hip_equals_the_oracle_statement_on_larger_tables_and_across_chunks(eng):
    from oracle_engine_stub import Engine as StubEngine

    from bench import grid64
    from farkle_ii_amd.rng_lags import LagSummary
    from tools.time_config import table_for

    stub = StubEngine()
    t64 = grid64()
    # one batch on the 64-strategy grid: the LDS-tally launch (result records are written beside it for the value matrix)
    lags = (1, 2, 7)
    want = stub.tournament_lags(t64, 2, 42, 100, 700, lags)
    _same(eng.tournament_lags(t64, 2, 42, 100, 700, lags), want)
    t5160 = table_for(5160)
    eng.set_option("chunk_bytes", 1 << 20)  # the smallest workspace: three chunks here, four-shuffle chunks on the 5 160 grid
    try:
        got = eng.tournament_lags(t64, 2, 42, 100, 700, lags, shuffles_per_batch=50)
        assert eng.timing()["play_launches"] >= 3
        # chunks SHORTER than the largest lag: the carry rows of a chunk come from several earlier chunks
        small = eng.tournament_lags(t5160, 4, 3, 10, 40, (1, 3, 6))
        assert eng.timing()["play_launches"] >= 7
    finally:
        eng.set_option("chunk_bytes", 48 << 30)
    _same(got, stub.tournament_lags(t64, 2, 42, 100, 700, lags, shuffles_per_batch=50))
    _same(small, stub.tournament_lags(t5160, 4, 3, 10, 40, (1, 3, 6)))
    # the 5 160-strategy grid at four and eight seats (hot / cold kernels), and a range cut into calls that merge back
    for k, n_sh in ((4, 40), (8, 24)):
        table = t5160[:5160 - 5160 % k]
        want = stub.tournament_lags(table, k, 0, 0, n_sh, (1, 3))
        whole = eng.tournament_lags(table, k, 0, 0, n_sh, (1, 3))
        _same(whole, want)
        merged = None
        for b, e in ((0, 2), (2, 3), (3, n_sh - 9), (n_sh - 9, n_sh)):
            part = LagSummary.from_engine(eng.tournament_lags(table, k, 0, b, e, (1, 3)), (1, 3))
            merged = part if merged is None else merged.merge(part)
        assert np.array_equal(merged.sums, whole["lag_sums"]) and np.array_equal(merged.tail, whole["lag_tail"])


def test_lag_request_validation(eng):
    from bench import grid64
    from farkle_ii_amd.backend import FarkleHipError

    for bad in ((), (0, 1), (2, 2), (3, 1), tuple(range(1, 18))):
        with pytest.raises(FarkleHipError):
            eng.tournament_lags(grid64(), 2, 42, 0, 10, bad)
    with pytest.raises(FarkleHipError):
        eng.tournament_lags(grid64(), 2, 42, 0, 10, (1,), max_rounds=40_000)
    res = eng.tournament_lags(grid64(), 2, 42, 5, 5, (1, 4))  # empty range
    assert res["n_shuffles"] == 0 and not res["lag_sums"].any() and len(res["lag_head"]) == 0
