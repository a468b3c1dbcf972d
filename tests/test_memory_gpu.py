"""Device-memory pressure (round-5 finding: four ranks sharing one device each planned two 48-GiB buffer sets and one died in hipMalloc).

The workspace of a call is sized from ``hipMemGetInfo`` — a share of what is free plus what the context already holds — and a call that
still meets ``hipErrorOutOfMemory`` gives its workspace back, halves the budget and is replayed.  Results never depend on the chunking:
checked against the CPU oracle with most of the device taken by somebody else.
"""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Hog:
    """Device memory held outside the engine under test: by a second context of the same library (``fk_debug_hold_memory``)."""

    def __init__(self):
        from farkle_ii_amd.backend import Engine

        self.holder = Engine(0)

    def leave(self, n_bytes: int) -> None:
        self.holder.hold_memory(n_bytes)

    def release(self) -> None:
        self.holder.hold_memory(-1)
        self.holder.close()


@pytest.fixture
def engine_and_hog():
    from farkle_ii_amd.engine import get_engine, set_engine

    set_engine(None)
    eng = get_engine()
    hog = _Hog()
    yield eng, hog
    hog.release()
    eng.set_option("workspace_percent", 80)
    set_engine(None)


def _case():
    import pyoracle as po

    from farkle_ii_amd.strategies import generate_strategy_grid, pack_strategies, prepare_public_helper_strategies

    table = pack_strategies(prepare_public_helper_strategies(generate_strategy_grid()[0]))  # the 5 160-strategy grid
    k, lo, hi = 4, 500, 900  # 400 shuffles x 1 290 games = 516 000 games: ~75 MB of workspace per 10^5 games
    want = po.tournament(table.view(po.STRATEGY_DTYPE), k, 7, lo, hi, shuffles_per_batch=100, n_threads=16)["tally"]
    return table, k, lo, hi, want


def test_chunked_under_memory_pressure_equals_the_oracle(engine_and_hog):
    eng, hog = engine_and_hog
    table, k, lo, hi, want = _case()
    roomy = eng.tournament(table, k, 7, lo, hi, shuffles_per_batch=100)["tally"]
    assert np.array_equal(roomy, want)
    budget_roomy = eng.get_option("last_budget")
    hog.leave(700 << 20)  # somebody else holds the device: ~0.7 GB left (+ what the context's own buffers hold)
    tight = eng.tournament(table, k, 7, lo, hi, shuffles_per_batch=100)
    assert np.array_equal(tight["tally"], want)
    assert eng.get_option("last_budget") < budget_roomy and eng.get_option("oom_replays") == 0  # sized from hipMemGetInfo: nothing failed
    rows = eng.tournament(table, k, 7, lo, lo + 40, want_rows=True)  # the rows path allocates its buffers under the same pressure
    assert np.array_equal(rows["tally"].sum(axis=0), eng.tournament(table, k, 7, lo, lo + 40)["tally"].sum(axis=0))


def test_out_of_memory_is_replayed_with_half_the_workspace(engine_and_hog):
    """An allocation that fails although the budget said it would fit (here: a budget of 2 000 % of what is free, 300 MB left for a call
    that wants ~450 MB in one chunk) releases the workspace, halves the budget and plays the call again — same results."""
    from farkle_ii_amd.backend import Engine
    from farkle_ii_amd.strategies import generate_strategy_grid, pack_strategies, prepare_public_helper_strategies

    _, hog = engine_and_hog
    table = pack_strategies(prepare_public_helper_strategies(generate_strategy_grid()[0]))
    k, lo, hi = 4, 0, 2000  # 2.58 million games: ~180 bytes of workspace each
    with Engine(0) as roomy:
        want = roomy.tournament(table, k, 7, lo, hi, shuffles_per_batch=500)["tally"]  # (this path is oracle-checked above and in test_hip_parity)
        assert roomy.get_option("oom_replays") == 0
    hog.leave(300 << 20)
    with Engine(0) as tight:
        tight.set_option("workspace_percent", 2000)
        got = tight.tournament(table, k, 7, lo, hi, shuffles_per_batch=500)["tally"]
        replays, budget = tight.get_option("oom_replays"), tight.get_option("last_budget")
        again = tight.tournament(table, k, 7, lo, hi, shuffles_per_batch=500)["tally"]  # the next call plans inside what the context now holds
    assert np.array_equal(got, want) and np.array_equal(again, want)
    assert replays >= 1 and budget < (6 << 30), (replays, budget)
