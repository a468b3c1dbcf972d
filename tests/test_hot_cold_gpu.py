"""GPU parity tests of the hot / cold game kernel (``fk_play_hc_kernel``, csrc/fk_play_hc.h) through the C-ABI, bit-exact
against the CPU oracle: generator state of every seat in LDS, the behaviour counters / banked totals in a per-lane plane.

The kernel is chosen by the launch plan for k >= 4 (option ``hot_cold`` = -1; k = 4: the cold-in-LDS instance; k = 5 .. 7: four
waves per SIMD with the increments in registers; k = 8: three; k = 9 .. 12 (round 5): one 768-thread block per CU) —
the instances the shipped library holds.  The variants that lost
or tied against them (profiles/HISTORY.md) left the tree in round 6 (logs under profiles/, sources in the repository's history)."""
from __future__ import annotations

import numpy as np
import pytest

import golden_util as gu
from test_state_store_gpu import _random_valid_table, _strats

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from farkle_ii_amd.backend import Engine

    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def po():
    import pyoracle

    return pyoracle


LDS_TABLE_BYTES = 12656  # LT_BYTES of csrc/fk_device.h (32-bit score entries since round 5)


FOUR_WAVE_BLOCK = {5: 256, 6: 512, 7: 1024}  # the register instances of k = 5 .. 7 run four waves per SIMD in these blocks
WIDE_BLOCK = {9: 768, 10: 768, 11: 768, 12: 768}  # nine to twelve seats: one block per CU, three waves per SIMD
COLD_IN_LDS_BLOCK = {3: 256, 4: 320, 5: 256}  # cold records in LDS: 32 bytes per seat and lane (the auto plan at k = 4)


def _ran_cold_in_lds(eng, k: int) -> bool:
    t = eng.timing()
    return t["play_block"] == COLD_IN_LDS_BLOCK[k] and t["play_lds_bytes"] == COLD_IN_LDS_BLOCK[k] * 32 * k


def _ran_hot_cold(eng, k: int, block: int | None = None, tables: int = 1) -> bool:
    t = eng.timing()
    if block is None and k == 4:
        return _ran_cold_in_lds(eng, 4)
    block = WIDE_BLOCK.get(k, FOUR_WAVE_BLOCK.get(k, 256)) if block is None else block
    # hot part: 16 bytes per seat and lane (generator state; the buffered half word rides in the cold-plane slot since round 5)
    return t["play_block"] == block and t["play_lds_bytes"] == block * 16 * k + (LDS_TABLE_BYTES if tables else 0)


@pytest.mark.parametrize("k", [3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_hot_cold_kernel_agrees_with_oracle(eng, po, k):
    """Per-batch tallies, rows and all-seat statistics of the same shuffles: hot / cold kernel, LDS-record kernel, oracle."""
    S = {3: 96, 4: 96, 5: 100, 6: 96, 7: 98, 8: 96, 9: 99, 10: 100, 11: 99, 12: 96}[k]
    table = _random_valid_table(S, 300 + k)
    n_sh = 36
    ref = po.tournament(table.view(po.STRATEGY_DTYPE), k, 6, 2, 2 + n_sh, shuffles_per_batch=16, want_rows=True, n_threads=8)
    try:
        for hc in (-1, 0):  # the launch plan's choice (hot / cold from four seats), then the LDS-record kernel on the same shuffles
            eng.set_option("hot_cold", hc)
            expect = hc == -1 and k >= 4
            got = eng.tournament(table, k, 6, 2, 2 + n_sh, shuffles_per_batch=16, want_rows=True, want_seat_stats=True)
            assert (k >= 4 and _ran_hot_cold(eng, k)) == expect, (k, hc, eng.timing())
            assert np.array_equal(got["tally"], ref["tally"]), (k, hc)
            assert got["rows"].tobytes() == ref["rows"].tobytes(), (k, hc)
            if hc == -1:
                stats = got["seat_stats"]
            else:
                assert np.array_equal(got["seat_stats"], stats), k
            eng.set_option("use_lds_tally", 0)  # one batch through result records, no final state records wanted
            rec = eng.tournament(table, k, 6, 2, 2 + n_sh)
            eng.set_option("use_lds_tally", -1)
            assert (k >= 4 and _ran_hot_cold(eng, k)) == expect
            assert np.array_equal(rec["tally"][0], ref["tally"].sum(axis=0)), (k, hc)
    finally:
        eng.set_option("hot_cold", -1)
        eng.set_option("use_lds_tally", -1)


def test_hot_cold_limits_overrides_and_safety_games(eng, po):
    """max_rounds / target variants (units-of-50 rounding), per-game overrides, never-banking tables, max_rounds = 0, and a
    target beyond the kernel's 12-bit totals (the launch plan then stays on the LDS-record kernel)."""
    from farkle_ii_amd.backend import make_overrides

    table = _strats(gu.load("grid_vectors.json")["g64"])
    never = table.copy()
    never["dice_threshold"], never["require_both"] = 0, 1
    k, gps = 4, 16
    ovs = [(9, 2, 1, k, 0), (9, 2, 3, k, 7), (9, 5, gps - 1, k, 1), (9, 0, 0, k, 3), (9, 7, 2, k, 250), (9, 2, 5, k, 5)]
    try:
        for tbl, target, mr, expect_hc in [(table, 10_000, 200, True), (table, 2_000, 5, True), (never, 10_000, 12, True),
                                           (table, 50, 200, True), (table, 10_000, 0, True), (table, 10_025, 200, True),
                                           (table, 1_030, 60, True), (table, 75, 200, True), (table, 1, 200, True),
                                           (table, 135_000, 40, True), (table, 135_001, 40, False), (table, 3_200_000, 40, False)]:
            ref = po.tournament(tbl.view(po.STRATEGY_DTYPE), k, 9, 0, 10, shuffles_per_batch=3, target_score=target, max_rounds=mr,
                                overrides=po.make_overrides(ovs), want_rows=True)
            got = eng.tournament(tbl, k, 9, 0, 10, shuffles_per_batch=3, target_score=target, max_rounds=mr,
                                 overrides=make_overrides(ovs), want_rows=True)
            assert _ran_hot_cold(eng, k) == expect_hc, (target, mr)
            assert np.array_equal(got["tally"], ref["tally"]), (target, mr)
            assert got["rows"].tobytes() == ref["rows"].tobytes(), (target, mr)
    finally:
        eng.set_option("hot_cold", -1)


@pytest.mark.parametrize("k", [5, 6, 7])
def test_four_wave_instances_limits_and_overrides(eng, po, k):
    """The four-wave register instances (k = 5 .. 7, the auto plan's choice): short targets, round limits and per-game
    overrides, final seat records (rows) and all-seat statistics against the oracle."""
    from farkle_ii_amd.backend import make_overrides
    from oracle_engine_stub import seat_stats_from_rows

    S = {5: 100, 6: 96, 7: 98}[k]
    gps = S // k
    table = _random_valid_table(S, 4100 + k)
    ovs = [(4, 1, 0, k, 0), (4, 1, gps - 1, k, 2), (4, 3, 2, k, 9), (4, 6, 1, k, 240)]  # (<= 255 rounds: the farkle field)
    for target, mr in [(10_000, 200), (1_500, 4), (50, 200), (10_000, 0), (135_000, 30)]:
        ref = po.tournament(table.view(po.STRATEGY_DTYPE), k, 4, 0, 9, shuffles_per_batch=4, target_score=target, max_rounds=mr,
                            overrides=po.make_overrides(ovs), want_rows=True, n_threads=8)
        got = eng.tournament(table, k, 4, 0, 9, shuffles_per_batch=4, target_score=target, max_rounds=mr,
                             overrides=make_overrides(ovs), want_rows=True, want_seat_stats=True)
        assert _ran_hot_cold(eng, k), (k, target, mr, eng.timing())  # auto plan
        assert np.array_equal(got["tally"], ref["tally"]), (k, target, mr)
        assert got["rows"].tobytes() == ref["rows"].tobytes(), (k, target, mr)
        assert np.array_equal(got["seat_stats"], seat_stats_from_rows(ref["rows"], k, S, gps, 4)), (k, target, mr)


@pytest.mark.parametrize("k", [9, 10, 11, 12])
def test_wide_table_instances_limits_and_overrides(eng, po, k):
    """Nine to twelve seats (the reference's production list holds 10 and 12, configs/farkle_mega_config.yaml:10): the one-block-per-CU
    register instances of the hot / cold kernel — short targets, round limits, per-game overrides, never-banking tables (safety-limit
    games), rows, per-batch tallies and all-seat statistics against the oracle; and on the reference's 5 160-strategy grid."""
    from farkle_ii_amd.backend import make_overrides
    from oracle_engine_stub import seat_stats_from_rows
    from test_state_store_gpu import _default_table

    S = {9: 99, 10: 100, 11: 99, 12: 96}[k]
    gps = S // k
    table = _random_valid_table(S, 7300 + k)
    never = table.copy()
    never["dice_threshold"], never["require_both"], never["consider_score"], never["consider_dice"] = 0, 1, 1, 1
    ovs = [(5, 1, 0, k, 0), (5, 1, gps - 1, k, 2), (5, 3, 2, k, 9), (5, 6, 1, k, 240)]
    for tbl, target, mr in [(table, 10_000, 200), (table, 1_500, 4), (table, 50, 200), (table, 10_000, 0), (table, 135_000, 30), (never, 10_000, 9)]:
        ref = po.tournament(tbl.view(po.STRATEGY_DTYPE), k, 5, 0, 9, shuffles_per_batch=4, target_score=target, max_rounds=mr,
                            overrides=po.make_overrides(ovs), want_rows=True, n_threads=8)
        got = eng.tournament(tbl, k, 5, 0, 9, shuffles_per_batch=4, target_score=target, max_rounds=mr,
                             overrides=make_overrides(ovs), want_rows=True, want_seat_stats=True)
        assert _ran_hot_cold(eng, k), (k, target, mr, eng.timing())  # auto plan
        assert np.array_equal(got["tally"], ref["tally"]), (k, target, mr)
        assert got["rows"].tobytes() == ref["rows"].tobytes(), (k, target, mr)
        assert np.array_equal(got["seat_stats"], seat_stats_from_rows(ref["rows"], k, S, gps, 4)), (k, target, mr)
    big = _default_table()[:5148 if k in (9, 11) else 5160].copy()  # (5 160 = 10 x 516 = 12 x 430; 5 148 = 9 x 572 = 11 x 468)
    big["strategy_id"] = np.arange(len(big))
    ref = po.tournament(big.view(po.STRATEGY_DTYPE), k, 0, 3, 7, shuffles_per_batch=3, n_threads=8)
    got = eng.tournament(big, k, 0, 3, 7, shuffles_per_batch=3)
    assert _ran_hot_cold(eng, k), eng.timing()
    assert np.array_equal(got["tally"], ref["tally"]), k


@pytest.mark.parametrize("k,waves", [(6, 3), (7, 3), (8, 2), (10, 2), (5, 4), (12, 3)])
def test_max_waves_option_never_reaches_an_instance_that_was_not_compiled(eng, po, k, waves):
    """Option ``max_waves`` below what the plan's hot / cold instance of k seats needs (round-4 advisor: max_waves = 3 at k = 6 / 7 on a wide
    table planned the 256-thread three-wave instance the shipped library no longer holds -> hipErrorInvalidValue): the call runs on the
    LDS-record kernel instead, results identical — on the 5 160-strategy grid, where the lane counts make the plan prefer the hot / cold kernel."""
    from test_state_store_gpu import _default_table

    big = _default_table()
    big = big[: len(big) // k * k].copy()  # (a table's size must be a multiple of the player count)
    big["strategy_id"] = np.arange(len(big))
    ref = po.tournament(big.view(po.STRATEGY_DTYPE), k, 0, 0, 3, shuffles_per_batch=2, n_threads=8)
    try:
        eng.set_option("max_waves", waves)
        got = eng.tournament(big, k, 0, 0, 3, shuffles_per_batch=2)
        need = 4 if 5 <= k <= 7 else 3
        assert _ran_hot_cold(eng, k) == (waves >= need), (k, waves, eng.timing())
        assert np.array_equal(got["tally"], ref["tally"]), (k, waves)
    finally:
        eng.set_option("max_waves", 6)


@pytest.mark.parametrize("k", [4])
def test_cold_records_in_lds_instance(eng, po, k):
    """Four seats: the cold records sit in LDS beside the hot part (32 bytes per seat and lane, no plane, global tables).
    Tallies, rows (final seat records), all-seat statistics, short targets, round limits and overrides against the oracle."""
    from farkle_ii_amd.backend import make_overrides
    from oracle_engine_stub import seat_stats_from_rows

    S = {3: 96, 4: 96, 5: 100}[k]
    gps = S // k
    table = _random_valid_table(S, 5200 + k)
    ovs = [(11, 0, 1, k, 0), (11, 2, gps - 1, k, 3), (11, 5, 0, k, 120)]
    try:
        for target, mr in [(10_000, 200), (2_000, 6), (50, 200), (10_000, 0), (135_000, 25)]:
            ref = po.tournament(table.view(po.STRATEGY_DTYPE), k, 11, 0, 8, shuffles_per_batch=3, target_score=target, max_rounds=mr,
                                overrides=po.make_overrides(ovs), want_rows=True, n_threads=8)
            got = eng.tournament(table, k, 11, 0, 8, shuffles_per_batch=3, target_score=target, max_rounds=mr,
                                 overrides=make_overrides(ovs), want_rows=True, want_seat_stats=True)
            assert _ran_cold_in_lds(eng, k), eng.timing()
            assert np.array_equal(got["tally"], ref["tally"]), (k, target, mr)
            assert got["rows"].tobytes() == ref["rows"].tobytes(), (k, target, mr)
            assert np.array_equal(got["seat_stats"], seat_stats_from_rows(ref["rows"], k, S, gps, 3)), (k, target, mr)
            eng.set_option("use_lds_tally", 0)  # counts only, through result records
            rec = eng.tournament(table, k, 11, 0, 8, target_score=target, max_rounds=mr, overrides=make_overrides(ovs))
            eng.set_option("use_lds_tally", -1)
            assert np.array_equal(rec["tally"][0], ref["tally"].sum(axis=0)), (k, target, mr)
    finally:
        eng.set_option("use_lds_tally", -1)


def test_hot_cold_counter_guard_replays_on_the_lds_record_kernel(eng, po):
    """A seat that rolls hot dice more than 255 times (or rolls more than 2 047 times) in one game reaches a guard bit of the
    hot / cold kernel's packed cold record: the call is replayed on the LDS-record kernel (16-bit fields) and still equals
    the oracle — with the cold records in LDS (four seats) and in the plane (eight)."""
    table = _strats(gu.load("grid_vectors.json")["g64"])[:8].copy()
    table["dice_threshold"], table["require_both"], table["auto_hot_dice"] = 0, 1, 1   # nobody banks: games run to the round limit
    table["strategy_id"] = np.arange(8)
    ref = po.tournament(table.view(po.STRATEGY_DTYPE), 4, 3, 0, 2, max_rounds=6000, want_rows=True)
    assert int(ref["rows"]["seats"]["hot_dice"].max()) > 250
    got = eng.tournament(table, 4, 3, 0, 2, max_rounds=6000, want_rows=True)
    t = eng.timing()  # the replay's kernel is the one the timing record describes
    assert not _ran_cold_in_lds(eng, 4), t
    assert np.array_equal(got["tally"], ref["tally"])
    assert got["rows"].tobytes() == ref["rows"].tobytes()
    # the same with eight seats (cold records in the per-lane plane)
    table8 = np.concatenate([table, table])
    table8["strategy_id"] = np.arange(16)
    ref8 = po.tournament(table8.view(po.STRATEGY_DTYPE), 8, 3, 0, 1, max_rounds=3000, want_rows=True)
    assert int(ref8["rows"]["seats"]["hot_dice"].max()) > 250
    got8 = eng.tournament(table8, 8, 3, 0, 1, max_rounds=3000, want_rows=True)
    assert not _ran_hot_cold(eng, 8), eng.timing()
    assert np.array_equal(got8["tally"], ref8["tally"]) and got8["rows"].tobytes() == ref8["rows"].tobytes()
    # the reference's default grid seats four never-banking strategies together now and then: 200 rounds, 940 rolls, 200
    # farkles, 142 hot-dice turns of one seat — inside the packed fields, no replay
    never = table.copy()
    never["strategy_id"] = np.arange(8)
    want = po.tournament(never.view(po.STRATEGY_DTYPE), 4, 3, 0, 3, max_rounds=200, want_rows=True)
    assert int(want["rows"]["seats"]["farkles"].max()) >= 190
    got = eng.tournament(never, 4, 3, 0, 3, max_rounds=200, want_rows=True)
    assert _ran_cold_in_lds(eng, 4)
    assert got["rows"].tobytes() == want["rows"].tobytes()


def test_hot_cold_is_the_default_for_wide_tables(eng, po):
    """k = 8 on the reference's 5 160-strategy grid (BASELINE config 4's widest table): auto plan = hot / cold kernel."""
    from test_state_store_gpu import _default_table

    table = _default_table()
    ref = po.tournament(table.view(po.STRATEGY_DTYPE), 8, 0, 0, 6, shuffles_per_batch=4, n_threads=8)
    got = eng.tournament(table, 8, 0, 0, 6, shuffles_per_batch=4)
    assert _ran_hot_cold(eng, 8)
    assert np.array_equal(got["tally"], ref["tally"])


@pytest.mark.parametrize("k", [4, 6])
def test_hot_cold_kernels_in_chunked_hinted_and_unpipelined_calls(eng, po, k):
    """The auto plan's hot / cold instances (k = 4: cold records in LDS; k = 6: registers + plane) under the host-side machinery:
    about ten chunks per call (1 MiB of workspace each), a hinted next call prepared behind this call's kernel, the pipeline
    off, rows and per-batch tallies."""
    S = 96
    table = _random_valid_table(S, 6100 + k)
    ranges = [(0, 1500), (1500, 2600)]
    ref = {rng: po.tournament(table.view(po.STRATEGY_DTYPE), k, 13, rng[0], rng[1], shuffles_per_batch=250, want_rows=True, n_threads=8)
           for rng in ranges}
    try:
        for pipeline in (1, 0):
            eng.set_option("pipeline", pipeline)
            for chunk in (48 << 30, 1 << 20):  # one chunk / roughly ten chunks per call
                eng.set_option("chunk_bytes", chunk)
                eng.hint_next(*ranges[1], need_state=True)
                for rng in ranges:
                    got = eng.tournament(table, k, 13, rng[0], rng[1], shuffles_per_batch=250, want_rows=True)
                    assert _ran_hot_cold(eng, k), eng.timing()
                    if chunk == 1 << 20:
                        assert eng.timing()["play_launches"] >= 5, eng.timing()
                    assert np.array_equal(got["tally"], ref[rng]["tally"]), (k, pipeline, chunk, rng)
                    assert got["rows"].tobytes() == ref[rng]["rows"].tobytes(), (k, pipeline, chunk, rng)
                counts = eng.tournament(table, k, 13, 0, 1500, shuffles_per_batch=250)
                assert np.array_equal(counts["tally"], ref[ranges[0]]["tally"]), (k, pipeline, chunk)
    finally:
        eng.set_option("pipeline", 1)
        eng.set_option("chunk_bytes", 48 << 30)


@pytest.mark.parametrize("k", [2, 4, 8])
def test_timing_carries_the_clock_the_launch_tail_and_the_flag_form(eng, po, k):
    """Option ``clock_stamps``: fk_timing reports the shader clock measured inside the game kernel, when its median and its last
    workgroup finished (the last wave of every block stamps its end), and which strategy-flag form of the instance ran; the stamps
    change no result.  ``want_seat_ratios=False`` returns the integer statistics alone (no float sums)."""
    S = 96
    table = _random_valid_table(S, 5200 + k)
    uniform = table.copy()
    for name in ("smart_five", "smart_one", "consider_score", "consider_dice", "auto_hot_dice", "run_up_score", "require_both", "favor_score"):
        uniform[name] = uniform[name][0]
    plain = eng.tournament(table, k, 3, 0, 24, want_seat_stats=True)
    try:
        eng.set_option("clock_stamps", 1)
        got = eng.tournament(table, k, 3, 0, 24, want_seat_stats=True, want_seat_ratios=False)
        t = eng.timing()
        assert np.array_equal(got["tally"], plain["tally"]) and np.array_equal(got["seat_stats"], plain["seat_stats"])
        assert got["seat_ratio_sums"] is None and plain["seat_ratio_sums"] is not None
        assert 500 < t["play_clock_mhz"] < 3000, t
        assert 0.0 < t["play_block_end_p50_ms"] <= t["play_block_end_max_ms"] <= t["play_ms"] * 1.05 + 0.05, t
        assert t["play_mixed_flags"] in (0xC000, 0xFF00), t  # a random table: some flag differs between its strategies
        eng.tournament(uniform, k, 3, 0, 8)
        assert eng.timing()["play_mixed_flags"] == 0, eng.timing()  # every flag shared by the whole table: the scalar form
    finally:
        eng.set_option("clock_stamps", 0)
    eng.tournament(table, k, 3, 0, 8)
    t = eng.timing()
    assert t["play_clock_mhz"] == 0 and t["play_block_end_max_ms"] == 0.0, t
