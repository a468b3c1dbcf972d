"""The Lemire-rejection DETOUR of the two game kernels, exercised on purpose (round-5 advisor finding).

A real rejection happens once in ~2^30 dice: the parity suites never reach the branch in which ``fk_play_kernel`` /
``fk_play_hc_kernel`` re-read the generator from the LDS seat record (or the hot planes, ``own_buf``, ``hasbuf``) and replay the roll
with ``roll_counts_sequential`` (csrc/fk_kernels.h:1149, csrc/fk_play_hc.h:386).  ``libfarkle_hip_detour.so`` is the same source
built with ``-DFK_FORCE_DETOUR=4`` (backend.VARIANTS): every fourth roll, chosen by bits of the advanced generator state so that the
lanes of a wave disagree, takes the detour.  The detour is exact whether or not a word was rejected, so tallies, rows and all-seat
statistics must stay bit-identical to the CPU oracle in every instance family: lean LDS records (k = 2, LDS tally), full records
(k = 3), the state-store instance, cold-in-LDS (k = 4), hot / cold with the buffered half word in the plane (k = 5 .. 12), H2H.
"""
from __future__ import annotations

import numpy as np
import pytest

from test_state_store_gpu import _random_valid_table

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def detour():
    from farkle_ii_amd import backend

    backend.build_library(variant="detour")  # prebuilt by __graft_entry__.build(); compiled here (hipcc, ~40 s) when the file did not travel
    eng = backend.Engine(0, variant="detour")
    yield eng
    eng.close()


@pytest.fixture(scope="module")
def po():
    import pyoracle

    return pyoracle


def test_the_variant_is_the_detour_build():
    """The variant library carries the forced-detour code: its kernels differ from the product library's (same exports)."""
    from farkle_ii_amd import backend

    product, variant = backend.library_path().read_bytes(), backend.library_path("detour").read_bytes()
    assert product != variant and abs(len(product) - len(variant)) < len(product) // 4


@pytest.mark.parametrize("k,S,options", [
    (2, 64, {}),                       # lean LDS records, tally in LDS, advance2
    (2, 64, {"use_lds_tally": 0}),     # result records
    (3, 96, {}),                       # fk_play_kernel, three seats
    (3, 96, {"lean": 0}),              # full 17-dword seat records
    (3, 96, {"state_store": 1}),       # one record per lane in LDS, exchanged with HBM per turn
    (4, 96, {}),                       # hot / cold, cold records in LDS
    (5, 100, {}), (6, 96, {}), (8, 96, {}), (10, 100, {}), (12, 96, {}),   # hot / cold, increments in registers, cold plane
    (6, 96, {"hot_cold": 0}),          # the LDS-record kernel at six seats
])
def test_forced_detours_change_nothing(detour, po, k, S, options):
    table = _random_valid_table(S, 900 + k)
    n_sh = 30
    ref = po.tournament(table.view(po.STRATEGY_DTYPE), k, 13, 4, 4 + n_sh, shuffles_per_batch=8, want_rows=True, n_threads=8)
    try:
        for name, value in options.items():
            detour.set_option(name, value)
        got = detour.tournament(table, k, 13, 4, 4 + n_sh, shuffles_per_batch=8, want_rows=True, want_seat_stats=True)
        counts = detour.tournament(table, k, 13, 4, 4 + n_sh)  # counts-only launch: LDS tally / 16-byte state records
    finally:
        for name in options:
            detour.set_option(name, {"use_lds_tally": -1, "lean": -1, "state_store": -1, "hot_cold": -1}[name])
    assert np.array_equal(got["tally"], ref["tally"]), (k, options)
    assert got["rows"].tobytes() == ref["rows"].tobytes(), (k, options)
    assert np.array_equal(counts["tally"][0], ref["tally"].sum(axis=0)), (k, options)
    assert int(got["seat_stats"][:, :, 0].sum()) == n_sh * S


def test_forced_detours_in_h2h_blocks(detour, po):
    table = _random_valid_table(8, 77)
    pairs = np.stack([table[[0, 1]], table[[2, 3]], table[[4, 5]], table[[6, 7]]])
    got = detour.h2h_blocks(pairs, 9, [0, 1, 2, 3], [0, 1, 0, 1], 400, 900)
    for b in range(4):
        want = po.h2h_block(pairs[b].view(po.STRATEGY_DTYPE), 9, b, [0, 1, 0, 1][b], 400, 900, 900)
        assert np.array_equal(got[b], want), b
