"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle and the golden
vectors (reference outputs).  Bit-exact: everything on this path is integer work.

Run on a MI355X box with ``python -m pytest tests -m gpu``.
"""
from __future__ import annotations

import numpy as np
import pytest

import golden_util as gu

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


def _strats(tuples):
    from farkle_ii_amd.strategies import STRATEGY_DTYPE

    return gu.strategies_from_tuples(tuples, STRATEGY_DTYPE)


def _coords(items):
    from farkle_ii_amd.backend import COORD_DTYPE

    out = np.zeros(len(items), dtype=COORD_DTYPE)
    for i, it in enumerate(items):
        out[i] = it
    return out


def _default_table():
    from farkle_ii_amd.strategies import default_grid_tuples

    return _strats(default_grid_tuples())


def _rows_equal(a: np.ndarray, b: np.ndarray) -> bool:
    return a.tobytes() == b.tobytes()


# ------------------------------------------------------------------ device info / loud failures
def test_device_is_gfx950(eng):
    info = eng.device_info()
    assert info["arch"].startswith("gfx950"), info
    assert info["wavefront_size"] == 64 and info["compute_units"] >= 200


# ------------------------------------------------------------------ RNG + dice
def test_streams_and_dice_match_golden(eng):
    data = gu.load("rng_vectors.json")
    for c in data["cases"]:
        coords = _coords([(c["purpose"], 0, c["root_seed"], c["k"], c["shuffle_index"], c["pair_id"], c["order"],
                           c["game_index"], c["seat_index"], 0)])
        faces, raw = eng.debug_dice(coords, c["sizes"])
        assert [int(v) for v in raw[0]] == c["raw64"][:4]
        assert faces[0].tolist() == c["dice"]


def test_dice_match_oracle_random_coordinates(eng, po):
    rs = np.random.default_rng(11)
    n = 4096
    items = [(int(rs.choice([10, 103, 203])), 0, int(rs.integers(0, 2**63)), int(rs.integers(2, 13)), int(rs.integers(0, 2**40)),
              int(rs.integers(0, 4000)), int(rs.integers(0, 2)), int(rs.integers(0, 2**34)), int(rs.integers(0, 12)), 0)
             for _ in range(n)]
    sizes = rs.integers(1, 7, size=64).astype(np.int32)
    faces, raw = eng.debug_dice(_coords(items), sizes)
    for i in range(0, n, 7):
        c = po.coord(items[i][0], *items[i][2:9])
        assert np.array_equal(po.dice_stream(c, sizes), faces[i])
        assert np.array_equal(po.stream64(c, 4), raw[i])


def test_coordinate_seeds_match_golden_oracle_and_host(eng, po):
    """fk_coordinate_seeds = coordinate_seed (utils/random.py:190-232): NumPy-generated goldens, the oracle on random
    coordinates of every purpose namespace (replicate_index included) and the host's vectorised fingerprints."""
    from farkle_ii_amd import random as urandom

    data = gu.load("rng_vectors.json")
    gold = _coords([(c["purpose"], 0, c["root_seed"], c["k"], c["shuffle_index"], c["pair_id"], c["order"], c["game_index"],
                     c["seat_index"], 0) for c in data["cases"]])
    s32, s64 = eng.coordinate_seeds(gold, want32=True, want64=True)
    assert s32.tolist() == [c["seed32"] for c in data["cases"]] and s64.tolist() == [c["seed64"] for c in data["cases"]]
    rs = np.random.default_rng(5)
    n = 2000
    items = [(int(rs.choice([1, 10, 11, 100, 101, 102, 103, 202, 203])), 0, int(rs.integers(0, 2**63)), int(rs.integers(0, 13)),
              int(rs.integers(0, 2**40)), int(rs.integers(0, 4000)), int(rs.integers(0, 2)), int(rs.integers(0, 2**34)),
              int(rs.integers(0, 12)), int(rs.integers(0, 3))) for _ in range(n)]
    s32, s64 = eng.coordinate_seeds(_coords(items), want32=True, want64=True)
    for i in range(0, n, 5):
        c = po.coord(items[i][0], *items[i][2:10])
        assert po.coordinate_seed32(c) == s32[i] and po.coordinate_seed64(c) == s64[i], items[i]
    # the game_seed column of a shuffle's rows (namespace 102), as the host computes it for row shards
    idx = np.arange(500, dtype=np.uint64)
    host = urandom.coordinate_seeds(urandom.RandomPurpose.TOURNAMENT_GAME, root_seed=42, k=2, shuffle_index=17, game_index=idx,
                                    dtype=np.uint32)
    dev, _ = eng.coordinate_seeds(_coords([(102, 0, 42, 2, 17, 0, 0, int(g), 0, 0) for g in idx]))
    assert np.array_equal(host, dev)


def _state_with_output(out64: int, lo: int) -> tuple[int, int]:
    """Invert the DXSM output function: a (hi, lo) state whose next 64-bit output is ``out64``."""
    M = 0xDA942042E4DD58B5
    mask = (1 << 64) - 1
    h = (out64 * pow(lo | 1, -1, 1 << 64)) & mask
    h ^= h >> 48
    h = (h * pow(M, -1, 1 << 64)) & mask
    h ^= h >> 32
    return h, lo


def test_lemire_rejection_slow_path(eng, po):
    """A die word r with low32(6r) < 4 must be redrawn (engine.py:101 -> Generator.integers).  Such words
    appear once per ~2^30 dice, so states that emit them on demand are constructed by inverting DXSM."""
    rejecting = [0, 715827883, 2147483648, 2863311531]
    assert all(((r * 6) & 0xFFFFFFFF) < 4 for r in rejecting)
    rs = np.random.default_rng(3)
    states = []
    for r in rejecting:
        for other in rejecting + [12345, 0xFFFFFFFF]:
            for has_buf in (0, 1):
                for lo_half_first in (0, 1):
                    out = (other << 32) | r if lo_half_first else (r << 32) | other
                    hi, lo = _state_with_output(out, int(rs.integers(0, 2**63)))
                    inc = int(rs.integers(0, 2**63)) * 2 + 1
                    states.append([hi, lo, int(rs.integers(0, 2**63)), inc, has_buf, rejecting[int(rs.integers(0, 4))] if has_buf else 0])
    states = np.array(states, dtype=np.uint64)
    for sizes in ([6, 6, 6], [1, 2, 3, 4, 5, 6, 1], [5, 1, 1, 6], [2, 2, 2, 2, 3]):
        faces, out = eng.debug_dice_state(states, sizes)
        for i, st in enumerate(states):
            f, o = po.dice_from_state(st, sizes)
            assert np.array_equal(f, faces[i]), (i, sizes)
            assert np.array_equal(o, out[i]), (i, sizes)
        assert faces.min() >= 1 and faces.max() <= 6
        # the game kernels' own instantiation: roll_counts<3> (18-bit count key, rejection test on the 6w product)
        keys, out3 = eng.debug_dice_keys(states, sizes)
        starts = np.concatenate([[0], np.cumsum(sizes)])
        for i, st in enumerate(states):
            f, o = po.dice_from_state(st, sizes)
            assert np.array_equal(o, out3[i]), (i, sizes)
            for c in range(len(sizes)):
                want = sum(1 << (3 * (int(face) - 1)) for face in f[starts[c]:starts[c + 1]])
                assert int(keys[i, c]) == want, (i, sizes, c)


def test_kernel_dice_instantiation_on_random_states(eng, po):
    """roll_counts<3> (fast path) against the oracle's sequential dice on random generator states, every roll size."""
    rs = np.random.default_rng(17)
    states = np.zeros((512, 6), dtype=np.uint64)
    states[:, :4] = rs.integers(0, 2**63, (512, 4), dtype=np.uint64)
    states[:, 3] |= np.uint64(1)
    states[:, 4] = rs.integers(0, 2, 512)
    states[:, 5] = np.where(states[:, 4] == 1, rs.integers(0, 2**32, 512, dtype=np.uint64), 0)
    sizes = [6, 5, 4, 3, 2, 1, 6, 1, 5, 2]
    keys, out = eng.debug_dice_keys(states, sizes)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    for i, st in enumerate(states):
        f, o = po.dice_from_state(st, sizes)
        assert np.array_equal(o, out[i]), i
        for c in range(len(sizes)):
            assert int(keys[i, c]) == sum(1 << (3 * (int(face) - 1)) for face in f[starts[c]:starts[c + 1]]), (i, c)


# ------------------------------------------------------------------ scoring / decisions
def test_score_table_all_923_patterns(eng):
    data = gu.load("scoring_vectors.json")
    table = data["table"]
    faces = np.zeros((len(table), 6), dtype=np.uint8)
    lens = np.zeros(len(table), dtype=np.int32)
    for i, row in enumerate(table):
        roll = [f + 1 for f in range(6) for _ in range(row[f])]
        faces[i, :len(roll)] = roll
        lens[i] = len(roll)
    plain = _strats([[300, 2, 0, 0, 1, 1, 0, 0, 0, 1, 0]] * len(table))
    out = eng.debug_score(faces, lens, np.zeros(len(table), dtype=np.int32), plain)
    for i, row in enumerate(table):
        assert out[i, 0] == row[6] and out[i, 1] == row[7] and out[i, 3] == 0 and out[i, 4] == 0, row


def test_score_matches_reference_csv_tables(eng):
    """The scoring tables the reference keeps as data files (its test CSV and the three data/*.csv files with explicit
    rolls, 448 rows): score, used and re-roll dice of every listed roll through the device score table."""
    import json

    rows = gu.load("scoring_vectors.json")["csv_rows"]
    assert len(rows) >= 448
    faces = np.zeros((len(rows), 6), dtype=np.uint8)
    lens = np.zeros(len(rows), dtype=np.int32)
    for i, row in enumerate(rows):
        roll = json.loads(row["Dice_Roll"])
        faces[i, :len(roll)] = roll
        lens[i] = len(roll)
    plain = _strats([[300, 2, 0, 0, 1, 1, 0, 0, 0, 1, 0]] * len(rows))
    out = eng.debug_score(faces, lens, np.zeros(len(rows), dtype=np.int32), plain)
    for i, row in enumerate(rows):
        assert (out[i, 0], out[i, 1], out[i, 2]) == (int(row["Score"]), int(row["Used_Dice"]), int(row["Reroll_Dice"])), row


def test_default_score_matches_golden_and_oracle(eng, po):
    data = gu.load("scoring_vectors.json")
    cases = data["default_score"]
    faces = np.zeros((len(cases), 6), dtype=np.uint8)
    for i, c in enumerate(cases):
        faces[i, :len(c["roll"])] = c["roll"]
    lens = np.array([len(c["roll"]) for c in cases], dtype=np.int32)
    pre = np.array([c["pre"] for c in cases], dtype=np.int32)
    out = eng.debug_score(faces, lens, pre, _strats([c["strategy"] for c in cases]))
    for i, c in enumerate(cases):
        assert out[i].tolist() == c["out"], c
    # exhaustive over the multiset table x a sample of the default grid x turn scores, against the oracle
    table = data["table"]
    grid = _default_table()
    rs = np.random.default_rng(5)
    pick = rs.choice(len(grid), size=48, replace=False)
    rolls, ls, pres, ss = [], [], [], []
    for row in table:
        roll = [f + 1 for f in range(6) for _ in range(row[f])]
        for gi in pick[: 12 if len(roll) < 3 else 48]:
            for p in (0, 250, 950):
                rolls.append(roll + [0] * (6 - len(roll)))
                ls.append(len(roll))
                pres.append(p)
                ss.append(gi)
    st = grid[np.array(ss)]
    out = eng.debug_score(np.array(rolls, dtype=np.uint8), ls, pres, st)
    for i in range(0, len(rolls), 5):
        exp = po.default_score(rolls[i][: ls[i]], pres[i], st[i : i + 1])
        assert tuple(out[i].tolist()) == exp, (rolls[i], pres[i], st[i])


def test_should_continue_matches_golden(eng):
    data = gu.load("scoring_vectors.json")
    cases = data["should_continue"]
    args = np.array([[c["turn_score"], c["dice_left"], c["has_scored"], c["final_round"], c["score_to_beat"],
                      c["player_score"]] for c in cases], dtype=np.int32)
    out = eng.debug_should_continue(args, _strats([c["strategy"] for c in cases]))
    assert out.tolist() == [c["out"] for c in cases]


# ------------------------------------------------------------------ single games
def test_game_rows_match_reference_vectors(eng):
    data = gu.load("game_vectors.json")
    tables = {"g64": _strats(data["grids"]["g64"]), "default": _default_table()}
    groups: dict = {}
    for g in data["games"]:
        groups.setdefault((g["grid"], g["k"], g["target"], g["max_rounds"]), []).append(g)
    for (grid, k, target, max_rounds), games in groups.items():
        table = tables[grid]
        coords = _coords([(g["purpose"], 0, g["root_seed"], k, g["shuffle"], g["pair"], g["order"], g["game"], 0, 0)
                          for g in games])
        seat = np.array([g["strategies"] for g in games], dtype=np.int32)
        rows = eng.play_games(coords, table, seat, k, target_score=target, max_rounds=max_rounds)
        for row, g in zip(rows, games):
            gu.assert_row_equal(gu.row_as_compact(row, k, lambda i: table[i]["strategy_id"]), g["row"], ctx=str(g["root_seed"]))


@pytest.mark.parametrize("k", [2, 3, 4, 5, 6, 8, 12])
def test_random_games_match_oracle(eng, po, k):
    table = _default_table()
    rs = np.random.default_rng(100 + k)
    n = 3000 if k <= 4 else 1200
    coords = _coords([(103, 0, int(rs.integers(0, 2**63)), k, int(rs.integers(0, 10**6)), 0, 0, int(rs.integers(0, 3000)), 0, 0)
                      for _ in range(n)])
    seat = np.stack([rs.choice(len(table), size=k, replace=False) for _ in range(n)]).astype(np.int32)
    rows = eng.play_games(coords, table, seat, k)
    ref = po.play_games(coords.view(po.COORD_DTYPE), table.view(po.STRATEGY_DTYPE), seat, k, n_threads=8)
    assert _rows_equal(rows, ref.view(rows.dtype))


def test_time_path_games(eng):
    data = gu.load("time_path_vectors.json")
    for block in [data["kat_counts"]] + data["many_games"]:
        table = _strats(block["strategies"])
        k = len(table)
        n = block["n_games"]
        coords = _coords([(10, 0, block["seed"], k, 0, 0, 0, i, 0, 0) for i in range(n)])
        rows = eng.play_games(coords, table, np.tile(np.arange(k, dtype=np.int32), n), k,
                              target_score=block.get("target", 10_000))
        for i, (row, gold) in enumerate(zip(rows, block["rows"])):
            gu.assert_row_equal(gu.row_as_compact(row, k, lambda j: table[j]["strategy_id"]), gold, ctx=f"game {i}")
    kat = data["kat_counts"]  # tests/unit/simulation/test_simulation.py:184-199
    table = _strats(kat["strategies"])
    coords = _coords([(10, 0, 123, 3, 0, 0, 0, i, 0, 0) for i in range(10)])
    rows = eng.play_games(coords, table, np.tile(np.arange(3, dtype=np.int32), 10), 3, target_score=5000)
    counts: dict = {}
    for row in rows:
        key = f"P{int(row['winner_seat']) + 1}"
        counts[key] = counts.get(key, 0) + 1
    assert counts == kat["expected"]


# ------------------------------------------------------------------ tournament
def test_tournament_matches_reference_vectors(eng):
    from farkle_ii_amd.backend import make_overrides

    data = gu.load("tournament_vectors.json")
    for case in data["cases"]:
        table = _strats(case["strategies"])
        ov = make_overrides([(11, 0, 0, 2, 0)]) if case["profile"] == "oracle" else None
        res = eng.tournament(table, case["k"], case["root_seed"], case["shuffle"], case["shuffle"] + 1,
                             target_score=case["target"], overrides=ov, want_rows=True, want_perms=True)
        assert res["perms"][0].tolist() == case["perm"], case["name"]
        ids = table["strategy_id"]
        gu.assert_tally_matches(res["tally"][0], ids, case["tally"], ctx=case["name"])
        for row, gold in zip(res["rows"], case["rows"]):
            gu.assert_row_equal(gu.row_as_compact(row, case["k"], lambda i: ids[i]), gold, ctx=case["name"])


def test_reference_expected_rows_on_gpu(eng):
    """EXPECTED_ROWS (tests/integration/test_raw_simulation_oracle.py:45-58) straight from the HIP path."""
    from farkle_ii_amd.backend import make_overrides

    data = gu.load("tournament_vectors.json")
    grid4 = _strats(gu.load("grid_vectors.json")["oracle4"])
    ov = make_overrides([(11, 0, 0, 2, 0)])
    for (root, k, shuffle, game), (seat_strats, status, winner_strategy, n_rounds, n_turns, scores) in data["EXPECTED_ROWS"]:
        res = eng.tournament(grid4, k, root, shuffle, shuffle + 1, target_score=100, overrides=ov, want_rows=True)
        row = res["rows"][game]
        assert [int(row["seats"][i]["strategy"]) for i in range(k)] == seat_strats
        assert ("completed", "safety_limit")[int(row["status"])] == status
        w = int(row["winner_seat"])
        assert (None if w < 0 else int(row["seats"][w]["strategy"])) == winner_strategy
        assert int(row["n_rounds"]) == n_rounds
        assert sum(int(row["seats"][i]["n_turns"]) for i in range(k)) == n_turns
        assert [int(row["seats"][i]["score"]) for i in range(k)] == scores


@pytest.mark.parametrize("k,n_shuffles", [(2, 600), (4, 300), (8, 150)])
def test_tournament_g64_matches_oracle(eng, po, k, n_shuffles):
    table = _strats(gu.load("grid_vectors.json")["g64"])
    res = eng.tournament(table, k, 42, 0, n_shuffles, want_rows=True, want_perms=True)
    ref = po.tournament(table.view(po.STRATEGY_DTYPE), k, 42, 0, n_shuffles, want_rows=True, want_perms=True, n_threads=8)
    assert np.array_equal(res["perms"], ref["perms"])
    assert _rows_equal(res["rows"], ref["rows"].view(res["rows"].dtype))
    assert np.array_equal(res["tally"], ref["tally"])


def test_tournament_default_grid_matches_oracle(eng, po):
    table = _default_table()
    for k, n_sh, root in [(4, 6, 0), (2, 4, 42), (6, 4, 9), (12, 3, 5)]:
        res = eng.tournament(table, k, root, 10, 10 + n_sh, want_rows=True, want_perms=True)
        ref = po.tournament(table.view(po.STRATEGY_DTYPE), k, root, 10, 10 + n_sh, want_rows=True, want_perms=True, n_threads=8)
        assert np.array_equal(res["perms"], ref["perms"]), k
        assert _rows_equal(res["rows"], ref["rows"].view(res["rows"].dtype)), k
        assert np.array_equal(res["tally"], ref["tally"]), k


def test_tournament_batches_chunks_options_and_overrides(eng, po):
    """Same results whatever the launch geometry: per-batch tallies, workspace chunking, LDS vs global
    tally, hand-over threshold, block size; max_rounds overrides land on the right (shuffle, game)."""
    from farkle_ii_amd.backend import make_overrides

    table = _strats(gu.load("grid_vectors.json")["g64"])
    ovs = [(42, 3, 5, 2, 0), (42, 17, 31, 2, 1), (42, 40, 0, 2, 7), (42, 40, 1, 4, 0), (7, 3, 5, 2, 0)]
    ref = po.tournament(table.view(po.STRATEGY_DTYPE), 2, 42, 0, 50, shuffles_per_batch=7, overrides=po.make_overrides(ovs),
                        want_rows=True, n_threads=8)
    base = eng.tournament(table, 2, 42, 0, 50, shuffles_per_batch=7, overrides=make_overrides(ovs), want_rows=True)
    assert np.array_equal(base["tally"], ref["tally"]) and base["tally"].shape[0] == 8
    assert _rows_equal(base["rows"], ref["rows"].view(base["rows"].dtype))
    try:
        for name, value in [("chunk_bytes", 1 << 20), ("batch_threshold", 1), ("batch_threshold", 64), ("block", 256),
                            ("block", 64), ("use_lds_tally", 0), ("lean", 0), ("lean", 1), ("block", 768), ("block", 1024),
                            ("longest_first", 0), ("uniform_flags", 0), ("block", 0), ("max_waves", 3), ("max_waves", 8)]:
            eng.set_option(name, value)
            got = eng.tournament(table, 2, 42, 0, 50, shuffles_per_batch=7, overrides=make_overrides(ovs), want_rows=True)
            assert np.array_equal(got["tally"], ref["tally"]), (name, value)
            assert _rows_equal(got["rows"], base["rows"]), (name, value)
            one = eng.tournament(table, 2, 42, 0, 50, overrides=make_overrides(ovs))
            assert np.array_equal(one["tally"][0], ref["tally"].sum(axis=0)), (name, value)
    finally:
        for name, value in [("chunk_bytes", 24 << 30), ("batch_threshold", 0), ("block", 0), ("use_lds_tally", -1), ("lean", -1),
                            ("longest_first", 1), ("uniform_flags", -1), ("max_waves", 6)]:
            eng.set_option(name, value)


def test_scalar_flag_kernel_instances_match_oracle(eng, po):
    """The three flag-dispatch instances of the game kernel (no flag varies / only require_both and favor vary / any
    flag varies) against the oracle, plus the generic instance forced on the same tables."""
    base = _strats(gu.load("grid_vectors.json")["g64"])
    uniform = base.copy()                      # every strategy shares all eight flags: scalar-flag instance
    uniform["require_both"], uniform["favor_score"] = 1, 0
    uniform["score_threshold"] = 250 + 25 * (np.arange(64) // 4)
    uniform["dice_threshold"] = np.arange(64) % 4
    mixed = base.copy()                        # a third flag varies: generic instance
    mixed["auto_hot_dice"][::3] = 0
    mixed["run_up_score"][::5] = 0
    for name, table in (("uniform", uniform), ("rb_fav", base), ("mixed", mixed)):
        ref = po.tournament(table.view(po.STRATEGY_DTYPE), 2, 9, 0, 40, n_threads=8)["tally"]
        assert np.array_equal(eng.tournament(table, 2, 9, 0, 40)["tally"], ref), name
        try:
            eng.set_option("uniform_flags", 0)
            assert np.array_equal(eng.tournament(table, 2, 9, 0, 40)["tally"], ref), name
        finally:
            eng.set_option("uniform_flags", -1)


def _random_valid_table(n: int, seed: int) -> np.ndarray:
    from farkle_ii_amd.strategies import STRATEGY_DTYPE

    rng = np.random.default_rng(seed)
    t = np.zeros(n, dtype=STRATEGY_DTYPE)
    t["score_threshold"] = rng.integers(2, 21, n) * 50
    t["dice_threshold"] = rng.integers(0, 5, n)
    for name in ("smart_five", "consider_score", "consider_dice", "auto_hot_dice", "run_up_score", "favor_score"):
        t[name] = rng.integers(0, 2, n)
    t["smart_one"] = t["smart_five"] & rng.integers(0, 2, n).astype(np.uint8)                     # strategies.py:198
    t["require_both"] = t["consider_score"] & t["consider_dice"] & rng.integers(0, 2, n).astype(np.uint8)  # :202
    t["strategy_id"] = np.arange(n)
    return t


def test_strategy_table_size_limits(eng, po):
    """Largest table the ABI accepts (S = 65 534 at k = 2), the sizes either side of the lean-record limit
    (strategy index in 14 bits: S <= 16 384) and one strategy too many."""
    from farkle_ii_amd.backend import FarkleHipError

    for S, k, n_sh in ((65_534, 2, 2), (16_384, 2, 3), (16_386, 2, 3), (16_386, 6, 4), (65_532, 4, 2)):  # (k = 4, 6: 16-bit indices of fk_play_hc_kernel)
        table = _random_valid_table(S, S + k)
        ref = po.tournament(table.view(po.STRATEGY_DTYPE), k, 5, 0, n_sh, want_rows=True, n_threads=8)
        got = eng.tournament(table, k, 5, 0, n_sh, want_rows=True)
        assert np.array_equal(got["tally"], ref["tally"]), (S, k)
        assert _rows_equal(got["rows"], ref["rows"].view(got["rows"].dtype)), (S, k)
    with pytest.raises(FarkleHipError):
        eng.tournament(_random_valid_table(65_536, 1), 2, 5, 0, 1)


def test_tournament_full_size_properties(eng):
    """BASELINE config 2 at full size (k=2, 64-strategy grid, 10^7 games, seed 42): size-independent
    properties of the tally (the oracle cannot play 10^7 games inside a test)."""
    table = _strats(gu.load("grid_vectors.json")["g64"])
    n_sh = 312_500
    full = eng.tournament(table, 2, 42, 0, n_sh)["tally"][0]
    assert np.all(full[:, 1] == n_sh)                              # every strategy seated once per shuffle
    assert np.array_equal(full[:, 1], full[:, 2] + full[:, 3])     # attempted = completed + safety
    assert full[:, 0].sum() * 2 == full[:, 2].sum()                # one winner per completed game
    assert 0 < full[:, 3].sum() < 0.05 * full[:, 1].sum()          # never-bank pairings reach the safety limit
    assert np.all(full[:, 14] == 0) and np.all(full[:, 25] == 0)   # winners never carry hit_max_rounds
    mean_score = full[:, 4] / np.maximum(full[:, 0], 1)
    assert np.all((mean_score >= 10_000) | (full[:, 0] == 0))      # winning_score >= target
    assert np.all(full[:, 15] >= full[:, 4])                       # sum of squares dominates the sum
    # additivity over a partition of the shuffle range + run-to-run determinism
    a = eng.tournament(table, 2, 42, 0, 100_000)["tally"][0]
    b = eng.tournament(table, 2, 42, 100_000, n_sh)["tally"][0]
    assert np.array_equal(a + b, full)
    again = eng.tournament(table, 2, 42, 0, n_sh)["tally"][0]
    assert np.array_equal(again, full)


# ------------------------------------------------------------------ H2H
def test_h2h_blocks_match_goldens_and_oracle(eng, po):
    from farkle_ii_amd.backend import make_overrides

    data = gu.load("h2h_vectors.json")
    grid4 = _strats(data["oracle4"])
    ov = make_overrides([(11, 0, 0, 0, 0), (11, 1, 0, 0, 0), (11, 1, 1, 0, 0)])
    expected = {tuple(k): v for k, v in data["EXPECTED_H2H_BLOCKS"]}
    for b in data["blocks"]:
        seats = grid4[[b["seat1_strategy"], b["seat2_strategy"]]]
        st = eng.h2h(seats, b["root_seed"], b["pair_id"], b["order"], 1, 2, 5000, target_score=100, overrides=ov)
        attempted, completed, safety, w1, w2 = (int(v) for v in st)
        wins_a, wins_b = (w1, w2) if b["order"] == 0 else (w2, w1)
        status = "complete" if completed >= 1 else ("unresolved_nonviable" if attempted >= 2 else "partial_resumable")
        assert [attempted, completed, safety, wins_a, wins_b, max(0, attempted - 1), status] == \
            expected[(b["pair_id"], b["root_seed"], b["order"])]
    g64 = _strats(gu.load("grid_vectors.json")["g64"])
    for b in data["g64_blocks"]:
        seats = g64[[b["seat1_strategy"], b["seat2_strategy"]]]
        state = None
        for step in b["trace"]:
            state = eng.h2h(seats, b["root_seed"], b["pair_id"], b["order"], b["n_completed_required"], b["max_attempts"],
                            b["chunk"], state=state)
            assert [int(v) for v in state] == step
    # a never-bank pairing: many safety-limit replacements, larger block, against the oracle
    never = [i for i, s in enumerate(g64) if s["dice_threshold"] == 0 and not s["require_both"]]
    for seats_idx, target, max_att in [((never[0], 5), 3000, 6000), ((never[0], never[1]), 10, 400), ((3, 40), 20_000, 40_000)]:
        seats = g64[list(seats_idx)]
        got = eng.h2h(seats, 42, 9, 1, target, max_att, 10**9)
        exp = po.h2h_block(seats.view(po.STRATEGY_DTYPE), 42, 9, 1, target, max_att, 10**9)
        assert np.array_equal(got, exp), seats_idx


# ------------------------------------------------------------------ error behaviour
def test_argument_errors(eng):
    from farkle_ii_amd.backend import FarkleHipError

    table = _strats(gu.load("grid_vectors.json")["g64"])
    with pytest.raises(FarkleHipError, match="n_players must divide"):
        eng.tournament(table, 3, 0, 0, 1)  # run_tournament.py:274
    with pytest.raises(FarkleHipError):
        eng.tournament(table, 2, 0, 0, 1, max_rounds=70_000)
    bad = table.copy()
    bad[0]["smart_one"], bad[0]["smart_five"] = 1, 0
    with pytest.raises(FarkleHipError, match="smart_one"):
        eng.tournament(bad, 2, 0, 0, 1)
    # require_both without both considerations: ThresholdStrategy.__post_init__ raises (strategies.py:201-207); the
    # branch-free discard search relies on this invariant, so every entry point must refuse the record
    for field in ("consider_score", "consider_dice"):
        bad = table.copy()
        bad[5]["require_both"], bad[5][field] = 1, 0
        with pytest.raises(FarkleHipError, match="require_both"):
            eng.tournament(bad, 2, 0, 0, 1)
        with pytest.raises(FarkleHipError, match="require_both"):
            eng.h2h(bad[4:6], 1, 0, 0, 10, 20, 20)
        with pytest.raises(FarkleHipError, match="require_both"):
            eng.debug_score(np.ones((1, 6), np.uint8), [6], [0], bad[5:6])
    empty = eng.tournament(table, 2, 0, 5, 5)
    assert empty["tally"].shape[0] == 0


# ------------------------------------------------------------------ BASELINE configs 3-5 at full size: properties
def test_config3_full_size_properties(eng):
    """BASELINE config 3 (k=4, 5 160-strategy grid, 10^8 games = 77 520 shuffles x 1 290, root seed 0): conservation,
    additivity over a partition of the shuffle range, per-batch tallies summing to the single-batch tally."""
    table = _default_table()
    n_sh = 77_520
    full = eng.tournament(table, 4, 0, 0, n_sh)["tally"][0]
    assert np.all(full[:, 1] == n_sh) and np.array_equal(full[:, 1], full[:, 2] + full[:, 3])
    assert full[:, 0].sum() * 4 == full[:, 2].sum() and full[:, 2].sum() + full[:, 3].sum() == n_sh * 5160
    assert np.all(full[:, 14] == 0) and np.all(full[:, 15] >= full[:, 4])
    won = full[:, 0] > 0
    assert np.all(full[won, 4] >= 10_000 * full[won, 0])                     # every winning score reaches the target
    assert np.all(full[won, 5] <= 200 * full[won, 0])                        # n_rounds <= max_rounds
    a = eng.tournament(table, 4, 0, 0, 30_000)["tally"][0]
    b = eng.tournament(table, 4, 0, 30_000, n_sh, shuffles_per_batch=10_000)["tally"]
    assert b.shape[0] == 5 and np.array_equal(a + b.sum(axis=0), full)


def test_config4_player_count_sweep_properties(eng):
    """BASELINE config 4 shape (k in {2,4,6,8} on the 5 160 grid) at 1/100 scale: exposure conservation per k."""
    table = _default_table()
    for k in (2, 4, 6, 8):
        n_sh = 2_500_000 // (5160 // k)
        t = eng.tournament(table, k, 7, 0, n_sh)["tally"][0]
        assert np.all(t[:, 1] == n_sh) and np.array_equal(t[:, 1], t[:, 2] + t[:, 3]), k
        assert t[:, 0].sum() * k == t[:, 2].sum(), k


def test_config4_full_size_launches_are_additive_over_a_split(eng):
    """BASELINE config 4 at FULL size for all four of its player counts: 2.5 x 10^8 games of one call (k = 2: one launch of
    2.5 x 10^8 tickets; k = 8: two chunks of 1.25 x 10^8) — the 32-bit ticket / game-id arithmetic at that size.  Exposure
    conservation, and the whole range against the sum of an uneven three-way split of it (a lost, duplicated or misaddressed
    game changes a tally).  (k = 4 and 6 joined in round 6: ~2 s of GPU time each.)"""
    table = _default_table()
    for k in (2, 4, 6, 8):
        n_sh = 250_000_000 // (5160 // k)
        whole = eng.tournament(table, k, 0, 0, n_sh)["tally"][0]
        assert eng.timing()["games"] == n_sh * (5160 // k)
        assert np.all(whole[:, 1] == n_sh) and np.array_equal(whole[:, 1], whole[:, 2] + whole[:, 3]), k
        assert whole[:, 0].sum() * k == whole[:, 2].sum(), k
        cuts = [0, n_sh // 3 + 1, n_sh - 12_345, n_sh]
        parts = sum(eng.tournament(table, k, 0, a, b)["tally"][0] for a, b in zip(cuts[:-1], cuts[1:]))
        assert np.array_equal(parts, whole), k


def test_config5_h2h_full_size_block(eng):
    """BASELINE config 5 shape: one pairing at 10^8 completed games; chunked execution reaches the same state."""
    g64 = _strats(gu.load("grid_vectors.json")["g64"])
    seats = g64[[3, 40]]
    target = 100_000_000
    one = eng.h2h(seats, 42, 5, 0, target, 2 * target, 10**12)
    attempted, completed, safety, w1, w2 = (int(v) for v in one)
    assert completed == target and attempted == completed + safety and w1 + w2 == completed
    state = None
    for _ in range(8):
        state = eng.h2h(seats, 42, 5, 0, target, 2 * target, 15_000_000, state=state)
    assert np.array_equal(state, one)
    swapped = eng.h2h(seats[::-1].copy(), 42, 5, 1, target, 2 * target, 10**12)  # order 1: strategy b in seat 1
    assert int(swapped[1]) == target and int(swapped[3]) + int(swapped[4]) == target
    # strategy a keeps winning the majority from either seat (it wins 76 % from seat 1)
    assert w1 > w2 and int(swapped[4]) > int(swapped[3])


# ------------------------------------------------------------------ randomized differential test
def test_fuzz_random_tables_limits_and_player_counts(eng, po):
    """Random legal strategy tables (extreme thresholds, every flag combination), player counts 1..12, targets and round
    limits, small and large shuffle counts: rows, permutations and tallies bit-identical to the oracle."""
    from farkle_ii_amd.strategies import STRATEGY_DTYPE

    rs = np.random.default_rng(2026)
    for trial in range(60):
        k = int(rs.choice([1, 2, 2, 3, 4, 5, 6, 7, 8, 10, 12]))
        S = k * int(rs.integers(1, 9))
        table = np.zeros(S, dtype=STRATEGY_DTYPE)
        for i in range(S):
            sf = int(rs.integers(0, 2))
            so = int(rs.integers(0, 2)) if sf else 0
            cs, cd = int(rs.integers(0, 2)), int(rs.integers(0, 2))
            rb = int(rs.integers(0, 2)) if (cs and cd) else 0
            table[i] = (int(rs.choice([0, 1, 49, 50, 51, 199, 250, 300, 500, 1000, 1001, 1350, 10_000])), int(rs.integers(-1, 7)), sf, so, cs, cd, rb,
                        int(rs.integers(0, 2)), int(rs.integers(0, 2)), int(rs.integers(0, 2)), 1000 + i)
        target = int(rs.choice([49, 100, 500, 1_234, 2000, 9_999, 10_000, 10_001, 20_000]))  # the kernels round to units of 50
        max_rounds = int(rs.choice([0, 1, 3, 50, 200, 300]))
        n_sh = int(rs.choice([1, 2, 7, 40]))
        root = int(rs.integers(0, 2**63))
        first = int(rs.integers(0, 2**40))
        got = eng.tournament(table, k, root, first, first + n_sh, shuffles_per_batch=3, target_score=target, max_rounds=max_rounds,
                             want_rows=True, want_perms=True)
        ref = po.tournament(table.view(po.STRATEGY_DTYPE), k, root, first, first + n_sh, shuffles_per_batch=3, target_score=target,
                            max_rounds=max_rounds, want_rows=True, want_perms=True, n_threads=4)
        ctx = (trial, k, S, target, max_rounds, n_sh)
        assert np.array_equal(got["perms"], ref["perms"]), ctx
        assert _rows_equal(got["rows"], ref["rows"].view(got["rows"].dtype)), ctx
        assert np.array_equal(got["tally"], ref["tally"]), ctx


def test_random_table_games_match_reference_vectors(eng):
    for g in gu.load("fuzz_vectors.json")["games"]:
        table = _strats(g["strategies"])
        k = g["k"]
        coords = _coords([(103, 0, g["root_seed"], k, g["shuffle"], 0, 0, g["game"], 0, 0)])
        row = eng.play_games(coords, table, np.arange(k, dtype=np.int32)[None, :], k, target_score=g["target"],
                             max_rounds=g["max_rounds"])[0]
        gu.assert_row_equal(gu.row_as_compact(row, k, lambda i: table[i]["strategy_id"]), g["row"], ctx=str(g["root_seed"]))
