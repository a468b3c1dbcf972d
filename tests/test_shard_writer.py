"""Row shards without a host-side conversion: the device's per-shuffle column images (``fk_tournament_run_columns``) and the library's own
Parquet writer (``fk_write_row_shards``, csrc/fk_shard_writer.h).

What the reference fixes is the TABLE of a shard and its Arrow schema (``raw_simulation_schema_for(k)``, utils/schema_helpers.py:23-90;
one file per shuffle, run_tournament.py:530-558) — `analyze ingest` reads the files with Arrow (oracle/gen_contract_v3.py runs it over
natively written shards).  So the bar here: a natively written file, read back by Arrow, EQUALS ``rows.rows_to_table`` of the same games
(the conversion every earlier round pinned against the reference's frozen rows), schema metadata included; sizes and SHA-256 are the
files'; the contract-v3 sidecar the library writes is the text ``contract_v3.fill_shard_template`` produces.  On the GPU: the kernel's
images are byte-equal to the NumPy restatement over the AoS rows of the same launch, for k in {2, 5, 12} with safety-limit games.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import struct

import numpy as np
import pytest

from oracle_engine_stub import column_images


def _random_rows(rng, k: int, gps: int, n_sh: int, null_rate: float):
    from farkle_ii_amd.backend import row_dtype

    n = gps * n_sh
    rows = np.zeros(n, dtype=row_dtype(k))
    rows["n_rounds"] = rng.integers(0, 201, n)
    rows["status"] = rng.random(n) < null_rate
    rows["seats"]["score"] = rng.integers(0, 40, (n, k)) * 50  # (ties are frequent: ranks and margins must follow the stable sort)
    for name in ("farkles", "rolls", "n_turns", "smart_five_uses", "n_smart_five_dice", "smart_one_uses", "n_smart_one_dice", "hot_dice"):
        rows["seats"][name] = rng.integers(0, 2048, (n, k))
    rows["seats"]["highest_turn"] = rng.integers(0, 600, (n, k)) * 50
    rows["seats"]["strategy"] = rng.integers(0, 5160, (n, k))
    completed = rows["status"] == 0
    rank = np.argsort(np.argsort(-rows["seats"]["score"].astype(np.int64), axis=1, kind="stable"), axis=1, kind="stable") + 1
    rows["seats"]["rank"] = np.where(completed[:, None], rank, 0)
    rows["winner_seat"] = np.where(completed, np.argmin(rank, axis=1), 255)
    rows["seats"]["hit_max_rounds"] = ~completed[:, None]
    return rows


def _reference_table(rows, k, ids, gps, sh, seeds, batch, game_seeds, root_seed=102):
    from farkle_ii_amd.rows import rows_to_table

    n_sh = len(sh)
    return rows_to_table(rows, k, ids, root_seed=root_seed, shuffle_index=np.repeat(sh, gps), game_index=np.tile(np.arange(gps, dtype=np.int32), n_sh),
                         deterministic_batch_id=np.repeat(batch, gps), shuffle_seed=np.repeat(seeds, gps),
                         game_seed=game_seeds.astype(np.int64), rng_purpose_namespace=102)


def test_sha256_of_the_shard_writer_matches_hashlib():
    from farkle_ii_amd.backend import load_library

    lib = load_library()
    rng = np.random.default_rng(7)
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 127, 128, 129, 257, 1000, 65536 + 17, 400_001):
        data = rng.integers(0, 256, n, dtype=np.uint8)
        for portable in (0, 1):  # SHA-NI (when the CPU has it) and the scalar rounds
            out = np.zeros(32, dtype=np.uint8)
            assert lib.fk_debug_sha256(data.ctypes.data_as(C.c_void_p), C.c_size_t(n), out.ctypes.data_as(C.c_void_p), C.c_int32(portable)) == 0
            assert out.tobytes() == hashlib.sha256(data.tobytes()).digest(), (n, portable)
        # the two-message lockstep form (what a writer thread runs over the two files it has just built): unequal halves, short tails
        pair = np.zeros(64, dtype=np.uint8)
        assert lib.fk_debug_sha256(data.ctypes.data_as(C.c_void_p), C.c_size_t(n), pair.ctypes.data_as(C.c_void_p), C.c_int32(2)) == 0
        blob = data.tobytes()
        assert pair[:32].tobytes() == hashlib.sha256(blob[:n // 2]).digest() and pair[32:].tobytes() == hashlib.sha256(blob[n // 2:]).digest(), n


@pytest.mark.parametrize("k,gps,n_sh,null_rate", [(2, 32, 6, 0.2), (5, 7, 3, 0.5), (12, 430, 3, 0.02), (3, 1, 4, 0.5), (2, 2580, 2, 0.0),
                                                  (4, 9, 3, 1.0), (8, 16, 2, 0.1), (1, 5, 2, 0.3)])
def test_native_shards_read_back_as_the_reference_tables(tmp_path, k, gps, n_sh, null_rate):
    import pyarrow as pa
    import pyarrow.parquet as pq

    from farkle_ii_amd import tournament as rt
    from farkle_ii_amd.backend import write_row_shards_native
    from farkle_ii_amd.parquet_template import footer_of, read_struct

    rng = np.random.default_rng(k * 1000 + gps)
    rows = _random_rows(rng, k, gps, n_sh, null_rate)
    ids = (np.arange(5160, dtype=np.int32) * 7 + 3)
    sh = np.arange(10**9, 10**9 + n_sh, dtype=np.int64)
    seeds = rng.integers(0, 2**32, n_sh).astype(np.int64)
    batch = (sh // 3 % 1000).astype(np.int32)
    game_seeds = rng.integers(0, 2**32, gps * n_sh, dtype=np.uint64).astype(np.uint32)
    want = _reference_table(rows, k, ids, gps, sh, seeds, batch, game_seeds)
    images = column_images(rows, k, ids, gps)
    res = write_row_shards_native(tmp_path, k, 102, images, sh, seeds, batch, game_seeds, 102, threads=3)
    assert not list(tmp_path.glob("*.tmp"))
    for i in range(n_sh):
        path = tmp_path / f"rows_102_{k}p_{sh[i]:012d}.parquet"
        blob = path.read_bytes()
        assert len(blob) == res["byte_length"][i] and hashlib.sha256(blob).digest() == res["sha256"][i].tobytes()
        got = pq.read_table(path)
        shard = want.slice(i * gps, gps)
        assert got.schema.equals(shard.schema, check_metadata=False) and got.equals(shard), (k, gps, i)
        # the Arrow schema a reader sees — what contract v3 fingerprints — is the Arrow-written shard's
        sink = pa.BufferOutputStream()
        pq.write_table(shard, sink, **rt.SHARD_WRITER_OPTIONS)
        assert pq.read_schema(path).equals(pq.read_schema(pa.BufferReader(sink.getvalue())), check_metadata=True)
        # the footer is well-formed Thrift: one row group of gps rows, 18 + 14 k uncompressed column chunks that tile the file
        meta, end = read_struct(footer_of(blob))
        assert end == struct.unpack("<I", blob[-8:-4])[0] and meta[3][0] == gps and meta[6][0].startswith(b"farkle_ii_amd shard writer")
        (group,) = meta[4][0]
        chunks = [c[3][0] for c in group[1][0]]
        assert len(chunks) == 18 + 14 * k and group[3][0] == gps
        cursor = 4
        for c in chunks:
            first = c[11][0] if 11 in c else c[9][0]
            assert c[4][0] == 0 and first == cursor and c[6][0] == c[7][0]
            cursor += c[6][0]
        assert cursor == len(blob) - 8 - end and group[2][0] == cursor - 4
    md = pq.read_metadata(tmp_path / f"rows_102_{k}p_{sh[0]:012d}.parquet")
    assert md.num_rows == gps and md.num_row_groups == 1 and md.num_columns == 18 + 14 * k


@pytest.mark.parametrize("k,gps", [(2, 1), (2, 2), (3, 33), (2, 129), (1, 130), (5, 257), (2, 2580)])
def test_delta_packed_columns_round_trip_any_int32(tmp_path, k, gps):
    """The int32 planes are written DELTA_BINARY_PACKED (blocks of 128 deltas, miniblock widths 0 / 8 / 16 / 32): every column reads back
    as the plane — full-range values whose deltas wrap modulo 2^32, constant runs (width 0), lengths around the block and miniblock
    edges, nullable columns with none, some or all rows null — in its logical type's range for the int8 / int16 columns."""
    import pyarrow.parquet as pq

    from farkle_ii_amd.backend import row_columns_bytes, write_row_shards_native
    from farkle_ii_amd.rows import raw_simulation_schema_for

    rng = np.random.default_rng(gps * 31 + k)
    n_sh = 4
    stride = row_columns_bytes(k, gps)
    n_planes = 4 + 13 * k
    names = ["winner_strategy", "winning_score", "victory_margin", "n_rounds"]
    for s in range(1, k + 1):
        names += [f"P{s}_{f}" for f in ("score", "farkles", "rolls", "highest_turn", "strategy", "rank", "loss_margin", "smart_five_uses",
                                        "n_smart_five_dice", "smart_one_uses", "n_smart_one_dice", "hot_dice", "n_turns")]
    schema = raw_simulation_schema_for(k)
    limits = {"int8": 2**7, "int16": 2**15, "int32": 2**31}
    images = np.zeros((n_sh, stride), dtype=np.uint8)
    planes = np.zeros((n_sh, n_planes, gps), dtype=np.int32)
    status = np.zeros((n_sh, gps), dtype=np.uint8)
    for i in range(n_sh):
        status[i] = (rng.random(gps) < (0.0, 0.3, 1.0, 0.05)[i])
        for c, name in enumerate(names):
            lim = limits[str(schema.field(name).type)]
            kind = (c + i) % 5
            if kind == 0:
                v = rng.integers(-lim, lim, gps)                      # the whole range: deltas wrap
            elif kind == 1:
                v = np.full(gps, rng.integers(-lim, lim))              # constant: every width 0
            elif kind == 2:
                v = rng.integers(0, min(lim, 200), gps)                # a counter: 8-bit miniblocks
            elif kind == 3:
                v = np.where(rng.random(gps) < 0.5, lim - 1, -lim)     # the extremes, alternating at random
            else:
                v = np.cumsum(rng.integers(0, 3, gps)) % lim           # slowly rising; one jump in some miniblock
                if gps > 40:
                    v[37] = lim - 1
            planes[i, c] = v
        ni = n_planes * 4 * gps
        images[i, :ni] = planes[i].view(np.uint8).reshape(-1)
        images[i, ni:ni + gps] = status[i]
        images[i, ni + gps:ni + 2 * gps] = 0                                            # winner seat
        images[i, ni + 2 * gps:ni + 2 * gps + gps * k] = np.tile(np.arange(k, dtype=np.uint8), gps)  # rank order
    sh = np.arange(n_sh, dtype=np.int64)
    write_row_shards_native(tmp_path, k, 9, images, sh, sh + 1, sh.astype(np.int32), rng.integers(0, 2**32, gps * n_sh, dtype=np.uint64).astype(np.uint32),
                            102, threads=2)
    nullable = {"winner_strategy", "winning_score", "victory_margin"} | {f"P{s}_{f}" for s in range(1, k + 1) for f in ("rank", "loss_margin")}
    for i in range(n_sh):
        got = pq.read_table(tmp_path / f"rows_9_{k}p_{i:012d}.parquet")
        assert got.schema.equals(schema, check_metadata=False)
        assert got.column("game_index").to_pylist() == list(range(gps))
        for c, name in enumerate(names):
            col = got.column(name)
            want = planes[i, c].tolist()
            if name in nullable:
                want = [None if st else v for v, st in zip(want, status[i].tolist())]
            assert col.to_pylist() == want, (k, gps, i, name)
        meta = pq.ParquetFile(tmp_path / f"rows_9_{k}p_{i:012d}.parquet").metadata.row_group(0)
        encodings = {meta.column(j).path_in_schema: meta.column(j).encodings for j in range(meta.num_columns)}
        assert "DELTA_BINARY_PACKED" in encodings["P1_score"] and "DELTA_BINARY_PACKED" in encodings["game_index"]


def test_native_writer_publishes_the_contract_v3_sidecar(tmp_path):
    """The sidecar the library writes per shard = contract_v3.fill_shard_template(template, name, size, sha256): same text, same digest."""
    import yaml

    from farkle_ii_amd import contract_v3 as c3
    from farkle_ii_amd.backend import write_row_shards_native
    from farkle_ii_amd.config import load_app_config
    from farkle_ii_amd.rows import raw_simulation_schema_for
    from farkle_ii_amd.runner import _stored_schema
    from farkle_ii_amd import tournament as rt

    cfg_path = tmp_path / "c.yaml"
    cfg_path.write_text(yaml.safe_dump({"io": {"results_dir_prefix": str(tmp_path / "out")}, "sim": {"n_players_list": [3], "seed_list": [5], "row_dir": "rows"}}))
    cfg = load_app_config(cfg_path, seed_list_len=1)
    sc = c3.SimulationContract(cfg, c3.make_code_identity("a" * 40))
    row_dir = cfg.simulation_row_dir(3)
    row_dir.mkdir(parents=True)
    template = sc.shard_template("row_shard", row_dir, _stored_schema(raw_simulation_schema_for(3), **rt.SHARD_WRITER_OPTIONS), n_players=3, sources=[])
    rng = np.random.default_rng(3)
    gps, n_sh = 11, 5
    rows = _random_rows(rng, 3, gps, n_sh, 0.3)
    ids = np.arange(5160, dtype=np.int32)
    sh = np.arange(n_sh, dtype=np.int64)
    res = write_row_shards_native(row_dir, 3, 5, column_images(rows, 3, ids, gps), sh, sh + 9, sh.astype(np.int32),
                                  rng.integers(0, 2**32, gps * n_sh, dtype=np.uint64).astype(np.uint32), 102, threads=2, sidecar=template)
    for i in range(n_sh):
        name = f"rows_5_3p_{i:012d}.parquet"
        text, side_sha = c3.fill_shard_template(template, name, int(res["byte_length"][i]), res["sha256"][i].tobytes().hex())
        assert (row_dir / (name + ".sidecar.json")).read_bytes() == text
        assert res["sidecar_sha256"][i].tobytes().hex() == side_sha
        doc = __import__("json").loads(text)
        assert doc["artifact"]["location"]["relative_path"] == f"3_players/3p_rows/{name}" and doc["artifact"]["byte_length"] == res["byte_length"][i]


def test_native_writer_reports_io_errors(tmp_path):
    from farkle_ii_amd.backend import write_row_shards_native

    rows = _random_rows(np.random.default_rng(1), 2, 4, 2, 0.0)
    images = column_images(rows, 2, np.arange(5160, dtype=np.int32), 4)
    with pytest.raises(OSError, match="open .*no_such_dir"):
        write_row_shards_native(tmp_path / "no_such_dir", 2, 1, images, [0, 1], [1, 2], [0, 0], np.zeros(8, dtype=np.uint32), 102)
    with pytest.raises(ValueError):
        write_row_shards_native(tmp_path, 2, 1, images, [0, 1, 2], [1, 2], [0, 0], np.zeros(8, dtype=np.uint32), 102)


@pytest.mark.gpu
@pytest.mark.parametrize("k,max_rounds", [(2, 200), (5, 3), (12, 200), (12, 2), (4, 0)])
def test_column_images_of_the_kernel_equal_the_rows_of_the_same_launch(tmp_path, k, max_rounds):
    """fk_tournament_run_columns vs fk_tournament_run(rows) on the 5 160-strategy grid (production shard shape) — max_rounds 3 / 2 / 0 make
    most or all games safety-limit games — then the native writer's files against rows_to_table."""
    import pyarrow.parquet as pq

    from farkle_ii_amd import random as urandom
    from farkle_ii_amd.backend import write_row_shards_native
    from farkle_ii_amd.engine import get_engine, set_engine
    from farkle_ii_amd.strategies import generate_strategy_grid, pack_strategies, prepare_public_helper_strategies

    set_engine(None)
    eng = get_engine()
    strategies = prepare_public_helper_strategies(generate_strategy_grid()[0])
    table = pack_strategies(strategies)
    ids = np.asarray([int(s.strategy_id) for s in strategies], dtype=np.int32)[::-1].copy()  # (any id per table row: not the row index)
    S, gps = len(table), len(table) // k
    lo, hi = 1000, 1007
    eng.set_option("rows_chunk_games", 3 * gps)  # several chunks: images of chunk i cross PCIe while chunk i + 1 plays
    try:
        rows = eng.tournament(table, k, 42, lo, hi, max_rounds=max_rounds, want_rows=True)
        cols = eng.tournament_columns(table, k, 42, lo, hi, ids, max_rounds=max_rounds)  # (one thread per (game, seat): the default to 16 seats)
        eng.set_option("columns_by_seat", 0)
        per_game = eng.tournament_columns(table, k, 42, lo, hi, ids, max_rounds=max_rounds)  # fk_row_columns_kernel: one thread per game
    finally:
        eng.set_option("rows_chunk_games", 4_000_000)
        eng.set_option("columns_by_seat", -1)
    assert np.array_equal(rows["tally"], cols["tally"])
    status = rows["rows"]["status"]
    if max_rounds <= 3:
        assert (status != 0).mean() > 0.5
    want_images = column_images(rows["rows"], k, ids, gps)
    assert cols["columns"].shape == want_images.shape
    defined = ((4 + 13 * k) * 4 + 2 + k) * gps  # (an image is padded to a multiple of 64 bytes; nothing reads the padding)
    cols["columns"][:, defined:] = 0
    assert np.array_equal(per_game["columns"][:, :defined], cols["columns"][:, :defined]) and np.array_equal(per_game["tally"], cols["tally"])
    if not np.array_equal(cols["columns"], want_images):  # name the first differing values: plane / byte array, game, both values
        sh_i, off = (int(v[0]) for v in np.nonzero(cols["columns"] != want_images))
        ni = (4 + 13 * k) * 4 * gps
        if off < ni:
            plane, g = off // (4 * gps), (off % (4 * gps)) // 4
            got_v, want_v = (int(a[sh_i, :ni].view(np.int32).reshape(-1, gps)[plane, g]) for a in (cols["columns"], want_images))
            where = f"int32 plane {plane} (seat {(plane - 4) // 13 if plane >= 4 else None}, field {(plane - 4) % 13 if plane >= 4 else plane}), game {g}"
        else:
            where, got_v, want_v = f"byte {off - ni} of the status / winner / rank-order arrays", int(cols["columns"][sh_i, off]), int(want_images[sh_i, off])
        n_bad = int((cols["columns"] != want_images).sum())
        raise AssertionError(f"k={k}: {n_bad} bytes differ; first in shuffle {sh_i}: {where}: kernel {got_v}, rows {want_v}; row = {rows['rows'][sh_i * gps + (g if off < ni else 0)]}")
    sh = np.arange(lo, hi, dtype=np.int64)
    seeds = urandom.coordinate_seeds(urandom.RandomPurpose.TOURNAMENT_SHUFFLE, root_seed=42, k=k, shuffle_index=sh.astype(np.uint64), dtype=np.uint32)
    game_seeds = eng.game_seeds(int(urandom.RandomPurpose.TOURNAMENT_GAME), 42, k, lo, hi, gps)
    res = write_row_shards_native(tmp_path, k, 42, cols["columns"], sh, seeds, (sh // 5).astype(np.int32), game_seeds, 102, threads=4)
    want = _reference_table(rows["rows"], k, ids, gps, sh, seeds.astype(np.int64), (sh // 5).astype(np.int32), np.asarray(game_seeds).reshape(-1), root_seed=42)
    for i in range(hi - lo):
        got = pq.read_table(tmp_path / f"rows_42_{k}p_{sh[i]:012d}.parquet")
        assert got.equals(want.slice(i * gps, gps)), (k, i)
    assert int(res["byte_length"].min()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("k", [2, 5, 12])
def test_async_rows_calls_deliver_the_same_images(k):
    """Option ``rows_async`` (`farkle run`, rows mode): four launch groups back to back into two page-locked buffers, each awaited with
    ``fk_rows_wait`` only after the NEXT call has been made — images and tallies equal those of the waiting form, also with several chunks
    per call and with the calls' chunk count odd (the device row buffers alternate over chunks and calls)."""
    from farkle_ii_amd.backend import row_columns_bytes
    from farkle_ii_amd.engine import get_engine, set_engine
    from farkle_ii_amd.strategies import generate_strategy_grid, pack_strategies, prepare_public_helper_strategies

    set_engine(None)
    eng = get_engine()
    table = pack_strategies(prepare_public_helper_strategies(generate_strategy_grid()[0]))
    ids = np.arange(len(table), dtype=np.int32)
    gps = len(table) // k
    stride = row_columns_bytes(k, gps)
    groups = [(0, 9), (9, 18), (18, 21), (21, 30)]
    want = [eng.tournament_columns(table, k, 7, lo, hi, ids) for lo, hi in groups]
    defined = ((4 + 13 * k) * 4 + 2 + k) * gps
    for chunk_games in (4_000_000, 3 * gps):  # one chunk per call / three (the 3-shuffle group: one)
        eng.set_option("rows_chunk_games", chunk_games)
        try:
            pins = [eng.pinned_empty(9 * stride, np.uint8) for _ in range(2)]
            waiting: list = []
            got = []
            for g, (lo, hi) in enumerate(groups):
                res = eng.tournament_columns(table, k, 7, lo, hi, ids, columns_out=pins[g & 1], async_rows=True)
                assert res["rows_event"] == g % 4 or res["rows_event"] in range(4)
                if waiting:  # the previous group: awaited after this call was made, copied out before its buffer is used again
                    prev, ev = waiting.pop()
                    eng.rows_wait(ev)
                    got.append((prev["tally"].copy(), prev["columns"].copy()))
                waiting.append((res, res["rows_event"]))
            prev, ev = waiting.pop()
            eng.rows_wait(ev)
            got.append((prev["tally"].copy(), prev["columns"].copy()))
        finally:
            eng.set_option("rows_chunk_games", 4_000_000)
        assert eng.get_option("rows_async") == 0  # the option is the call's, not the engine's
        for (tally, columns), w in zip(got, want):
            assert np.array_equal(tally, w["tally"])
            assert np.array_equal(columns[:, :defined], w["columns"][:, :defined])
    with pytest.raises(RuntimeError):
        eng.rows_wait(4)
    # fk_tournament_run_columns_seeds: the shards' fingerprints with the images, in the waiting and the async form, one and several chunks
    from farkle_ii_amd import random as urandom
    from farkle_ii_amd.backend import make_coords

    lo, hi = 40, 49
    want_sh = eng.coordinate_seeds(make_coords(int(urandom.RandomPurpose.TOURNAMENT_SHUFFLE), 7, k, np.arange(lo, hi, dtype=np.uint64)), want32=True)[0]
    assert np.array_equal(want_sh, urandom.coordinate_seeds(urandom.RandomPurpose.TOURNAMENT_SHUFFLE, root_seed=7, k=k,
                                                            shuffle_index=np.arange(lo, hi, dtype=np.uint64), dtype=np.uint32))
    want_games = eng.game_seeds(int(urandom.RandomPurpose.TOURNAMENT_GAME), 7, k, lo, hi, gps)
    plain = eng.tournament_columns(table, k, 7, lo, hi, ids)
    for async_rows, chunk_games in ((False, 4_000_000), (True, 4_000_000), (True, 2 * gps)):
        eng.set_option("rows_chunk_games", chunk_games)
        try:
            sh_out, g_out = np.zeros(hi - lo, dtype=np.uint32), np.zeros((hi - lo) * gps, dtype=np.uint32)
            res = eng.tournament_columns(table, k, 7, lo, hi, ids, columns_out=pins[0], async_rows=async_rows, shuffle_seeds_out=sh_out,
                                         game_seeds_out=g_out)
            assert np.array_equal(sh_out, want_sh) and np.array_equal(g_out.reshape(hi - lo, gps), want_games)  # complete on return
            if async_rows:
                eng.rows_wait(res["rows_event"])
            assert np.array_equal(res["tally"], plain["tally"]) and np.array_equal(res["columns"][:, :defined], plain["columns"][:, :defined])
        finally:
            eng.set_option("rows_chunk_games", 4_000_000)
    with pytest.raises(ValueError):
        eng.tournament_columns(table, k, 7, lo, hi, ids, shuffle_seeds_out=np.zeros(3, dtype=np.uint32))
