"""ctypes binding of the C-ABI library (``include/farkle_hip.h`` -> ``farkle_ii_amd/libfarkle_hip.so``).

This is the only door from Python into the simulation: there is no CPU fallback.  If the shared
library has not been built, or no HIP device is usable, constructing an :class:`Engine` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from .strategies import STRATEGY_DTYPE

PKG_DIR = Path(__file__).resolve().parent
# FARKLE_HIP_LIB: load another build of the same C-ABI (A/B timing of kernel variants); the default is the in-tree library
LIB_PATH = Path(os.environ.get("FARKLE_HIP_LIB", PKG_DIR / "libfarkle_hip.so"))
SRC_DIR = PKG_DIR / "csrc"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

COORD_DTYPE = np.dtype(
    [("purpose", "<u4"), ("pad", "<u4"), ("root_seed", "<u8"), ("k", "<u8"), ("shuffle_index", "<u8"),
     ("pair_id", "<u8"), ("order", "<u8"), ("game_index", "<u8"), ("seat_index", "<u8"), ("replicate_index", "<u8")]
)
OVERRIDE_DTYPE = np.dtype(
    [("root_seed", "<u8"), ("a", "<u8"), ("b", "<u8"), ("k_or_order", "<u4"), ("max_rounds", "<u4")]
)
SEAT_DTYPE = np.dtype(
    [("score", "<i4"), ("strategy", "<i4"), ("farkles", "<u2"), ("rolls", "<u2"), ("n_turns", "<u2"),
     ("highest_turn", "<u2"), ("smart_five_uses", "<u2"), ("n_smart_five_dice", "<u2"), ("smart_one_uses", "<u2"),
     ("n_smart_one_dice", "<u2"), ("hot_dice", "<u2"), ("rank", "u1"), ("hit_max_rounds", "u1")]
)
H2H_BLOCK_DTYPE = np.dtype(
    [("seats", STRATEGY_DTYPE, (2,)), ("pair_id", "<u8"), ("order", "<u4"), ("pad", "<u4"), ("target", "<u8"),
     ("max_attempts", "<u8"), ("state", "<u8", (5,))]
)
TALLY_COLS = 26
LAG_COLS = 11  # FK_LAG_COLS: pairs | win: sx sy sxx syy sxy | n_rounds: sx sy sxx syy sxy
SEAT_STAT_COLS = 31
SEAT_RATIO_COLS = 4  # FK_SEAT_RATIO_COLS: sum / sum of squares of score / n_turns, of score / n_rounds (float64, (shuffle, game, seat) order)
SEAT_STAT_NAMES = ("exposures", "completed_exposures", "safety_limit_exposures", "wins", "final_score_sum", "final_score_square_sum",
                   "n_turns_sum", "n_turns_square_sum", "turn_round_mismatch_count", "turn_minus_rounds_sum",
                   "turn_minus_rounds_square_sum") + tuple(
    f"{name}_{kind}" for name in ("rank", "loss_margin", "rolls", "farkles", "highest_turn", "hot_dice", "smart_five_uses",
                                   "n_smart_five_dice", "smart_one_uses", "n_smart_one_dice") for kind in ("sum", "square_sum"))
# tally columns (run_tournament.py:109-121, 165-195)
COL_WINS, COL_ATTEMPTED, COL_COMPLETED, COL_SAFETY, COL_SUMS, COL_SQ_SUMS = 0, 1, 2, 3, 4, 15

FK_ERR_ROLL_LIMIT, FK_ERR_ARG, FK_ERR_COUNTER_OVERFLOW, FK_ERR_HIP, FK_ERR_NO_DEVICE, FK_ERR_COMM = -1, -2, -3, -4, -5, -6
COMM_ID_BYTES = 128


def row_dtype(k: int) -> np.dtype:
    """Structured dtype of one game row: 4-byte header + k 28-byte seat records."""
    return np.dtype([("n_rounds", "<u2"), ("status", "u1"), ("winner_seat", "i1"), ("seats", SEAT_DTYPE, (k,))])


class _DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("arch", C.c_char * 32), ("compute_units", C.c_int32), ("clock_mhz", C.c_int32),
                ("wavefront_size", C.c_int32), ("lds_bytes_per_cu", C.c_int32), ("hbm_bytes", C.c_uint64)]


class _Timing(C.Structure):
    _fields_ = [("perm_ms", C.c_float), ("seed_ms", C.c_float), ("play_ms", C.c_float), ("total_ms", C.c_float),
                ("play_launches", C.c_int32), ("play_block", C.c_int32), ("play_grid", C.c_int32),
                ("play_lds_bytes", C.c_int32), ("games", C.c_int64), ("prefetched_chunks", C.c_int32), ("play_clock_mhz", C.c_int32),
                ("play_block_end_p50_ms", C.c_float), ("play_block_end_max_ms", C.c_float), ("play_mixed_flags", C.c_int32)]


class FarkleHipError(RuntimeError):
    """Raised for every non-zero return of the C-ABI; ``code`` is the FK_ERR_* value."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


def hip_sources() -> list[Path]:
    return [SRC_DIR / "farkle_hip.hip"]


VARIANTS = {"detour": ["-DFK_FORCE_DETOUR=4"]}  # test builds: libfarkle_hip_<variant>.so beside the product library


def library_path(variant: str | None = None) -> Path:
    return LIB_PATH if not variant else LIB_PATH.with_name(f"libfarkle_hip_{variant}.so")


def build_library(force: bool = False, verbose: bool = False, variant: str | None = None) -> Path:
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU).  ``variant``: a test build with extra
    definitions (``VARIANTS``), never loaded by the product."""
    out = library_path(variant)
    deps = hip_sources() + [SRC_DIR / "fk_device.h", SRC_DIR / "fk_kernels.h", SRC_DIR / "fk_play_hc.h", SRC_DIR / "fk_shard_writer.h", SRC_DIR / "fk_perm_wave.h", SRC_DIR / "fk_row_columns_seats.h", PKG_DIR.parent / "include" / "farkle_hip.h"]
    if out.exists() and not force and all(out.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return out
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *(VARIANTS[variant] if variant else []), "-o", str(out),
           *[str(s) for s in hip_sources()]]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(" ".join(cmd))
        print(res.stdout, res.stderr)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed ({res.returncode}): {res.stderr[-2000:]}")
    return out


_EXPORTS = ["fk_init", "fk_destroy", "fk_last_error", "fk_get_device_info", "fk_get_timing", "fk_set_option",
            "fk_tournament_run", "fk_tournament_run_stats", "fk_tournament_run_all_player", "fk_tournament_run_lags", "fk_tournament_hint_next", "fk_play_games", "fk_h2h_run", "fk_h2h_run_blocks", "fk_coordinate_seeds", "fk_debug_score", "fk_debug_should_continue",
            "fk_debug_dice", "fk_debug_dice_state", "fk_debug_dice_keys", "fk_comm_unique_id", "fk_comm_init", "fk_reduce_tally", "fk_comm_destroy", "fk_tally_resident_reduce", "fk_comm_ranks", "fk_host_alloc", "fk_host_free", "fk_game_seeds",
            "fk_tournament_run_columns", "fk_row_columns_bytes", "fk_write_row_shards", "fk_debug_sha256", "fk_get_option", "fk_debug_deadline_handshake", "fk_debug_hold_memory", "fk_rows_wait", "fk_tournament_run_columns_seeds"]
_libs: dict = {}


def load_library(variant: str | None = None) -> C.CDLL:
    """dlopen the in-tree library; loud failure when it has not been built."""
    if variant not in _libs:
        path = library_path(variant)
        if not path.exists():
            raise FileNotFoundError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the simulation path)")
        lib = C.CDLL(str(path))
        for name in _EXPORTS:
            getattr(lib, name)  # AttributeError if a declared symbol is not exported
        lib.fk_last_error.restype = C.c_char_p
        lib.fk_last_error.argtypes = [C.c_void_p]
        lib.fk_destroy.restype = None
        lib.fk_destroy.argtypes = [C.c_void_p]
        _libs[variant] = lib
    return _libs[variant]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def row_columns_bytes(k: int, games_per_shuffle: int) -> int:
    """``fk_row_columns_bytes``: bytes of one shuffle's column image."""
    lib = load_library()
    lib.fk_row_columns_bytes.restype = C.c_size_t
    return int(lib.fk_row_columns_bytes(C.c_int32(k), C.c_int32(games_per_shuffle)))


class _ShardJob(C.Structure):
    _fields_ = [("k", C.c_int32), ("games_per_shuffle", C.c_int32), ("n_shuffles", C.c_int32), ("threads", C.c_int32), ("atomic", C.c_int32),
                ("rng_purpose_namespace", C.c_int32), ("root_seed", C.c_uint64), ("shuffle_index", C.c_void_p), ("shuffle_seed", C.c_void_p),
                ("batch_id", C.c_void_p), ("game_seed", C.c_void_p), ("columns", C.c_void_p), ("shard_stride", C.c_size_t),
                ("directory", C.c_char_p), ("footer_head", C.c_char_p), ("footer_head_len", C.c_size_t), ("footer_kv", C.c_char_p),
                ("footer_kv_len", C.c_size_t), ("footer_orders", C.c_char_p), ("footer_orders_len", C.c_size_t), ("leaf_type", C.c_void_p),
                ("leaf_paths", C.c_char_p), ("n_leaves", C.c_int32), ("side_body", C.POINTER(C.c_char_p)), ("side_full", C.POINTER(C.c_char_p)),
                ("side_directory", C.c_char_p)]


def write_row_shards_native(row_dir, k: int, root_seed: int, columns: np.ndarray, shuffle_index, shuffle_seed, batch_id, game_seeds,
                            rng_purpose_namespace: int, *, threads: int = 1, atomic: bool = True, sidecar: dict | None = None) -> dict:
    """``fk_write_row_shards``: the row-shard files of ``columns`` (``[n_shuffles][stride]`` column images) in ``row_dir``, written by
    ``threads`` host threads of the library.  Returns ``byte_length`` int64 ``[n]``, ``sha256`` uint8 ``[n][32]`` and — with a contract-v3
    ``sidecar`` template (``contract_v3.SimulationContract.shard_template``) — ``sidecar_sha256``."""
    return prepare_row_shards_native(row_dir, k, root_seed, columns, shuffle_index, shuffle_seed, batch_id, game_seeds, rng_purpose_namespace,
                                     threads=threads, atomic=atomic, sidecar=sidecar)()


def prepare_row_shards_native(row_dir, k: int, root_seed: int, columns: np.ndarray, shuffle_index, shuffle_seed, batch_id, game_seeds,
                              rng_purpose_namespace: int, *, threads: int = 1, atomic: bool = True, sidecar: dict | None = None):
    """``write_row_shards_native`` in two steps: the job is checked and laid out here; the returned callable makes the one library call
    (``columns`` may be filled in between: `farkle run` hands the callable to its shard thread, which then holds the interpreter lock
    for a few bytecodes per launch group)."""
    from .parquet_template import shard_footer_template

    lib = load_library()
    columns = np.ascontiguousarray(columns, dtype=np.uint8)
    n = len(columns)
    stride = columns.shape[1] if columns.ndim == 2 else 0
    gps = np.asarray(game_seeds).size // max(n, 1)
    tpl = shard_footer_template(k)
    sh = np.ascontiguousarray(shuffle_index, dtype=np.int64)
    seeds = np.ascontiguousarray(shuffle_seed, dtype=np.int64)
    batch = np.ascontiguousarray(batch_id, dtype=np.int32)
    gs = np.ascontiguousarray(game_seeds, dtype=np.uint32).reshape(-1)
    if not (len(sh) == len(seeds) == len(batch) == n) or gs.size != n * gps or stride < row_columns_bytes(k, gps):
        raise ValueError("shard job arrays disagree about the number of shuffles / games per shuffle")
    leaf_type = np.ascontiguousarray(tpl["leaf_type"], dtype=np.int32)
    leaf_paths = b"".join(b"".join(part.encode("utf-8") + b"\0" for part in path) + b"\0" for path in tpl["leaf_paths"])
    job = _ShardJob(k=k, games_per_shuffle=gps, n_shuffles=n, threads=max(1, int(threads)), atomic=1 if atomic else 0,
                    rng_purpose_namespace=int(rng_purpose_namespace), root_seed=int(root_seed), shuffle_index=sh.ctypes.data,
                    shuffle_seed=seeds.ctypes.data, batch_id=batch.ctypes.data, game_seed=gs.ctypes.data, columns=columns.ctypes.data,
                    shard_stride=stride, directory=os.fspath(row_dir).encode("utf-8"), footer_head=tpl["footer_head"],
                    footer_head_len=len(tpl["footer_head"]), footer_kv=tpl["footer_kv"], footer_kv_len=len(tpl["footer_kv"]),
                    footer_orders=tpl["footer_orders"], footer_orders_len=len(tpl["footer_orders"]), leaf_type=leaf_type.ctypes.data,
                    leaf_paths=leaf_paths, n_leaves=len(leaf_type))
    side_sha = None
    if sidecar is not None:
        body = (C.c_char_p * 4)(*[piece.encode("utf-8") for piece in sidecar["body"]])
        full = (C.c_char_p * 5)(*[piece.encode("utf-8") for piece in sidecar["full"]])
        job.side_body, job.side_full, job.side_directory = body, full, sidecar["directory"].encode("utf-8")
        side_sha = np.zeros((n, 32), dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.int64)
    sha = np.zeros((n, 32), dtype=np.uint8)
    err = C.create_string_buffer(512)
    lib.fk_write_row_shards.restype = C.c_int
    keep = (columns, sh, seeds, batch, gs, leaf_type, leaf_paths, tpl, sidecar)  # what the job points into

    def run() -> dict:
        rc = lib.fk_write_row_shards(C.byref(job), _p(sizes), _p(sha), _p(side_sha), err, C.c_size_t(len(err)))
        if rc != 0:
            raise OSError(f"fk_write_row_shards failed ({rc}): {err.value.decode('utf-8', 'replace')}")
        return {"byte_length": sizes, "sha256": sha, "sidecar_sha256": side_sha}

    run.keep = keep  # (alive as long as the callable is)
    return run


def make_overrides(items) -> np.ndarray:
    """items: (root_seed, a, b, k_or_order, max_rounds) tuples -> fk_override[]."""
    items = list(items)
    out = np.zeros(len(items), dtype=OVERRIDE_DTYPE)
    for i, it in enumerate(items):
        out[i] = tuple(int(v) for v in it)
    return out


def make_coords(purpose, root_seed, k, shuffle_index=0, pair_id=0, order=0, game_index=0, n: int | None = None) -> np.ndarray:
    """Broadcast scalars/arrays into an fk_coord[] (seat_index = replicate_index = 0)."""
    arrs = [np.atleast_1d(np.asarray(v)) for v in (purpose, root_seed, k, shuffle_index, pair_id, order, game_index)]
    n = max(len(a) for a in arrs) if n is None else n
    out = np.zeros(n, dtype=COORD_DTYPE)
    for name, a in zip(("purpose", "root_seed", "k", "shuffle_index", "pair_id", "order", "game_index"), arrs):
        out[name] = a
    return out


class Engine:
    """One context (HIP stream + device workspace) on one GPU."""

    def __init__(self, device: int = 0, variant: str | None = None):
        self._lib = load_library(variant)  # (variant: a test build of the library, backend.VARIANTS)
        self._ctx = C.c_void_p()
        rc = self._lib.fk_init(C.c_int(device), C.byref(self._ctx))
        if rc != 0:
            self._ctx = C.c_void_p()
            names = {FK_ERR_NO_DEVICE: "no usable HIP device (this engine has no CPU fallback)", FK_ERR_ARG: "bad device ordinal",
                     FK_ERR_HIP: "HIP runtime failure during fk_init"}
            raise FarkleHipError(rc, f"fk_init({device}) failed: {names.get(rc, rc)}")
        self.device = device
        self.comm_world = 1
        self.comm_rank = 0

    # -- plumbing -------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.fk_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int) -> None:
        if rc != 0:
            msg = self._lib.fk_last_error(self._ctx)
            raise FarkleHipError(rc, (msg or b"").decode() or f"error {rc}")

    def set_option(self, name: str, value: int) -> None:
        self._check(self._lib.fk_set_option(self._ctx, name.encode(), C.c_int64(int(value))))

    def device_info(self) -> dict:
        info = _DeviceInfo()
        self._check(self._lib.fk_get_device_info(self._ctx, C.byref(info)))
        return {"name": info.name.decode(), "arch": info.arch.decode(), "compute_units": info.compute_units,
                "clock_mhz": info.clock_mhz, "wavefront_size": info.wavefront_size,
                "lds_bytes_per_cu": info.lds_bytes_per_cu, "hbm_bytes": int(info.hbm_bytes)}

    def timing(self) -> dict:
        t = _Timing()
        self._check(self._lib.fk_get_timing(self._ctx, C.byref(t)))
        return {f: getattr(t, f) for f, _ in _Timing._fields_}

    # -- hot path ------------------------------------------------------------------------
    def tournament(self, table: np.ndarray, k: int, root_seed: int, shuffle_begin: int, shuffle_end: int,
                   shuffles_per_batch: int | None = None, target_score: int = 10_000, max_rounds: int = 200,
                   overrides: np.ndarray | None = None, want_rows: bool = False, want_perms: bool = False,
                   want_seat_stats: bool = False, rows_out: np.ndarray | None = None, want_seat_ratios: bool = True) -> dict:
        """All games of shuffles ``[shuffle_begin, shuffle_end)``: per-batch tallies (+ rows / permutations / the all-seat
        integer statistics ``[n_batches][S][SEAT_STAT_COLS]``, columns ``SEAT_STAT_NAMES``, with the four float64 ratio sums
        ``seat_ratio_sums [n_batches][S][SEAT_RATIO_COLS]`` of the same table unless ``want_seat_ratios`` is off: they are one
        sequential sum per (batch, strategy), which a caller that only reads the integer columns need not wait for)."""
        table = np.ascontiguousarray(table, dtype=STRATEGY_DTYPE)
        S = len(table)
        n_sh = int(shuffle_end) - int(shuffle_begin)
        spb = n_sh if not shuffles_per_batch else int(shuffles_per_batch)
        spb = max(spb, 1)
        n_batches = max((n_sh + spb - 1) // spb, 0)
        gps = S // k if k > 0 else 0
        tally = np.zeros((max(n_batches, 1), S, TALLY_COLS), dtype=np.int64)
        rows = None
        if want_rows:
            n_rows = max(n_sh, 0) * gps
            if rows_out is not None:  # e.g. a page-locked buffer from pinned_empty(): rows cross PCIe by DMA while the next chunk plays
                if rows_out.dtype != row_dtype(k) or len(rows_out) < n_rows or not rows_out.flags["C_CONTIGUOUS"]:
                    raise ValueError("rows_out must be a contiguous array of row_dtype(k) with room for every game")
                rows = rows_out[:n_rows]
            else:
                rows = np.zeros(n_rows, dtype=row_dtype(k))
        perms = np.zeros((max(n_sh, 0), S), dtype=np.int32) if want_perms else None
        ov = overrides if overrides is not None else np.zeros(0, dtype=OVERRIDE_DTYPE)
        ov = np.ascontiguousarray(ov, dtype=OVERRIDE_DTYPE)
        stats = np.zeros((max(n_batches, 1), S, SEAT_STAT_COLS), dtype=np.int64) if want_seat_stats else None
        ratios = np.zeros((max(n_batches, 1), S, SEAT_RATIO_COLS), dtype=np.float64) if want_seat_stats and want_seat_ratios else None
        if ratios is not None:  # the all-player accumulators: 31 integer sums + the four float64 sums in (shuffle, game, seat) order
            self._check(self._lib.fk_tournament_run_all_player(
                self._ctx, _p(table), C.c_int32(S), C.c_int32(k), C.c_uint64(root_seed), C.c_uint64(shuffle_begin),
                C.c_uint64(shuffle_end), C.c_uint32(spb), C.c_int32(target_score), C.c_int32(max_rounds), _p(ov),
                C.c_int32(len(ov)), _p(tally), _p(rows), _p(perms), _p(stats), _p(ratios)))
        else:
            self._check(self._lib.fk_tournament_run_stats(
                self._ctx, _p(table), C.c_int32(S), C.c_int32(k), C.c_uint64(root_seed), C.c_uint64(shuffle_begin),
                C.c_uint64(shuffle_end), C.c_uint32(spb), C.c_int32(target_score), C.c_int32(max_rounds), _p(ov),
                C.c_int32(len(ov)), _p(tally), _p(rows), _p(perms), _p(stats)))
        return {"tally": tally[:n_batches], "rows": rows, "perms": perms,
                "seat_stats": None if stats is None else stats[:n_batches],
                "seat_ratio_sums": None if ratios is None else ratios[:n_batches]}

    def tournament_columns(self, table: np.ndarray, k: int, root_seed: int, shuffle_begin: int, shuffle_end: int, strategy_ids,
                           shuffles_per_batch: int | None = None, target_score: int = 10_000, max_rounds: int = 200,
                           overrides: np.ndarray | None = None, columns_out: np.ndarray | None = None, async_rows: bool = False,
                           shuffle_seeds_out: np.ndarray | None = None, game_seeds_out: np.ndarray | None = None) -> dict:
        """``tournament`` with the rows as per-shuffle COLUMN IMAGES (``fk_tournament_run_columns``): ``columns`` uint8
        ``[n_shuffles][fk_row_columns_bytes(k, S / k)]`` — what ``write_row_shards_native`` frames as the row-shard Parquet files.
        ``async_rows``: return when the last copy to the host is queued; read ``columns`` after ``rows_wait(result["rows_event"])``.
        ``shuffle_seeds_out`` uint32 ``[n_shuffles]`` / ``game_seeds_out`` uint32 ``[n_shuffles * S / k]``: also filled, complete on return
        (``fk_tournament_run_columns_seeds``: the fingerprints a shard's manifest record and its game_seed column carry)."""
        table = np.ascontiguousarray(table, dtype=STRATEGY_DTYPE)
        S = len(table)
        ids = np.ascontiguousarray(strategy_ids, dtype=np.int32)
        if len(ids) != S:
            raise ValueError("strategy_ids must name every strategy of the table")
        n_sh = max(int(shuffle_end) - int(shuffle_begin), 0)
        spb = max(n_sh if not shuffles_per_batch else int(shuffles_per_batch), 1)
        n_batches = (n_sh + spb - 1) // spb
        stride = row_columns_bytes(k, S // k)
        tally = np.zeros((max(n_batches, 1), S, TALLY_COLS), dtype=np.int64)
        if columns_out is not None:
            if columns_out.dtype != np.uint8 or columns_out.size < n_sh * stride or not columns_out.flags["C_CONTIGUOUS"]:
                raise ValueError("columns_out must be a contiguous uint8 array with room for every shuffle's image")
            columns = columns_out.reshape(-1)[:n_sh * stride].reshape(n_sh, stride)
        else:
            columns = np.zeros((n_sh, stride), dtype=np.uint8)
        ov = np.ascontiguousarray(overrides if overrides is not None else np.zeros(0, dtype=OVERRIDE_DTYPE), dtype=OVERRIDE_DTYPE)
        rows_event = None
        if n_sh:
            if async_rows:
                self.set_option("rows_async", 1)
            try:
                for name, arr, want in (("shuffle_seeds_out", shuffle_seeds_out, n_sh), ("game_seeds_out", game_seeds_out, n_sh * (S // k))):
                    if arr is not None and (arr.dtype != np.uint32 or arr.size != want or not arr.flags["C_CONTIGUOUS"]):
                        raise ValueError(f"{name} must be a contiguous uint32 array of {want} elements")
                self._run_columns(table, S, k, root_seed, shuffle_begin, shuffle_end, spb, target_score, max_rounds, ov, tally, ids, columns,
                                  shuffle_seeds_out, game_seeds_out)
                if async_rows:
                    rows_event = self.get_option("rows_event")
            finally:
                if async_rows:
                    self.set_option("rows_async", 0)
        return {"tally": tally[:n_batches], "columns": columns, "rows_event": rows_event}

    def rows_wait(self, slot: int) -> None:
        """``fk_rows_wait``: the column images of the ``async_rows`` call that reported ``rows_event == slot`` are in the host buffer."""
        self._check(self._lib.fk_rows_wait(self._ctx, C.c_int32(slot)))

    columns_with_seeds = True  # (tournament_columns takes shuffle_seeds_out / game_seeds_out)

    def _run_columns(self, table, S, k, root_seed, shuffle_begin, shuffle_end, spb, target_score, max_rounds, ov, tally, ids, columns,
                     shuffle_seeds=None, game_seeds=None) -> None:
        args = (self._ctx, _p(table), C.c_int32(S), C.c_int32(k), C.c_uint64(root_seed), C.c_uint64(shuffle_begin), C.c_uint64(shuffle_end),
                C.c_uint32(spb), C.c_int32(target_score), C.c_int32(max_rounds), _p(ov), C.c_int32(len(ov)), _p(tally), _p(ids), _p(columns))
        if shuffle_seeds is None and game_seeds is None:
            self._check(self._lib.fk_tournament_run_columns(*args))
        else:
            self._check(self._lib.fk_tournament_run_columns_seeds(*args, _p(shuffle_seeds), _p(game_seeds)))

    def tournament_lags(self, table: np.ndarray, k: int, root_seed: int, shuffle_begin: int, shuffle_end: int, lags,
                        shuffles_per_batch: int | None = None, target_score: int = 10_000, max_rounds: int = 200,
                        overrides: np.ndarray | None = None) -> dict:
        """``tournament`` + the lag sufficient statistics of the strategy family (``fk_tournament_run_lags``):
        ``lag_sums [S][n_lags][LAG_COLS]`` int64 and the first / last ``min(max lag, n_shuffles)`` series rows
        (``lag_head`` / ``lag_tail``, uint16 ``n_rounds | won << 15``, ``[m][S]``) that ``rng_lags.LagSummary`` merges ranges with."""
        table = np.ascontiguousarray(table, dtype=STRATEGY_DTYPE)
        lags = np.ascontiguousarray(list(lags), dtype=np.int32)
        S = len(table)
        n_sh = max(int(shuffle_end) - int(shuffle_begin), 0)
        spb = max(n_sh if not shuffles_per_batch else int(shuffles_per_batch), 1)
        n_batches = (n_sh + spb - 1) // spb
        tally = np.zeros((max(n_batches, 1), S, TALLY_COLS), dtype=np.int64)
        m = min(int(lags.max()) if len(lags) else 0, n_sh)
        sums = np.zeros((S, len(lags), LAG_COLS), dtype=np.int64)
        head = np.zeros((max(m, 1), S), dtype=np.uint16)
        tail = np.zeros((max(m, 1), S), dtype=np.uint16)
        ov = np.ascontiguousarray(overrides if overrides is not None else np.zeros(0, dtype=OVERRIDE_DTYPE), dtype=OVERRIDE_DTYPE)
        self._check(self._lib.fk_tournament_run_lags(
            self._ctx, _p(table), C.c_int32(S), C.c_int32(k), C.c_uint64(root_seed), C.c_uint64(shuffle_begin),
            C.c_uint64(shuffle_end), C.c_uint32(spb), C.c_int32(target_score), C.c_int32(max_rounds), _p(ov), C.c_int32(len(ov)),
            _p(tally), _p(lags), C.c_int32(len(lags)), _p(sums), _p(head), _p(tail)))
        return {"tally": tally[:n_batches], "lag_sums": sums, "lag_head": head[:m], "lag_tail": tail[:m], "n_shuffles": n_sh}

    def get_option(self, name: str) -> int:
        """``fk_get_option``: an option's value or a figure of the last call (``last_budget``, ``oom_replays``, the effective ``comm_timeout_ms``)."""
        value = C.c_int64(0)
        self._check(self._lib.fk_get_option(self._ctx, name.encode("utf-8"), C.byref(value)))
        return int(value.value)

    def hold_memory(self, leave_free: int) -> tuple[int, int]:
        """``fk_debug_hold_memory``: take device memory until about ``leave_free`` bytes are free (< 0: give it back); (free, total) after."""
        free, total = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.fk_debug_hold_memory(self._ctx, C.c_int64(leave_free), C.byref(free), C.byref(total)))
        return int(free.value), int(total.value)

    def pinned_empty(self, n: int, dtype) -> np.ndarray:
        """``n`` elements of ``dtype`` in page-locked host memory (``fk_host_alloc``), freed when the array is collected — the
        ``rows_out`` buffer of rows mode.  Keep the engine alive while the array is."""
        import weakref

        dtype = np.dtype(dtype)
        nbytes = max(int(n) * dtype.itemsize, 1)
        ptr = C.c_void_p()
        self._check(self._lib.fk_host_alloc(self._ctx, C.c_size_t(nbytes), C.byref(ptr)))
        raw = (C.c_char * nbytes).from_address(ptr.value)
        arr = np.frombuffer(raw, dtype=dtype, count=int(n))
        lib, ctx, addr = self._lib, self._ctx, ptr.value
        weakref.finalize(raw, lambda: lib.fk_host_free(ctx, C.c_void_p(addr)) if ctx else None)
        return arr

    def hint_next(self, shuffle_begin: int, shuffle_end: int, need_state: bool = False) -> None:
        """The call after the next ``tournament`` call will play this shuffle range of the same table, k and root: its
        permutations and seat seeding are then prepared in the drain tail of the next call's game kernel."""
        self._check(self._lib.fk_tournament_hint_next(self._ctx, C.c_uint64(shuffle_begin), C.c_uint64(shuffle_end),
                                                      C.c_int32(1 if need_state else 0)))

    def play_games(self, coords: np.ndarray, table: np.ndarray, seat_strategy, k: int, target_score: int = 10_000,
                   max_rounds: int = 200) -> np.ndarray:
        coords = np.ascontiguousarray(coords, dtype=COORD_DTYPE)
        table = np.ascontiguousarray(table, dtype=STRATEGY_DTYPE)
        ss = np.ascontiguousarray(seat_strategy, dtype=np.int32).reshape(-1)
        n = len(coords)
        if ss.size != n * k:
            raise ValueError("seat_strategy must hold n_games * k entries")
        rows = np.zeros(n, dtype=row_dtype(k))
        self._check(self._lib.fk_play_games(self._ctx, _p(coords), C.c_int64(n), _p(table), C.c_int32(len(table)), _p(ss),
                                            C.c_int32(k), C.c_int32(target_score), C.c_int32(max_rounds), _p(rows)))
        return rows

    def h2h(self, seats: np.ndarray, root_seed: int, pair_id: int, order: int, target: int, max_attempts: int,
            chunk_games: int, target_score: int = 10_000, max_rounds: int = 200, overrides: np.ndarray | None = None,
            state=None) -> np.ndarray:
        seats = np.ascontiguousarray(seats, dtype=STRATEGY_DTYPE)
        if len(seats) != 2:
            raise ValueError("H2H blocks seat exactly two strategies")
        st = np.zeros(5, dtype=np.uint64) if state is None else np.ascontiguousarray(state, dtype=np.uint64).copy()
        ov = overrides if overrides is not None else np.zeros(0, dtype=OVERRIDE_DTYPE)
        ov = np.ascontiguousarray(ov, dtype=OVERRIDE_DTYPE)
        self._check(self._lib.fk_h2h_run(self._ctx, _p(seats), C.c_uint64(root_seed), C.c_uint64(pair_id), C.c_uint32(order),
                                         C.c_uint64(target), C.c_uint64(max_attempts), C.c_uint64(chunk_games),
                                         C.c_int32(target_score), C.c_int32(max_rounds), _p(ov), C.c_int32(len(ov)), _p(st)))
        return st

    def h2h_blocks(self, seats: np.ndarray, root_seed: int, pair_ids, orders, target, max_attempts, chunk_games=None,
                   target_score: int = 10_000, max_rounds: int = 200, overrides: np.ndarray | None = None,
                   states: np.ndarray | None = None) -> np.ndarray:
        """Many (pair, order) blocks of one root advanced in shared kernel launches (``fk_h2h_run_blocks``): ``seats[b]`` =
        the two seated strategies of block b; ``target`` / ``max_attempts`` scalars or per-block arrays; ``states`` the
        blocks' progress so far (default: fresh).  Returns uint64 ``[n_blocks][5]`` = attempted, completed, safety,
        wins_seat1, wins_seat2 — each row equals the serial ``h2h`` result of that block."""
        seats = np.ascontiguousarray(seats, dtype=STRATEGY_DTYPE).reshape(-1, 2)
        n = len(seats)
        blocks = np.zeros(n, dtype=H2H_BLOCK_DTYPE)
        blocks["seats"] = seats
        blocks["pair_id"] = np.asarray(pair_ids, dtype=np.uint64).reshape(-1)
        blocks["order"] = np.asarray(orders, dtype=np.uint32).reshape(-1)
        blocks["target"] = np.asarray(target, dtype=np.uint64)
        blocks["max_attempts"] = np.asarray(max_attempts, dtype=np.uint64)
        if states is not None:
            blocks["state"] = np.asarray(states, dtype=np.uint64).reshape(n, 5)
        chunk = int(blocks["max_attempts"].max()) if chunk_games is None and n else int(chunk_games or 0)
        ov = overrides if overrides is not None else np.zeros(0, dtype=OVERRIDE_DTYPE)
        ov = np.ascontiguousarray(ov, dtype=OVERRIDE_DTYPE)
        self._check(self._lib.fk_h2h_run_blocks(self._ctx, _p(blocks), C.c_int64(n), C.c_uint64(root_seed), C.c_uint64(chunk),
                                                C.c_int32(target_score), C.c_int32(max_rounds), _p(ov), C.c_int32(len(ov))))
        return blocks["state"].copy()

    # -- multi-GPU: the one exchange of the path, RCCL through the C-ABI ---------------------
    def comm_unique_id(self) -> bytes:
        """Rank 0: a fresh RCCL communicator id (128 bytes) to ship to every rank."""
        buf = C.create_string_buffer(COMM_ID_BYTES)
        rc = self._lib.fk_comm_unique_id(buf)
        if rc != 0:
            raise FarkleHipError(rc, "fk_comm_unique_id failed: librccl.so.1 could not be loaded or ncclGetUniqueId failed")
        return buf.raw

    def comm_init(self, comm_id: bytes, rank: int, world_size: int) -> None:
        """Collective: every rank joins the communicator on its own GPU."""
        if len(comm_id) != COMM_ID_BYTES:
            raise ValueError("communicator id must be 128 bytes")
        self._check(self._lib.fk_comm_init(self._ctx, C.create_string_buffer(comm_id, COMM_ID_BYTES), C.c_int32(rank), C.c_int32(world_size)))
        self.comm_world = world_size
        self.comm_rank = rank

    def reduce_tally(self, tally: np.ndarray, root: int = 0) -> np.ndarray:
        """Collective: int64 SUM over the communicator; the total on ``root`` (the rank's own tally elsewhere)."""
        t = np.ascontiguousarray(tally, dtype=np.int64).copy()
        self._check(self._lib.fk_reduce_tally(self._ctx, _p(t), C.c_int64(t.size), C.c_int32(root)))
        return t

    def reduce_resident_tally(self, shape, root: int = 0) -> np.ndarray | None:
        """Collective: the device-resident tally accumulator (option ``resident_tally``) summed over the communicator on the
        device; the total on ``root`` (None elsewhere).  Clears the accumulator."""
        out = np.zeros(shape, dtype=np.int64)
        self._check(self._lib.fk_tally_resident_reduce(self._ctx, _p(out), C.c_int64(out.size), C.c_int32(root)))
        return out if (self.comm_world == 1 or self.comm_rank == root) else None

    def comm_ranks(self) -> int:
        """Ranks of the engine's RCCL communicator as RCCL counts them (1 without one)."""
        return int(self._lib.fk_comm_ranks(self._ctx))

    def comm_destroy(self) -> None:
        self._check(self._lib.fk_comm_destroy(self._ctx))
        self.comm_world = 1

    def coordinate_seeds(self, coords: np.ndarray, want32: bool = True, want64: bool = False):
        """SeedSequence fingerprints of whole coordinates (``coordinate_seed``, utils/random.py:190-232)."""
        coords = np.ascontiguousarray(coords, dtype=COORD_DTYPE)
        n = len(coords)
        s32 = np.zeros(n, dtype=np.uint32) if want32 else None
        s64 = np.zeros(n, dtype=np.uint64) if want64 else None
        self._check(self._lib.fk_coordinate_seeds(self._ctx, C.c_int64(n), _p(coords), _p(s32), _p(s64)))
        return s32, s64

    def game_seeds(self, purpose: int, root_seed: int, k: int, shuffle_begin: int, shuffle_end: int, games_per_shuffle: int) -> np.ndarray:
        """uint32 ``coordinate_seed`` fingerprints ``[n_shuffles][games_per_shuffle]`` of (purpose, root, k, shuffle, game_index)."""
        n_sh = max(int(shuffle_end) - int(shuffle_begin), 0)
        out = np.zeros((n_sh, int(games_per_shuffle)), dtype=np.uint32)
        self._check(self._lib.fk_game_seeds(self._ctx, C.c_uint32(int(purpose)), C.c_uint64(root_seed), C.c_uint64(k), C.c_uint64(shuffle_begin),
                                            C.c_uint64(n_sh), C.c_uint32(int(games_per_shuffle)), _p(out)))
        return out

    # -- single-op probes ------------------------------------------------------------------
    def debug_score(self, faces: np.ndarray, lens, pre, strategies: np.ndarray) -> np.ndarray:
        faces = np.ascontiguousarray(faces, dtype=np.uint8).reshape(-1, 6)
        n = len(faces)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        pre = np.ascontiguousarray(pre, dtype=np.int32)
        strategies = np.ascontiguousarray(strategies, dtype=STRATEGY_DTYPE)
        out = np.zeros((n, 5), dtype=np.int32)
        self._check(self._lib.fk_debug_score(self._ctx, C.c_int64(n), _p(faces), _p(lens), _p(pre), _p(strategies), _p(out)))
        return out

    def debug_should_continue(self, args: np.ndarray, strategies: np.ndarray) -> np.ndarray:
        args = np.ascontiguousarray(args, dtype=np.int32).reshape(-1, 6)
        strategies = np.ascontiguousarray(strategies, dtype=STRATEGY_DTYPE)
        out = np.zeros(len(args), dtype=np.int32)
        self._check(self._lib.fk_debug_should_continue(self._ctx, C.c_int64(len(args)), _p(args), _p(strategies), _p(out)))
        return out

    def debug_dice(self, coords: np.ndarray, sizes, want_raw: bool = True):
        coords = np.ascontiguousarray(coords, dtype=COORD_DTYPE)
        sizes = np.ascontiguousarray(sizes, dtype=np.int32)
        n = len(coords)
        faces = np.zeros((n, int(sizes.sum())), dtype=np.uint8)
        raw = np.zeros((n, 4), dtype=np.uint64) if want_raw else None
        self._check(self._lib.fk_debug_dice(self._ctx, C.c_int64(n), _p(coords), C.c_int32(len(sizes)), _p(sizes), _p(faces), _p(raw)))
        return faces, raw

    def debug_dice_state(self, state: np.ndarray, sizes):
        state = np.ascontiguousarray(state, dtype=np.uint64).reshape(-1, 6)
        sizes = np.ascontiguousarray(sizes, dtype=np.int32)
        n = len(state)
        faces = np.zeros((n, int(sizes.sum())), dtype=np.uint8)
        out = np.zeros((n, 6), dtype=np.uint64)
        self._check(self._lib.fk_debug_dice_state(self._ctx, C.c_int64(n), _p(state), C.c_int32(len(sizes)), _p(sizes), _p(faces), _p(out)))
        return faces, out

    def debug_dice_keys(self, state: np.ndarray, sizes):
        """The game kernels' dice instantiation (``roll_counts<3>``) from explicit generator states: ``(keys[n][n_calls],
        state_out[n][6])``; key = six 3-bit face counts."""
        state = np.ascontiguousarray(state, dtype=np.uint64).reshape(-1, 6)
        sizes = np.ascontiguousarray(sizes, dtype=np.int32)
        n = len(state)
        keys = np.zeros((n, len(sizes)), dtype=np.uint32)
        out = np.zeros((n, 6), dtype=np.uint64)
        self._check(self._lib.fk_debug_dice_keys(self._ctx, C.c_int64(n), _p(state), C.c_int32(len(sizes)), _p(sizes), _p(keys), _p(out)))
        return keys, out
