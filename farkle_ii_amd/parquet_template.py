"""The schema-only parts of a row shard's Parquet footer, taken from a file Arrow wrote for the same schema.

``csrc/fk_shard_writer.h`` frames the device's column images as Parquet files itself (3 - 5 ms of Arrow encoding per shard is what made
rows mode host-bound).  What a reader needs to restore the EXACT Arrow schema of ``raw_simulation_schema_for(k)``
(``src/farkle/utils/schema_helpers.py:79-90`` of the reference: int16 / int8 logical types, ``list<item: string>`` with its field name
and nullability) lives in three members of ``FileMetaData`` that depend on the schema alone: the ``SchemaElement`` list, the
``ARROW:schema`` key-value entry and the column orders.  Rather than re-deriving them, this module asks Arrow once per player count —
an empty table of the schema, written to memory with the shard writer options — and cuts those members out of the footer as raw Thrift
spans, which the native writer splices around the row-group metadata it generates.  A compact-protocol reader of the footer is all that
takes (``read_struct``); it doubles as the checker of the native writer's own footers in the tests.
"""
from __future__ import annotations

import struct
from functools import lru_cache
from typing import Any

_EXPECTED_BASE = ("root_seed", "k", "shuffle_index", "game_index", "deterministic_batch_id", "shuffle_seed", "termination_status",
                  "hit_safety_limit", "outcome_schema_version", "winner_seat", "winner_strategy", "game_seed", "rng_scheme_version",
                  "rng_purpose_namespace", "seat_ranks", "winning_score", "victory_margin", "n_rounds")
_EXPECTED_SEAT = ("score", "farkles", "rolls", "highest_turn", "strategy", "rank", "loss_margin", "smart_five_uses", "n_smart_five_dice",
                  "smart_one_uses", "n_smart_one_dice", "hot_dice", "n_turns", "hit_max_rounds")


def _varint(b: bytes, p: int) -> tuple[int, int]:
    result = shift = 0
    while True:
        c = b[p]
        p += 1
        result |= (c & 0x7F) << shift
        shift += 7
        if not c & 0x80:
            return result, p


def _zigzag(v: int) -> int:
    return (v >> 1) ^ -(v & 1)


def _read_value(b: bytes, p: int, t: int) -> tuple[Any, int]:
    if t in (1, 2):  # BOOL_TRUE / BOOL_FALSE (the value lives in the field header)
        return t == 1, p
    if t == 3:
        return b[p], p + 1
    if t in (4, 5, 6):
        v, p = _varint(b, p)
        return _zigzag(v), p
    if t == 7:
        return struct.unpack("<d", b[p:p + 8])[0], p + 8
    if t == 8:
        n, p = _varint(b, p)
        return bytes(b[p:p + n]), p + n
    if t in (9, 10):
        head = b[p]
        p += 1
        n, et = head >> 4, head & 15
        if n == 15:
            n, p = _varint(b, p)
        out = []
        for _ in range(n):
            if et in (1, 2):
                out.append(b[p] == 1)
                p += 1
            else:
                v, p = _read_value(b, p, et)
                out.append(v)
        return out, p
    if t == 12:
        return read_struct(b, p)
    raise ValueError(f"unsupported Thrift compact type {t}")


def read_struct(b: bytes, p: int = 0) -> tuple[dict[int, tuple[Any, int, int, int]], int]:
    """Thrift compact struct at ``b[p:]`` -> ({field id: (value, span start incl. field header, span end, type)}, end offset)."""
    out: dict[int, tuple[Any, int, int, int]] = {}
    fid = 0
    while True:
        start = p
        head = b[p]
        p += 1
        if head == 0:
            return out, p
        delta, t = head >> 4, head & 15
        if delta:
            fid += delta
        else:
            v, p = _varint(b, p)
            fid = _zigzag(v)
        value, p = _read_value(b, p, t)
        out[fid] = (value, start, p, t)


def footer_of(blob: bytes) -> bytes:
    if blob[:4] != b"PAR1" or blob[-4:] != b"PAR1":
        raise ValueError("not a Parquet file")
    n = struct.unpack("<I", blob[-8:-4])[0]
    return blob[-8 - n:-8]


@lru_cache(maxsize=None)
def shard_footer_template(k: int) -> dict[str, Any]:
    """{footer_head, footer_kv, footer_orders: raw Thrift spans; leaf_type: physical type per leaf; leaf_paths: path_in_schema per leaf}
    of a row shard of ``k`` players."""
    import pyarrow as pa
    import pyarrow.parquet as pq

    from .rows import raw_simulation_schema_for
    from .tournament import SHARD_WRITER_OPTIONS

    sink = pa.BufferOutputStream()
    pq.write_table(raw_simulation_schema_for(k).empty_table(), sink, **SHARD_WRITER_OPTIONS)
    footer = footer_of(sink.getvalue().to_pybytes())
    meta, end = read_struct(footer)
    if end != len(footer) or sorted(meta) != [1, 2, 3, 4, 5, 6, 7]:
        raise ValueError(f"unexpected FileMetaData members {sorted(meta)} in Arrow's footer: the shard writer splices fields 1, 2, 5 and 7")
    spans = {fid: footer[s:e] for fid, (_, s, e, _) in meta.items()}
    if meta[1][2] != meta[2][1] or meta[2][2] != meta[3][1]:
        raise ValueError("FileMetaData fields 1-3 are not adjacent")
    columns = meta[4][0][0][1][0]
    leaf_type = [col[3][0][1][0] for col in columns]
    leaf_paths = [[part.decode("utf-8") for part in col[3][0][3][0]] for col in columns]
    expected = list(_EXPECTED_BASE) + [f"P{s}_{name}" for s in range(1, k + 1) for name in _EXPECTED_SEAT]
    if [p[0] for p in leaf_paths] != expected or len(leaf_paths) != 18 + 14 * k:
        raise ValueError("the raw simulation schema's columns are not the ones the shard writer encodes")
    return {"footer_head": spans[1] + spans[2], "footer_kv": spans[5], "footer_orders": spans[7], "leaf_type": leaf_type, "leaf_paths": leaf_paths}
