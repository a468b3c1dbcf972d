"""Diagnostic: what one 32-row, 46-column row shard costs on one core — Arrow encoding, file creation, the tmp + rename.
usage: python tools/time_shard_write.py [n_shards=2048] [dir]"""
import os, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, pyarrow as pa, pyarrow.parquet as pq
from farkle_ii_amd.backend import row_dtype
from farkle_ii_amd.tournament import rows_to_table, _shard_worker_init

_shard_worker_init()
k, gps = 2, 32
n_sh = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
d = tempfile.mkdtemp(dir=sys.argv[2] if len(sys.argv) > 2 else None)
rows = np.zeros(n_sh * gps, dtype=row_dtype(k))
sh = np.repeat(np.arange(n_sh, dtype=np.int64), gps)
gi = np.tile(np.arange(gps, dtype=np.int32), n_sh)
kw = dict(root_seed=1, shuffle_index=sh, game_index=gi, deterministic_batch_id=np.zeros(n_sh * gps, dtype=np.int32), shuffle_seed=sh,
          game_seed=sh, rng_purpose_namespace=102)
table = rows_to_table(rows, k, np.arange(64, dtype=np.int32), **kw)
t0 = time.perf_counter()
table = rows_to_table(rows, k, np.arange(64, dtype=np.int32), **kw)
print(f"convert (vectorised, warm)      {(time.perf_counter() - t0) / n_sh * 1e3:7.3f} ms per shard, {table.num_columns} columns")
opts = dict(write_statistics=False, use_dictionary=False)

def encode_only(i):
    pq.write_table(table.slice(i * gps, gps), pa.BufferOutputStream(), **opts)

def arrow_tmp_rename(i):
    out = os.path.join(d, f"a{i}.parquet")
    pq.write_table(table.slice(i * gps, gps), out + ".tmp", **opts)
    os.replace(out + ".tmp", out)

def buffer_tmp_rename(i):
    sink = pa.BufferOutputStream()
    pq.write_table(table.slice(i * gps, gps), sink, **opts)
    out = os.path.join(d, f"b{i}.parquet")
    fd = os.open(out + ".tmp", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    os.write(fd, sink.getvalue())
    os.close(fd)
    os.replace(out + ".tmp", out)

def buffer_direct(i):
    sink = pa.BufferOutputStream()
    pq.write_table(table.slice(i * gps, gps), sink, **opts)
    fd = os.open(os.path.join(d, f"c{i}.parquet"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    os.write(fd, sink.getvalue())
    os.close(fd)

for fn in (encode_only, arrow_tmp_rename, buffer_tmp_rename, buffer_direct):
    t0 = time.perf_counter()
    for i in range(n_sh):
        fn(i)
    print(f"{fn.__name__:31s} {(time.perf_counter() - t0) / n_sh * 1e3:7.3f} ms per shard", flush=True)
