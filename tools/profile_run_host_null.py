"""Diagnostic (CPU): host time of `farkle run --metrics` with a NULL engine (a plausible tally made up in no time): what the Python side costs
per player count on wide grids, where the oracle-backed stub would take minutes.  usage: python tools/profile_run_host_null.py [config] [top=30] [extra run args ...]   (rows on: column images of random games, so that the
native shard writer, manifests, sidecars and the completion stamp are timed at production shape)"""
import cProfile, io, json, pstats, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import yaml
from farkle_ii_amd import engine as eng_mod
from farkle_ii_amd.cli import main


class NullEngine:
    def set_option(self, *a): pass
    def hint_next(self, *a, **k): pass
    def device_info(self): return {"name": "null"}
    def timing(self): return {}
    def coordinate_seeds(self, coords, want32=False):
        n = len(coords)
        return (np.arange(n, dtype=np.uint32),) if want32 else np.arange(n, dtype=np.uint64)
    def tournament(self, table, k, root, lo, hi, shuffles_per_batch=None, **kw):
        S, n = len(table), hi - lo
        spb = shuffles_per_batch or n
        nb = (n + spb - 1) // spb
        t = np.zeros((nb, S, 26), dtype=np.int64)
        per = np.minimum(spb, n - np.arange(nb) * spb)
        t[:, :, 1] = per[:, None]; t[:, :, 2] = per[:, None]
        rng = np.random.default_rng(lo)
        t[:, :, 0] = rng.integers(0, per[:, None] // k + 1, (nb, S))
        t[:, :, 4:15] = t[:, :, 0:1] * 7; t[:, :, 15:26] = t[:, :, 0:1] * 50
        return {"tally": t, "rows": None, "perms": None, "seat_stats": None, "seat_ratio_sums": None}
    def game_seeds(self, purpose, root, k, lo, hi, gps):
        return np.arange((hi - lo) * gps, dtype=np.uint32)
    columns_with_seeds = True
    def tournament_columns(self, table, k, root, lo, hi, ids, shuffles_per_batch=None, columns_out=None, shuffle_seeds_out=None, game_seeds_out=None, **kw):
        from farkle_ii_amd.backend import row_columns_bytes
        S, n = len(table), hi - lo
        gps = S // k
        stride = row_columns_bytes(k, gps)
        if shuffle_seeds_out is not None: shuffle_seeds_out[...] = np.arange(lo, hi, dtype=np.uint32)
        if game_seeds_out is not None: game_seeds_out[...] = np.arange(n * gps, dtype=np.uint32)
        if getattr(self, "_img", None) is None or self._img.shape != (n, stride):
            rng = np.random.default_rng(k)
            one = rng.integers(0, 200, stride, dtype=np.uint8)  # one random shard, repeated: the writer's cost does not depend on the values
            ni = (4 + 13 * k) * 4 * gps
            one[ni:ni + gps] = rng.random(gps) < 0.01
            one[ni + gps:] %= k
            self._img = np.broadcast_to(one, (n, stride)).copy()
        if columns_out is not None:  # (the runner lays the shard job out around ITS buffer)
            columns_out.reshape(-1)[:n * stride].reshape(n, stride)[...] = self._img
        return {"tally": self.tournament(table, k, root, lo, hi, shuffles_per_batch)["tally"], "columns": self._img}


eng_mod.set_engine(NullEngine())
cfg_path = ROOT / (sys.argv[1] if len(sys.argv) > 1 else "configs/bench_mega_rows_off.yaml")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
base = yaml.safe_load(cfg_path.read_text())
with tempfile.TemporaryDirectory(prefix="fk_hostprof_") as tmp:
    def run(name, profile=False):
        cfg = json.loads(json.dumps(base)); cfg["io"]["results_dir_prefix"] = str(Path(tmp) / name)
        p = Path(tmp) / f"{name}.yaml"; p.write_text(yaml.safe_dump(cfg))
        argv = ["--config", str(p), "--log-level", "WARNING", "run", "--metrics", *sys.argv[3:]]
        t0 = time.perf_counter()
        if profile:
            pr = cProfile.Profile(); pr.enable(); main(argv); pr.disable()
            s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(top); print(s.getvalue())
        else:
            main(argv)
        print(name, f"{(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    import shutil
    for name, prof in (("warm", False), ("timed", False), ("profiled", True)):
        run(name, prof)
        shutil.rmtree(Path(tmp) / f"{name}_seed_{base['sim']['seed_list'][0]}", ignore_errors=True)
