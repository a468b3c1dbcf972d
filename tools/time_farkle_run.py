"""Diagnostic: `farkle run --metrics` end to end (AppConfig -> plan -> launches -> artifacts) on BASELINE config 2.
Three runs: rows off (counts + metrics: the whole 10^7 games), metric chunk files on (per-batch tallies), rows on (row shards +
manifest, on a reduced shuffle count: the reference's format is one parquet file per shuffle).  Prints one JSON object with
wall time, engine time (inside Engine.tournament) and host time per phase.
usage: python tools/time_farkle_run.py [rows_shuffles=6400] [out.json] [only=name,name,...]
(names: rows_off, rows_off_metric_chunks, rows_on, config3_rows_off, config3_rows_off_metric_chunks, mega_rows_off, mega_rows_on, mega_rows_on_v3)"""
import json, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import yaml
from farkle_ii_amd import runner, tournament as rt
from farkle_ii_amd.cli import main
from farkle_ii_amd.engine import get_engine

ROOT = Path(__file__).resolve().parent.parent
rows_shuffles = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
ONLY = set(sys.argv[3].split(",")) if len(sys.argv) > 3 else None
want = lambda *names: ONLY is None or any(n in ONLY for n in names)  # noqa: E731
FIXTURE_IDENTITY = "%s:%s" % ("ab" * 20, "cd" * 32)  # a code identity to sign contract-v3 documents with (any well-formed one costs the same)
base = yaml.safe_load((ROOT / "configs" / "bench_config2.yaml").read_text())
eng = get_engine()  # (module-level script: the row-shard writers are separate `python -m farkle_ii_amd.shard_writer` processes, not forks of this one)
acc = {"engine_s": 0.0, "shard_s": 0.0, "calls": 0}
real_t, real_w = eng.tournament, rt.write_row_shards

def timed_tournament(*a, **kw):
    t0 = time.perf_counter()
    try:
        return real_t(*a, **kw)
    finally:
        acc["engine_s"] += time.perf_counter() - t0
        acc["calls"] += 1

def timed_shard(*a, **kw):
    t0 = time.perf_counter()
    try:
        return real_w(*a, **kw)
    finally:
        acc["shard_s"] += time.perf_counter() - t0

eng.tournament = timed_tournament
rt.write_row_shards = timed_shard
real_tc, real_wc = eng.tournament_columns, rt.write_row_shards_from_columns

def timed_tournament_columns(*a, **kw):
    t0 = time.perf_counter()
    try:
        return real_tc(*a, **kw)
    finally:
        acc["engine_s"] += time.perf_counter() - t0
        acc["calls"] += 1

def timed_shard_columns(*a, **kw):
    t0 = time.perf_counter()
    try:
        out = real_wc(*a, **kw)
    finally:
        acc["shard_s"] += time.perf_counter() - t0
    if not kw.get("deferred_write"):
        return out

    def timed_write():  # (the prepared library call, made on the shard thread)
        t1 = time.perf_counter()
        try:
            return out()
        finally:
            acc["shard_s"] += time.perf_counter() - t1

    return timed_write

eng.tournament_columns = timed_tournament_columns
rt.write_row_shards_from_columns = timed_shard_columns
out = {"config": "configs/bench_config2.yaml (k=2, 64-strategy grid, root seed 42), `farkle run --metrics`", "runs": {}}
import os
with tempfile.TemporaryDirectory(prefix="fk_e2e_", dir=os.environ.get("FK_E2E_DIR")) as tmp:  # FK_E2E_DIR=/dev/shm: file IO into memory
    out["results_dir"] = tmp
    def run(name, sim_extra, batching=None, screening=None):
        cfg = json.loads(json.dumps(base))
        cfg["io"]["results_dir_prefix"] = str(Path(tmp) / name)
        cfg["sim"].update(sim_extra)
        if batching: cfg["batching"].update(batching)
        if screening: cfg["screening"].update(screening)
        path = Path(tmp) / f"{name}.yaml"
        path.write_text(yaml.safe_dump(cfg))
        for key in acc: acc[key] = 0
        t0 = time.perf_counter()
        main(["--config", str(path), "--log-level", "WARNING", "run", "--metrics"])
        wall = time.perf_counter() - t0
        plan = json.loads(next((Path(tmp)).glob(f"{name}_seed_42/2_players/simulation_workload_plan.json")).read_text())
        games = plan["required_games"]
        out["runs"][name] = {"games": games, "shuffles": plan["required_shuffles"], "wall_s": wall, "games_per_s": games / wall,
                             "engine_s": acc["engine_s"], "engine_calls": acc["calls"], "host_s": wall - acc["engine_s"],
                             "row_shard_write_s": acc["shard_s"], "row_writer_threads": runner.ROW_WRITER_THREADS}
        print(name, json.dumps(out["runs"][name]), flush=True)
    run("warm", {}, {"target_batches": 4, "min_shuffles_per_batch": 8}, {"resolution_delta": 0.3})  # import / first-launch costs out of the way
    del out["runs"]["warm"]
    if want("rows_off"): run("rows_off", {})
    if want("rows_off_metric_chunks"): run("rows_off_metric_chunks", {"metric_chunk_dir": "metric_chunks"})
    per_batch = max(1, rows_shuffles // 100)
    if want("rows_on"): run("rows_on", {"row_dir": "rows", "metric_chunk_dir": "metric_chunks"}, {"target_batches": 100, "min_shuffles_per_batch": per_batch}, {"resolution_delta": 0.5})
    # BASELINE config 3 through the same command: k = 4, the default 5 160-strategy grid, 10^8 games (configs/bench_config3.yaml)
    base3 = yaml.safe_load((ROOT / "configs" / "bench_config3.yaml").read_text())

    def run3(name, sim_extra):
        cfg = json.loads(json.dumps(base3))
        cfg["io"]["results_dir_prefix"] = str(Path(tmp) / name)
        cfg["sim"].update(sim_extra)
        path = Path(tmp) / f"{name}.yaml"
        path.write_text(yaml.safe_dump(cfg))
        for key in acc: acc[key] = 0
        t0 = time.perf_counter()
        main(["--config", str(path), "--log-level", "WARNING", "run", "--metrics"])
        wall = time.perf_counter() - t0
        plan = json.loads(next((Path(tmp)).glob(f"{name}_seed_0/4_players/simulation_workload_plan.json")).read_text())
        games = plan["required_games"]
        out["runs"][name] = {"config": "configs/bench_config3.yaml (k=4, 5 160-strategy grid, root seed 0)", "games": games,
                             "shuffles": plan["required_shuffles"], "wall_s": wall, "games_per_s": games / wall, "engine_s": acc["engine_s"],
                             "engine_calls": acc["calls"], "host_s": wall - acc["engine_s"], "host_share": (wall - acc["engine_s"]) / wall}
        print(name, json.dumps(out["runs"][name]), flush=True)
    if want("config3_rows_off", "config3_rows_off_metric_chunks"):
        run3("config3_warm", {})
        del out["runs"]["config3_warm"]
    if want("config3_rows_off"): run3("config3_rows_off", {})
    if want("config3_rows_off_metric_chunks"): run3("config3_rows_off_metric_chunks", {"metric_chunk_dir": "metric_chunks"})
    # the reference's production list of player counts through one `farkle run --metrics` (configs/bench_mega_rows_off.yaml)
    base_m = yaml.safe_load((ROOT / "configs" / "bench_mega_rows_off.yaml").read_text())

    def run_mega(name, config="bench_mega_rows_off.yaml", extra_args=()):
        import shutil
        cfg = yaml.safe_load((ROOT / "configs" / config).read_text())
        cfg["io"]["results_dir_prefix"] = str(Path(tmp) / name)
        path = Path(tmp) / f"{name}.yaml"
        path.write_text(yaml.safe_dump(cfg))
        for key in acc: acc[key] = 0
        t0 = time.perf_counter()
        main(["--config", str(path), "--log-level", "WARNING", "run", "--metrics", *extra_args])
        wall = time.perf_counter() - t0
        root = Path(tmp) / f"{name}_seed_102"
        plans = {int(p.parent.name.split("_")[0]): json.loads(p.read_text()) for p in root.glob("*_players/simulation_workload_plan.json")}
        games = sum(p["required_games"] for p in plans.values())
        shards = list(root.glob("*_players/*_rows/rows_*.parquet"))
        out["runs"][name] = {"config": f"configs/{config} (k in {{2,3,4,5,6,8,10,12}}, 5 160-strategy grid, root seed 102, default screening resolution)"
                                       + (" + contract-v3 sidecars / sealed manifests / authenticated completion" if extra_args else ""),
                             "games": games, "games_per_k": {str(k): plans[k]["required_games"] for k in sorted(plans)},
                             "shuffles_per_k": {str(k): plans[k]["required_shuffles"] for k in sorted(plans)}, "wall_s": wall, "games_per_s": games / wall,
                             "engine_s": acc["engine_s"], "engine_calls": acc["calls"], "host_s": wall - acc["engine_s"],
                             "host_share": (wall - acc["engine_s"]) / wall, "row_shard_write_s": acc["shard_s"],
                             "row_shards": len(shards), "row_shard_bytes": sum(p.stat().st_size for p in shards),
                             "sidecars": len(list(root.rglob("*.sidecar.json"))), "row_writer_processes": runner.ROW_WRITER_THREADS,
                             "reference_cpu_hours_at_its_published_12_worker_rate": games / 1142.9 / 3600}
        print(name, json.dumps(out["runs"][name]), flush=True)
        shutil.rmtree(root, ignore_errors=True)  # (rows on: gigabytes of shards per run)
    if want("mega_rows_off", "mega_rows_on", "mega_rows_on_v3"):
        run_mega("mega_warm")
        del out["runs"]["mega_warm"]
    if want("mega_rows_off"): run_mega("mega_rows_off")
    # the reference's production command AS SHIPPED: rows on (configs/bench_mega_rows_on.yaml) — and the same with the contract-v3
    # documents the reference's `analyze ingest` needs
    if want("mega_rows_on"): run_mega("mega_rows_on", "bench_mega_rows_on.yaml")
    if want("mega_rows_on_v3"): run_mega("mega_rows_on_v3", "bench_mega_rows_on.yaml", ("--code-identity", FIXTURE_IDENTITY))
for rec in out["runs"].values():
    rec.setdefault("host_share", rec["host_s"] / rec["wall_s"])
r = out["runs"]
if "rows_on" in r:
    out["host_bottleneck"] = ("rows on: one parquet file + one manifest line per shuffle of 32 games (the reference's row-shard format, "
                              "run_tournament.py:530-558): %.2f ms of host wall time per shard, %.2f ms of it Arrow conversion (one per 1 024 "
                              "shuffles) + parquet encoding + file creation on %d writer processes, against %.4f ms of engine time per shuffle"
                              % (1e3 * r["rows_on"]["host_s"] / r["rows_on"]["shuffles"],
                                 1e3 * r["rows_on"]["row_shard_write_s"] / r["rows_on"]["shuffles"], runner.ROW_WRITER_THREADS,
                                 1e3 * r["rows_on"]["engine_s"] / r["rows_on"]["shuffles"]))
out["reference_published"] = {"games_per_s_1_worker": 279.0, "games_per_s_12_workers": 1142.9,
                              "where": "docs/remediation/task4c_simulation_execution_report.md:109-116 (Ryzen 7 3700X, rows + metrics)"}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    Path(sys.argv[2]).write_text(json.dumps(out, indent=1) + "\n")
