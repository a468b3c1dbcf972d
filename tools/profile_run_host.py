"""Diagnostic (CPU): where `farkle run --metrics` spends HOST time.  The engine is the oracle-backed test stub with its answers
memoised, so the second run of the same configuration costs the host path only; that run is profiled (cProfile, cumulative).
usage: python tools/profile_run_host.py [config=configs/bench_config2.yaml] [top=45]"""
import cProfile, io, json, pstats, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import yaml
import oracle_engine_stub as stub
from farkle_ii_amd import engine as eng_mod
from farkle_ii_amd.cli import main

cfg_path = ROOT / (sys.argv[1] if len(sys.argv) > 1 else "configs/bench_config2.yaml")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
e = stub.Engine(0)
memo, real = {}, e.tournament
def memo_tournament(table, k, root, lo, hi, **kw):
    key = (k, root, lo, hi, tuple(sorted((a, repr(b)) for a, b in kw.items())))
    if key not in memo:
        memo[key] = real(table, k, root, lo, hi, **kw)
    return memo[key]
e.tournament = memo_tournament
eng_mod.set_engine(e)
base = yaml.safe_load(cfg_path.read_text())
with tempfile.TemporaryDirectory(prefix="fk_hostprof_") as tmp:
    def run(name, profile=False):
        cfg = json.loads(json.dumps(base)); cfg["io"]["results_dir_prefix"] = str(Path(tmp) / name)
        p = Path(tmp) / f"{name}.yaml"; p.write_text(yaml.safe_dump(cfg))
        argv = ["--config", str(p), "--log-level", "WARNING", "run", "--metrics"]
        t0 = time.perf_counter()
        if profile:
            pr = cProfile.Profile(); pr.enable(); main(argv); pr.disable()
            s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(top); print(s.getvalue())
        else:
            main(argv)
        print(name, f"{(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    run("fill"); run("warm"); run("timed"); run("timed2"); run("profiled", True)

# timeline: where the engine call sits inside the run (memoised: its own duration is ~0)
marks = []
orig = e.tournament
def marked(*a, **kw):
    marks.append(("engine call", time.perf_counter()))
    time.sleep(0.025)  # the real call's duration at this container's speed (GIL released, as inside ctypes): what the helper thread can use
    r = orig(*a, **kw)
    marks.append(("engine return", time.perf_counter()))
    return r
e.tournament = marked
with tempfile.TemporaryDirectory(prefix="fk_hostprof_") as tmp:
    for rep in range(3):
        marks.clear()
        cfg = json.loads(json.dumps(base)); cfg["io"]["results_dir_prefix"] = str(Path(tmp) / f"t{rep}")
        p = Path(tmp) / f"t{rep}.yaml"; p.write_text(yaml.safe_dump(cfg))
        t0 = time.perf_counter()
        main(["--config", str(p), "--log-level", "WARNING", "run", "--metrics"])
        t1 = time.perf_counter()
        print(f"timeline: before the engine call {(marks[0][1] - t0) * 1e3:.1f} ms, after it {(t1 - marks[-1][1]) * 1e3:.1f} ms, total {(t1 - t0) * 1e3:.1f} ms")
