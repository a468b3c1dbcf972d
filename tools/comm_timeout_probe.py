"""Diagnostic: fk_comm_init for a world of two with ONE rank present, under a 4-second deadline; progress on stderr."""
import faulthandler, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
faulthandler.dump_traceback_later(40, exit=True)
from farkle_ii_amd.backend import Engine, FarkleHipError
eng = Engine(0)
eng.set_option("comm_timeout_ms", 4000)
t0 = time.time()
print("unique id ...", file=sys.stderr, flush=True)
cid = eng.comm_unique_id()
print("comm_init ...", file=sys.stderr, flush=True)
try:
    eng.comm_init(cid, 0, 2)
    print("comm_init returned success?!", file=sys.stderr, flush=True)
except FarkleHipError as exc:
    print(f"refused after {time.time() - t0:.1f} s: code {exc.code}: {exc}", file=sys.stderr, flush=True)
print("ranks", eng.comm_ranks(), file=sys.stderr, flush=True)
eng.close()
print("closed", file=sys.stderr, flush=True)
