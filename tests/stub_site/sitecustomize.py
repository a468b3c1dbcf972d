"""TEST INFRASTRUCTURE ONLY.  With FK_TEST_STUB_ENGINE=1 in the environment and this directory on PYTHONPATH, every Python
process started by a test (e.g. the ranks of `python -m torch.distributed.run -m farkle_ii_amd ... run`) gets the oracle-backed
engine stub as its process-wide engine, so the real CLI runs on a GPU-less host.  The product never imports this."""
import os

if os.environ.get("FK_TEST_STUB_ENGINE") == "1":
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent.parent
    for p in (root, root / "oracle", root / "tests"):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    import oracle_engine_stub
    from farkle_ii_amd import engine

    engine.set_engine(oracle_engine_stub.Engine(0))
