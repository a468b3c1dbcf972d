"""Artifact-contract version 3 of a STANDALONE `farkle run`: authenticated sidecars, sealed shard manifests and
the authenticated ``simulation.done.json``.

The fixture (tests/golden/contract_v3_vectors.json, oracle/gen_contract_v3.py) holds what the reference did in the build container:
its validators, ``simulation_is_complete``, ingest's source snapshot and ``analyze ingest`` itself ACCEPTED a standalone tree of this
package, and its own writers' documents for the same configuration.  Here, without the reference:

* ``farkle_ii_amd.contract_v3`` reproduces every document of the REFERENCE's own run byte for byte from the identities of its artifacts;
* a standalone ``farkle run --code-identity`` — on the oracle stub and (-m gpu) on the HIP engine — writes exactly the documents the
  reference accepted, validates them itself, notices tampering and a foreign code identity, and resumes as a no-op.
"""
from __future__ import annotations

import base64
import hashlib
import json
from pathlib import Path

import pytest
import yaml

import golden_util as gu

GOLD = gu.load("contract_v3_vectors.json")
COMMIT, DIRTY = GOLD["code_identity"]["commit"], GOLD["code_identity"]["dirty_fingerprint_sha256"]
SIDE = ".sidecar.json"


@pytest.fixture(params=["oracle-stub", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request):
    from farkle_ii_amd import engine as eng_mod

    if request.param == "hip":
        eng_mod.set_engine(None)
        yield eng_mod.get_engine()
    else:
        import oracle_engine_stub

        stub = oracle_engine_stub.Engine(0)
        eng_mod.set_engine(stub)
        yield stub
    eng_mod.set_engine(None)


def _config(tmp_path: Path, **extra) -> Path:
    payload = {key: dict(val) for key, val in GOLD["config"].items()}
    payload["io"] = {"results_dir_prefix": str(tmp_path / "out"), "analysis_subdir": "analysis"}
    payload.update(extra)
    path = tmp_path / "tiny.yaml"
    path.write_text(yaml.safe_dump(payload))
    return path


def _kind(rel: str) -> str:
    name = Path(rel).name
    return ("strategy_manifest" if name == "strategy_manifest.parquet" else "workload_plan" if name == "simulation_workload_plan.json"
            else "checkpoint" if name.endswith("_checkpoint.pkl") else "checkpoint_summary" if name.endswith("_checkpoint.parquet")
            else "metrics_summary" if name.endswith("_metrics.parquet") else "row_shard" if name.startswith("rows_") else "metric_chunk")


def test_reference_documents_are_reproduced_byte_for_byte(tmp_path):
    """The reference's own run of the tiny oracle configuration: its 61 sidecars, 4 sealed manifests and 2 completions from the identities
    of the artifacts they bind (five whole artifacts travel, so the byte / schema / format inspection is exercised too)."""
    from farkle_ii_amd import contract_v3 as c3
    from farkle_ii_amd.config import load_app_config

    cfg = load_app_config(_config(tmp_path), seed_list_len=1)
    root = cfg.results_root
    docs = GOLD["reference_documents"]
    sc = c3.SimulationContract(cfg, c3.make_code_identity(COMMIT, DIRTY))
    for rel, blob in GOLD["reference_artifact_samples_b64"].items():
        (root / rel).parent.mkdir(parents=True, exist_ok=True)
        (root / rel).write_bytes(base64.b64decode(blob))
    manifest = root / "strategy_manifest.parquet"
    # inputs first: later sidecars name them as sources (their identities are read back from the sidecars on disk)
    assert sc.write_sidecar(manifest, "strategy_manifest", n_players=2, sources=(), support_counts=[2, 4]).decode() == docs["strategy_manifest.parquet"]["sidecar"]
    for k in (2, 4):
        plan = root / f"{k}_players" / "simulation_workload_plan.json"
        assert sc.write_sidecar(plan, "workload_plan", n_players=k, sources=[manifest]).decode() == docs[f"{k}_players/simulation_workload_plan.json"]["sidecar"]
    n_checked = 3
    templates: dict = {}
    for rel, rec in docs.items():
        name = Path(rel).name
        k = int(rel.split("_players/")[0]) if "_players/" in rel else 2
        src = [manifest, root / f"{k}_players" / "simulation_workload_plan.json"]
        if name in ("strategy_manifest.parquet", "simulation_workload_plan.json", "simulation.done.json"):
            continue
        if name in ("manifest.jsonl", "metrics_manifest.jsonl"):
            native = GOLD["reference_manifest_records"][rel]
            (root / rel).parent.mkdir(parents=True, exist_ok=True)
            sc.publish_manifest(root / rel, [json.loads(line) for line in native.splitlines()], n_players=k)
            assert (root / rel).read_text() == native and (root / (rel + SIDE)).read_text() == rec["sidecar"], rel
            n_checked += 1
            continue
        kind = _kind(rel)
        if "sidecar" in rec:  # a whole document: identity in, text out
            want = json.loads(rec["sidecar"])["artifact"]
            got = sc.sidecar_for_identity(rel, kind, byte_length=rec["byte_length"], content_sha256=rec["content_sha256"], arrow_schema=want["arrow_schema"],
                                          format_identity=want["format_identity"], n_players=k, sources=src)
            assert got.decode() == rec["sidecar"], rel
            if (root / rel).exists():  # ... and from the artifact's own bytes
                assert sc.sidecar_bytes(root / rel, kind, n_players=k, sources=src).decode() == rec["sidecar"], rel
            if kind in ("row_shard", "metric_chunk"):
                import pyarrow as pa

                fields = want["arrow_schema"]
                assert c3.identity_sha256({"schema_version": 2, "fields": fields["fields"]}) == fields["fingerprint_sha256"]
                templates[(kind, k)] = (rec["sidecar"], fields)
        n_checked += 1
    # the other shards of a directory: the template text around three values hashes to the reference's sidecar
    for rel, rec in docs.items():
        if "sidecar_sha256" not in rec:
            continue
        kind = _kind(rel)
        k = int(rel.split("_players/")[0])
        whole, fields = templates[(kind, k)]
        first = json.loads(whole)["artifact"]
        text = whole.replace(first["content_sha256"], "@sha@").replace(f'"byte_length":{first["byte_length"]}', '"byte_length":@len@')
        text = text.replace(first["location"]["relative_path"], rel)
        body = json.loads(text.replace("@sha@", rec["content_sha256"]).replace("@len@", str(rec["byte_length"])))
        digest = body.pop("sidecar_contract_sha256")
        assert digest != c3.identity_sha256(body)  # (the first shard's digest does not fit another shard)
        doc = c3.canonical_json_bytes({**body, "sidecar_contract_sha256": c3.identity_sha256(body)}) + b"\n"
        assert hashlib.sha256(doc).hexdigest() == rec["sidecar_sha256"], rel
    # completions: from the sidecars on disk (artifact bytes are not read: placeholders of the right length stand for them)
    for rel, rec in docs.items():
        if "sidecar" in rec and not (root / (rel + SIDE)).exists():
            (root / rel).parent.mkdir(parents=True, exist_ok=True)
            (root / (rel + SIDE)).write_text(rec["sidecar"])
        if "sidecar" in rec and not (root / rel).exists():
            (root / rel).write_bytes(b"\0" * rec["byte_length"])
    for k in (2, 4):
        nd = root / f"{k}_players"
        outputs = [nd / f"{k}p_checkpoint.pkl", nd / "simulation_workload_plan.json", nd / f"{k}p_checkpoint.parquet", nd / f"{k}p_metrics.parquet",
                   manifest, nd / f"{k}p_rows" / "manifest.jsonl", nd / f"{k}p_metric_chunks" / "metrics_manifest.jsonl"]
        assert (c3.canonical_json_bytes(sc.completion(outputs)) + b"\n").decode() == docs[f"{k}_players/simulation.done.json"]["completion"]
        n_checked += 1
    assert n_checked >= 3 + 4 + 2 + 8


def test_standalone_run_writes_what_the_reference_accepted(engine, tmp_path):
    """`farkle run --code-identity`: every sidecar, sealed manifest and completion has the SHA-256 of the document the reference's
    validators, simulation_is_complete, ingest snapshot and `analyze ingest` accepted (fixture: reference_verdict_on_standalone_tree)."""
    from farkle_ii_amd import contract_v3 as c3
    from farkle_ii_amd.cli import main
    from farkle_ii_amd.config import load_app_config

    verdict = GOLD["reference_verdict_on_standalone_tree"]
    assert verdict["simulation_is_complete"] == {"2": True, "4": True} and verdict["other_code_identity_is_stale"] is True
    assert verdict["ingest_source_snapshot"]["2"]["shards"] == 21 and set(verdict["analyze_ingest"]["rows"].values()) == {42, 21}
    assert len(verdict["validate_artifact_sidecar"]) == 57 and len(verdict["load_immutable_manifest_sidecar"]) == 4
    cfg_path = _config(tmp_path)
    main(["--config", str(cfg_path), "--log-level", "WARNING", "run", "--code-identity", f"{COMMIT}:{DIRTY}"])
    cfg = load_app_config(cfg_path, seed_list_len=1)
    root = cfg.results_root
    want = GOLD["standalone_documents"]
    got = {str(p.relative_to(root))[:-len(SIDE)] for p in root.rglob("*" + SIDE)} | {str(p.relative_to(root)) for p in root.rglob("simulation.done.json")}
    assert got == set(want)
    for rel, rec in want.items():
        if "completion" in rec:
            assert (root / rel).read_text() == rec["completion"], rel
            continue
        data = (root / rel).read_bytes()
        assert (len(data), hashlib.sha256(data).hexdigest()) == (rec["byte_length"], rec["content_sha256"]), f"{rel}: artifact bytes differ from the accepted run's"
        assert hashlib.sha256((root / (rel + SIDE)).read_bytes()).hexdigest() == rec["sidecar_sha256"], rel
    # the package's own validator agrees, under this identity only, and notices a changed sidecar / manifest
    cfg._code_identity = c3.make_code_identity(COMMIT, DIRTY)
    sc = c3.SimulationContract(cfg, cfg._code_identity)
    done = root / "2_players" / "simulation.done.json"
    assert sc.is_complete(done) and sc.is_complete(root / "4_players" / "simulation.done.json")
    assert not c3.SimulationContract(cfg, c3.make_code_identity("0" * 40)).is_complete(done)
    victim = root / "2_players" / "2p_rows" / "manifest.jsonl"
    original = victim.read_bytes()
    victim.write_bytes(original.replace(b'"rows":2', b'"rows":3', 1))
    assert not sc.is_complete(done)
    victim.write_bytes(original)
    assert sc.is_complete(done)
    # a second invocation finds both player counts complete and touches nothing
    before = {p: p.stat().st_mtime_ns for p in root.rglob("*") if p.is_file() and p.name != "active_config.yaml"}
    main(["--config", str(cfg_path), "--log-level", "WARNING", "run", "--code-identity", f"{COMMIT}:{DIRTY}"])
    assert {p: p.stat().st_mtime_ns for p in before} == before
    # without the identity the stamp cannot be judged: a clear refusal, not a silent re-run
    with pytest.raises(c3.ContractError, match="code identity"):
        main(["--config", str(cfg_path), "--log-level", "WARNING", "run"])


def test_contract_selection_and_code_identity_parsing(tmp_path):
    from farkle_ii_amd import contract_v3 as c3
    from farkle_ii_amd.config import load_app_config
    from farkle_ii_amd.runner import _Sidecars

    cfg = load_app_config(_config(tmp_path), seed_list_len=1)
    assert cfg.artifact_contract_version == 2 and _Sidecars(cfg, 2, [], True).v3 is None  # --sidecars alone: the structural contract
    cfg._code_identity = c3.parse_code_identity(f"{COMMIT}:{DIRTY}")
    assert cfg.artifact_contract_version == 3 and _Sidecars(cfg, 2, [], True).v3 is not None
    assert cfg._code_identity == {"commit": COMMIT, "policy": "development_dirty", "state": "development_dirty", "dirty_fingerprint_sha256": DIRTY}
    assert c3.parse_code_identity(COMMIT) == {"commit": COMMIT, "policy": "release_clean", "state": "clean", "dirty_fingerprint_sha256": None}
    assert c3.parse_code_identity(f"{COMMIT}::development_dirty")["policy"] == "development_dirty"
    for bad in ("abc", COMMIT + ":xyz", f"{COMMIT}:{DIRTY}:nonsense", f"{COMMIT}:{DIRTY}:a:b"):
        with pytest.raises(c3.ContractError):
            c3.parse_code_identity(bad)
    stated = load_app_config(_config(tmp_path, artifact_contract={"artifact_contract_version": 3}), seed_list_len=1)
    with pytest.raises(c3.ContractError, match="--code-identity"):
        _Sidecars(stated, 2, [], True)  # version 3 asked for, nothing to sign with
    # private identities never enter a configuration digest or the persisted configuration
    from farkle_ii_amd.sidecars import config_hash

    plain = load_app_config(_config(tmp_path), seed_list_len=1)
    assert config_hash(plain) == config_hash(cfg)


def test_resolve_code_identity_follows_git(tmp_path):
    """``--reference-checkout``: HEAD for a clean tree; for a dirty one the SHA-256 over the staged diff, the worktree diff and the
    inventoried untracked files (authenticated_contract.py:408-462)."""
    import subprocess

    from farkle_ii_amd import contract_v3 as c3

    repo = tmp_path / "checkout"
    (repo / "src").mkdir(parents=True)
    (repo / "src" / "a.py").write_text("x = 1\n")
    env = {"GIT_AUTHOR_NAME": "t", "GIT_AUTHOR_EMAIL": "t@t", "GIT_COMMITTER_NAME": "t", "GIT_COMMITTER_EMAIL": "t@t", "HOME": str(tmp_path), "PATH": "/usr/bin:/bin"}
    for cmd in (["git", "init", "-q"], ["git", "add", "-A"], ["git", "commit", "-qm", "one"]):
        subprocess.run(cmd, cwd=repo, check=True, env=env)
    head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=repo, check=True, capture_output=True, text=True).stdout.strip()
    assert c3.resolve_code_identity(repo) == {"commit": head, "policy": "development_dirty", "state": "clean", "dirty_fingerprint_sha256": None}
    (repo / "src" / "a.py").write_text("x = 2\n")
    (repo / "src" / "new.py").write_text("y = 1\n")
    (repo / "notes.txt").write_text("outside the inventory\n")
    dirty = c3.resolve_code_identity(repo)
    diff = subprocess.run(["git", "diff", "--binary", "--no-ext-diff"], cwd=repo, check=True, capture_output=True).stdout
    want = hashlib.sha256(b"tracked-index\0" + b"tracked-worktree\0" + diff + b"untracked\0src/new.py\0"
                          + hashlib.sha256(b"y = 1\n").hexdigest().encode()).hexdigest()
    assert dirty == {"commit": head, "policy": "development_dirty", "state": "development_dirty", "dirty_fingerprint_sha256": want}


def test_interrupted_v3_run_resumes_to_the_documents_of_an_uninterrupted_one(tmp_path, monkeypatch):
    """A contract-v3 run cut down before its second checkpoint (one launch group per deterministic batch), then resumed: the row shards
    and sidecars of the owned batches stay, the replayed ones are rewritten, and the finished tree carries exactly the documents of an
    uninterrupted run — the ones the reference accepted."""
    import oracle_engine_stub

    from farkle_ii_amd import contract_v3 as c3
    from farkle_ii_amd import engine as eng_mod
    from farkle_ii_amd import runner
    from farkle_ii_amd.config import load_app_config

    eng_mod.set_engine(oracle_engine_stub.Engine(0))
    try:
        cfg_path = _config(tmp_path)
        cfg = load_app_config(cfg_path, seed_list_len=1)
        cfg._code_identity = c3.make_code_identity(COMMIT, DIRTY)
        cfg.sim.sidecars = True
        monkeypatch.setattr(runner, "MAX_GAMES_PER_LAUNCH", 1)  # one deterministic batch per launch group
        real_write = runner._atomic_write_bytes
        saves = {"n": 0}

        def dying_write(path, content):
            if path.name.endswith("checkpoint.pkl"):
                saves["n"] += 1
                if saves["n"] == 2:
                    raise KeyboardInterrupt("power cut before the second checkpoint")
            real_write(path, content)

        monkeypatch.setattr(runner, "_atomic_write_bytes", dying_write)
        with pytest.raises(KeyboardInterrupt):
            runner.run_single_n(cfg, 2)
        n_dir = cfg.n_dir(2)
        assert not (n_dir / "simulation.done.json").exists() and not (n_dir / "2p_rows" / ("manifest.jsonl" + SIDE)).exists()
        monkeypatch.setattr(runner, "_atomic_write_bytes", real_write)
        runner.run_single_n(cfg, 2)
        runner.run_single_n(cfg, 4)
    finally:
        eng_mod.set_engine(None)
    root = cfg.results_root
    for rel, rec in GOLD["standalone_documents"].items():
        if "completion" in rec:
            assert (root / rel).read_text() == rec["completion"], rel
        else:
            assert hashlib.sha256((root / (rel + SIDE)).read_bytes()).hexdigest() == rec["sidecar_sha256"], rel
    sc = c3.SimulationContract(cfg, cfg._code_identity)
    assert sc.is_complete(root / "2_players" / "simulation.done.json") and sc.is_complete(root / "4_players" / "simulation.done.json")


def _v3_rank_main(rank: int, world: int, port: int, cfg_path: str) -> None:
    import os
    import sys

    root = Path(__file__).resolve().parent.parent
    for p in (root, root / "oracle", root / "tests"):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist

    import oracle_engine_stub
    from farkle_ii_amd import contract_v3 as c3
    from farkle_ii_amd import engine as eng_mod
    from farkle_ii_amd import runner
    from farkle_ii_amd.config import load_app_config

    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng_mod.set_engine(oracle_engine_stub.Engine(0))
    runner.MAX_GAMES_PER_LAUNCH = 40  # several launch groups, each cut over the two ranks
    cfg = load_app_config(Path(cfg_path), seed_list_len=1)
    cfg._code_identity = c3.make_code_identity(COMMIT, DIRTY)
    cfg.sim.sidecars = True
    for k in (2, 4):
        runner.run_single_n(cfg, k)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_write_the_same_v3_documents(tmp_path):
    """Two gloo ranks: each writes the row shards (and their sidecars) of its own batches from the same template, rank 0 seals the
    manifests and publishes the completion — the documents of the single-process run."""
    import socket

    import torch.multiprocessing as mp

    from farkle_ii_amd import contract_v3 as c3
    from farkle_ii_amd.config import load_app_config

    cfg_path = _config(tmp_path)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_v3_rank_main, args=(2, port, str(cfg_path)), nprocs=2, join=True)
    cfg = load_app_config(cfg_path, seed_list_len=1)
    root = cfg.results_root
    for rel, rec in GOLD["standalone_documents"].items():
        if "completion" in rec:
            assert (root / rel).read_text() == rec["completion"], rel
        else:
            assert hashlib.sha256((root / (rel + SIDE)).read_bytes()).hexdigest() == rec["sidecar_sha256"], rel


def test_manifest_sealed_from_its_text_equals_the_parsed_route(tmp_path):
    """``publish_manifest(path)`` seals a manifest whose lines are already canonical from its TEXT (one regular expression per line); the
    document and the sealed file equal those of the parsed-records route — also for lines out of coordinate order — and a manifest with a
    process id, white space or a torn last line takes the parsed route."""
    import yaml

    from farkle_ii_amd import contract_v3 as c3
    from farkle_ii_amd.config import load_app_config

    cfg_path = tmp_path / "c.yaml"
    cfg_path.write_text(yaml.safe_dump({"io": {"results_dir_prefix": str(tmp_path / "out")}, "sim": {"n_players_list": [3], "seed_list": [5], "row_dir": "rows"}}))
    cfg = load_app_config(cfg_path, seed_list_len=1)
    sc = c3.SimulationContract(cfg, c3.make_code_identity("b" * 40))
    row_dir = cfg.simulation_row_dir(3)
    row_dir.mkdir(parents=True)
    records = [{"path": f"rows_5_3p_{i:012d}.parquet", "rows": 7, "root_seed": 5, "n_players": 3, "shuffle_index": i, "shuffle_seed": 2**32 - 1 - i,
                "deterministic_batch_id": i // 2, "byte_length": 1000 + i, "data_sha256": hashlib.sha256(b"d%d" % i).hexdigest(),
                "sidecar_sha256": hashlib.sha256(b"s%d" % i).hexdigest(), "schema_fingerprint_sha256": "0f" * 32} for i in range(9)]
    canonical = [json.dumps(r, sort_keys=True, separators=(",", ":")) for r in records]

    def sealed(lines, parsed: bool):
        path = row_dir / "manifest.jsonl"
        path.write_text("\n".join(lines) + "\n")
        if parsed:
            recs = []
            for line in lines:
                try:
                    recs.append(json.loads(line))
                except json.JSONDecodeError:
                    pass
            doc = sc.publish_manifest(path, recs, n_players=3)
        else:
            doc = sc.publish_manifest(path, n_players=3)
        return json.dumps(doc, sort_keys=True), path.read_bytes(), path.with_name(path.name + ".sidecar.json").read_bytes()

    want = sealed(canonical, parsed=True)
    assert sealed(canonical, parsed=False) == want and c3._canonical_manifest_text("\n".join(canonical) + "\n") is not None
    shuffled = canonical[4:] + canonical[:4]
    assert sealed(shuffled, parsed=False) == want
    for other in ([json.dumps({**r, "pid": 77}, sort_keys=True) for r in records],                       # a process id, default separators
                  canonical[:-1] + [canonical[-1][:40]],                                                   # a torn last line
                  canonical[:3] + [json.dumps({**records[3], "note": "a b"}, sort_keys=True, separators=(",", ":"))] + canonical[4:]):  # another shape
        assert c3._canonical_manifest_text("\n".join(other) + "\n") is None
    assert sealed([json.dumps({**r, "pid": 77}, sort_keys=True) for r in records], parsed=False) == want  # (pid dropped, lines re-encoded)
    torn = canonical[:-1] + [canonical[-1][:40]]
    assert sealed(torn, parsed=False) == sealed(torn, parsed=True)
    with pytest.raises(c3.ContractError):
        sealed(canonical + canonical[:1], parsed=False)  # the same coordinate twice
