"""TEST INFRASTRUCTURE ONLY — let the REFERENCE judge a STANDALONE ``farkle run`` tree written under artifact-contract version 3, in the
build container, and freeze the proof.

The reference is imported from /root/reference (oracle/ref_import.py); the engine behind the standalone run is the CPU oracle stub
(tests/oracle_engine_stub.py — there is no GPU in this container).  Four things happen on the reference's tiny oracle configuration
(tests/helpers/raw_simulation_oracle.py:80-190: k in {2, 4}, rows + metric chunks + expanded metrics):

(a) ``python -m farkle_ii_amd --config tiny.yaml run --code-identity <fixture identity>`` writes a results tree on its own.
(b) THE REFERENCE'S VALIDATORS read that tree with the reference's own AppConfig of the same YAML and the same supplied code identity
    (``cfg._code_identity``, the API ``release_identity.publish_staged_v3_from_metadata`` :683 and ``_completion_contract`` :1115 take):
      * ``validate_artifact_sidecar`` (utils/artifact_contract.py:629; the v3 branch = ``release_identity.validate_v3_compat``: bytes,
        Arrow schema / format identity, location) on every ordinary artifact;
      * ``load_immutable_manifest_sidecar`` on both sealed manifests;
      * ``simulation.runner.simulation_is_complete`` (:274-317 -> ``resolve_v3_stage_state`` -> ``classify_authenticated_lifecycle``):
        the authenticated completion, stage identity recomputed by the reference, must classify COMPLETE_VALID;
      * ``analysis.ingest._canonical_row_shards`` (:162-336): ingest's whole source snapshot — completion contract, manifest sidecar and
        SHA-256, every shard's sidecar hash / byte length / schema fingerprint, the coordinate-sorted manifest root, directory = manifest;
      * ``analysis.ingest.run`` (:915): the reference's ``analyze ingest`` itself, end to end, over the standalone tree.
(c) The reference's OWN ``run_single_n`` writes the same configuration; ``farkle_ii_amd.contract_v3`` is then pointed at the reference's
    artifact bytes and must reproduce every one of its sidecars, both sealed manifests and both completion stamps BYTE FOR BYTE.
(d) tests/golden/contract_v3_vectors.json: the configuration, the code identity, every accepted document of (a) (sidecar / manifest
    sidecar / completion text with the identity of the artifact it binds) and of (c), and what the validators returned.  Replayed
    without the reference by tests/test_contract_v3.py on the oracle stub and (-m gpu) on the HIP engine.  Only data travels.

    python oracle/gen_contract_v3.py
"""
from __future__ import annotations

import base64
import hashlib
import json
import shutil
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
for p in (HERE, ROOT, ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import ref_import  # noqa: E402

ref_import.import_reference()

import yaml  # noqa: E402
from farkle.analysis import ingest as ref_ingest  # noqa: E402
from farkle.config import load_app_config as ref_load_app_config  # noqa: E402
from farkle.simulation import runner as ref_runner  # noqa: E402
from farkle.utils.artifact_contract import sidecar_path as ref_sidecar_path  # noqa: E402
from farkle.utils.artifact_contract import validate_artifact_sidecar  # noqa: E402
from farkle.utils.authenticated_contract import CodeIdentity, load_immutable_manifest_sidecar  # noqa: E402

OUT = ROOT / "tests" / "golden" / "contract_v3_vectors.json"
# the reference's tiny oracle configuration (tests/helpers/raw_simulation_oracle.py:80-190), simulation part (as oracle/gen_binding.py)
TINY_SIM = {"n_players_list": [2, 4], "seed": 11, "seed_list": [11], "n_jobs": 1, "expanded_metrics": True, "row_dir": "rows",
            "metric_chunk_dir": "metric_chunks", "desired_sec_per_chunk": 1, "ckpt_every_sec": 1, "score_thresholds": [500],
            "dice_thresholds": [2], "smart_five_opts": [False], "smart_one_opts": [False], "consider_score_opts": [True],
            "consider_dice_opts": [True], "auto_hot_dice_opts": [False, True], "run_up_score_opts": [False],
            "include_stop_at": False, "include_stop_at_heuristic": False}
TINY_CONFIG = {"sim": TINY_SIM, "screening": {"resolution_delta": 0.4, "interval_confidence": 0.95},
               "batching": {"target_batches": 3, "min_shuffles_per_batch": 2}}
TAG = b"farkle_ii_amd contract-v3 fixture (oracle/gen_contract_v3.py)"
COMMIT, DIRTY = hashlib.sha1(TAG).hexdigest(), hashlib.sha256(TAG).hexdigest()


def write_config(tmp: Path, name: str) -> Path:
    payload = {key: dict(val) for key, val in TINY_CONFIG.items()}
    payload["io"] = {"results_dir_prefix": str(tmp / name / "out"), "analysis_subdir": "analysis"}
    (tmp / name).mkdir(parents=True, exist_ok=True)
    path = tmp / name / "tiny.yaml"
    path.write_text(yaml.safe_dump(payload))
    return path


def reference_config(cfg_path: Path):
    cfg = ref_load_app_config(cfg_path, seed_list_len=1)
    cfg._code_identity = CodeIdentity(commit=COMMIT, policy="development_dirty", state="development_dirty", dirty_fingerprint_sha256=DIRTY)
    # an execution-only budget (config.py:406-411; no identity reads it): this generator's own process — both packages, Arrow, pandas — sits
    # at the default 768-MiB process-tree warning level, where the reference's scheduler stops admitting work
    cfg.resources.process_tree_warning_threshold_mb = 1536
    return cfg


def documents(root: Path, digests_only: bool = False) -> dict[str, dict]:
    """Every contract document under ``root`` with the identity of what it binds (relative path -> record)."""
    out = {}
    for side in sorted(root.rglob("*.sidecar.json")):
        artifact = side.with_name(side.name[:-len(".sidecar.json")])
        data = artifact.read_bytes()
        record = {"byte_length": len(data), "content_sha256": hashlib.sha256(data).hexdigest()}
        text = side.read_text(encoding="utf-8")
        # shards differ in three values only: the first shard of a directory travels whole, the others as the SHA-256 of their sidecar
        whole = not digests_only and not (artifact.name.startswith(("rows_", "metrics_0")) and not artifact.name.endswith(("_000000000000.parquet", "_000001.parquet")))
        record["sidecar" if whole else "sidecar_sha256"] = text if whole else hashlib.sha256(text.encode("utf-8")).hexdigest()
        out[str(artifact.relative_to(root))] = record
    for done in sorted(root.rglob("simulation.done.json")):
        out[str(done.relative_to(root))] = {"completion": done.read_text(encoding="utf-8")}
    return out


def standalone_run(tmp: Path) -> tuple[Path, Path]:
    """(a): the package's own CLI, oracle stub engine, contract v3."""
    from farkle_ii_amd import engine as eng_mod
    from farkle_ii_amd.cli import main
    from oracle_engine_stub import Engine as StubEngine

    cfg_path = write_config(tmp, "standalone")
    eng_mod.set_engine(StubEngine(0))
    try:
        main(["--config", str(cfg_path), "--log-level", "WARNING", "run", "--code-identity", f"{COMMIT}:{DIRTY}"])
    finally:
        eng_mod.set_engine(None)
    return cfg_path, tmp / "standalone" / "out_seed_11"


def reference_judges(cfg_path: Path, root: Path) -> dict:
    """(b): the reference's validators over the standalone tree."""
    cfg = reference_config(cfg_path)
    assert cfg.results_root.resolve() == root.resolve(), (cfg.results_root, root)
    verdict: dict = {"validate_artifact_sidecar": [], "load_immutable_manifest_sidecar": [], "simulation_is_complete": {}, "ingest_source_snapshot": {}}
    for side in sorted(root.rglob("*.sidecar.json")):
        artifact = side.with_name(side.name[:-len(".sidecar.json")])
        rel = str(artifact.relative_to(root))
        if artifact.name in ("manifest.jsonl", "metrics_manifest.jsonl"):
            loaded = load_immutable_manifest_sidecar(artifact)
            assert loaded.manifest_sha256 == hashlib.sha256(artifact.read_bytes()).hexdigest(), rel
            verdict["load_immutable_manifest_sidecar"].append([rel, loaded.summary.entry_count])
        else:
            view = validate_artifact_sidecar(artifact)  # v3 branch: full byte / schema / location validation
            assert view.artifact_contract_version == 3 and view.code_revision == COMMIT, rel
            verdict["validate_artifact_sidecar"].append([rel, view.operation])
    for k in (2, 4):
        ok = ref_runner.simulation_is_complete(cfg, k)
        assert ok is True, f"the reference's simulation_is_complete rejects the standalone {k}p completion"
        verdict["simulation_is_complete"][str(k)] = ok
        snap = ref_ingest._canonical_row_shards(cfg.n_dir(k), cfg, k)
        verdict["ingest_source_snapshot"][str(k)] = {"shards": len(snap.shards), "manifest_sha256": snap.manifest_sha256,
                                                    "manifest_root": snap.manifest_root.summary.root_sha256}
    # tampering is noticed: one flipped byte in a row shard -> the validator refuses that shard; restored afterwards
    victim = sorted(root.rglob("rows_*.parquet"))[3]
    original = victim.read_bytes()
    victim.write_bytes(original[:-9] + bytes([original[-9] ^ 1]) + original[-8:])
    try:
        validate_artifact_sidecar(victim)
        raise SystemExit("the reference accepted a tampered row shard")
    except Exception as exc:  # noqa: BLE001 - ArtifactContractError / ArtifactMismatchError
        verdict["tampered_shard_refused"] = type(exc).__name__
    victim.write_bytes(original)
    # and a stamp written under ANOTHER code identity is stale for this consumer
    other = reference_config(cfg_path)
    other._code_identity = CodeIdentity(commit="0" * 40, policy="release_clean", state="clean", dirty_fingerprint_sha256=None)
    assert ref_runner.simulation_is_complete(other, 2) is False
    verdict["other_code_identity_is_stale"] = True
    # the reference's `analyze ingest`, end to end
    ref_ingest.run(cfg)
    ingested = sorted(str(p.relative_to(cfg.analysis_dir)) for p in cfg.analysis_dir.rglob("*.parquet"))
    assert ingested, "analyze ingest wrote nothing"
    import pyarrow.parquet as pq

    verdict["analyze_ingest"] = {"files": ingested, "rows": {rel: pq.read_metadata(cfg.analysis_dir / rel).num_rows for rel in ingested}}
    return verdict


def reference_writes(tmp: Path) -> tuple[Path, Path]:
    """(c), first half: the reference's own run of the same configuration."""
    cfg_path = write_config(tmp, "reference")
    cfg = reference_config(cfg_path)
    for k in (2, 4):
        ref_runner.run_single_n(cfg, k)
    return cfg_path, cfg.results_root


def reproduce_reference_documents(cfg_path: Path, root: Path) -> int:
    """(c), second half: contract_v3 over the reference's artifact bytes == the reference's documents, byte for byte."""
    import pyarrow.parquet as pq

    from farkle_ii_amd import contract_v3 as c3
    from farkle_ii_amd.config import load_app_config

    cfg = load_app_config(cfg_path, seed_list_len=1)
    sc = c3.SimulationContract(cfg, c3.make_code_identity(COMMIT, DIRTY))
    checked = 0

    def same(path: Path, kind: str, **kw) -> None:
        nonlocal checked
        assert sc.sidecar_bytes(path, kind, **kw) == c3.sidecar_path(path).read_bytes(), f"sidecar of {path} differs from the reference's"
        checked += 1

    manifest = root / "strategy_manifest.parquet"
    same(manifest, "strategy_manifest", n_players=2, sources=(), support_counts=[2, 4])
    for k in (2, 4):
        nd = root / f"{k}_players"
        plan = nd / "simulation_workload_plan.json"
        same(plan, "workload_plan", n_players=k, sources=[manifest])
        src = [manifest, plan]
        same(nd / f"{k}p_checkpoint.pkl", "checkpoint", n_players=k, sources=src)
        same(nd / f"{k}p_checkpoint.parquet", "checkpoint_summary", n_players=k, sources=src)
        same(nd / f"{k}p_metrics.parquet", "metrics_summary", n_players=k, sources=src)
        sealed = []
        for d, kind, pattern, name in ((nd / f"{k}p_rows", "row_shard", "rows_*.parquet", "manifest.jsonl"),
                                       (nd / f"{k}p_metric_chunks", "metric_chunk", "metrics_0*.parquet", "metrics_manifest.jsonl")):
            files = sorted(d.glob(pattern))
            template = sc.shard_template(kind, d, pq.read_schema(files[0]), n_players=k, sources=src)
            for f in files:
                same(f, kind, n_players=k, sources=src)
                data = f.read_bytes()
                text, side_sha = c3.fill_shard_template(template, f.name, len(data), hashlib.sha256(data).hexdigest())
                assert text == c3.sidecar_path(f).read_bytes(), f"template sidecar of {f} differs from the reference's"
            man = d / name
            native, side = man.read_bytes(), c3.sidecar_path(man).read_bytes()
            sc.publish_manifest(man, [json.loads(line) for line in native.decode("utf-8").splitlines()], n_players=k)
            assert man.read_bytes() == native and c3.sidecar_path(man).read_bytes() == side, f"sealed manifest {man} differs from the reference's"
            sealed.append(man)
            checked += 1
        done = nd / "simulation.done.json"
        outputs = [nd / f"{k}p_checkpoint.pkl", plan, nd / f"{k}p_checkpoint.parquet", nd / f"{k}p_metrics.parquet", manifest, *sealed]
        assert c3.canonical_json_bytes(sc.completion(outputs)) + b"\n" == done.read_bytes(), f"completion of {k}p differs from the reference's"
        assert sc.is_complete(done)
        checked += 1
    return checked


def main() -> None:
    tmp = Path(tempfile.mkdtemp(prefix="fk_contract_v3_"))
    try:
        # (the reference's own run goes first: its memory guard counts this process's RSS — 768 MiB — and (a) + (b) leave Arrow pools behind)
        ref_cfg_path, ref_root = reference_writes(tmp)
        cfg_path, root = standalone_run(tmp)
        standalone_docs = documents(root, digests_only=True)  # (what the reference accepts below; the texts that travel whole are its own)
        verdict = reference_judges(cfg_path, root)
        print(f"(b) the reference accepts the standalone tree: {len(verdict['validate_artifact_sidecar'])} sidecars, "
              f"{len(verdict['load_immutable_manifest_sidecar'])} sealed manifests, simulation_is_complete {verdict['simulation_is_complete']}, "
              f"ingest snapshot {({k: v['shards'] for k, v in verdict['ingest_source_snapshot'].items()})} shards, "
              f"analyze ingest wrote {verdict['analyze_ingest']['rows']}")
        reference_docs = documents(ref_root)
        checked = reproduce_reference_documents(ref_cfg_path, ref_root)
        print(f"(c) contract_v3 reproduces {checked} of the reference's own documents byte for byte")
        # a few whole artifacts travel (small ones), so the replay also exercises the file-reading path
        samples = {}
        for rel in ("strategy_manifest.parquet", "2_players/simulation_workload_plan.json", "4_players/simulation_workload_plan.json", "2_players/2p_metric_chunks/metrics_000001.parquet",
                    "2_players/2p_rows/rows_11_2p_000000000000.parquet", "2_players/2p_checkpoint.parquet"):
            samples[rel] = base64.b64encode((ref_root / rel).read_bytes()).decode("ascii")
        doc = {"generated_by": "oracle/gen_contract_v3.py (reference imported in the build container; engine = CPU oracle stub)",
               "config": TINY_CONFIG, "code_identity": {"commit": COMMIT, "dirty_fingerprint_sha256": DIRTY},
               "reference_verdict_on_standalone_tree": verdict, "standalone_documents": standalone_docs,
               "reference_documents": reference_docs, "reference_artifact_samples_b64": samples,
               "reference_manifest_records": {rel: (ref_root / rel).read_text(encoding="utf-8") for rel in (
                   "2_players/2p_rows/manifest.jsonl", "2_players/2p_metric_chunks/metrics_manifest.jsonl",
                   "4_players/4p_rows/manifest.jsonl", "4_players/4p_metric_chunks/metrics_manifest.jsonl")}}
        OUT.write_text(json.dumps(doc, sort_keys=True))
        print(OUT.name, OUT.stat().st_size)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
