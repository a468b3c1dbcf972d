"""``simulation.done.json`` in the reference's shape (non-v3 payload of ``write_stage_done``, ``utils/stage_completion.py:391-512``).

The reference's ``analysis/ingest.py:191-215`` reads the simulation contract (``shuffle_index_start``, ``shuffle_index_end``,
``num_shuffles``, ...) from the TOP LEVEL of the stamp — ``write_stage_done`` merges its ``metadata`` mapping into the payload
(:505-512) — next to ``schema_version``, ``completion_state``, ``inputs`` / ``input_identities``, ``outputs`` / ``output_identities``,
``code_identity`` and ``stage_identity_sha256``.  Everything that is a formula is the reference's formula; the values that name
the PRODUCER — ``config_sha`` / ``stage_config_sha`` (digests of the reference's whole AppConfig serialisation,
``config.py:2107-2129``) and ``code_identity`` — are this engine's own digests and revision.

What this stamp is NOT: the artifact-contract version 3 stamp (``release_identity.write_v3_stage_completion``).  The reference's
``ingest`` refuses anything below v3 (``analysis/ingest.py:216-217``); the route to it is the binding of INTEGRATION.md §0, where
the reference's own writers publish v3 sidecars and the authenticated stamp from this engine's results."""
from __future__ import annotations

import hashlib
import json
import os
from pathlib import Path
from typing import Any, Iterable, Mapping, Sequence

SCHEMA_VERSION = 4              # stage_completion.py:38
LIFECYCLE_CONTRACT_VERSION = 1  # :40
SIMULATION_CACHE_KEY_VERSION = 4  # analysis/stage_registry: cache_key_version of the "simulation" stage


def freshness_sha256(freshness_key: Mapping[str, Any]) -> str:
    """stage_completion.py:63-72: canonical JSON (sorted keys, compact separators, UTF-8) -> SHA-256."""
    canonical = json.dumps(dict(freshness_key), sort_keys=True, separators=(",", ":"), ensure_ascii=False)
    return hashlib.sha256(canonical.encode("utf-8")).hexdigest()


def _sha256_file(path: Path) -> str:
    digest = hashlib.sha256()
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1024 * 1024), b""):
            digest.update(chunk)
    return digest.hexdigest()


def path_content_identity(path: Path, *, logical_role: str, known: Mapping[str, tuple] | None = None) -> dict[str, object]:
    """stage_completion.py:150-181: path-independent exact-byte identity of one file or directory.  (Plain os calls: a rows-on run
    stamps one shard per shuffle, and pathlib costs more per file than reading and hashing a 5-KB shard does.)"""
    import stat as _stat

    name = os.fspath(path)
    if known is not None and name in known:  # written by this run: (bytes, sha256) came back from the writer with the manifest line
        size, digest = known[name]
        return {"logical_role": logical_role, "kind": "file", "byte_length": int(size), "content_sha256": digest, "sidecar_sha256": None}
    try:
        st = os.stat(name)
    except FileNotFoundError:
        return {"logical_role": logical_role, "kind": "missing"}
    if _stat.S_ISREG(st.st_mode):
        if st.st_size <= (8 << 20):
            with open(name, "rb") as fh:
                digest = hashlib.sha256(fh.read()).hexdigest()
        else:
            digest = _sha256_file(Path(name))
        sidecar = name + ".sidecar.json"
        return {"logical_role": logical_role, "kind": "file", "byte_length": st.st_size, "content_sha256": digest,
                "sidecar_sha256": _sha256_file(Path(sidecar)) if os.path.isfile(sidecar) else None}
    path = Path(name)
    entries = [{"relative_path": child.relative_to(path).as_posix(), "byte_length": child.stat().st_size,
                "content_sha256": _sha256_file(child)}
               for child in sorted((item for item in path.rglob("*") if item.is_file()), key=lambda p: p.as_posix())]
    return {"logical_role": logical_role, "kind": "directory", "entry_count": len(entries), "tree_sha256": freshness_sha256({"entries": entries})}


def path_identities(paths: Sequence[Path], *, prefix: str, known: Mapping[str, tuple] | None = None) -> list[dict[str, object]]:
    """In path order.  One thread: a shard is read and hashed in ~25 us, and a pool of threads taking turns at the GIL for work
    items that small measured 20x SLOWER inside a `farkle run` (11 s for 20 000 shards against 0.5 s)."""
    return [path_content_identity(p, logical_role=f"{prefix}_{i:04d}", known=known) for i, p in enumerate(paths)]


def stage_identity_sha256(*, stage: str | None, stage_config_sha: str | None, cache_key_version: int,
                          freshness_key: Mapping[str, Any] | None, code_identity: Mapping[str, object],
                          run_lineage_sha256: str | None, input_identities: Sequence[Mapping[str, object]]) -> str:
    """stage_completion.py:214-236."""
    return freshness_sha256({"lifecycle_contract_version": LIFECYCLE_CONTRACT_VERSION, "stage_key": stage,
                             "stage_cache_key_version": cache_key_version, "stage_config_identity": stage_config_sha,
                             "method_versions": dict(freshness_key or {}), "code_identity": dict(code_identity),
                             "run_lineage_sha256": run_lineage_sha256, "upstream_identities": list(input_identities)})


_STAGING_PREFIXES = ("._tmp_", "._artifact_v3_", "._sidecar_v3_", "._manifest_v3_", "._manifest_sidecar_v3_")


def completion_output_files(paths: Iterable[Path], done_path: Path) -> list[str]:
    """simulation/runner.py:434-461: directories are expanded (sorted by POSIX path) so that a stamp never authenticates itself;
    a directory whose manifest has a sidecar is represented by that manifest alone; sidecars and staging files are skipped."""
    done = os.path.realpath(done_path)
    files: list[str] = []  # (plain strings: a rows-on run lists one shard per shuffle, and 51 200 Path objects cost 0.3 s)
    for path in paths:
        path = Path(path)
        if path.is_dir():
            sealed = [c for c in (path / "manifest.jsonl", path / "metrics_manifest.jsonl")
                      if c.is_file() and c.with_name(f"{c.name}.sidecar.json").is_file()]
            if sealed:
                files.extend(str(c) for c in sealed)
                continue
            found = []
            for d, _, names in os.walk(path):  # (os.walk, one realpath per DIRECTORY: a rows-on run lists one shard per shuffle)
                real_dir = os.path.realpath(d)
                for name in names:
                    if name.endswith(".sidecar.json") or name.startswith(_STAGING_PREFIXES):
                        continue
                    if os.path.join(real_dir, name) != done:
                        found.append(os.path.join(d, name))
            files.extend(sorted(found))
        else:
            files.append(str(path))
    return list(dict.fromkeys(files))


def write_stage_done(done_path: Path, *, inputs: Iterable[Path], outputs: Iterable[Path], stage: str, config_sha: str | None,
                     stage_config_sha: str | None, cache_key_version: int, freshness_key: Mapping[str, Any] | None,
                     code_identity: Mapping[str, object], run_lineage_sha256: str | None = None, status: str = "success",
                     reason: str | None = None, metadata: Mapping[str, Any] | None = None,
                     known_identities: Mapping[str, tuple] | None = None) -> dict[str, Any]:
    """The payload of stage_completion.py:473-512 for a successful stage, written atomically.  Returns the payload.
    ``known_identities``: ``{path: (bytes, sha256)}`` of files this run wrote and hashed while it had their bytes in memory (row
    shards without sidecars); every other path is read."""
    input_paths, output_paths = [os.fspath(p) for p in inputs], [os.fspath(p) for p in outputs]
    if status == "success":
        missing = [p for p in (*input_paths, *output_paths) if not (known_identities and p in known_identities) and not os.path.exists(p)]
        if missing:
            raise FileNotFoundError(f"cannot publish successful completion with missing paths: {missing}")
    completion_state = "complete_valid" if status == "success" else "blocked_by_cap" if status == "blocked_by_cap" else "partial_resumable"
    fresh = None if freshness_key is None else dict(freshness_key)
    input_ids = path_identities(input_paths, prefix="input")
    output_ids = path_identities(output_paths, prefix="output", known=known_identities)
    resolved_stage_sha = stage_config_sha if stage_config_sha is not None else config_sha
    payload: dict[str, Any] = {
        "schema_version": SCHEMA_VERSION, "lifecycle_contract_version": LIFECYCLE_CONTRACT_VERSION, "stage": stage,
        "config_sha": config_sha, "stage_config_sha": resolved_stage_sha, "cache_key_version": cache_key_version,
        "freshness_key": fresh, "freshness_sha256": None if fresh is None else freshness_sha256(fresh),
        "completion_state": completion_state, "inputs": input_paths, "input_identities": input_ids,
        "outputs": output_paths, "output_identities": output_ids, "code_identity": dict(code_identity),
        "run_lineage_sha256": run_lineage_sha256,
        "stage_identity_sha256": stage_identity_sha256(stage=stage, stage_config_sha=resolved_stage_sha, cache_key_version=cache_key_version,
                                                       freshness_key=fresh, code_identity=code_identity,
                                                       run_lineage_sha256=run_lineage_sha256, input_identities=input_ids),
        "status": status, "reason": reason, "blocking_dependency": None, "upstream_stage": None}
    if metadata:
        collisions = set(payload).intersection(metadata)
        if collisions:
            raise ValueError(f"completion metadata collides with reserved fields: {sorted(collisions)}")
        payload.update(dict(metadata))
    done_path = Path(done_path)
    done_path.parent.mkdir(parents=True, exist_ok=True)
    tmp = done_path.with_name(f"._tmp_{done_path.name}")
    tmp.write_text(json.dumps(payload, indent=2, sort_keys=True) + "\n", encoding="utf-8")
    os.replace(tmp, done_path)
    return payload


def simulation_stage_config_sha(base_stage_config_sha: str, root_seed: int, n_players: int, game_profile_sha256: str | None) -> str:
    """simulation/runner.py:325-335: the shared simulation scope bound to one concrete (root, k) cell."""
    identity: dict[str, object] = {"base_stage_config_sha": base_stage_config_sha, "root_seed": int(root_seed), "n_players": int(n_players)}
    if game_profile_sha256 is not None:
        identity["game_profile_sha256"] = game_profile_sha256
    return freshness_sha256(identity)


def freshness_key(cfg, game_profile_sha256: str | None = None) -> dict[str, Any]:
    """``AppConfig.freshness_key`` (config.py:527-573): the versioned statistical contract of the run.  ``artifact_contract`` and
    ``k_aggregation`` are sections this package carries opaquely; the reference's defaults apply to what they do not set, except
    ``artifact_contract_version``: this engine's writers implement contract version 2 (sidecars.py), so that is what they state."""
    from .rows import OUTCOME_SCHEMA_VERSION, TOURNAMENT_METHOD_VERSION

    contract = dict(cfg.opaque.get("artifact_contract") or {})
    kagg = dict(cfg.opaque.get("k_aggregation") or {})
    counts: set = set()
    for value in cfg.sim.n_players_list:
        try:
            counts.add(int(value))
        except (TypeError, ValueError):
            counts.add(str(value))
    weights = kagg.get("k_weights")
    out: dict[str, Any] = {
        "artifact_contract_version": 2, "estimand_version": int(contract.get("estimand_version", 2)),
        "schema_version": int(contract.get("schema_version", 2)), "rng_scheme_version": int(cfg.rng.scheme_version),
        "outcome_schema_version": OUTCOME_SCHEMA_VERSION, "tournament_method_version": TOURNAMENT_METHOD_VERSION,
        "baseline_version": int(contract.get("baseline_version", 1)), "k_support_version": int(contract.get("k_support_version", 1)),
        "weighting_version": int(contract.get("weighting_version", 1)), "conditioning_version": int(contract.get("conditioning_version", 2)),
        "multiplicity_version": int(contract.get("multiplicity_version", 1)),
        "candidate_family_version": int(contract.get("candidate_family_version", 1)), "baseline": "chance_rate_by_k",
        "required_player_counts": sorted(counts, key=lambda v: (isinstance(v, str), str(v))),
        "k_aggregation_method": kagg.get("method", "equal-k"),
        "k_weights": None if weights is None else {str(k): float(v) for k, v in sorted(weights.items())},
        "conditioning": "unconditional_default", "multiplicity": "holm_h2h"}
    if game_profile_sha256 is not None:
        out["game_profile_sha256"] = game_profile_sha256
    return out
