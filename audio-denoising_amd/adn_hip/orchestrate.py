"""Phase orchestrators of the stage scripts: command lines, log files, resolved-config dump and run
manifest of the reference's src/pipeline.py:10-78 and src/pipeline_v2.py:17-435.

Host-only: every stage is one subprocess of a script that lies next to the orchestrator script,
so the orchestrators keep working when only some of the stage scripts are the MI355X ones.
A selected phase whose script is not in that directory fails the run, as launching the missing
script does in the reference (src/pipeline_v2.py:123-135); ADN_ALLOW_MISSING_STAGES=1 records it as
skipped instead (partial installs), and the run then still ends with a non-zero exit code.
"""
import argparse
import hashlib
import json
import os
import platform
import subprocess
import sys
import time
from datetime import datetime, timezone
from pathlib import Path

from .stages import run_step as _run_logged

ALL_PHASES = [1, 2, 3, 4, 5, 6, "metrics"]

# phase -> (headline, [(result key, script, log file)])  (src/pipeline_v2.py:284-402)
PHASE_PLAN = {
    1: ("[PHASE 1] Analysis & Noise Profile",
        [("phase1_analyze", "analyze_audio.py", "phase1_analyze_stdout.log"),
         ("phase1_noise_profile", "noise_profile.py", "phase1_noise_profile_stdout.log")]),
    2: ("[PHASE 2] Primary Noise Reduction",
        [("phase2_wavelet", "wavelet_denoise.py", "phase2_wavelet_stdout.log"),
         ("phase2b_spectral_subtract", "spectral_subtract.py", "phase2b_spectral_subtract_stdout.log"),
         ("phase2c_wiener", "wiener_filter.py", "phase2c_wiener_stdout.log")]),
    3: ("[PHASE 3] Harmonic-Percussive Separation",
        [("phase3_hpss", "hpss_separate.py", "phase3_hpss_stdout.log")]),
    4: ("[PHASE 4] AI Enhancement (Optional)",
        [("phase4_ai", "ai_separate.py", "phase4_ai_stdout.log")]),
    5: ("[PHASE 5] Artifact Control",
        [("phase5_artifact", "artifact_control.py", "phase5_artifact_stdout.log")]),
    6: ("[PHASE 6] Mastering & Normalization",
        [("phase6_mastering", "mastering_chain.py", "phase6_mastering_stdout.log")]),
    "metrics": ("[METRICS] Quality Evaluation",
                [("metrics", "metrics_eval.py", "metrics_eval_stdout.log")]),
}


def file_sha256(path):
    """Hex SHA-256 of a file, None when it does not exist (src/pipeline_v2.py:41-50)."""
    if path is None or not os.path.exists(path):
        return None
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        while True:
            chunk = fh.read(1 << 20)
            if not chunk:
                break
            h.update(chunk)
    return h.hexdigest()


def git_info():
    """Branch and 8-character commit of the working directory, "unknown" outside a checkout."""
    def ask(*cmd):
        return subprocess.check_output(["git", *cmd], stderr=subprocess.DEVNULL).decode().strip()
    try:
        return {"branch": ask("branch", "--show-current"), "commit": ask("rev-parse", "HEAD")[:8]}
    except Exception:
        return {"branch": "unknown", "commit": "unknown"}


def system_info():
    return {"platform": platform.platform(), "python_version": platform.python_version(),
            "processor": platform.processor(), "hostname": platform.node(),
            "timestamp": datetime.now(timezone.utc).isoformat()}


def _importable(name):
    import importlib.util
    try:
        return importlib.util.find_spec(name) is not None
    except (ImportError, ValueError):
        return False


def check_dependencies(phase, strict=False):
    """Optional-package probe per phase (src/pipeline_v2.py:52-112): warnings only, except that
    strict mode turns a phase 4 without any separation package into an error."""
    missing = []
    if phase == 3 and not _importable("scipy"):
        missing.append("scipy")
    elif phase == 4:
        if not any(_importable(m) for m in ("demucs", "spleeter", "speechbrain")):
            print("[WARN] Phase 4: No AI dependencies found. Install with: pip install -r requirements-ai.txt")
            if strict:
                missing.append("demucs/spleeter/speechbrain (at least one)")
    elif phase == 6 and not _importable("pyloudnorm"):
        print("[WARN] Phase 6: pyloudnorm not found, using fallback LUFS calculation")
    elif phase == "metrics":
        for mod in ("pesq", "pystoi"):
            if not _importable(mod):
                print(f"[WARN] Metrics: {mod} not found. Optional install: pip install -r requirements-metrics.txt")
    if missing and strict:
        raise RuntimeError(f"Missing required dependencies for phase {phase}: {', '.join(missing)}")
    return not missing


def run_step(cmd, log_path, dry_run=False):
    """One stage subprocess with the result record of src/pipeline_v2.py:114-139."""
    if dry_run:
        print(f"[DRY-RUN] Would execute: {' '.join(cmd)}")
        return {"status": "dry-run", "elapsed": 0, "returncode": 0}
    t0 = time.time()
    try:
        _run_logged(cmd, log_path)
    except SystemExit as exc:                         # v2 raises RuntimeError where v1 exits
        raise RuntimeError(str(exc)) from None
    return {"status": "success", "elapsed": time.time() - t0, "returncode": 0}


def load_config(config_path):
    """YAML config with the required sections; `noise_sample` accepted for `noise`
    (src/pipeline_v2.py:141-160)."""
    import yaml
    with open(config_path, "r") as fh:
        cfg = yaml.safe_load(fh)
    for field in ("paths", "phases"):
        if field not in cfg:
            raise ValueError(f"Missing required config field: {field}")
    paths = cfg.get("paths", {})
    if "noise" not in paths and "noise_sample" in paths:
        paths["noise"] = paths["noise_sample"]
        cfg["paths"] = paths
    return cfg


def create_run_manifest(cfg, args, phase_results):
    """Provenance record of one run (src/pipeline_v2.py:162-211)."""
    paths = cfg["paths"]
    manifest = {
        "version": cfg.get("version", "2.0.0"),
        "run_id": f"run_{int(time.time())}",
        "timestamp": datetime.now(timezone.utc).isoformat(),
        "configuration": {
            "config_file": args.config,
            "config_hash": file_sha256(args.config),
            "seed": cfg.get("global", {}).get("seed", 42),
            "phases_requested": args.phases.split(",") if args.phases else [],
            "dry_run": args.dry_run,
            "strict_mode": args.strict,
        },
        "inputs": {"audio": {"path": paths["input"], "hash": file_sha256(paths["input"])},
                   "noise": {"path": paths["noise"], "hash": file_sha256(paths["noise"])}},
        "system": system_info(),
        "git": git_info(),
        "phases": phase_results,
        "outputs": {},
        "metrics": {},
        "total_elapsed_sec": sum(r.get("elapsed", 0) for r in phase_results.values()),
    }
    final_output = paths.get("final_output")
    if final_output and os.path.exists(final_output):
        manifest["outputs"]["final"] = {"path": final_output, "hash": file_sha256(final_output),
                                        "size_bytes": os.path.getsize(final_output)}
    metrics_path = paths.get("metrics_path")
    if metrics_path and os.path.exists(metrics_path):
        with open(metrics_path, "r") as fh:
            manifest["metrics"] = json.load(fh)
    return manifest


def select_phases(arg, cfg):
    """`--phases` list ("all", "1,2,metrics"), else the phases the config enables
    (src/pipeline_v2.py:240-264)."""
    if arg:
        if arg.lower() == "all":
            return list(ALL_PHASES)
        out = []
        for tok in arg.split(","):
            tok = tok.strip()
            if tok == "metrics":
                out.append("metrics")
                continue
            try:
                out.append(int(tok))
            except ValueError:
                print(f"[WARN] Invalid phase: {tok}")
        return out
    out = [i for i in range(1, 7) if cfg.get("phases", {}).get(f"phase{i}", {}).get("enabled", False)]
    if cfg.get("metrics", {}).get("enabled", False):
        out.append("metrics")
    return out


def pipeline_v2_main(argv=None, src_dir=None):
    ap = argparse.ArgumentParser(description="Audio Restoration Pipeline Orchestrator v2")
    ap.add_argument("--config", required=True, help="Path to YAML config")
    ap.add_argument("--input", default=None, help="Override input audio path")
    ap.add_argument("--noise", default=None, help="Override noise sample path")
    ap.add_argument("--reference", default=None, help="Reference audio for metrics")
    ap.add_argument("--phases", default=None, help="Comma list of phases (e.g., 1,2,3 or all)")
    ap.add_argument("--dry-run", action="store_true", help="Plan execution without running")
    ap.add_argument("--strict", action="store_true", help="Fail on missing dependencies")
    ap.add_argument("--profile", default=None, help="Mastering profile")
    args = ap.parse_args(argv)

    cfg = load_config(args.config)
    for key, val in (("input", args.input), ("noise", args.noise), ("reference", args.reference)):
        if val:
            cfg["paths"][key] = val
    paths = cfg["paths"]
    dry_run = args.dry_run or cfg.get("global", {}).get("dry_run", False)
    strict = args.strict or cfg.get("global", {}).get("strict_mode", False)
    selected = select_phases(args.phases, cfg)
    print("[INFO] Pipeline v2 starting")
    print(f"[INFO] Phases to run: {selected}")
    print(f"[INFO] Mode: {'DRY-RUN' if dry_run else 'EXECUTE'}")
    print(f"[INFO] Strict mode: {strict}")
    for key in ("outputs_dir", "logs_dir", "reports_dir", "plots_dir"):
        if key in paths:
            Path(paths[key]).mkdir(parents=True, exist_ok=True)
    with open(os.path.join(paths["logs_dir"], "resolved_config.json"), "w") as fh:
        json.dump(cfg, fh, indent=2)

    src = Path(src_dir) if src_dir else Path(__file__).resolve().parents[1] / "src"
    extra_args = {"mastering_chain.py": ["--profile", args.profile] if args.profile else [],
                  "metrics_eval.py": ["--reference", args.reference] if args.reference else []}
    results, missing = {}, []
    for phase in ALL_PHASES:
        if phase not in selected:
            continue
        headline, steps = PHASE_PLAN[phase]
        print("\n" + headline)
        deps_ok = check_dependencies(phase, strict)
        for key, script, log in steps:
            if phase == 4 and strict and not deps_ok:
                print("[SKIP] Phase 4 skipped due to missing dependencies")
                results[key] = {"status": "skipped", "reason": "missing_deps"}
                continue
            if not (src / script).exists() and not dry_run:
                if strict or os.environ.get("ADN_ALLOW_MISSING_STAGES", "0") != "1":
                    raise RuntimeError(f"Step failed: stage script not found: {src / script}")
                print(f"[SKIP] {script} is not installed next to the orchestrator")
                results[key] = {"status": "skipped", "reason": "script_missing"}
                missing.append(script)
                continue
            cmd = [sys.executable, str(src / script), "--config", args.config] + extra_args.get(script, [])
            results[key] = run_step(cmd, os.path.join(paths["logs_dir"], log), dry_run)

    print("\n[INFO] Creating run manifest...")
    manifest = create_run_manifest(cfg, args, results)
    manifest_path = paths.get("manifest_path", os.path.join(paths.get("reports_dir", "."), "run_manifest.json"))
    Path(os.path.dirname(manifest_path) or ".").mkdir(parents=True, exist_ok=True)
    with open(manifest_path, "w") as fh:
        json.dump(manifest, fh, indent=2)
    print(f"[INFO] Run manifest saved to {manifest_path}")

    total = sum(r.get("elapsed", 0) for r in results.values())
    rule = "=" * 60
    print("\n" + rule + "\nPIPELINE COMPLETE\n" + rule)
    print(f"Total time: {total:.1f}s")
    print(f"Phases run: {sum(1 for r in results.values() if r.get('status') == 'success')}")
    print(f"Phases skipped: {sum(1 for r in results.values() if r.get('status') == 'skipped')}")
    final_output = paths.get("final_output")
    if not dry_run and final_output and os.path.exists(final_output):
        print(f"Final output: {final_output}")
        print(f"Output size: {os.path.getsize(final_output) / (1024 * 1024):.2f} MB")
    print(rule)
    if missing:
        raise SystemExit("selected phases did not run, their scripts are missing: " + ", ".join(missing))
    return manifest


def pipeline_main(argv=None, src_dir=None):
    """Phases 1-2 orchestrator (src/pipeline.py:25-78): the input and noise overrides are passed on
    to the scripts that take them."""
    import yaml
    ap = argparse.ArgumentParser(description="Audio Restoration Pipeline Orchestrator (Phases 1-2)")
    ap.add_argument("--config", required=True, help="Path to YAML config")
    ap.add_argument("--input", default=None, help="Override input audio path")
    ap.add_argument("--noise", default=None, help="Override noise sample path")
    ap.add_argument("--phases", default="1,2", help="Comma list of phases to execute, e.g., 1,2 or 1")
    args = ap.parse_args(argv)
    with open(args.config, "r") as fh:
        cfg = yaml.safe_load(fh)
    paths = cfg["paths"]
    if args.input:
        paths["input"] = args.input
    if args.noise:
        paths["noise"] = args.noise
    for key in ("outputs_dir", "logs_dir", "reports_dir"):
        Path(paths[key]).mkdir(parents=True, exist_ok=True)
    with open(os.path.join(paths["logs_dir"], "resolved_config.json"), "w") as fh:
        json.dump(cfg, fh, indent=2)
    src = Path(src_dir) if src_dir else Path(__file__).resolve().parents[1] / "src"
    overrides = {"analyze_audio.py": ["--input", paths["input"], "--noise", paths["noise"]],
                 "noise_profile.py": ["--noise", paths["noise"]],
                 "wavelet_denoise.py": ["--input", paths["input"]]}
    selected = [p.strip() for p in args.phases.split(",") if p.strip()]
    for phase in (1, 2):
        if str(phase) not in selected:
            continue
        for _key, script, log in PHASE_PLAN[phase][1]:
            _run_logged([sys.executable, str(src / script), "--config", args.config] + overrides.get(script, []),
                        os.path.join(paths["logs_dir"], log))
    print("Pipeline phases complete.")
