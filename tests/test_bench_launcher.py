"""`python bench.py --gpus N` launches its N ranks itself: run on the CPU with gloo ranks and the
stand-in step of --dry-run, the launcher must print ONE JSON line with n_gpus = N; a rank that fails makes it exit non-zero."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=timeout)


def test_gpus_2_spawns_two_ranks_and_relays_one_json_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["config"]["global_batch"] == 64
    assert j["scaling"] == "weak" and j["dry_run"] is True


def test_single_rank_dry_run_needs_no_launcher():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1


def test_mismatched_world_size_is_refused():
    r = _run(["--gpus", "2", "--dry-run"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_committed_traffic_is_reported_only_for_the_sources_it_was_measured_on(monkeypatch):
    """bench.py prints roofline.traffic from profiles/r04_pmc_traffic_p<precision>.json only when the file carries the hash of the
    kernel sources that are running; a profile of other sources is named as stale and not reported."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    f = os.path.join(root, "profiles", "r04_pmc_traffic_p1.json")
    rec = json.load(open(f))
    assert len(bench.source_sha16()) == 16 and rec.get("commit") and rec.get("src_sha16")
    monkeypatch.setattr(bench, "source_sha16", lambda: rec["src_sha16"])
    j, note = bench.committed_traffic(1)
    assert j is not None and j["total_bytes_per_step"] == rec["total_bytes_per_step"] and rec["commit"] in note
    monkeypatch.setattr(bench, "source_sha16", lambda: "0" * 16)
    j, note = bench.committed_traffic(1)
    assert j is None and "stale" in note
    j, note = bench.committed_traffic(2)
    assert j is None and "no counter profile" in note
