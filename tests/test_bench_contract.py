"""bench.py's output contract (one JSON line with the fields the driver reads), on a small frame so that it runs in seconds."""
import json
import os
import subprocess
import sys

import pytest

import util


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--width", "1280", "--height", "1024", "--cpu-seconds", "0.5"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    # the driver keeps 8 KB of stdout: the line has to fit, and nothing may follow it (round 4's 23 KB line came back unparsed)
    assert len(lines[0]) < 8000, len(lines[0])
    assert r.stdout.rstrip("\n").splitlines()[-1] == lines[0]
    d = json.loads(lines[0])
    full = json.load(open(os.path.join(util.ROOT, "bench_legs.json")))      # every leg in full next to bench.py
    assert full["value"] == d["value"] and "full_table" in full and "stages_ms" in full["full_table"]
    # roofline.traffic is counted in the run itself (rocprofv3 children over --step-only) when the profiler is there
    assert "counted in this run" in d["roofline"]["traffic_source"] or "failed" in d["roofline"]["traffic_source"], d["roofline"]["traffic_source"]
    if "counted in this run" in d["roofline"]["traffic_source"]:
        assert d["roofline"]["traffic"] > 0 and d["roofline"]["path_traffic"] > 0
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and d["config"]["parity"].startswith("bit-exact")
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1
    cpu = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] == 1 and cpu["value"] > 0
    assert d["value"] > cpu["value"]
    assert d["legs"]["full_table"]["parity"] == "bit-exact" and full["full_table"]["parity"].startswith("every stage bit-exact")


@pytest.mark.gpu
def test_bench_tile_column_mode_on_one_gpu():
    """--shard tile-cols with one rank: torch-owned pictures behind the C ABI, the column split (one column) and the gather
    (a no-op) — the path the N > 1 tile-column runs take, bit-exact against the oracle."""
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--width", "1280", "--height", "1024", "--no-cpu", "--shard", "tile-cols"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["scaling"] == "strong" and "tile-columns" in d["config"]["parallelism"]
    assert d["config"]["parity"].startswith("bit-exact")


@pytest.mark.gpu
def test_bench_tile_column_mode_with_in_loop_filters_on_one_gpu():
    """--shard tile-cols --tc-filters with one rank: the halo exchange is a no-op, the rank's share of the filter tasks is all of
    them, and the gathered picture equals the oracle's deblock + CDEF + restoration of the whole frame."""
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--width", "1280", "--height", "1024", "--no-cpu", "--shard", "tile-cols", "--tc-filters"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "in-loop filters per column" in d["config"]["parallelism"]
    assert d["config"]["parity"].startswith("bit-exact") and "restoration" in d["config"]["parity"]
